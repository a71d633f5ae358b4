// Cross-attention core of CASync (reference module/unet.py:207-218), per frame:
//
//   E[i][j] = sum_d Q[i][d] K[j][d]          i: 100 face positions, j: 100 audio positions
//   P       = softmax_j(E)                   (no 1/sqrt(d) scale, single head)
//   out[i][c] = gamma * sum_j P[i][j] V[j][c] + res[i][c]
//
// Q, K, V come from the 1x1 projections (NHWC rows: position-major, channel contiguous).
// One workgroup = 32 query rows of one frame x 512/nz value channels, walked in 128-channel chunks
// (grid = B x 4 x nz).  The 32x100 score tile lives in LDS only and is computed once per
// workgroup; nz = 4 (one chunk each, scores recomputed: QK^T is ~10 % of the work) fills the chip
// at small batches (<= 32 frames), nz = 1 avoids the 4x Q/K re-reads once B x 4 workgroups are plenty.
// Each V chunk is fetched into registers ahead of time (the first at kernel entry, the next under
// the previous chunk's PV) and parked in LDS over the dead K tile, so its HBM/L2 latency hides.  QK^T and PV run on
// v_mfma_f32_32x32x2_f32; the row softmax keeps a row on eight lanes (16 keys each) and reduces them by DPP.
#include "common.h"

namespace {

constexpr int NP = 100;   // positions per frame (10 x 10)
constexpr int DQK = 64;   // query/key channels
constexpr int CV = 512;   // value channels
constexpr int QLD = 68;   // LDS row strides (floats): 16-B aligned, b128 reads conflict-free
constexpr int SLD = 132;

// Value of lane (l ^ 1), (l ^ 2) and (l ^ 7 within its group of eight) by DPP (no LDS round trip): the reduction over the
// eight lanes that share a softmax row.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141;   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float row8_max(float v) {
  v = fmaxf(v, dpp_f<DPP_XOR1>(v));
  v = fmaxf(v, dpp_f<DPP_XOR2>(v));
  return fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
}
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_f<DPP_XOR1>(v);
  v += dpp_f<DPP_XOR2>(v);
  return v + dpp_f<DPP_HALF_MIRROR>(v);
}

// LDS carve (floats): Q block, score tile, then K (QK^T phase) / V chunk (PV phase) share one area
constexpr int VLD = 132;                       // V chunk rows: 128 channels + 4
constexpr int oQ = 0, oS = oQ + 32 * QLD, oKV = oS + 32 * SLD;
constexpr int KV_FLOATS = 128 * QLD > NP * VLD ? 128 * QLD : NP * VLD;
constexpr int ATT_LDS_BYTES = (oKV + KV_FLOATS) * 4;
constexpr int VREGS = (NP * 128 / 4 + 255) / 256;   // float4 per thread holding the V chunk in flight

template <typename T>
__global__ __launch_bounds__(256) void cross_attention_kernel(
    const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
    const T* __restrict__ v, int ldv, const T* __restrict__ res, int ld_res,
    const float* __restrict__ gamma, T* __restrict__ out, int ld_out) {
  extern __shared__ __attribute__((aligned(16))) float att_lds[];
  float* Qs = att_lds + oQ;
  float* Ss = att_lds + oS;
  float* Ks = att_lds + oKV;   // [128][QLD] during QK^T
  float* Vs = att_lds + oKV;   // [100][VLD] during PV (after the scores are done with K)
  const int b = blockIdx.x, qb = blockIdx.y;                    // frame, 32-query block
  const int cpw = 4 / (int)gridDim.z, cz0 = blockIdx.z * cpw;   // this workgroup's 128-channel chunks
  const size_t row0 = (size_t)b * NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kh = lane >> 5;

  // ---- the V chunk [100 keys][128 channels] starts its trip from HBM/L2 now and is parked in
  //      registers until K is dead (it arrives under the staging, QK^T and softmax phases) ----
  f32x4 vreg[VREGS];
  auto vfetch = [&](int cz) {
#pragma unroll
    for (int j = 0; j < VREGS; ++j) {
      const int idx = tid + 256 * j;              // float4 index: key = idx/32, 4 channels at (idx%32)*4
      if (idx < NP * 32) vreg[j] = ld4(v + (row0 + (idx >> 5)) * ldv + cz * 128 + (idx & 31) * 4);
    }
  };
  vfetch(cz0);
  const float gam = gamma[0];   // (requested here, not behind two barriers)

  // ---- stage Q block and all K rows (rows >= 100 are zero) ----
  {
    const int r = tid >> 4, c4 = (tid & 15) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lr = r + 16 * p, qi = qb * 32 + lr;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (qi < NP) val = ld4(q + (row0 + qi) * ldq + c4);
      *reinterpret_cast<f32x4*>(Qs + lr * QLD + c4) = val;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int kr = r + 16 * p;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (kr < NP) val = ld4(k + (row0 + kr) * ldk + c4);
      *reinterpret_cast<f32x4*>(Ks + kr * QLD + c4) = val;
    }
  }
  __syncthreads();

  // ---- scores: wave w owns keys 32w .. 32w+31 ----
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* qa = Qs + r32 * QLD + 4 * kh;
    const float* kb = Ks + (wave * 32 + r32) * QLD + 4 * kh;
#pragma unroll
    for (int g = 0; g < DQK / 8; ++g) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(qa + 8 * g);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(kb + 8 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * kh;
      Ss[i * SLD + wave * 32 + r32] = acc[r];
    }
  }
  __syncthreads();   // scores complete, K dead

  for (int ci = 0; ci < cpw; ++ci) {
    const int cz = cz0 + ci;
    // ---- park this chunk's V in LDS (over K / the previous chunk) ----
#pragma unroll
    for (int j = 0; j < VREGS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NP * 32) *reinterpret_cast<f32x4*>(Vs + (idx >> 5) * VLD + (idx & 31) * 4) = vreg[j];
    }
    if (ci == 0) {
      // ---- row softmax: wave w owns rows 8w .. 8w+7, EIGHT LANES PER ROW: lane (row l>>3, slice s = l&7) holds keys
      //      4s + 32j + e (j, e < 4; four conflict-free 16-B reads), reduces its 16 values locally and the eight partial
      //      results with three DPP moves (quad xor 1, xor 2, half-row mirror).  (Round 4: one row per wave pass cost 12
      //      dependent ds_bpermute round trips per row, 96 per wave.) ----
      float* srow = Ss + (wave * 8 + (lane >> 3)) * SLD + 4 * (lane & 7);
      f32x4 x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = *reinterpret_cast<const f32x4*>(srow + 32 * j);
      if ((lane & 7) != 0) x[3] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // keys 100..127 do not exist
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, x[j][e]);
      mx = row8_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[j][e] = expf(x[j][e] - mx);   // exp(-inf) = 0: the missing keys become exact zeros (PV reads them)
          sum += x[j][e];
        }
      const float inv = 1.f / row8_sum(sum);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(srow + 32 * j) = x[j] * inv;
    }
    __syncthreads();
    if (ci + 1 < cpw) vfetch(cz + 1);   // next chunk's V travels under this chunk's PV
    const int c = cz * 128 + wave * 32 + r32;
    float rres[16];                     // ... and so do the residual values of this chunk's outputs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      rres[r] = qi < NP ? (float)res[(row0 + qi) * ld_res + c] : 0.f;
    }

    // ---- out = P V: wave w owns channels 32w .. 32w+31 of the 128-channel chunk ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* pa = Ss + r32 * SLD + 4 * kh;
    const float* vb = Vs + wave * 32 + r32;
#pragma unroll
    for (int g = 0; g < 13; ++g) {  // 13 groups of 8 keys cover 0..103; P is zero past 99
      const f32x4 fa = *reinterpret_cast<const f32x4*>(pa + 8 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int key = 8 * g + 4 * kh + s;
        key = key < NP ? key : NP - 1;  // stay inside the staged rows (weight is 0 there)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], vb[key * VLD], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (qi < NP) out[(row0 + qi) * ld_out + c] = (T)(gam * acc[r] + rres[r]);
    }
    if (ci + 1 < cpw) __syncthreads();   // everyone is done with Vs before the next chunk is parked
  }
}

}  // namespace

const char* cross_attention_kernel_name(int dtype) {
  if (dtype == DT_BF16) return casync_opts().att_bf16 ? "cross_attention_bf16_kernel" : "cross_attention_kernel<__bf16>";
  return "cross_attention_kernel<float>";
}

int launch_cross_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                           const void* res, int ld_res, const float* gamma_dev, void* out,
                           int ld_out, int batch, hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(q && k && v && res && gamma_dev && out, "cross_attention: null pointer");
  CASYNC_REQUIRE(batch > 0, "cross_attention: batch %d", batch);
  CASYNC_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldq >= DQK && ldk >= DQK && ldv >= CV &&
                     ld_res >= CV && ld_out >= CV,
                 "cross_attention: bad leading dimensions");
  CASYNC_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0, "cross_attention: Q/K alignment");
  // the bf16 engine's own kernel: both products on bf16 matrix instructions, one workgroup per frame (no channel split);
  // a forced `att_nz` asks for the split kernel below, in bf16 too
  if (dtype == DT_BF16 && casync_opts().att_bf16 && casync_opts().att_nz == 0)
    return launch_cross_attention_bf16(q, ldq, k, ldk, v, ldv, res, ld_res, gamma_dev, out, ld_out, batch, stream);
  static unsigned long long once_f32 = 0, once_bf16 = 0;
  if (int st = dtype == DT_BF16
                   ? casync_ensure_dyn_lds(&once_bf16, reinterpret_cast<const void*>(cross_attention_kernel<bf16_t>), ATT_LDS_BYTES)
                   : casync_ensure_dyn_lds(&once_f32, reinterpret_cast<const void*>(cross_attention_kernel<float>), ATT_LDS_BYTES))
    return st;
  // frame x 32-query block x channel split: enough workgroups to fill 256 CUs x ~4, no more
  const int forced_nz = casync_opts().att_nz;
  int nz = batch <= 32 ? 4 : (batch <= 160 ? 2 : 1);   // measured (round 4 kernel): 4 up to a lane of B=64 (+0.15 % over 2 there), 1 loses 0.8 %
  if (forced_nz == 1 || forced_nz == 2 || forced_nz == 4) nz = forced_nz;
  const dim3 grid(batch, 4, nz);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(cross_attention_kernel<bf16_t>, grid, dim3(256), ATT_LDS_BYTES, stream, (const bf16_t*)q,
                       ldq, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)res, ld_res, gamma_dev,
                       (bf16_t*)out, ld_out);
  else
    hipLaunchKernelGGL(cross_attention_kernel<float>, grid, dim3(256), ATT_LDS_BYTES, stream, (const float*)q,
                       ldq, (const float*)k, ldk, (const float*)v, ldv, (const float*)res, ld_res, gamma_dev,
                       (float*)out, ld_out);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}
