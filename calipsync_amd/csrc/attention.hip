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
// at small batches (<= 24 frames), nz = 1 avoids the 4x Q/K re-reads once B x 4 workgroups are plenty.
// Each V chunk is fetched into registers ahead of time (the first at kernel entry, the next under
// the previous chunk's PV) and parked in LDS over the dead K tile, so its HBM/L2 latency hides.  QK^T and PV run on
// v_mfma_f32_32x32x2_f32; the row softmax is a 64-lane shuffle reduction (two keys per lane).
#include "common.h"

namespace {

constexpr int NP = 100;   // positions per frame (10 x 10)
constexpr int DQK = 64;   // query/key channels
constexpr int CV = 512;   // value channels
constexpr int QLD = 68;   // LDS row strides (floats): 16-B aligned, b128 reads conflict-free
constexpr int SLD = 132;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
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

  const float gam = gamma[0];
  for (int ci = 0; ci < cpw; ++ci) {
    const int cz = cz0 + ci;
    // ---- park this chunk's V in LDS (over K / the previous chunk) ----
#pragma unroll
    for (int j = 0; j < VREGS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < NP * 32) *reinterpret_cast<f32x4*>(Vs + (idx >> 5) * VLD + (idx & 31) * 4) = vreg[j];
    }
    if (ci == 0) {
      // ---- row softmax: wave w owns rows 8w .. 8w+7 ----
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        float* srow = Ss + (wave * 8 + rr) * SLD;
        const bool has2 = lane + 64 < NP;
        const float e0 = srow[lane];
        const float e1 = has2 ? srow[lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(e0, e1));
        const float p0 = expf(e0 - mx);
        const float p1 = has2 ? expf(e1 - mx) : 0.f;
        const float inv = 1.f / wave_sum(p0 + p1);
        srow[lane] = p0 * inv;
        srow[lane + 64] = p1 * inv;  // keys 100..127 become exact zeros
      }
    }
    __syncthreads();
    if (ci + 1 < cpw) vfetch(cz + 1);   // next chunk's V travels under this chunk's PV

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
    const int c = cz * 128 + wave * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (qi < NP) out[(row0 + qi) * ld_out + c] = (T)(gam * acc[r] + (float)res[(row0 + qi) * ld_res + c]);
    }
    if (ci + 1 < cpw) __syncthreads();   // everyone is done with Vs before the next chunk is parked
  }
}

}  // namespace

int launch_cross_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                           const void* res, int ld_res, const float* gamma_dev, void* out,
                           int ld_out, int batch, hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(q && k && v && res && gamma_dev && out, "cross_attention: null pointer");
  CASYNC_REQUIRE(batch > 0, "cross_attention: batch %d", batch);
  CASYNC_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldq >= DQK && ldk >= DQK && ldv >= CV &&
                     ld_res >= CV && ld_out >= CV,
                 "cross_attention: bad leading dimensions");
  CASYNC_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0, "cross_attention: Q/K alignment");
  static unsigned long long once_f32 = 0, once_bf16 = 0;
  if (int st = dtype == DT_BF16
                   ? casync_ensure_dyn_lds(&once_bf16, reinterpret_cast<const void*>(cross_attention_kernel<bf16_t>), ATT_LDS_BYTES)
                   : casync_ensure_dyn_lds(&once_f32, reinterpret_cast<const void*>(cross_attention_kernel<float>), ATT_LDS_BYTES))
    return st;
  // frame x 32-query block x channel split: enough workgroups to fill 256 CUs x ~4, no more
  const int forced_nz = casync_opts().att_nz;
  int nz = batch <= 24 ? 4 : (batch <= 160 ? 2 : 1);   // measured: 4 wins up to 24 frames, 2 from 32 (a lane of B=64)
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
