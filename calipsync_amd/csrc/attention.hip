// Cross-attention core of CASync (reference module/unet.py:207-218), per frame:
//
//   E[i][j] = sum_d Q[i][d] K[j][d]          i: 100 face positions, j: 100 audio positions
//   P       = softmax_j(E)                   (no 1/sqrt(d) scale, single head)
//   out[i][c] = gamma * sum_j P[i][j] V[j][c] + res[i][c]
//
// Q, K, V come from the 1x1 projections (NHWC rows: position-major, channel contiguous).
// One workgroup = 32 query rows of one frame (grid = B x 4); the 32x100 score tile lives in
// LDS only.  QK^T and PV run on v_mfma_f32_32x32x2_f32; the row softmax is a 64-lane
// shuffle reduction (two keys per lane).
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

template <typename T>
__global__ __launch_bounds__(256) void cross_attention_kernel(
    const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
    const T* __restrict__ v, int ldv, const T* __restrict__ res, int ld_res,
    const float* __restrict__ gamma, T* __restrict__ out, int ld_out) {
  __shared__ __attribute__((aligned(16))) float Qs[32 * QLD];
  __shared__ __attribute__((aligned(16))) float Ks[128 * QLD];
  __shared__ __attribute__((aligned(16))) float Ss[32 * SLD];
  const int b = blockIdx.x, qb = blockIdx.y;
  const size_t row0 = (size_t)b * NP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, kh = lane >> 5;

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
  __syncthreads();

  // ---- row softmax over the 100 keys: wave w owns rows 8w .. 8w+7 ----
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
  __syncthreads();

  // ---- out = P V: wave w owns value channels 128w .. 128w+127 (4 MFMA column tiles) ----
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* pa = Ss + r32 * SLD + 4 * kh;
  const T* vb = v + wave * 128 + r32;
#pragma unroll 1
  for (int g = 0; g < 13; ++g) {  // 13 groups of 8 keys cover 0..103; P is zero past 99
    const f32x4 fa = *reinterpret_cast<const f32x4*>(pa + 8 * g);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int key = 8 * g + 4 * kh + s;
      key = key < NP ? key : NP - 1;  // stay inside this frame's rows (weight is 0 there)
      const T* vrow = vb + (row0 + key) * ldv;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], (float)vrow[j * 32], acc[j], 0, 0, 0);
    }
  }
  const float gam = gamma[0];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = wave * 128 + j * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (qi < NP)
        out[(row0 + qi) * ld_out + c] = (T)(gam * acc[j][r] + (float)res[(row0 + qi) * ld_res + c]);
    }
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
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(cross_attention_kernel<bf16_t>, dim3(batch, 4), dim3(256), 0, stream, (const bf16_t*)q, ldq,
                       (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)res, ld_res, gamma_dev,
                       (bf16_t*)out, ld_out);
  else
    hipLaunchKernelGGL(cross_attention_kernel<float>, dim3(batch, 4), dim3(256), 0, stream, (const float*)q, ldq,
                       (const float*)k, ldk, (const float*)v, ldv, (const float*)res, ld_res, gamma_dev,
                       (float*)out, ld_out);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}
