// Premise test: can fp32 MFMA and packed-fp32 VALU FMAs run concurrently at full rate on gfx950?
// 8 waves per workgroup (2 per SIMD): waves 0-3 loop on v_mfma_f32_32x32x2_f32, waves 4-7 on
// v_pk_fma_f32.  mode 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/experiments/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int viters) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    f32x2 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x2{(float)j, (float)threadIdx.x};
    f32x2 x = {1.0001f, 0.9999f}, y = {threadIdx.x * 1e-7f, blockIdx.x * 1e-7f};
    for (int i = 0; i < viters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[j]) : "v"(x), "v"(y));
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, wgs = 256 * 4;
  for (int vm = 1; vm <= 4; ++vm)
  for (int mode = 1; mode <= 3; ++mode) {
    const int viters = iters * vm;
    hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, d, 100, mode, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, d, iters, mode, viters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (mode & 1) ? (double)wgs * 4 * iters * 16 * 4096.0 : 0;          // 32x32x2x2 flops per MFMA
    const double vf = (mode & 2) ? (double)wgs * 4 * viters * 64 * 64 * 4.0 : 0;       // 64 pk_fma x 64 lanes x 4 flops
    printf("vm %d mode %d: %.3f ms  mfma %.1f TF  valu %.1f TF  total %.1f TF\n", vm, mode, ms, mf / ms * 1e-9, vf / ms * 1e-9,
           (mf + vf) / ms * 1e-9);
  }
  return 0;
}
