// Do fp32 MFMA and fp32 VALU FMAs run concurrently on gfx950?  One 512-thread workgroup per CU: waves 0-3
// (one per SIMD) loop on v_mfma_f32_32x32x2_f32, waves 4-7 (their SIMD partners) on v_pk_fma_f32 or v_fma_f32.
// mode 1 = MFMA waves only, 2 = VALU waves only, 3 = both.  Clock warmed by ~1 s of the same kernel first; the
// shader clock is read back (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/experiments/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters, int mode, int viters) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f - 0.1f, y = 1.0f - blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
      }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0) {
      stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
      stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
  } else {
    if (!(mode & 2)) return;
    f32x2 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x2{(float)j, (float)threadIdx.x};
    f32x2 x = {1.0001f, 0.9999f}, y = {threadIdx.x * 1e-7f, blockIdx.x * 1e-7f};
    for (int i = 0; i < viters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (PACKED) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[j]) : "v"(x), "v"(y));
          else asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[j][0]) : "v"(x[0]), "v"(y[0]));
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 256 && !(mode & 1)) {
      stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
      stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
  }
}

template <bool PACKED>
void sweep() {
  float* d;
  unsigned long long* st;
  hipMalloc(&d, 4096);
  hipMalloc(&st, 2 * 256 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 3000, wgs = 256;      // 3000 x 16 MFMA x 64 cycles = 1.3 ms at 2.39 GHz
  for (int rep = 0; rep < 400; ++rep) hipLaunchKernelGGL(k<PACKED>, dim3(wgs), dim3(512), 0, 0, d, st, iters, 1, 0);   // warm
  hipDeviceSynchronize();
  for (int vm = 1; vm <= 4; vm *= 2)
    for (int mode = 1; mode <= 3; ++mode) {
      // VALU ops per MFMA op issued by the partner wave: 64 x vm per 16 MFMAs
      const int viters = iters * vm;
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k<PACKED>, dim3(wgs), dim3(512), 0, 0, d, st, iters, mode, viters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      unsigned long long h[2];
      hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
      const double mf = (mode & 1) ? (double)wgs * 4 * iters * 16 * 4096.0 : 0;
      const double vf = (mode & 2) ? (double)wgs * 4 * viters * 64 * 64 * (PACKED ? 4.0 : 2.0) : 0;
      printf("%s vm %d mode %d: %7.3f ms  mfma %6.1f TF  valu %6.1f TF  total %6.1f TF  clock %.2f GHz\n", PACKED ? "v_pk_fma_f32" : "v_fma_f32   ",
             vm, mode, ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9, (double)h[0] / (double)h[1] * 0.1);
    }
}

int main() {
  sweep<true>();
  sweep<false>();
  return 0;
}
