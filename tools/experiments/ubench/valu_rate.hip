// Issue rate of the vector instructions the depthwise phases are built from (one wave per SIMD, 16 independent
// accumulators, s_memtime around 4096 x 16 instructions): cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void rate(float* out, unsigned long long* cyc, unsigned a0, unsigned b0) {
  float acc[16];
  f32x2 pacc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (float)i, pacc[i] = f32x2{(float)i, 1.f};
  const bf16x2 a = __builtin_bit_cast(bf16x2, a0 + threadIdx.x), b = __builtin_bit_cast(bf16x2, b0);
  const float fa = __builtin_bit_cast(float, a0), fb = __builtin_bit_cast(float, b0);
  const f32x2 pa = {fa, fb}, pb = {fb, fa};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 4096; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc[i], false);
      if (MODE == 1) acc[i] = __builtin_fmaf(fa, fb, acc[i]);
      if (MODE == 2) pacc[i] = __builtin_elementwise_fma(pa, pb, pacc[i]);
      if (MODE == 3) acc[i] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, acc[i]) << 16);
      if (MODE == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(acc[i]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + pacc[i][0] + pacc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 << 20); hipHostMalloc((void**)&cyc, 8);
  const char* names[] = {"v_dot2c_f32_bf16", "v_fma_f32", "v_pk_fma_f32", "v_lshlrev_b32", "v_and_b32 (literal)"};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // 256 CUs x 4 workgroups of 256 threads: 4 waves per SIMD, every SIMD busy; 4 x 4096 x 16 instructions per SIMD
  for (int m = 0; m < 5; ++m) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      dim3 g(1024), b(256);
      hipEventRecord(e0, 0);
      if (m == 0) hipLaunchKernelGGL(rate<0>, g, b, 0, 0, out, cyc, 0x3f803f80u, 0x3f803f80u);
      if (m == 1) hipLaunchKernelGGL(rate<1>, g, b, 0, 0, out, cyc, 0x3f803f80u, 0x3f803f80u);
      if (m == 2) hipLaunchKernelGGL(rate<2>, g, b, 0, 0, out, cyc, 0x3f803f80u, 0x3f803f80u);
      if (m == 3) hipLaunchKernelGGL(rate<3>, g, b, 0, 0, out, cyc, 0x3f803f80u, 0x3f803f80u);
      if (m == 4) hipLaunchKernelGGL(rate<4>, g, b, 0, 0, out, cyc, 0x3f803f80u, 0x3f803f80u);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double per_simd = 4.0 * 4096 * 16;
    printf("%-20s %.3f ms for %.0f wave-instructions per SIMD: %.2f ns each = %.2f cycles at 2.4 GHz\n", names[m], ms, per_simd,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
  return 0;
}
