// Round 6: a register-only matrix-instruction loop as a co-runner (tools/experiments/op_beside_model.py BURN=1): which instruction of ANOTHER
// wave on the SIMD disturbs the fp32 fused Up block?  Built on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/ubench/mfma_burn.hip -o /tmp/libburn.so
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void burn_kernel(float* sink, int iters, unsigned seed) {
  const unsigned t = threadIdx.x + seed;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)(0.001f * ((t * 7 + j) % 13) - 0.006f), b[j] = (__bf16)(0.001f * ((t * 5 + j) % 11) - 0.005f);
  const float fa = 0.001f * (t % 17), fb = 0.002f * (t % 5);
  f32x16 c16[2] = {};
  f32x4 c4[4] = {};
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {          // v_mfma_f32_32x32x16_bf16 (the 128x128 GEMMs)
      c16[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[0], 0, 0, 0);
      c16[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c16[1], 0, 0, 0);
    } else if constexpr (KIND == 1) {   // v_mfma_f32_16x16x32_bf16
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[k], 0, 0, 0);
    } else if constexpr (KIND == 2) {   // v_mfma_f32_32x32x2_f32
      c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c16[0], 0, 0, 0);
      c16[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c16[1], 0, 0, 0);
    } else if constexpr (KIND == 3) {   // v_mfma_f32_16x16x4_f32
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c4[k], 0, 0, 0);
    } else {                            // packed fp32 FMAs only
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = c4[k] * fa + fb;
    }
  }
  float s = 0;
  for (int k = 0; k < 16; ++k) s += c16[0][k] + c16[1][k];
  for (int k = 0; k < 4; ++k) s += c4[k][0] + c4[k][3];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

extern "C" int burn(int kind, int iters, int blocks, void* stream, float* sink) {
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(burn_kernel<0>, dim3(blocks), dim3(256), 0, s, sink, iters, 1u); break;
    case 1: hipLaunchKernelGGL(burn_kernel<1>, dim3(blocks), dim3(256), 0, s, sink, iters, 1u); break;
    case 2: hipLaunchKernelGGL(burn_kernel<2>, dim3(blocks), dim3(256), 0, s, sink, iters, 1u); break;
    case 3: hipLaunchKernelGGL(burn_kernel<3>, dim3(blocks), dim3(256), 0, s, sink, iters, 1u); break;
    default: hipLaunchKernelGGL(burn_kernel<4>, dim3(blocks), dim3(256), 0, s, sink, iters, 1u); break;
  }
  return (int)hipGetLastError();
}
