// Write bandwidth of the store patterns the engine's kernels use, on a [M][128] fp32 tensor (512-B rows, M = 204800: 105 MB, six
// buffers in rotation so that every launch writes HBM, not the 256-MB Infinity Cache).  hipcc --offload-arch=gfx950 -O3.
//   0  linear: a wave instruction writes 1 KB contiguous (what a fill does)
//   1  GEMM epilogue: 16 lanes = 256 B of one row, the four 16-lane phases = four rows (64-column tile)
//   2  MFMA D[channel][pixel] direct (ir_fused, first skinny GEMM): a lane = 16 B of one pixel, the 16 lanes of a phase = 16 pixels
//   3  MFMA D[pixel][channel] direct (second skinny GEMM): dword stores, the 16 lanes of a phase = 64 B of one pixel
//   4  as 2, but the eight 16-channel tiles of a pixel row written back to back by one wave (full 512-B rows per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(256) void wr(float* C, int M) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  if (P == 0) {
    const long long n16 = (long long)M * 32;   // 16-B pieces
    for (long long i = gw * 64 + lane; i < n16; i += nw * 64) *reinterpret_cast<f32x4*>(C + i * 4) = v;
  } else if (P == 1) {   // tiles of 64 rows x 64 columns; wave = 16 rows; per instruction 4 rows x 256 B
    const long long ntile = (long long)(M / 64) * 2;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
      const long long m0 = (t >> 1) * 64 + (threadIdx.x >> 6) * 16, n0 = (t & 1) * 64;
#pragma unroll
      for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(C + (m0 + 4 * p + q) * 128 + n0 + l15 * 4) = v;
    }
  } else {
    const long long ntile = M / 16;   // 16-pixel tiles, all 128 channels by one wave
    for (long long t = gw; t < ntile; t += nw) {
      if (P == 2 || P == 4) {
        float* dst = C + (t * 16 + l15) * 128 + 4 * q;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) *reinterpret_cast<f32x4*>(dst + 16 * tt) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* dst = C + (t * 16 + 4 * q + r) * 128 + l15;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) dst[16 * tt] = v[r];
        }
      }
    }
  }
}
int main() {
  const int M = 204800, NB = 6;
  float* buf[NB];
  for (int i = 0; i < NB; ++i) hipMalloc(&buf[i], (size_t)M * 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"linear 1 KB per wave instruction", "GEMM epilogue (4 rows x 256 B)", "D[ch][px] direct (16 px x 16 B per phase)",
                         "D[px][ch] direct (dword, 64 B per phase)"};
  for (int p = 0; p < 4; ++p)
    for (int grid : {1024, 4096}) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 24; ++i) {
          float* c = buf[i % NB];
          if (p == 0) hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, c, M);
          if (p == 1) hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, c, M);
          if (p == 2) hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(256), 0, 0, c, M);
          if (p == 3) hipLaunchKernelGGL(wr<3>, dim3(grid), dim3(256), 0, 0, c, M);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("%-44s grid %5d: %6.1f us per 105 MB = %5.2f TB/s\n", names[p], grid, ms / 24 * 1e3, (double)M * 512 / (ms / 24) / 1e9);
    }
  return 0;
}
