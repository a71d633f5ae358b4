// What does one k-tile of the fp32 ring GEMM cost, piece by piece?  A stripped main loop (no global
// traffic except the optional LDS-DMA, no epilogue): 4 waves, per-wave 32*TM x 32*TN outputs, BK = 32 floats.
//   flags:  R = ds_read_b128 fragment reads (software-pipelined one column pair ahead)
//           B = one s_barrier per k-tile
//           D = LDS-DMA of the next k-tile issued by the MFMA waves, between the MFMA groups
//           L = ... issued by four extra loader waves instead (512-thread workgroup)
// Reports shader cycles per k-tile per wave (s_memtime) against the MFMA-only floor 64 * 4*TM*TN*4 / ... .
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/gla tools/experiments/gemm_loop_ablate.hip && /tmp/gla
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROWB = 128;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, bool R, bool B, bool D, bool L, int WPB, int NST>
__global__ __launch_bounds__(L ? 512 : 256) void loop_kernel(const float* __restrict__ src, float* sink,
                                                             unsigned long long* cycles, int nk, int shared_src) {
  constexpr int BM = 64 * TM, BN = 64 * TN, ROWS = BM + BN, STAGE = ROWS * ROWB, LPT = ROWS / 32;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // fill the ring once (any data)
  for (int i = tid; i < NST * STAGE / 16; i += blockDim.x)
    reinterpret_cast<f32x4*>(ring)[i] = f32x4{(float)(i & 15) * 0.125f - 1.f, 0.5f, -0.25f, (float)(tid & 7) * 0.1f};
  __syncthreads();
  const int lrow8 = lane >> 3, lcol = lane & 7;
  auto dma = [&](int kt, int lw, int j0, int j1) {   // pieces [j0, j1) of loader-wave lw for k-tile kt
    char* st = ring + (kt % NST) * STAGE;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      if (j < j0 || j >= j1) continue;
      const int r = (j * 4 + lw) * 8 + lrow8;
      const float* p = src + ((size_t)(shared_src ? (blockIdx.x & 7) : blockIdx.x) * ROWS + r) * 1024 + (size_t)(kt & 31) * 32 + (lcol ^ ((r >> 1) & 7)) * 4;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p,
                                       (void __attribute__((address_space(3)))*)(st + (j * 4 + lw) * 8 * ROWB), 16, 0, 0);
    }
  };
  if (L && wave >= 4) {   // loader waves
    for (int t = 0; t < NST - 1; ++t) dma(t, wave - 4, 0, LPT);
    for (int kt = 0; kt < nk; ++kt) {
      wait_vmcnt<(NST - 3) * LPT>();
      if (B) __builtin_amdgcn_s_barrier();
      dma(kt + NST - 1, wave - 4, 0, LPT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, kh = lane >> 5;
  int a_off[TM], a_key[TM], b_off[TN], b_key[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (BM / 2) + i * 32 + r32;
    a_off[i] = r * ROWB;
    a_key[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = BM + wn * (BN / 2) + j * 32 + r32;
    b_off[j] = r * ROWB;
    b_key[j] = (r >> 1) & 7;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa0[i] = fa1[i] = f32x4{0.1f * lane, 0.2f, 0.3f, 0.4f};
#pragma unroll
  for (int j = 0; j < TN; ++j) fb0[j] = fb1[j] = f32x4{0.5f, 0.01f * lane, 0.7f, 0.8f};
  auto frag = [&](const char* st, int g, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    if (!R) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + (((2 * g + kh) ^ a_key[i]) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + (((2 * g + kh) ^ b_key[j]) << 4));
  };
  auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
  };
  if (D) for (int t = 0; t < NST - 1; ++t) dma(t, wave, 0, LPT);
  frag(ring, 0, fa0, fb0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int kt = 0; kt < nk; ++kt) {
    if (D) wait_vmcnt<(NST - 3) * LPT>();
    if (B) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* st = ring + (kt % NST) * STAGE;
    const char* st_next = ring + ((kt + 1) % NST) * STAGE;
    frag(st, 1, fa1, fb1);
    if (D) dma(kt + NST - 1, wave, 0, LPT / 4);
    mma(fa0, fb0);
    frag(st, 2, fa0, fb0);
    if (D) dma(kt + NST - 1, wave, LPT / 4, LPT / 2);
    mma(fa1, fb1);
    frag(st, 3, fa1, fb1);
    if (D) dma(kt + NST - 1, wave, LPT / 2, 3 * LPT / 4);
    mma(fa0, fb0);
    frag(st_next, 0, fa0, fb0);
    if (D) dma(kt + NST - 1, wave, 3 * LPT / 4, LPT);
    mma(fa1, fb1);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[tid] = s;
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int TM, int TN, bool R, bool B, bool D, bool L, int WPB, int NST = 3>
void run(const char* tag, const float* src, float* sink, unsigned long long* cyc, int shared_src = 0) {
  constexpr int STAGE = (64 * TM + 64 * TN) * ROWB;
  constexpr size_t lds = (size_t)NST * STAGE;
  auto kern = loop_kernel<TM, TN, R, B, D, L, WPB, NST>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nk = 256, grid = 256 * WPB;
  for (int rep = 0; rep < 60; ++rep)   // ~warm clock
    hipLaunchKernelGGL(kern, dim3(grid), dim3(L ? 512 : 256), lds, 0, src, sink, cyc, nk, shared_src);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, grid * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per = (double)h[h.size() / 2] / nk, floor_c = 64.0 * 16 * TM * TN;
  printf("%-34s NST %d src %s tile %3dx%-3d  WG/CU %d : %7.0f cycles per k-tile per wave  (MFMA floor %5.0f x %d WG = %5.0f) -> %5.1f %%\n", tag, NST, shared_src ? "L2 " : "HBM", 64 * TM,
         64 * TN, WPB, per, floor_c, WPB, floor_c * WPB, 100.0 * floor_c * WPB / per);
}

int main() {
  float *src, *sink;
  unsigned long long* cyc;
  hipMalloc(&src, (size_t)1024 * 384 * 1024 * sizeof(float));   // 1024 workgroups x 384 rows x 1024 floats (1.5 GB)
  hipMemset(src, 0, (size_t)1024 * 384 * 1024 * sizeof(float));
  hipMalloc(&sink, 4096);
  hipMalloc(&cyc, 4096 * sizeof(unsigned long long));
#define ALL(TM, TN, W)                                                            \
  run<TM, TN, false, false, false, false, W>("MFMA only", src, sink, cyc);        \
  run<TM, TN, true, false, false, false, W>("+ LDS fragment reads", src, sink, cyc); \
  run<TM, TN, true, true, false, false, W>("+ barrier", src, sink, cyc);
#define DMA(TM, TN, W, N)                                                                           \
  run<TM, TN, true, true, true, false, W, N>("+ LDS-DMA by the MFMA waves", src, sink, cyc, 0);     \
  run<TM, TN, true, true, true, false, W, N>("+ LDS-DMA by the MFMA waves", src, sink, cyc, 1);     \
  run<TM, TN, true, true, false, true, W, N>("+ LDS-DMA by loader waves", src, sink, cyc, 0);       \
  run<TM, TN, true, true, false, true, W, N>("+ LDS-DMA by loader waves", src, sink, cyc, 1);
  ALL(2, 1, 1)
  DMA(2, 1, 1, 3)
  DMA(2, 1, 1, 4)
  DMA(2, 1, 1, 6)
  ALL(2, 1, 2)
  DMA(2, 1, 2, 3)
  ALL(2, 2, 1)
  DMA(2, 2, 1, 3)
  DMA(2, 2, 1, 4)
  ALL(1, 1, 1)
  DMA(1, 1, 1, 3)
  DMA(1, 1, 1, 6)
  return 0;
}
