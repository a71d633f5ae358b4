// How fast can a CU fill LDS by LDS-DMA (buffer_load_dwordx4 ... lds), as the GEMM rings do it?  One workgroup per CU,
// NW waves, a ring of NST stages of `rows` 128-B rows; every k-tile reads one 128-B column of `rows` source rows that
// are `pitch` bytes apart (pitch = 128: contiguous; 1024: the bf16 K = 512 operand).  No MFMA, no LDS reads: counted
// vmcnt -> barrier -> issue, exactly the ring's synchronisation.  Source: the `share` workgroups of a group sit on ONE
// XCD (blockIdx % 8) and read the same rows (8 = the N-tiles of one A tile).  `stream` = 1: after every nk k-tiles a
// group moves on to fresh rows (an A operand streamed from HBM, first touch), 0: the same rows again (L2-resident).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/lfr tools/experiments/lds_fill_rate.hip && /tmp/lfr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const void* base, unsigned bytes, void __attribute__((address_space(3)))* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), lds, 16, voff, soff, 0, 0);
#endif
}

template <int NW, int ROWS, int NST>
__global__ __launch_bounds__(64 * NW) void fill_kernel(const char* src, unsigned src_bytes, int pitch, int nk, int share, int iters,
                                                       unsigned long long* cycles, int stream, int arows) {
  constexpr int LPT = ROWS / (8 * NW), STAGE = ROWS * 128;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;     // 32 workgroups per XCD
  const int grp = xcd * (32 / share) + idx / share, ngrp = 8 * (32 / share);
  int voff[LPT];
#pragma unroll
  for (int j = 0; j < LPT; ++j) voff[j] = ((grp * ROWS + (j * NW + wave) * 8 + (lane >> 3)) * pitch + (lane & 7) * 16);
  // rows [0, arows) of a stage are the streamed operand ("A"), the rest the resident one ("W")
  auto issue = [&](int it) {
    const int st = it % NST;
    const int pass = stream ? it / nk : 0;
    const long long a_adv = (long long)pass * ngrp * ROWS * pitch;       // fresh rows every pass
    const int soff = (it % nk) * 128;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const bool is_a = (j * NW + wave) * 8 < arows;
      const int so = soff + (is_a ? (int)(a_adv % (long long)(src_bytes / 2)) : 0);
      dma16(src, src_bytes, (void __attribute__((address_space(3)))*)(ring + st * STAGE + (j * NW + wave) * 8 * 128), voff[j], so);
    }
  };
  for (int t = 0; t < NST - 1; ++t) issue(t);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    wait_vmcnt<(NST - 2) * LPT>();
    __builtin_amdgcn_s_barrier();
    issue(it + NST - 1);
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NW, int ROWS, int NST>
void run(const char* src, size_t src_bytes, int pitch, int nk, int share, unsigned long long* cyc, int stream = 0, int arows = 0) {
  auto kern = fill_kernel<NW, ROWS, NST>;
  const size_t lds = (size_t)NST * ROWS * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 2000, grid = 256;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, src, (unsigned)src_bytes, pitch, nk, share, iters, cyc, stream, arows);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, src, (unsigned)src_bytes, pitch, nk, share, iters, cyc, stream, arows);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * iters * ROWS * 128;
  printf("waves %d  stage %3d rows (%2d KB) x %d stages  pitch %5d  nk %3d  share %2d  streamed rows %3d : %6.1f GB/s per CU, %5.2f TB/s chip, %5.2f us per stage\n",
         NW, ROWS, ROWS * 128 / 1024, NST, pitch, nk, share, stream ? arows : 0, bytes / ms / 1e6 / 256, bytes / ms / 1e9, ms * 1e3 / iters);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  char* src; unsigned long long* cyc;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&cyc, 4096 * 8);
  // all resident (L2): contiguous vs 1 KB pitch, sharing 1 / 8
  for (int share : {1, 8}) for (int pitch : {128, 1024}) run<8, 384, 3>(src, bytes, pitch, pitch >= 1024 ? 8 : 1, share, cyc);
  // the 256x128 bf16 GEMM's mix: 256 streamed A rows shared by 8 workgroups + 128 resident W rows
  run<8, 384, 3>(src, bytes, 1024, 8, 8, cyc, 1, 256);
  run<8, 384, 3>(src, bytes, 1024, 8, 1, cyc, 1, 256);
  run<8, 384, 3>(src, bytes, 1024, 8, 8, cyc, 1, 384);
  run<8, 192, 6>(src, bytes, 1024, 8, 8, cyc, 1, 128);
  run<8, 128, 8>(src, bytes, 1024, 8, 8, cyc, 1, 64);
  run<4, 256, 3>(src, bytes, 1024, 8, 8, cyc, 1, 128);       // 128x128 tile
  return 0;
}
