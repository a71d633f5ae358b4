// Device helpers shared by the fp32 and the bf16 expand + depthwise kernels (pw_dw.hip, pw_dw_bf16.hip): the run
// picker of the depthwise epilogue, the buffer-addressed LDS-DMA request and the swizzle key of a k-tile ring row.
#pragma once
#include "common.h"

namespace {

// Output rows are cut into RS runs per frame / strip so that NQ x WO x runs work items fill the 256 threads evenly.
constexpr int pick_runs(int per_run_items, int rows, int blocks) {
  int best = 1;
  double best_u = 0;
  for (int rs = 1; rs <= 8 && rs <= rows; ++rs) {
    const int items = per_run_items * rs * blocks, rounds = (items + 255) / 256;
    const double u = (double)items / (256.0 * rounds);
    if (u > best_u + 0.02) best = rs, best_u = u;
  }
  return best;
}

__device__ __forceinline__ void ft_dma16(const void* base, unsigned bytes, void* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (void __attribute__((address_space(3)))*)lds, 16, voff, soff, 0, 0);
#endif
}

// swizzle key of stage row r: 128-B rows: (r >> 1) & 7 (gemm.hip); 64-B rows: a 4-entry table by (r >> 2) & 3 chosen so
// that the sixteen rows of a ds_read_b128 service group fall on sixteen different 16-B bank slots
template <int KF>
__device__ __forceinline__ int ft_key(int r) {
  if constexpr (KF == 32) return (r >> 1) & 7;
  else return (0x1320 >> (4 * ((r >> 2) & 3))) & 3;      // {0, 2, 3, 1}
}

}  // namespace
