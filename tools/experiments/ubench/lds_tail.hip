// Round 6 (profiles/r6_two_models.txt section 9): does LDS read data reach the registers LATE for part of a wave when another
// wave of the SIMD runs matrix instructions?  The debug build of the failing fused Up block showed: accumulators right, G taps right
// when stored a few instructions later, weights right -- and ONE component of the sum wrong on lanes 48..63 only, i.e. a vector
// instruction issued right behind `s_waitcnt lgkmcnt(n)` computed with something else than what the LDS read delivered.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/ubench/lds_tail.hip -o /tmp/lds_tail && /tmp/lds_tail
//
// VICTIM: each lane reads 16 B from LDS into the SAME registers every iteration, alternately from two images with known,
// different contents, waits (lgkmcnt) and consumes the value in the next instruction (an FMA with an exact result); a consumer that
// sees the previous iteration's register contents -- or anything else -- is counted, with the lanes it happened on.
//   form 0  one ds_read_b128, s_waitcnt lgkmcnt(0), FMAs
//   form 1  four ds_read_b128 in flight, consumed behind lgkmcnt(3) / (2) / (1) / (0) as the kernel does
//   form 2  as 1 with a ds_write_b128 in front of the reads (the kernel's E store)
// CO-RUNNER on a second stream: a register-only loop of one matrix instruction (or packed FMAs, or nothing).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// value of dword e of 16-B slot s of image img: exact small integers
__device__ __forceinline__ float val(int img, int s, int e) { return (float)((img ? 4096 : 0) + 4 * s + e); }

template <int FORM>
__global__ __launch_bounds__(256, 4) void victim(unsigned* counts, unsigned long long* lanes_hit, int iters) {
  __shared__ f32x4 img[2][4][256 + 8];
  __shared__ f32x4 scratch[256];
  const int tid = threadIdx.x;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 2; ++i) img[i][k][tid] = f32x4{val(i, 256 * k + tid, 0), val(i, 256 * k + tid, 1), val(i, 256 * k + tid, 2), val(i, 256 * k + tid, 3)};
  __syncthreads();
  unsigned bad = 0;
  const float w = 2.f;
  for (int it = 0; it < iters; ++it) {
    const int im = it & 1, slot = (tid + 17 * it) & 255;       // every lane its own 16 B, a different slot every iteration
    if constexpr (FORM == 0) {
      const f32x4 r = img[im][0][slot];
      const f32x4 x = r * w + 1.f;                                // consumed right behind the wait
#pragma unroll
      for (int e = 0; e < 4; ++e) bad += x[e] != val(im, slot, e) * w + 1.f;
    } else {
      if constexpr (FORM == 2) scratch[tid] = f32x4{(float)it, 1.f, 2.f, 3.f};
      f32x4 r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = img[im][k][slot];
      f32x4 acc = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = r[k] * w + acc;           // the compiler waits lgkmcnt(3), (2), (1), (0)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float expect = 1.f;
        for (int k = 0; k < 4; ++k) expect = val(im, 256 * k + slot, e) * w + expect;
        bad += acc[e] != expect;
      }
    }
    asm volatile("" ::: "memory");
  }
  if (bad) {
    atomicAdd(&counts[0], bad);
    atomicOr(lanes_hit, 1ull << (tid & 63));
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void burn_kernel(float* sink, int iters) {
  const unsigned t = threadIdx.x + 1;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)(0.001f * ((t * 7 + j) % 13) - 0.006f), b[j] = (__bf16)(0.001f * ((t * 5 + j) % 11) - 0.005f);
  const float fa = 0.001f * (t % 17), fb = 0.002f * (t % 5);
  f32x16 c16[2] = {};
  f32x4 c4[4] = {};
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {
      c16[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[0], 0, 0, 0);
      c16[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c16[1], 0, 0, 0);
    } else if constexpr (KIND == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[k], 0, 0, 0);
    } else if constexpr (KIND == 2) {
      c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c16[0], 0, 0, 0);
      c16[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, c16[1], 0, 0, 0);
    } else if constexpr (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c4[k], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) c4[k] = c4[k] * fa + fb;
    }
  }
  float s = 0;
  for (int k = 0; k < 16; ++k) s += c16[0][k] + c16[1][k];
  for (int k = 0; k < 4; ++k) s += c4[k][0] + c4[k][3];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

static void launch_victim(int form, hipStream_t s, unsigned* counts, unsigned long long* lanes, int iters, int blocks) {
  switch (form) {
    case 0: hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, s, counts, lanes, iters); break;
    case 1: hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, s, counts, lanes, iters); break;
    default: hipLaunchKernelGGL(victim<2>, dim3(blocks), dim3(256), 0, s, counts, lanes, iters); break;
  }
}
static void launch_burn(int kind, hipStream_t s, float* sink, int iters, int blocks) {
  switch (kind) {
    case 0: hipLaunchKernelGGL(burn_kernel<0>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 1: hipLaunchKernelGGL(burn_kernel<1>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 2: hipLaunchKernelGGL(burn_kernel<2>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    case 3: hipLaunchKernelGGL(burn_kernel<3>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    default: hipLaunchKernelGGL(burn_kernel<4>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100, iters = argc > 2 ? atoi(argv[2]) : 4000, blocks = 1024;
  static const char* form_name[3] = {"ds_read_b128, lgkmcnt(0), FMA", "4 x ds_read_b128, lgkmcnt(3..0), FMAs", "ds_write_b128 + 4 x ds_read_b128, FMAs"};
  static const char* burn_name[6] = {"32x32x16_bf16", "16x16x32_bf16", "32x32x2_f32", "16x16x4_f32", "pk_fma only", "nothing"};
  float* sink;
  unsigned* counts;
  unsigned long long* lanes;
  CHECK(hipMalloc(&sink, 1024));
  CHECK(hipMalloc(&counts, 16));
  CHECK(hipMalloc(&lanes, 8));
  hipStream_t sv, sb;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%d launches of %d workgroups x 256 lanes x %d iterations per cell; cells: launches with a mismatch / mismatching values (lanes hit, hex mask)\n", launches, blocks, iters);
  printf("%-42s", "victim form \\ co-runner");
  for (int k = 0; k < 6; ++k) printf(" %22s", burn_name[k]);
  printf("\n");
  for (int form = 0; form < 3; ++form) {
    printf("%-42s", form_name[form]);
    for (int kind = 0; kind < 6; ++kind) {
      launch_victim(form, sv, counts, lanes, iters, blocks);
      CHECK(hipStreamSynchronize(sv));
      CHECK(hipEventRecord(e0, sv));
      launch_victim(form, sv, counts, lanes, iters, blocks);
      CHECK(hipEventRecord(e1, sv));
      CHECK(hipStreamSynchronize(sv));
      float vms = 0, bms = 0;
      CHECK(hipEventElapsedTime(&vms, e0, e1));
      if (kind < 5) {
        CHECK(hipEventRecord(e0, sb));
        launch_burn(kind, sb, sink, 20000, blocks);
        CHECK(hipEventRecord(e1, sb));
        CHECK(hipStreamSynchronize(sb));
        CHECK(hipEventElapsedTime(&bms, e0, e1));
      }
      int bad_launches = 0;
      unsigned long long bad_items = 0, mask = 0;
      for (int l = 0; l < launches; ++l) {
        CHECK(hipMemsetAsync(counts, 0, 16, sv));
        CHECK(hipMemsetAsync(lanes, 0, 8, sv));
        CHECK(hipStreamSynchronize(sv));
        if (kind < 5) {
          const int nb = (int)(3.0f * vms / (bms > 0.01f ? bms : 0.01f)) + 2;
          for (int b = 0; b < nb; ++b) launch_burn(kind, sb, sink, 20000, blocks);
        }
        launch_victim(form, sv, counts, lanes, iters, blocks);
        CHECK(hipStreamSynchronize(sv));
        unsigned h[4];
        unsigned long long m;
        CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&m, lanes, 8, hipMemcpyDeviceToHost));
        if (h[0]) ++bad_launches, bad_items += h[0], mask |= m;
        CHECK(hipStreamSynchronize(sb));
      }
      char cell[80];
      if (bad_launches) snprintf(cell, sizeof cell, "%d / %llu (%llx)", bad_launches, bad_items, mask);
      else snprintf(cell, sizeof cell, "0");
      printf(" %22s", cell);
      fflush(stdout);
    }
    printf("\n");
  }
  return 0;
}
