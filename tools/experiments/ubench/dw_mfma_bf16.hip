// Round 6, VERDICT r5 #3 step 1: the depthwise 3x3 phase (P2) of ir_fused_bf16_kernel on the matrix pipe.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/ubench/dw_mfma_bf16.hip -o /tmp/dw_mfma && /tmp/dw_mfma
//
// One workgroup = the fused block's 8 x 16 output tile of a 32-channel chunk (up4.1 / up3.1: 64 expanded channels = two
// chunks): E [10][18][32] bf16 in LDS in the kernel's own layout (64-B pixels, 16-B columns XOR-keyed by hx, rows padded
// by 16 B), taps [9][32] + bd fp32 in LDS; a wave owns output rows 2 w, 2 w + 1 (16 pixels each).  Per "chunk" iteration a
// wave runs P2 (depthwise + bias + LeakyReLU -> D as bf16 in registers) and P3's four project MFMAs on D, exactly the
// instruction mix between the shipped kernel's two barriers.  Three P2 variants:
//   MODE 0  shipped: widen bf16 -> fp32, 9 packed FMAs per channel pair on the VALU (csrc/ir_fused.hip P2)
//   MODE 1  v_mfma_f32_16x16x32_bf16 with a BLOCK-DIAGONAL A operand: output tile = 16 channels x 16 pixels,
//           K = 2 taps x 16 channels; lane (q, n) supplies B = ONE 16-byte read of E (8 channels of pixel n + tap);
//           A[m][(tap, c')] = wd[tap][16 h + m] * delta(c', m), built per chunk from the fp32 taps in LDS
//           (1 ds_read_b32 + 1 cvt + 4 v_and per fragment); 9 taps -> 5 MFMAs per tile, bias = accumulator init
//   MODE 2  the same with the ten A fragments of a chunk prebuilt (what a pack-time image + LDS-DMA would give):
//           one ds_read_b128 per fragment, no vector arithmetic
// and the error of each variant's D against float64 on the same bf16 E and fp32 taps.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int IH = 10, IW = 18, EROWB = IW * 64 + 16, EBYTES = IH * EROWB;
constexpr int oTAPS = (EBYTES + 15) / 16 * 16;          // [9][32] taps, [32] b1 (unused here), [32] bd : fp32
constexpr int oAIMG = oTAPS + 11 * 32 * 4;              // MODE 2: [2 h][5 u][64 lanes] x 16 B
constexpr int oW2 = oAIMG + 10 * 64 * 16;               // [32 out][32 k] bf16 project weights (P3)
constexpr int LDS_BYTES = oW2 + 32 * 64;

// byte offset of 16-B column `col` of halo pixel (hy, hx): csrc/ir_common.h e_off<1, 16, IW>() x 4
__device__ __host__ inline int e_byte(int hy, int hx, int col) { return hy * EROWB + hx * 64 + ((col ^ ((hx >> 1) & 3)) << 4); }

__device__ __forceinline__ f32x4 mfma16b(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x4 lrelu4(f32x4 v) {
  const f32x4 s = v * 0.01f;
  return f32x4{vmax_raw(v.x, s.x), vmax_raw(v.y, s.y), vmax_raw(v.z, s.z), vmax_raw(v.w, s.w)};
}

// dump [wave][j][lane][8] floats: the eight D values a lane hands to P3 (bf16, widened), in the lane's K-slot order
template <int MODE>
__global__ __launch_bounds__(256, 4) void dw_kernel(const bf16_t* __restrict__ e_img, const float* __restrict__ taps,
                                                   const bf16_t* __restrict__ w2, const u32x4* __restrict__ a_img, int iters,
                                                   float* __restrict__ dump, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sE = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
  for (int i = tid; i < EBYTES / 16; i += 256) reinterpret_cast<u32x4*>(sE)[i] = reinterpret_cast<const u32x4*>(e_img)[i];
  for (int i = tid; i < 11 * 32; i += 256) reinterpret_cast<float*>(smem + oTAPS)[i] = taps[i];
  for (int i = tid; i < 10 * 64; i += 256) reinterpret_cast<u32x4*>(smem + oAIMG)[i] = a_img[i];
  for (int i = tid; i < 32 * 64 / 16; i += 256) reinterpret_cast<u32x4*>(smem + oW2)[i] = reinterpret_cast<const u32x4*>(w2)[i];
  __syncthreads();
  const float* wf = reinterpret_cast<const float*>(smem + oTAPS);
  constexpr int NPX = 2;
  f32x4 acc3[NPX][2];
  for (int i = 0; i < NPX; ++i) acc3[i][0] = acc3[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // MODE 0 addressing: three tap-column bases of this lane's first pixel (row 2 w, column l15), 16-B column q
  const char* ebk[3];
  for (int kx = 0; kx < 3; ++kx) ebk[kx] = sE + e_byte(NPX * wave, l15 + kx, q);
  // MODE 1 / 2 addressing: tap pair u -> this lane's tap t = 2 u + (q >> 1) (tap 9 does not exist: weight 0, address of tap 8),
  // 16-B column 2 h + (q & 1); bases for h = 0 and 1
  const char* ebu[5][2];
  for (int u = 0; u < 5; ++u) {
    int t = 2 * u + (q >> 1);
    t = t > 8 ? 8 : t;
    const int ky = t / 3, kx = t - 3 * ky;
    for (int h = 0; h < 2; ++h) ebu[u][h] = sE + e_byte(NPX * wave + ky, l15 + kx, 2 * h + (q & 1));
  }
  // MODE 1: lane masks of the block-diagonal fragment.  Lane (q, m = l15) holds K-slots j = 0..7 = channels 8 (q & 1) + j of
  // its tap; the only non-zero one is channel m: dword (m & 7) >> 1, half m & 1 -- when m >> 3 == q & 1
  unsigned amask[4];
  {
    const bool act = (l15 >> 3) == (q & 1);
    for (int d = 0; d < 4; ++d) amask[d] = act && d == ((l15 & 7) >> 1) ? ((l15 & 1) ? 0xFFFF0000u : 0x0000FFFFu) : 0u;
  }
  const int tap_hi = q >> 1;

#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
    bf16x8 fd[NPX];
    if constexpr (MODE == 0) {
      const int c8 = 8 * q;
      f32x4 a0[NPX], a1[NPX];
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        a0[j] = *reinterpret_cast<const f32x4*>(wf + 10 * 32 + c8);
        a1[j] = *reinterpret_cast<const f32x4*>(wf + 10 * 32 + c8 + 4);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        f32x4 w0[3], w1v[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          w0[ky] = *reinterpret_cast<const f32x4*>(wf + (ky * 3 + kx) * 32 + c8);
          w1v[ky] = *reinterpret_cast<const f32x4*>(wf + (ky * 3 + kx) * 32 + c8 + 4);
        }
#pragma unroll
        for (int r = 0; r < NPX + 2; ++r) {
          const bf16x8 e = *reinterpret_cast<const bf16x8*>(ebk[kx] + r * EROWB);
          const f32x4 elo = {(float)e[0], (float)e[1], (float)e[2], (float)e[3]}, ehi = {(float)e[4], (float)e[5], (float)e[6], (float)e[7]};
#pragma unroll
          for (int j = 0; j < NPX; ++j) {
            const int ky = r - j;
            if (ky >= 0 && ky < 3) {
              a0[j] += elo * w0[ky];
              a1[j] += ehi * w1v[ky];
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        const f32x4 l0 = lrelu4(a0[j]), l1 = lrelu4(a1[j]);
        const bf16x4 h0 = __builtin_convertvector(l0, bf16x4), h1 = __builtin_convertvector(l1, bf16x4);
        fd[j] = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      }
    } else {
      f32x4 acc[NPX][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(wf + 10 * 32 + 16 * h + 4 * q);   // rows 4 q .. + 3 = channels 16 h + 4 q + i
#pragma unroll
        for (int j = 0; j < NPX; ++j) acc[j][h] = bias;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          bf16x8 fa;
          if constexpr (MODE == 1) {
            float w = wf[(2 * u + tap_hi) * 32 + 16 * h + l15];          // (u = 4, tap_hi = 1: row 9 is not a tap)
            if (u == 4) w = tap_hi ? 0.f : w;
            const bf16x4 wb = __builtin_convertvector(f32x4{w, w, 0.f, 0.f}, bf16x4);
            const unsigned ww = __builtin_bit_cast(unsigned, bf16x2{wb[0], wb[1]});
            fa = __builtin_bit_cast(bf16x8, u32x4{ww & amask[0], ww & amask[1], ww & amask[2], ww & amask[3]});
          } else {
            fa = *reinterpret_cast<const bf16x8*>(smem + oAIMG + ((h * 5 + u) * 64 + lane) * 16);
          }
#pragma unroll
          for (int j = 0; j < NPX; ++j) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ebu[u][h] + j * EROWB);
            acc[j][h] = mfma16b(fa, fb, acc[j][h]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        const f32x4 l0 = lrelu4(acc[j][0]), l1 = lrelu4(acc[j][1]);
        const bf16x4 h0 = __builtin_convertvector(l0, bf16x4), h1 = __builtin_convertvector(l1, bf16x4);
        fd[j] = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};   // K-slot s = channel 16 (s >> 2) + 4 q + (s & 3)
      }
    }
    // P3: four project MFMAs on D (W2c rows as the A operand, one 16-B read each)
    {
      bf16x8 fb[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(smem + oW2 + (16 * n + l15) * 64 + ((q ^ (l15 & 3)) << 4));
#pragma unroll
      for (int i = 0; i < NPX; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc3[i][n] = mfma16b(fb[n], fd[i], acc3[i][n]);
    }
    if (dump && it == 0) {
#pragma unroll
      for (int j = 0; j < NPX; ++j)
#pragma unroll
        for (int s = 0; s < 8; ++s) dump[(((size_t)blockIdx.x * 4 + wave) * NPX + j) * 512 + lane * 8 + s] = (float)fd[j][s];
    }
  }
  float s = 0.f;
  for (int i = 0; i < NPX; ++i)
    for (int n = 0; n < 2; ++n) s += acc3[i][n].x + acc3[i][n].y + acc3[i][n].z + acc3[i][n].w;
  if (s == 123.456f) sink[tid] = s;
}

static float bf16_round(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  u &= 0xFFFF0000u;
  memcpy(&v, &u, 4);
  return v;
}
static unsigned short bf16_bits(float v) {
  v = bf16_round(v);
  unsigned u;
  memcpy(&u, &v, 4);
  return (unsigned short)(u >> 16);
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  srand(7);
  auto rnd = [] { return (float)(rand() / (double)RAND_MAX * 2.0 - 1.0); };
  // E: LReLU-like activations (mostly positive, some small negatives), bf16; taps ~ the recipe's scale; bd
  std::vector<float> e(IH * IW * 32), taps(11 * 32);
  for (auto& v : e) {
    const float x = rnd() * 1.5f;
    v = bf16_round(x > 0 ? x : 0.01f * x);
  }
  for (int i = 0; i < 9 * 32; ++i) taps[i] = rnd() * 0.6f;
  for (int i = 9 * 32; i < 11 * 32; ++i) taps[i] = rnd() * 0.3f;
  std::vector<unsigned short> eimg(EBYTES / 2, 0), w2(32 * 32);
  for (int hy = 0; hy < IH; ++hy)
    for (int hx = 0; hx < IW; ++hx)
      for (int c = 0; c < 32; ++c) eimg[(e_byte(hy, hx, c >> 3) >> 1) + (c & 7)] = bf16_bits(e[(hy * IW + hx) * 32 + c]);
  for (auto& v : w2) v = bf16_bits(rnd() * 0.2f);
  // MODE 2's prebuilt fragments: [h][u][lane] 8 bf16
  std::vector<unsigned short> aimg(10 * 64 * 8, 0);
  for (int h = 0; h < 2; ++h)
    for (int u = 0; u < 5; ++u)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, q = lane >> 4, t = 2 * u + (q >> 1);
        if (t < 9 && (m >> 3) == (q & 1)) aimg[((h * 5 + u) * 64 + lane) * 8 + (m & 7)] = bf16_bits(taps[t * 32 + 16 * h + m]);
      }
  // float64 reference of D on the 8 x 16 tile, and the same with bf16-rounded taps (what the matrix pipe multiplies by)
  std::vector<double> ref(8 * 16 * 32);
  for (int oy = 0; oy < 8; ++oy)
    for (int ox = 0; ox < 16; ++ox)
      for (int c = 0; c < 32; ++c) {
        double s = taps[10 * 32 + c];
        for (int t = 0; t < 9; ++t) s += (double)taps[t * 32 + c] * e[((oy + t / 3) * IW + ox + t % 3) * 32 + c];
        ref[(oy * 16 + ox) * 32 + c] = s > 0 ? s : 0.01 * s;
      }

  bf16_t *d_e, *d_w2;
  float *d_taps, *d_dump, *d_sink;
  u32x4* d_aimg;
  const int nwg = 256 * 4 * 4;
  CK(hipMalloc(&d_e, EBYTES));
  CK(hipMalloc(&d_w2, 32 * 64));
  CK(hipMalloc(&d_taps, 11 * 32 * 4));
  CK(hipMalloc(&d_aimg, 10 * 64 * 16));
  CK(hipMalloc(&d_dump, 4 * 2 * 512 * 4));
  CK(hipMalloc(&d_sink, 1024));
  CK(hipMemcpy(d_e, eimg.data(), EBYTES, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w2, w2.data(), 32 * 64, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_taps, taps.data(), 11 * 32 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_aimg, aimg.data(), 10 * 64 * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"VALU (shipped P2)", "MFMA, A built per chunk", "MFMA, A prebuilt (LDS image)"};
  double base_ms = 0;
  printf("workgroup = 8x16 output pixels x 32 channels per chunk; %d workgroups (4 per CU), LDS %d B\n", nwg, LDS_BYTES);
  for (int mode = 0; mode < 3; ++mode) {
    auto launch = [&](int grid, int iters, float* dump) {
      if (mode == 0) hipLaunchKernelGGL(dw_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, 0, d_e, d_taps, d_w2, d_aimg, iters, dump, d_sink);
      if (mode == 1) hipLaunchKernelGGL(dw_kernel<1>, dim3(grid), dim3(256), LDS_BYTES, 0, d_e, d_taps, d_w2, d_aimg, iters, dump, d_sink);
      if (mode == 2) hipLaunchKernelGGL(dw_kernel<2>, dim3(grid), dim3(256), LDS_BYTES, 0, d_e, d_taps, d_w2, d_aimg, iters, dump, d_sink);
    };
    // correctness: one workgroup, one chunk, D dumped in the lane's K-slot order
    launch(1, 1, d_dump);
    CK(hipDeviceSynchronize());
    std::vector<float> dump(4 * 2 * 512);
    CK(hipMemcpy(dump.data(), d_dump, dump.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, esum = 0, rsum = 0;
    for (int wave = 0; wave < 4; ++wave)
      for (int j = 0; j < 2; ++j)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 8; ++s) {
            const int q = lane >> 4, ox = lane & 15, oy = 2 * wave + j;
            const int c = mode == 0 ? 8 * q + s : 16 * (s >> 2) + 4 * q + (s & 3);
            const double d = fabs((double)dump[((wave * 2 + j) * 64 + lane) * 8 + s] - ref[(oy * 16 + ox) * 32 + c]);
            emax = d > emax ? d : emax;
            esum += d;
            rsum += fabs(ref[(oy * 16 + ox) * 32 + c]);
          }
    // timing: 128 chunks per workgroup, best of 5
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      launch(nwg, 128, nullptr);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    if (mode == 0) base_ms = best;
    const double chunks_per_simd = (double)nwg / 256 * 128;   // a workgroup's 4 waves sit on the CU's 4 SIMDs: one wave-chunk per SIMD, workgroup and chunk
    printf("%-30s %.3f ms  %.0f ns = %.0f cycles at 2.4 GHz per wave-chunk and SIMD  (%.2fx)   "
           "D error vs float64: max %.3e  mean %.3e (mean |D| %.3e)\n",
           names[mode], best, best * 1e6 / chunks_per_simd, best * 1e6 / chunks_per_simd * 2.4, base_ms / best, emax, esum / 4096,
           rsum / 4096);
  }
  return 0;
}
