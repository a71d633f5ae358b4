// Expand 1x1 conv + BN + LeakyReLU + depthwise 3x3 + BN + LeakyReLU in ONE kernel for the bf16 engine (BASELINE
// configs[2]; reference module/unet.py:17-30 with BN folded) -- the inverted residuals of the low-resolution stages
// (10x10, 16x16, 20x20 as whole-frame tiles, 40x40 as row strips), the bf16 counterpart of pw_dw.hip:
//
//   D[b, oy, ox, n] = lrelu( sum_taps wd[tap][n] * E[b, oy*s + ky - 1, ox*s + kx - 1, n] + bd[n] ),
//   E[b, y, x, n]   = lrelu( A[b, y, x, :] . W1[n, :] + b1[n] )                    (zero outside the frame)
//
// Why it exists (round 5).  In bf16 HBM governs every stage, and the un-fused chain moves the expanded
// tensor E -- the widest tensor of the block, 2 x Cin channels -- through HBM twice: the expand GEMM writes it, the
// depthwise kernel reads it back (30 launches, 2.0 ms of a 14.5 ms step at B = 512, profiles/r4_launch_table_bf16_b512.txt).
// Here the GEMM's output tile is WHOLE FRAMES (or an 8-row strip of a 40x40 frame with its one-row halo recomputed) x BN
// channels, so the 3x3 neighbourhood of every output is inside the tile and the depthwise conv runs on the tile while it
// sits in LDS: the block reads A once and writes D once.  Round 3 tried this with the fp32 kernel's 32-channel tiles and
// an fp32 E image and lost 2 % end to end: a 32-channel tile streams its A rows through LDS N / 32 times where the
// 128 x 128 GEMM does N / 128 times.  This kernel takes 64 (128) channels per tile with a bf16 E image -- byte for byte
// the LDS footprint of the fp32 kernel's 32-channel tile -- which brings the operand bytes staged per MAC to 1.16 x
// (0.84 x) those of the 128 x 128 GEMM:  (M + BN) / (M * BN)  =  464 / 25,600  against  256 / 16,384.
//
// GEMM part: pw_dw.hip's LDS-DMA ring byte for byte (64-B k-tile rows = 32 bf16 = the K of ONE v_mfma_f32_16x16x32_bf16,
// 16-B columns XOR-swizzled through the source address, two stages, buffer_load ... lds with per-lane constant offsets
// and the k position in an SGPR).  The weight fragment is the MFMA A operand and the pixels the B operand: a lane ends
// up with four consecutive channels of one pixel.  The four waves are NG x MG = (BN / 32) channel groups x pixel groups:
// a wave owns two 16-channel tiles and every MG-th pixel tile, so a k-tile costs it MTW + 2 fragment reads for 2 MTW
// MFMAs (a wave that owned one channel tile and all pixel tiles would be bound by its ds_read_b128s).
//
// Epilogue 1: + b1, LeakyReLU -> bf16 E image in LDS over the dead ring: rows of HW + 2 pixels (a zero column either
// side of the frame), 16-B columns (eight channels) XOR-keyed by the pixel column, plus ONE zero row behind the image.
// Epilogue 2: a thread owns eight channels (their 72 tap weights in registers) and one output column and walks down a
// run of output rows with a ROLLING window: every input row of the run is read from LDS and widened to fp32 once (three
// 16-B reads, 24 shifts / masks) and used by the three output rows it touches; tap rows above / below the frame read
// the zero row.  fp32 accumulate, + bd, LeakyReLU, one 16-B store of eight bf16 channels per output.
#include <stdlib.h>

#include "common.h"
#include "ir_common.h"
#include "pw_dw_common.h"

namespace {

constexpr int KROWB = 64;             // bytes of a k-tile row: 32 bf16
constexpr int RPI = 1024 / KROWB;     // rows one LDS-DMA instruction of a wave fills

__device__ __forceinline__ f32x4 mfma16b(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// The bf16 E image: NR rows of WP = HW + 2 pixels of BN bf16 (a zero column left and right of the frame) + one zero row
// (index NR).  The NC 16-B columns of a pixel are XOR-keyed by its column: the sixteen lanes of a ds_read_b128 service
// group read the same column of 16 / NC ... consecutive pixels x all their 16-B columns = one contiguous window.
template <int HW, int BN, int NR>
struct ETileB {
  static constexpr int WP = HW + 2, PIXB = BN * 2, ROWB = WP * PIXB, NC = PIXB / 16;
  static constexpr int ZROW = NR;
  static constexpr size_t bytes = (size_t)(NR + 1) * ROWB;
  static_assert((NC & (NC - 1)) == 0, "16-B columns per pixel: a power of two");
  // byte offset of 16-B column `c` of the pixel in image row R, column x (-1 .. HW)
  static __device__ __forceinline__ int at(int R, int x, int c) { return (R * WP + x + 1) * PIXB + ((c ^ ((x + 1) & (NC - 1))) << 4); }
};

// GEMM tile of M_ pixel rows x BN channels
template <int M_, int BN>
struct GemmGeomB {
  static constexpr int MT = (M_ + 15) / 16, M_PAD = 16 * MT;
  static constexpr int NT = BN / 16, NG = NT / 2, MG = 4 / NG;   // channel groups (two 16-channel tiles each) x pixel groups
  static constexpr int MTW = (MT + MG - 1) / MG;                 // pixel tiles per wave
  static constexpr int ROWS = M_PAD + BN, LPT = (ROWS + 4 * RPI - 1) / (4 * RPI), STAGE = LPT * 4 * RPI * KROWB;
  static_assert(NG == 1 || NG == 2 || NG == 4, "BN = 32, 64 or 128");
};

// acc[i][j] (i-th pixel tile of this wave, channel tile 2 ng + j) = W1 tile x A rows over the whole K; ends with the
// ring consumed (barrier).  voff[j]: this lane's source offset of the wave's j-th LDS-DMA instruction (rows [0, M_PAD)
// of a stage are A rows, then BN rows of W1).
template <class G>
__device__ __forceinline__ void pw_dw_gemm_b(char* ring, const bf16_t* __restrict__ A, const bf16_t* __restrict__ W1, unsigned a_bytes,
                                             unsigned w_bytes, const int (&voff)[G::LPT], int nk, int wave, int l15, int q,
                                             f32x4 (&acc)[G::MTW][2]) {
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < G::LPT; ++j) {
      char* dst = ring + stage * G::STAGE + (j * 4 + wave) * RPI * KROWB;
      if ((j * 4 + wave) * RPI < G::M_PAD) ft_dma16(A, a_bytes, dst, voff[j], kt * KROWB);
      else ft_dma16(W1, w_bytes, dst, voff[j], kt * KROWB);
    }
  };
  const int ng = wave % G::NG, mg = wave / G::NG;
  const int frag = l15 * KROWB + ((q ^ ft_key<16>(l15)) << 4);   // this lane's 16 B of fragment row 16 t + l15 (16 t adds nothing to the key)
#pragma unroll
  for (int i = 0; i < G::MTW; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Two k-tiles in flight on a two-stage ring: every fragment of k-tile kt is read into registers first, a second barrier
  // says "everyone has read stage kt & 1", the fill of k-tile kt + 2 goes into that stage at once and the MFMAs of k-tile kt
  // run under it, so a fill has two iterations to arrive.  Measured against the fp32 kernel's schedule (k-tile kt + 1 requested
  // at the top of iteration kt): no difference end to end (profiles/r5_ab_bf16_pw_dw.txt) -- with matrix instructions 8 x
  // faster than fp32 the loop is bound by the BYTES its fills move through LDS (hence the 64- / 128-channel tiles), not by
  // their latency.  Kept: it costs nothing and is the safer schedule at low occupancy.
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::LPT) : "memory");   // this wave's part of k-tile kt has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // (k-tile kt + 1 may still be in flight)
    __syncthreads();                                                                  // ... everyone's
    const char* st = ring + (kt & 1) * G::STAGE;
    bf16x8 fw[2], fa[G::MTW];
#pragma unroll
    for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(st + (G::M_PAD + 16 * (2 * ng + j)) * KROWB + frag);
#pragma unroll
    for (int i = 0; i < G::MTW; ++i) {
      // a tile index past the end (odd tile count) recomputes the last tile and never stores it
      const int t = mg + G::MG * i < G::MT ? mg + G::MG * i : G::MT - 1;
      fa[i] = *reinterpret_cast<const bf16x8*>(st + 16 * t * KROWB + frag);
    }
    if (kt + 2 < nk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments are in registers
      __syncthreads();                                     // everyone's: the stage is free
      issue(kt + 2, kt & 1);
    }
#pragma unroll
    for (int i = 0; i < G::MTW; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16b(fw[j], fa[i], acc[i][j]);
  }
  __syncthreads();   // the ring is consumed: it becomes the E image
}

// this lane's source offsets of the wave's LDS-DMA instructions; a_row(r) = the A row (GEMM row index in the whole
// operand) that stage row r < M_PAD shows
template <class G, class RowFn>
__device__ __forceinline__ void dma_offsets(int (&voff)[G::LPT], int wave, int lane, int lda, int K, int n0, int BN, RowFn a_row) {
  constexpr int CPR = KROWB / 16;
  const int lrow = lane / CPR, lcol = lane % CPR;
#pragma unroll
  for (int j = 0; j < G::LPT; ++j) {
    const int r = (j * 4 + wave) * RPI + lrow;             // row inside the stage
    const int cs = lcol ^ ft_key<16>(r);                   // swizzled source column
    if ((j * 4 + wave) * RPI < G::M_PAD) {
      voff[j] = (int)__umul24((unsigned)a_row(r), (unsigned)(lda * 2)) + cs * 16;   // rows < 2^24, row bytes < 2^24; the operand is < 2 GiB (launcher)
    } else {
      const int wr = r - G::M_PAD < BN ? r - G::M_PAD : 0; // rows behind the tile (LPT rounds up) re-read W row 0
      voff[j] = (n0 + wr) * K * 2 + cs * 16;
    }
  }
}

// zero columns (x = -1 and x = HW of the NR image rows) and the zero row of the E image
template <class E, int NR, int HW>
__device__ __forceinline__ void e_zero_border(char* sE, int tid) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < NR * 2 * E::NC; i += 256) {
    const int c = i % E::NC, r2 = i / E::NC, R = r2 >> 1, x = (r2 & 1) ? HW : -1;
    *reinterpret_cast<f32x4*>(sE + E::at(R, x, c)) = z;
  }
  for (int i = tid; i < E::ROWB / 16; i += 256) *reinterpret_cast<f32x4*>(sE + E::ZROW * E::ROWB + i * 16) = z;
}

// the accumulators of one pixel (four consecutive channels of channel tiles 2 ng, 2 ng + 1) -> bf16 -> the E image
template <class E>
__device__ __forceinline__ void e_store(char* sE, int R, int x, int ng, int q, f32x4 v0, f32x4 v1) {
  // channels 16 n + 4 q .. + 3 = half (q & 1) of 16-B column 2 n + (q >> 1)
  *reinterpret_cast<bf16x4*>(sE + E::at(R, x, 4 * ng + (q >> 1)) + 8 * (q & 1)) = __builtin_convertvector(v0, bf16x4);
  *reinterpret_cast<bf16x4*>(sE + E::at(R, x, 4 * ng + 2 + (q >> 1)) + 8 * (q & 1)) = __builtin_convertvector(v1, bf16x4);
}

// ---- the upsampled addend of an Up block's commuted expand conv (common.h GemmEpilogue::ups_src; pw_dw.hip does the same in
// fp32): W1 . cat(up(lo), skip) = up(W1a . lo) + W1b . skip, so the tile's GEMM runs over the skip half only and epilogue 1
// adds the bilinear x2 upsample (align_corners=True) of G = W1a . lo, whose low-resolution rows under the tile are parked
// in LDS behind the ring / E image by LDS-DMA while the GEMM runs: GR rows of HL pixels x 64 channels (128 B per pixel,
// 16 KB), the eight 16-B channel groups of pixel p XOR-keyed by p & 7 on the source side ----
constexpr int kUpsTileBytes = 16384;   // 6 rows x 20 pixels (40x40 strips) or 10 x 10 (20x20 frames) x 128 B, rounded up to 4 issues
template <int HL>
__device__ __forceinline__ void ups_tile_load_b(char* sG, const bf16_t* g, int ld, int gy0, int n_rows, int wave, int lane) {
  // g: this frame's channel slice (64 channels from the workgroup's n0); rows past the tile / frame re-read the last one
#pragma unroll
  for (int j = 0; j < kUpsTileBytes / 4096; ++j) {
    const int piece = (j * 4 + wave) * 64 + lane, p = piece >> 3, ql = piece & 7;
    int gr = p / HL;
    const int gx = p - gr * HL;
    gr = gr < n_rows ? gr : n_rows - 1;
    const int gy = gy0 + gr < HL ? gy0 + gr : HL - 1;
    const bf16_t* src = g + (size_t)(gy * HL + gx) * ld + 8 * (ql ^ (p & 7));
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(sG + (j * 4 + wave) * 1024), 16, 0, 0);
  }
}
// up(G) at high-resolution pixel (y, x): the four low-resolution corners (pixel index in the tile) and their weights, worked
// out once per pixel; then four channels (16-B group c8, half `half`) per call.  Weighted sum of the four corners with fused
// multiply-adds (the fp32 kernels keep ATen's unfused three-lerp form so that every fp32 kernel produces the same bits; at
// bf16 storage precision that buys nothing and costs 18 instead of 8 packed operations per four channels).
struct UpsCorners {
  int p[4];
  float w[4];
};
template <int HW>
__device__ __forceinline__ UpsCorners ups_corners(int gy0, int y, int x) {
  constexpr int HL = HW / 2;
  const UpsTap ty = ups_tap((float)(HL - 1) / (float)(HW - 1), y, HL), tx = ups_tap((float)(HL - 1) / (float)(HW - 1), x, HL);
  UpsCorners c;
  c.p[0] = (ty.i0 - gy0) * HL + tx.i0;
  c.p[1] = (ty.i0 - gy0) * HL + tx.i1;
  c.p[2] = (ty.i1 - gy0) * HL + tx.i0;
  c.p[3] = (ty.i1 - gy0) * HL + tx.i1;
  c.w[0] = ty.l0 * tx.l0;
  c.w[1] = ty.l0 * tx.l1;
  c.w[2] = ty.l1 * tx.l0;
  c.w[3] = ty.l1 * tx.l1;
  return c;
}
__device__ __forceinline__ f32x4 ups_at_lds_b(const char* sG, const UpsCorners& c, int c8, int half) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(sG + c.p[k] * 128 + 16 * (c8 ^ (c.p[k] & 7)) + 8 * half);
    r += c.w[k] * f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  return r;
}

struct Taps8 {          // nine taps + the bias of eight consecutive channels
  f32x4 w[9][2], b[2];
};
__device__ __forceinline__ void load_taps(Taps8& t, const float* __restrict__ wd, const float* __restrict__ bd, int c, int N) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    t.w[k][0] = *reinterpret_cast<const f32x4*>(wd + (size_t)k * N + c);
    t.w[k][1] = *reinterpret_cast<const f32x4*>(wd + (size_t)k * N + c + 4);
  }
  t.b[0] = *reinterpret_cast<const f32x4*>(bd + c);
  t.b[1] = *reinterpret_cast<const f32x4*>(bd + c + 4);
}

// One work item of the depthwise epilogue: 16-B column `c` (eight channels), output column ox, `nrows` (<= RPS) output
// rows downwards.  Input row `rel` of the run (rel = 0 is tap row 0 of the first output) is image row Rfirst + rel and
// frame row yfirst + rel; EDGE: rows outside [0, HW) are the zero row (whole-frame images carry no zero rows of their
// own; strips do).  Rolling window: slot rel % 3 holds input row rel, widened to fp32.
template <class E, int S, int HW, bool EDGE, int RPS>
__device__ __forceinline__ void dw_run_b(const char* sE, int Rfirst, int yfirst, int ox, int c, const Taps8& t, bf16_t* d,
                                         size_t d_row, int nrows) {
  int pk[3], zk[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    pk[kx] = E::at(Rfirst, ox * S + kx - 1, c);
    zk[kx] = E::at(E::ZROW, ox * S + kx - 1, c);
  }
  f32x4 win[3][3][2];
#pragma unroll
  for (int r = 0; r < RPS; ++r) {
    if (r < nrows) {
#pragma unroll
      for (int rel = (r == 0 ? 0 : r * S + 3 - S); rel <= r * S + 2; ++rel) {
        const bool ok = !EDGE || (unsigned)(yfirst + rel) < (unsigned)HW;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bf16x8 e = *reinterpret_cast<const bf16x8*>(sE + (ok ? pk[kx] + rel * E::ROWB : zk[kx]));
          win[rel % 3][kx][0] = f32x4{(float)e[0], (float)e[1], (float)e[2], (float)e[3]};
          win[rel % 3][kx][1] = f32x4{(float)e[4], (float)e[5], (float)e[6], (float)e[7]};
        }
      }
      f32x4 a0 = t.b[0], a1 = t.b[1];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          a0 += win[(r * S + ky) % 3][kx][0] * t.w[ky * 3 + kx][0];
          a1 += win[(r * S + ky) % 3][kx][1] * t.w[ky * 3 + kx][1];
        }
      const bf16x4 h0 = __builtin_convertvector(lrelu4(a0), bf16x4), h1 = __builtin_convertvector(lrelu4(a1), bf16x4);
      *reinterpret_cast<bf16x8*>(d) = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      d += d_row;
    }
  }
}

// Output rows are cut into runs of RPS rows; a thread's cost is ~ (rows it loads + widens) + 2 x (outputs it computes) per
// run, and the rounds of 256 threads a workgroup needs follow from the runs that really exist (ceil(rows / RPS)).
constexpr int pick_rps(int per_run_items, int rows, int blocks, int stride) {
  int best = rows, best_cost = 1 << 30;
  for (int rps = 1; rps <= rows && rps <= 10; ++rps) {
    const int runs = (rows + rps - 1) / rps, items = per_run_items * runs * blocks, rounds = (items + 255) / 256;
    const int cost = rounds * ((rps - 1) * stride + 3 + 2 * rps);
    if (cost < best_cost) best = rps, best_cost = cost;
  }
  return best;
}

// XCD-aware tile order (speed only): workgroups b, b + 8, ... share an XCD; each XCD gets a contiguous run of tiles so
// that the channel tiles of one pixel tile (same A rows) meet in one L2
__device__ __forceinline__ int xcd_tile(int t, int nwg) {
  const int qn = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
  return (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + idx;
}

// ---- whole frames: F frames of HW x HW pixels x BN channels per workgroup ----
template <int HW, int F, int BN, int S>
struct FTGeomB : GemmGeomB<F * HW * HW, BN> {
  using Base = GemmGeomB<F * HW * HW, BN>;
  static constexpr int P = HW * HW, M = F * P;
  static constexpr int NQ = BN / 8;                            // 16-B channel columns
  static constexpr int HO = (HW + 2 - 3) / S + 1;              // output rows = columns
  using E = ETileB<HW, BN, F * HW>;                            // frame f occupies image rows f HW .. + HW - 1
  static constexpr int RPS = pick_rps(NQ * HO, HO, F, S), RS = (HO + RPS - 1) / RPS;   // rows per run, runs per frame
  static constexpr size_t lds = (2 * (size_t)Base::STAGE > E::bytes ? 2 * (size_t)Base::STAGE : E::bytes + 15) / 16 * 16;
  static constexpr int occ = 160 * 1024 / lds >= 2 ? 2 : 1;
  static_assert(lds <= 160 * 1024, "LDS budget");
};

template <int HW, int F, int BN, int S>
__global__ __launch_bounds__(256, (FTGeomB<HW, F, BN, S>::occ)) void pw_dw_bf16_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ wd, const float* __restrict__ bd, bf16_t* __restrict__ D, int ldd, int frames, int K, int N,
    int n_ntiles, int nwg, unsigned a_bytes, unsigned w_bytes, const bf16_t* __restrict__ ups, int ld_ups) {
  using G = FTGeomB<HW, F, BN, S>;
  using E = typename G::E;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;

  const int bid = xcd_tile(blockIdx.x, nwg);
  const int ft = bid / n_ntiles, nt = bid - ft * n_ntiles;
  const int f0 = ft * F, nf = frames - f0 < F ? frames - f0 : F;   // frames of this tile
  const int m0 = f0 * G::P, m_valid = nf * G::P, n0 = nt * BN;

  int voff[G::LPT];
  dma_offsets<G>(voff, wave, lane, lda, K, n0, BN, [&](int r) { return m0 + (r < m_valid ? r : m_valid - 1); });   // pad rows re-read the last pixel
  // the low-resolution tile of the upsampled addend travels under the GEMM (F = 1 frame, BN = 64: the whole HW/2 x HW/2 frame)
  char* sG = ring + G::lds;
  if (ups) ups_tile_load_b<HW / 2>(sG, ups + (size_t)f0 * (HW / 2) * (HW / 2) * ld_ups + n0, ld_ups, 0, HW / 2, wave, lane);
  // (the expand biases are requested before the K loop and the depthwise taps before epilogue 1 -- both are the head of a
  // latency chain otherwise: a global load in front of the first instruction that needs it)
  const int ng = wave % G::NG, mg = wave / G::NG;
  const f32x4 bias0 = *reinterpret_cast<const f32x4*>(b1 + n0 + 32 * ng + 4 * q);
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + n0 + 32 * ng + 16 + 4 * q);
  f32x4 acc[G::MTW][2];
  pw_dw_gemm_b<G>(ring, A, W1, a_bytes, w_bytes, voff, K / 32, wave, l15, q, acc);
  Taps8 taps;
  load_taps(taps, wd, bd, n0 + 8 * (tid % G::NQ), N);   // 256 % NQ == 0: a thread keeps its channel column

  // ---- epilogue 1: + b1 (+ the upsampled addend), LeakyReLU -> the zero-bordered bf16 E image ----
  char* sE = ring;
  e_zero_border<E, F * HW, HW>(sE, tid);
  {
#pragma unroll
    for (int i = 0; i < G::MTW; ++i) {
      const int t = mg + G::MG * i;
      const int px = 16 * t + l15;
      if (t < G::MT && px < m_valid) {
        const int f = px / G::P, rem = px - f * G::P, y = rem / HW, x = rem - y * HW;
        f32x4 v0 = acc[i][0] + bias0, v1 = acc[i][1] + bias1;
        if (ups) {
          const UpsCorners uc = ups_corners<HW>(0, y, x);
          v0 += ups_at_lds_b(sG, uc, 4 * ng + (q >> 1), q & 1);
          v1 += ups_at_lds_b(sG, uc, 4 * ng + 2 + (q >> 1), q & 1);
        }
        e_store<E>(sE, f * HW + y, x, ng, q, lrelu4(v0), lrelu4(v1));
      }
    }
  }
  __syncthreads();

  // ---- epilogue 2: depthwise 3x3 (zero columns / the zero row = the padding), + bd, LeakyReLU -> D ----
  {
    constexpr int HO = G::HO, NITEM = G::NQ * HO * G::RS * F;
    static_assert(256 % G::NQ == 0, "a thread keeps its channel column");
    for (int id = tid; id < NITEM; id += 256) {
      const int c = id % G::NQ, rest = id / G::NQ, ox = rest % HO, run = rest / HO, f = run / G::RS, r0 = (run - f * G::RS) * G::RPS;
      if (f >= nf) continue;
      const int nrows = HO - r0 < G::RPS ? HO - r0 : G::RPS;
      dw_run_b<E, S, HW, true, G::RPS>(sE, f * HW + r0 * S - 1, r0 * S - 1, ox, c, taps,
                                       D + ((size_t)(f0 + f) * HO * HO + (size_t)r0 * HO + ox) * ldd + n0 + 8 * c, (size_t)HO * ldd, nrows);
    }
  }
}

// ---- 40 x 40 frames: the tile is a STRIP of SR output rows of one frame -- its (SR - 1) * STRIDE + 3 input rows x 40
// columns are contiguous rows of A; the one-row halo of the neighbouring strips is recomputed (10 rows for 8: x 1.25;
// stride 2: 9 for 8: x 1.125), rows above / below the frame become zero rows of E (pw_dw.hip, pw_dw_strip_kernel) ----
template <int HW, int SR, int STRIDE, int BN>
struct FSGeomB : GemmGeomB<((SR - 1) * STRIDE + 3) * HW, BN> {
  using Base = GemmGeomB<((SR - 1) * STRIDE + 3) * HW, BN>;
  static constexpr int P = HW * HW, RIN = (SR - 1) * STRIDE + 3, M = RIN * HW;
  static constexpr int HO = (HW + 2 - 3) / STRIDE + 1, NS = (HO + SR - 1) / SR;   // output rows / strips per frame
  static constexpr int NQ = BN / 8;
  using E = ETileB<HW, BN, RIN>;                               // image row = strip-local input row (zero rows by epilogue 1)
  static constexpr int RPS = pick_rps(NQ * HO, SR, 1, STRIDE), RS = (SR + RPS - 1) / RPS;   // output rows per run, runs per strip
  static constexpr size_t lds = (2 * (size_t)Base::STAGE > E::bytes ? 2 * (size_t)Base::STAGE : E::bytes + 15) / 16 * 16;
  static constexpr int occ = 160 * 1024 / lds >= 2 ? 2 : 1;
  static_assert(lds <= 160 * 1024, "LDS budget");
};

template <int HW, int SR, int STRIDE, int BN>
__global__ __launch_bounds__(256, (FSGeomB<HW, SR, STRIDE, BN>::occ)) void pw_dw_bf16_strip_kernel(
    const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ wd, const float* __restrict__ bd, bf16_t* __restrict__ D, int ldd, int frames, int K, int N,
    int n_ntiles, int nwg, unsigned a_bytes, unsigned w_bytes, const bf16_t* __restrict__ ups, int ld_ups) {
  using G = FSGeomB<HW, SR, STRIDE, BN>;
  using E = typename G::E;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;

  const int bid = xcd_tile(blockIdx.x, nwg);
  const int ft = bid / n_ntiles, nt = bid - ft * n_ntiles;
  const int fr = ft / G::NS, st = ft - fr * G::NS;           // frame, strip
  const int y0 = st * SR * STRIDE - 1;                       // first input row of the strip (-1: the zero row above the frame)
  const int row_lo = fr * G::P, row_base = row_lo + y0 * HW; // A row of strip pixel 0 (may lie before the frame)
  const int n0 = nt * BN;

  int voff[G::LPT];
  dma_offsets<G>(voff, wave, lane, lda, K, n0, BN, [&](int r) {   // rows outside the frame / pad rows: any row of the frame
    const int row = row_base + r;
    return row < row_lo ? row_lo : (row > row_lo + G::P - 1 ? row_lo + G::P - 1 : row);
  });
  // the low-resolution rows under this strip (at most 6: the tap rows of its first and last row inside the frame)
  char* sG = ring + G::lds;
  const int gy0 = ups_tap((float)(HW / 2 - 1) / (float)(HW - 1), y0 < 0 ? 0 : y0, HW / 2).i0;
  if (ups) ups_tile_load_b<HW / 2>(sG, ups + (size_t)fr * (HW / 2) * (HW / 2) * ld_ups + n0, ld_ups, gy0, kUpsTileBytes / (HW / 2 * 128), wave, lane);
  const int ng = wave % G::NG, mg = wave / G::NG;
  const f32x4 bias0 = *reinterpret_cast<const f32x4*>(b1 + n0 + 32 * ng + 4 * q);   // (requested ahead: see pw_dw_bf16_kernel)
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + n0 + 32 * ng + 16 + 4 * q);
  f32x4 acc[G::MTW][2];
  pw_dw_gemm_b<G>(ring, A, W1, a_bytes, w_bytes, voff, K / 32, wave, l15, q, acc);
  Taps8 taps;
  load_taps(taps, wd, bd, n0 + 8 * (tid % G::NQ), N);

  // ---- epilogue 1: + b1 (+ the upsampled addend), LeakyReLU -> the E image; rows outside the frame are zero ----
  char* sE = ring;
  e_zero_border<E, G::RIN, HW>(sE, tid);
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < G::MTW; ++i) {
      const int t = mg + G::MG * i;
      const int px = 16 * t + l15;
      if (t < G::MT && px < G::M) {
        const int yl = px / HW, x = px - yl * HW, y = y0 + yl;
        const bool inside = y >= 0 && y < HW;
        f32x4 v0 = acc[i][0] + bias0, v1 = acc[i][1] + bias1;
        if (ups && inside) {
          const UpsCorners uc = ups_corners<HW>(gy0, y, x);
          v0 += ups_at_lds_b(sG, uc, 4 * ng + (q >> 1), q & 1);
          v1 += ups_at_lds_b(sG, uc, 4 * ng + 2 + (q >> 1), q & 1);
        }
        e_store<E>(sE, yl, x, ng, q, inside ? lrelu4(v0) : z, inside ? lrelu4(v1) : z);
      }
    }
  }
  __syncthreads();

  // ---- epilogue 2: depthwise 3x3 over the strip, + bd, LeakyReLU -> D ----
  {
    constexpr int HO = G::HO, NITEM = G::NQ * HO * G::RS;
    const int oy_first = st * SR, rows_out = HO - oy_first < SR ? HO - oy_first : SR;
    static_assert(256 % G::NQ == 0, "a thread keeps its channel column");
    for (int id = tid; id < NITEM; id += 256) {
      const int c = id % G::NQ, rest = id / G::NQ, ox = rest % HO, r0 = (rest / HO) * G::RPS;
      if (r0 >= rows_out) continue;
      const int nrows = rows_out - r0 < G::RPS ? rows_out - r0 : G::RPS;
      dw_run_b<E, STRIDE, HW, false, G::RPS>(sE, r0 * STRIDE, 0, ox, c, taps,
                                             D + (((size_t)fr * HO + oy_first + r0) * HO + ox) * ldd + n0 + 8 * c, (size_t)HO * ldd, nrows);
    }
  }
}

template <class G, class Kern>
int launch_common(Kern kern, unsigned long long* attr_once, bool ups_ok, long long nwg, const bf16_t* a, int lda, const bf16_t* w1,
                  const float* b1, const float* wd, const float* bd, bf16_t* d, int ldd, int frames, int k, int n, int n_nt,
                  const bf16_t* ups, int ld_ups, hipStream_t stream) {
  CASYNC_REQUIRE(!ups || ups_ok, "pw_dw (bf16): this instance takes no upsampled addend");
  if (int st = casync_ensure_dyn_lds(attr_once, reinterpret_cast<const void*>(kern), (int)G::lds + (ups_ok ? kUpsTileBytes : 0))) return st;
  const unsigned long long ab = ((unsigned long long)((long long)frames * G::P - 1) * lda + k) * 2, wb = (unsigned long long)n * k * 2;
  CASYNC_REQUIRE(nwg < (1ll << 31) && ab < (1ull << 31) && wb < (1ull << 31), "pw_dw (bf16): operand larger than 2 GiB");
  CASYNC_REQUIRE((long long)frames * G::P < (1ll << 24) && (long long)lda * 2 < (1ll << 24),
                 "pw_dw (bf16): %lld rows of %d elements exceed the kernel's 24-bit row arithmetic", (long long)frames * G::P, lda);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), G::lds + (ups ? kUpsTileBytes : 0), stream, a, lda, w1, b1, wd, bd, d, ldd, frames,
                     k, n, n_nt, (int)nwg, (unsigned)ab, (unsigned)wb, ups, ld_ups);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

template <int HW, int SR, int STRIDE, int BN>
int launch_fs(const bf16_t* a, int lda, const bf16_t* w1, const float* b1, const float* wd, const float* bd, bf16_t* d, int ldd,
              int frames, int k, int n, const bf16_t* ups, int ld_ups, hipStream_t stream) {
  using G = FSGeomB<HW, SR, STRIDE, BN>;
  static unsigned long long attr_once = 0;
  constexpr bool UPS_OK = BN == 64 && STRIDE == 1 && G::lds % 16 == 0 && (HW / 2) * 128 * 6 <= kUpsTileBytes && G::lds + kUpsTileBytes <= 160 * 1024;
  const int n_nt = n / BN;
  return launch_common<G>(pw_dw_bf16_strip_kernel<HW, SR, STRIDE, BN>, &attr_once, UPS_OK, (long long)frames * G::NS * n_nt, a, lda, w1, b1,
                          wd, bd, d, ldd, frames, k, n, n_nt, ups, ld_ups, stream);
}

template <int HW, int F, int BN, int S>
int launch_ft(const bf16_t* a, int lda, const bf16_t* w1, const float* b1, const float* wd, const float* bd, bf16_t* d, int ldd,
              int frames, int k, int n, const bf16_t* ups, int ld_ups, hipStream_t stream) {
  using G = FTGeomB<HW, F, BN, S>;
  static unsigned long long attr_once = 0;
  constexpr bool UPS_OK = F == 1 && BN == 64 && S == 1 && HW % 2 == 0 && G::lds % 16 == 0 && (HW / 2) * (HW / 2) * 128 <= kUpsTileBytes &&
                          G::lds + kUpsTileBytes <= 160 * 1024;
  const int n_nt = n / BN;
  return launch_common<G>(pw_dw_bf16_kernel<HW, F, BN, S>, &attr_once, UPS_OK, (long long)((frames + F - 1) / F) * n_nt, a, lda, w1, b1, wd,
                          bd, d, ldd, frames, k, n, n_nt, ups, ld_ups, stream);
}

// channel tile of the whole-frame instances below 20x20: 64, or 128 when the option asks for it and N allows
int bn_small(int cexp) {
  const int want = casync_opts().fuse_dw_bf16_bn;
  return want >= 128 && cexp % 128 == 0 ? 128 : 64;
}

}  // namespace

bool pw_dw_bf16_supported(int hw, int cin, int cexp, int stride) {
  if (cin % 32 || cexp % 64) return false;
  if (hw == 10 || hw == 16) return stride == 1;
  return (hw == 20 || hw == 40) && (stride == 1 || stride == 2);
}

const char* pw_dw_bf16_kernel_name(int hw, int cexp, int frames, int stride) {
  static thread_local char buf[64];
  (void)frames;
  if (hw == 40) snprintf(buf, sizeof(buf), "pw_dw_bf16_strip_kernel<40, %d, %d, 64>", stride == 1 ? 8 : 4, stride);
  else if (hw == 20) snprintf(buf, sizeof(buf), "pw_dw_bf16_kernel<20, 1, 64, %d>", stride);
  else snprintf(buf, sizeof(buf), "pw_dw_bf16_kernel<%d, %d, %d, 1>", hw, hw == 10 ? 2 : 1, bn_small(cexp));
  return buf;
}

bool pw_dw_bf16_takes_ups(int hw, int stride) { return stride == 1 && (hw == 20 || hw == 40); }

int launch_pw_dw_bf16(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd, void* d, int ldd,
                      int frames, int hw, int stride, int cin, int cexp, hipStream_t stream, const void* ups, int ld_ups) {
  CASYNC_REQUIRE(a && w1 && b1 && wd && bd && d && frames > 0, "pw_dw (bf16): bad args");
  CASYNC_REQUIRE(pw_dw_bf16_supported(hw, cin, cexp, stride), "pw_dw (bf16): no instance for %dx%d cin=%d cexp=%d stride=%d", hw, hw, cin,
                 cexp, stride);
  CASYNC_REQUIRE(lda >= cin && lda % 8 == 0 && ldd >= cexp && ldd % 8 == 0, "pw_dw (bf16): bad leading dimensions");
  CASYNC_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)d % 16) == 0 && ((uintptr_t)b1 % 16) == 0 &&
                     ((uintptr_t)wd % 16) == 0 && ((uintptr_t)bd % 16) == 0,
                 "pw_dw (bf16): pointers must be 16-B aligned");
  CASYNC_REQUIRE(!ups || (pw_dw_bf16_takes_ups(hw, stride) && ld_ups >= cexp && ld_ups % 8 == 0 && (uintptr_t)ups % 16 == 0),
                 "pw_dw (bf16): bad upsampled addend");
  const bf16_t* af = static_cast<const bf16_t*>(a);
  const bf16_t* wf = static_cast<const bf16_t*>(w1);
  const bf16_t* uf = static_cast<const bf16_t*>(ups);
  bf16_t* df = static_cast<bf16_t*>(d);
  if (hw == 40)
    return stride == 1 ? launch_fs<40, 8, 1, 64>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                       : launch_fs<40, 4, 2, 64>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  if (hw == 20)
    return stride == 1 ? launch_ft<20, 1, 64, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                       : launch_ft<20, 1, 64, 2>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  const bool wide = bn_small(cexp) == 128;
  if (hw == 10)
    return wide ? launch_ft<10, 2, 128, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                : launch_ft<10, 2, 64, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  return wide ? launch_ft<16, 1, 128, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
              : launch_ft<16, 1, 64, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
}
