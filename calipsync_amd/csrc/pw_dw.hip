// Expand 1x1 conv + BN + LeakyReLU + depthwise 3x3 + BN + LeakyReLU in ONE kernel, for the inverted residuals of the
// low-resolution stages (10x10, 16x16, 20x20 as whole-frame tiles, 40x40 as row strips -- pw_dw_strip_kernel below:
// reference module/unet.py:17-30 with BN folded):
//
//   D[b, oy, ox, n] = lrelu( sum_taps wd[tap][n] * E[b, oy*s + ky - 1, ox*s + kx - 1, n] + bd[n] ),
//   E[b, y, x, n]   = lrelu( A[b, y, x, :] . W1[n, :] + b1[n] )                    (zero outside the frame)
//
// Below 32x32 the expanded tensor (2 x Cin channels, up to 2048) is too wide for the fully fused block of
// ir_fused.hip (its project accumulators would not fit the registers), so round 2 ran these blocks as GEMM ->
// depthwise -> GEMM: 30 depthwise launches per forward, each a serialisation point of its lane (11-14 us alone,
// 26-49 us beside the other lane's GEMMs: profiles/r2_f32_b64_kernel_stats_timed.csv) and one HBM/L2 round trip
// of E each.  Here an output tile of the expand GEMM is WHOLE FRAMES x BN channels: the 3x3 neighbourhood of every
// pixel is inside the tile, so the depthwise conv runs on the accumulator tile while it sits in LDS -- no halo, no
// recompute, E never leaves the CU.  The project GEMM (pw_gemm) then reads D as before.
//
// GEMM part: the LDS-DMA ring of gemm.hip (128-B k-tile rows, XOR-swizzled through the source address, two stages,
// buffer_load ... lds with per-lane constant offsets and the k position in an SGPR), computed with
// v_mfma_f32_16x16x4_f32 so that the row count only has to be a multiple of 16: 2 frames of 10x10 = 200 rows -> 208
// (4 % padding), 16x16 = 256, 20x20 = 400 rows exactly.  The weight fragment is the MFMA A operand and the pixels the
// B operand: a lane ends up with four consecutive channels of one pixel, one 16-B LDS store per tile.
//
// Epilogue (round 4): + b1, LeakyReLU -> E tile in LDS (over the dead ring) as an IMAGE with a zero column on either
// side: rows of HW + 2 pixels.  Then every thread owns one channel quad (its nine tap weights live in registers) and one
// output column and walks down a run of output rows: the nine taps of an output are nine ds_read_b128 at COMPILE-TIME row
// offsets from three column pointers that advance by a constant per row -- no column tests, no per-tap address
// arithmetic, no divisions in the loop (whole frames skip the tap row above the first / below the last image row).  Round 3's epilogue recomputed
// (frame, y, x) of every output by two divisions, tested and addressed each tap on its own and XOR-keyed it by its pixel:
// ~150 vector instructions per output quad, 3.0-3.3 vector instructions per MFMA over the whole kernel on the 16x16 /
// 20x20 / 40x40 instances (profiles/r4_mfma_busy.json), on the issue slots the fp32 MFMAs need.
#include <stdlib.h>

#include "common.h"
#include "ir_common.h"
#include "pw_dw_common.h"

namespace {

// The E image both kernels build: NR rows of WP = HW + 2 pixels (a zero column left and right of the frame) of BN floats;
// the 16-B columns of a pixel are XOR-keyed by its column (the stores of eight consecutive pixels of one channel quad fall
// on eight different 16-B bank groups; a reader that walks down ONE column pays the key once).  No pixel padding and no
// zero rows: the image must not outgrow the k-tile ring it overlays -- a workgroup with a larger LDS footprint loses more
// in the two-lane schedule than a cheaper epilogue gains (first version: 144-B pixels + zero rows, -0.5 % end to end).
template <int HW, int BN, int NR>
struct ETile {
  static constexpr int WP = HW + 2, ROWF = WP * BN;
  static constexpr size_t bytes = (size_t)NR * ROWF * sizeof(float);
  // float offset of channel quad `cq` of the pixel in image row R, column x (-1 .. HW)
  static __device__ __forceinline__ int at(int R, int x, int cq) { return (R * WP + x + 1) * BN + ((cq ^ ((x + 1) & 7)) << 2); }
};
// KF = floats per k-tile row: 32 (128-B rows, gemm.hip's ring) or 16 (64-B rows: half the ring, so that the whole
// working set of a workgroup stays near the 32 KB of the 64x64 GEMM tiles it shares the CUs with -- a workgroup that
// fills a CU's LDS shuts the other lane's kernels out, measured -2 % end to end with 74 KB rings)
template <int HW, int F, int BN, int KF, int S = 1, int NST_ = 2>
struct FTGeom {
  static constexpr int NST = NST_;                             // ring stages
  static constexpr int ROWB = KF * 4, RPI = 1024 / ROWB;       // row bytes; rows one LDS-DMA instruction fills (8 / 16)
  static constexpr int P = HW * HW, M = F * P, MT = (M + 15) / 16, M_PAD = 16 * MT;
  static constexpr int NT = BN / 16, WPN = 4 / NT;             // n-tiles; waves that share an n-tile
  static constexpr int MTW = (MT + WPN - 1) / WPN;             // m-tiles per wave
  static constexpr int ROWS = M_PAD + BN, LPT = (ROWS + 4 * RPI - 1) / (4 * RPI), STAGE = LPT * 4 * RPI * ROWB;
  static constexpr int NQ = BN / 4;                            // channel quads
  static constexpr int HO = (HW + 2 - 3) / S + 1;              // output rows = columns
  using E = ETile<HW, BN, F * HW>;                             // frame f occupies image rows f HW .. + HW - 1
  static constexpr int RS = pick_runs(NQ * HO, HO, F), RPS = (HO + RS - 1) / RS;   // runs per frame, rows per run
  static constexpr size_t etile = E::bytes;
  static constexpr size_t lds = NST * (size_t)STAGE > etile ? NST * (size_t)STAGE : etile;
  static constexpr int occ = (int)(160 * 1024 / lds) >= 4 ? 4 : (int)(160 * 1024 / lds);
  static_assert(NT == 2 || NT == 4, "BN = 32 or 64");
  static_assert(KF == 16 || KF == 32, "64-B or 128-B k-tile rows");
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert(2 * E::ROWF * 4 < 65536, "tap row offsets are DS immediates");
};

// four channels of the bilinear x2 upsample (align_corners=True) of a low-resolution tensor g [HL x HL pixels][ld] at
// high-resolution pixel (y, x): the addend of an Up block's commuted expand conv (common.h GemmEpilogue::ups_src)
template <int HW>
__device__ __forceinline__ f32x4 ups_at(const float* g, int ld, int y, int x) {
  constexpr int HL = HW / 2;
  const UpsTap ty = ups_tap((float)(HL - 1) / (float)(HW - 1), y, HL), tx = ups_tap((float)(HL - 1) / (float)(HW - 1), x, HL);
  return ups_lerp(ty, tx, *reinterpret_cast<const f32x4*>(g + (size_t)(ty.i0 * HL + tx.i0) * ld),
                  *reinterpret_cast<const f32x4*>(g + (size_t)(ty.i0 * HL + tx.i1) * ld),
                  *reinterpret_cast<const f32x4*>(g + (size_t)(ty.i1 * HL + tx.i0) * ld),
                  *reinterpret_cast<const f32x4*>(g + (size_t)(ty.i1 * HL + tx.i1) * ld));
}

// The same addend from a tile of g parked in LDS (round 4).  ups_at() costs a lane 52 gathered 16-B loads in its first
// epilogue, every low-resolution value fetched 4-8 times from L2 by the lanes around it; from LDS the 32-frame launches
// take 118 instead of 130 us (up2.0, 40x40 strips) and 82 instead of 87 us (up1.0), i.e. what the same GEMM + depthwise
// cost without an addend (twice down2.1's 54 us / down3.1's 40 us at twice their channels): B=64 +0.95 %.  The tile holds GR rows
// of HL pixels x 32 channels from row gy0 on, linear in pixels (LDS-DMA: a wave writes 8 pixels x 128 B per instruction),
// the 16-B channel quads of pixel p XOR-keyed by p & 7 on the SOURCE side, so that the 16 lanes of a read -- the same
// quad of ~8 consecutive pixels -- land on different banks.
constexpr int kUpsTileBytes = 16384;   // 6 rows x 20 pixels (40x40 strips) or 10 x 10 (20x20 frames) x 128 B, rounded up to 4 issues
template <int HL>
__device__ __forceinline__ void ups_tile_load(float* sG, const float* g, int ld, int gy0, int n_rows, int wave, int lane) {
  // g: this frame's channel slice (32 channels from the workgroup's n0); rows past the frame re-read its last row
#pragma unroll
  for (int j = 0; j < kUpsTileBytes / 4096; ++j) {
    const int piece = (j * 4 + wave) * 64 + lane, p = piece >> 3, ql = piece & 7;
    int gr = p / HL;
    const int gx = p - gr * HL;
    gr = gr < n_rows ? gr : n_rows - 1;
    const int gy = gy0 + gr < HL ? gy0 + gr : HL - 1;
    const float* src = g + (size_t)(gy * HL + gx) * ld + 4 * (ql ^ (p & 7));
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(reinterpret_cast<char*>(sG) + (j * 4 + wave) * 1024), 16, 0, 0);
  }
}
// (four corner weights per pixel and fused multiply-adds, as P1 of the fused block does for its G slice: 8 instead of the
// 18 packed operations of ATen's three-lerp form per four channels -- round 5; same value to fp32 rounding)
template <int HW>
__device__ __forceinline__ f32x4 ups_at_lds(const float* sG, int gy0, int y, int x, int cq) {
  constexpr int HL = HW / 2;
  const UpsTap ty = ups_tap((float)(HL - 1) / (float)(HW - 1), y, HL), tx = ups_tap((float)(HL - 1) / (float)(HW - 1), x, HL);
  auto at = [&](int gy, int gx) {
    const int p = (gy - gy0) * HL + gx;
    return *reinterpret_cast<const f32x4*>(sG + p * 32 + 4 * (cq ^ (p & 7)));
  };
  return (ty.l0 * tx.l0) * at(ty.i0, tx.i0) + (ty.l0 * tx.l1) * at(ty.i0, tx.i1) + (ty.l1 * tx.l0) * at(ty.i1, tx.i0) +
         (ty.l1 * tx.l1) * at(ty.i1, tx.i1);
}

// The GEMM part both kernels share: acc[i] (i-th pixel tile of this wave) = W1 tile x A rows over the whole K, k-tiles
// arriving by LDS-DMA into a two-stage ring.  voff[j]: this lane's source offset of the wave's j-th LDS-DMA instruction
// (rows [0, M_PAD) of a stage are A rows, then BN rows of W1); ends with the ring consumed (barrier).
template <class G, int BN, int KF>
__device__ __forceinline__ void pw_dw_gemm(char* ring, const float* __restrict__ A, const float* __restrict__ W1, unsigned a_bytes,
                                           unsigned w_bytes, const int (&voff)[G::LPT], int nk, int wave, int l15, int q,
                                           f32x4 (&acc)[G::MTW]) {
  constexpr int ROWB = G::ROWB, RPI = G::RPI;
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < G::LPT; ++j) {
      char* dst = ring + stage * G::STAGE + (j * 4 + wave) * RPI * ROWB;
      if ((j * 4 + wave) * RPI < G::M_PAD) ft_dma16(A, a_bytes, dst, voff[j], kt * ROWB);
      else ft_dma16(W1, w_bytes, dst, voff[j], kt * ROWB);
    }
  };
  // wave -> channel tile `wn` and the pixel tiles wm, wm + WPN, ...
  const int wn = wave % G::NT, wm = wave / G::NT;
  const int key = ft_key<KF>(l15);                        // key of every fragment row 16 t + l15 (16 t adds nothing)
  int frag[KF / 16];                                      // this lane's 16 B of each 16-float k-group of a row
#pragma unroll
  for (int g = 0; g < KF / 16; ++g) frag[g] = l15 * ROWB + (((4 * g + q) ^ key) << 4);
#pragma unroll
  for (int i = 0; i < G::MTW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the MFMAs of one k-tile: all its fragments are requested up front, then the MFMAs run back to back: s outer, tile
  // inner, so an accumulator is reused only every MTW instructions (no dependent-issue stalls)
  auto compute = [&](const char* st) __attribute__((always_inline)) {
    f32x4 fw[KF / 16], fa[KF / 16][G::MTW];
#pragma unroll
    for (int g = 0; g < KF / 16; ++g) {
      fw[g] = *reinterpret_cast<const f32x4*>(st + (G::M_PAD + 16 * wn) * ROWB + frag[g]);
#pragma unroll
      for (int i = 0; i < G::MTW; ++i) {
        // a tile index past the end (BN = 32, second wave of the pair) recomputes the last tile and never stores it
        const int t = wm + G::WPN * i < G::MT ? wm + G::WPN * i : G::MT - 1;
        fa[g][i] = *reinterpret_cast<const f32x4*>(st + 16 * t * ROWB + frag[g]);
      }
    }
#pragma unroll
    for (int g = 0; g < KF / 16; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < G::MTW; ++i) acc[i] = mfma16(fw[g][s], fa[g][i][s], acc[i]);
  };
  if constexpr (G::NST == 2) {
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of k-tile kt has landed
      __syncthreads();                                   // ... everyone's; and everyone is done reading the other stage
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      compute(ring + (kt & 1) * G::STAGE);
    }
  } else {
    // Deep ring (round 5, small batches): NST - 1 k-tiles in flight.  With a handful of workgroups per CU nobody hides a
    // k-tile's fill latency (two stages: ~0.45 us per k-iteration at M = 800 whatever the tile, profiles/r4_ab_small_batch.txt);
    // here a fill has NST - 2 iterations to arrive.  Still ONE barrier per k-tile: k-tile kt + NST - 1 goes into the stage
    // that k-tile kt - 1 was read from, and everyone has left iteration kt - 1 once they meet at the barrier of iteration kt.
    constexpr int NST = G::NST;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < nk) issue(p, p);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * G::LPT) : "memory");   // k-tile kt landed (newer ones may fly)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + NST - 1 < nk) issue(kt + NST - 1, (kt + NST - 1) % NST);
      compute(ring + (kt % NST) * G::STAGE);
    }
  }
  __syncthreads();   // the ring is consumed: it becomes the E tile
}

// The two zero columns of the E image (x = -1 and x = HW of all NR rows), all BN / 4 quads.
template <class E, int NR, int NQ, int HW>
__device__ __forceinline__ void e_zero_columns(float* sE, int tid) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < NR * 2 * NQ; i += 256) {
    const int cq = i % NQ, r2 = i / NQ, R = r2 >> 1, x = (r2 & 1) ? HW : -1;
    *reinterpret_cast<f32x4*>(sE + E::at(R, x, cq)) = z;
  }
}

// One work item of the depthwise epilogue: channel quad cq, output column ox, `nrows` output rows downwards from output
// row oy0.  R0 = image row of tap row ky = 0 of the first output (may be -1 for a whole frame's first row: skipped);
// d = the first output's place in D.  EDGE: the image holds whole frames of HW rows without zero rows -- the tap row above
// the first and below the last image row of the frame is left out; strips carry their zero rows in the image.
template <class E, int S, int HW, bool EDGE>
__device__ __forceinline__ void dw_run(const float* sE, int R0, int ox, int cq, int oy0, const f32x4 (&wt)[9], f32x4 bv, float* d,
                                       size_t d_row, int nrows) {
  int p[3];   // float offsets into the image (integers, so that the loads stay LDS instructions with immediate row offsets)
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) p[kx] = E::at(R0, ox * S + kx - 1, cq);
  for (int r = 0; r < nrows; ++r) {
    f32x4 a = bv;
    const int y0 = (oy0 + r) * S - 1;                       // frame row of tap row 0
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (EDGE && (y0 + ky < 0 || y0 + ky >= HW)) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) a += *reinterpret_cast<const f32x4*>(sE + p[kx] + ky * E::ROWF) * wt[ky * 3 + kx];
    }
    *reinterpret_cast<f32x4*>(d) = lrelu4(a);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) p[kx] += S * E::ROWF;
    d += d_row;
  }
}

template <int HW, int F, int BN, int KF, int S, int NST = 2>
__global__ __launch_bounds__(256, (FTGeom<HW, F, BN, KF, S, NST>::occ)) void pw_dw_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ wd, const float* __restrict__ bd, float* __restrict__ D, int ldd, int frames, int K, int N,
    int n_ntiles, int nwg, unsigned a_bytes, unsigned w_bytes, const float* __restrict__ ups, int ld_ups) {
  using G = FTGeom<HW, F, BN, KF, S, NST>;
  using E = typename G::E;
  constexpr int ROWB = G::ROWB, RPI = G::RPI, CPR = ROWB / 16;   // 16-B columns per row
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4, lrow = lane / CPR, lcol = lane % CPR;

  // XCD-aware tile order (speed only): workgroups b, b + 8, ... share an XCD; give each XCD a contiguous run of tiles so
  // the channel tiles of one frame group (same A rows) meet in one L2
  int ft, nt;
  {
    const int t = blockIdx.x, qn = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + idx;
    ft = bid / n_ntiles;
    nt = bid - ft * n_ntiles;
  }
  const int f0 = ft * F, nf = frames - f0 < F ? frames - f0 : F;   // frames of this tile
  const int m0 = f0 * G::P, m_valid = nf * G::P, n0 = nt * BN;
  const int nk = K / KF;

  // ---- LDS-DMA: this lane's source offset of each of the wave's LPT instructions (8 rows x 128 B each) ----
  int voff[G::LPT];
#pragma unroll
  for (int j = 0; j < G::LPT; ++j) {
    const int r = (j * 4 + wave) * RPI + lrow;             // row inside the stage: [0, M_PAD) = A, then BN rows of W1
    const int cs = lcol ^ ft_key<KF>(r);                   // swizzled source column
    if ((j * 4 + wave) * RPI < G::M_PAD) {
      const int row = m0 + (r < m_valid ? r : m_valid - 1);   // pad rows re-read the last pixel; never used
      voff[j] = (int)__umul24((unsigned)row, (unsigned)(lda * 4)) + cs * 16;   // rows < 2^24, row bytes < 2^24, operand < 2 GiB (launcher)
    } else {
      const int wr = r - G::M_PAD < BN ? r - G::M_PAD : 0;    // rows behind the tile (LPT rounds up) re-read W row 0
      voff[j] = (n0 + wr) * K * 4 + cs * 16;
    }
  }
  const int wn = wave % G::NT, wm = wave / G::NT;
  // the low-resolution tile of the upsampled addend travels under the GEMM (its first vmcnt(0) + barrier cover it); it
  // lives behind the ring / E image: the launch adds kUpsTileBytes of LDS when `ups` is set (F = 1 frame, BN = 32 only)
  float* sG = reinterpret_cast<float*>(ring + G::lds);
  if (ups) ups_tile_load<HW / 2>(sG, ups + (size_t)f0 * (HW / 2) * (HW / 2) * ld_ups + n0, ld_ups, 0, HW / 2, wave, lane);
  f32x4 acc[G::MTW];
  pw_dw_gemm<G, BN, KF>(ring, A, W1, a_bytes, w_bytes, voff, nk, wave, l15, q, acc);

  // ---- epilogue 1: + b1, LeakyReLU -> the zero-bordered E image ----
  float* sE = reinterpret_cast<float*>(ring);
  e_zero_columns<E, F * HW, G::NQ, HW>(sE, tid);
  {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(b1 + n0 + 16 * wn + 4 * q);
#pragma unroll
    for (int i = 0; i < G::MTW; ++i) {
      const int t = wm + G::WPN * i;
      const int px = 16 * t + l15;
      if (t < G::MT && px < m_valid) {
        const int f = px / G::P, rem = px - f * G::P, y = rem / HW, x = rem - y * HW;
        f32x4 v = acc[i] + bias;
        if (ups) {
          // + the bilinear x2 upsample of the low-resolution half of an Up block's expand conv (it commutes with the
          // 1x1 conv: common.h GemmEpilogue::ups_src), HW/2 x HW/2 frames of ld_ups channels
          v += ups_at_lds<HW>(sG, 0, y, x, 4 * wn + q);
        }
        *reinterpret_cast<f32x4*>(sE + E::at(f * HW + y, x, 4 * wn + q)) = lrelu4(v);
      }
    }
  }
  __syncthreads();

  // ---- epilogue 2: depthwise 3x3 (zero columns = the horizontal padding; the vertical one by skipping tap rows), + bd, LeakyReLU -> D ----
  {
    constexpr int HO = G::HO, NITEM = G::NQ * HO * G::RS * F;
    f32x4 wt[9];
    int cq_of = -1;
    for (int id = tid; id < NITEM; id += 256) {
      const int cq = id % G::NQ, rest = id / G::NQ, ox = rest % HO, run = rest / HO, f = run / G::RS, r0 = (run - f * G::RS) * G::RPS;
      if (f >= nf) continue;
      if (cq != cq_of) {   // (256 % NQ == 0: a thread keeps its channel quad; loaded once)
        const float* wq = wd + n0 + 4 * cq;
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(wq + (size_t)t * N);
        cq_of = cq;
      }
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bd + n0 + 4 * cq);
      const int nrows = HO - r0 < G::RPS ? HO - r0 : G::RPS;
      dw_run<E, S, HW, true>(sE, f * HW + r0 * S - 1, ox, cq, r0, wt, bv,
                             D + ((size_t)(f0 + f) * HO * HO + (size_t)r0 * HO + ox) * ldd + n0 + 4 * cq, (size_t)HO * ldd, nrows);
    }
  }
}

// ---- 40 x 40 frames: a whole frame does not fit a tile (1600 pixels x 32 channels = 200 KB), so the tile is a STRIP of
// SR output rows of one frame: its (SR - 1) * STRIDE + 3 input rows x 40 columns are contiguous rows of A, the expand
// GEMM recomputes the one-row halo of the neighbouring strips (10 rows for 8: x 1.25; stride 2: 9 for 8: x 1.125),
// rows above / below the frame become zero rows of E (the depthwise conv zero-pads the EXPANDED tensor).  Everything
// else is pw_dw_kernel.
template <int HW, int SR, int STRIDE, int BN, int KF>
struct FSGeom {
  static constexpr int NST = 2;
  static constexpr int ROWB = KF * 4, RPI = 1024 / ROWB;
  static constexpr int P = HW * HW, RIN = (SR - 1) * STRIDE + 3, M = RIN * HW, MT = (M + 15) / 16, M_PAD = 16 * MT;
  static constexpr int HO = (HW + 2 - 3) / STRIDE + 1, NS = (HO + SR - 1) / SR;   // output rows / strips per frame
  static constexpr int NT = BN / 16, WPN = 4 / NT;
  static constexpr int MTW = (MT + WPN - 1) / WPN;
  static constexpr int ROWS = M_PAD + BN, LPT = (ROWS + 4 * RPI - 1) / (4 * RPI), STAGE = LPT * 4 * RPI * ROWB;
  static constexpr int NQ = BN / 4;
  using E = ETile<HW, BN, RIN>;                                // image row = strip-local input row (zero rows by epilogue 1)
  static constexpr int RS = pick_runs(NQ * HO, SR, 1), RPS = (SR + RS - 1) / RS;   // runs per strip, output rows per run
  static constexpr size_t etile = E::bytes;
  static constexpr size_t lds = 2 * (size_t)STAGE > etile ? 2 * (size_t)STAGE : etile;
  static constexpr int occ = (int)(160 * 1024 / lds) >= 4 ? 4 : (int)(160 * 1024 / lds);
  static_assert(NT == 2 || NT == 4, "BN = 32 or 64");
  static_assert(KF == 16 || KF == 32, "64-B or 128-B k-tile rows");
  static_assert(lds <= 160 * 1024, "LDS budget");
};

template <int HW, int SR, int STRIDE, int BN, int KF>
__global__ __launch_bounds__(256, (FSGeom<HW, SR, STRIDE, BN, KF>::occ)) void pw_dw_strip_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ wd, const float* __restrict__ bd, float* __restrict__ D, int ldd, int frames, int K, int N,
    int n_ntiles, int nwg, unsigned a_bytes, unsigned w_bytes, const float* __restrict__ ups, int ld_ups) {
  using G = FSGeom<HW, SR, STRIDE, BN, KF>;
  constexpr int ROWB = G::ROWB, RPI = G::RPI, CPR = ROWB / 16;
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4, lrow = lane / CPR, lcol = lane % CPR;

  int ft, nt;   // XCD-aware order: the channel tiles of one strip (same A rows) meet in one L2
  {
    const int t = blockIdx.x, qn = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + idx;
    ft = bid / n_ntiles;
    nt = bid - ft * n_ntiles;
  }
  const int fr = ft / G::NS, st = ft - fr * G::NS;           // frame, strip
  const int y0 = st * SR * STRIDE - 1;                       // first input row of the strip (-1: the zero row above the frame)
  const int row_lo = fr * G::P, row_base = row_lo + y0 * HW; // A row of strip pixel 0 (may lie before the frame)
  const int n0 = nt * BN, nk = K / KF;

  int voff[G::LPT];
#pragma unroll
  for (int j = 0; j < G::LPT; ++j) {
    const int r = (j * 4 + wave) * RPI + lrow;
    const int cs = lcol ^ ft_key<KF>(r);
    if ((j * 4 + wave) * RPI < G::M_PAD) {
      int row = row_base + r;                                // rows outside the frame / pad rows: any row of the frame
      row = row < row_lo ? row_lo : (row > row_lo + G::P - 1 ? row_lo + G::P - 1 : row);
      voff[j] = (int)__umul24((unsigned)row, (unsigned)(lda * 4)) + cs * 16;   // rows < 2^24, row bytes < 2^24, operand < 2 GiB (launcher)
    } else {
      const int wr = r - G::M_PAD < BN ? r - G::M_PAD : 0;
      voff[j] = (n0 + wr) * K * 4 + cs * 16;
    }
  }
  const int wn = wave % G::NT, wm = wave / G::NT;
  // the low-resolution rows under this strip (at most 6 of them: tap rows of the strip's first and last row inside the
  // frame), parked behind the ring / E image while the GEMM runs (see ups_tile_load)
  float* sG = reinterpret_cast<float*>(ring + G::lds);
  const int gy0 = ups_tap((float)(HW / 2 - 1) / (float)(HW - 1), y0 < 0 ? 0 : y0, HW / 2).i0;
  if (ups) ups_tile_load<HW / 2>(sG, ups + (size_t)fr * (HW / 2) * (HW / 2) * ld_ups + n0, ld_ups, gy0, kUpsTileBytes / (HW / 2 * 128), wave, lane);
  f32x4 acc[G::MTW];
  pw_dw_gemm<G, BN, KF>(ring, A, W1, a_bytes, w_bytes, voff, nk, wave, l15, q, acc);

  // ---- epilogue 1: + b1 (+ the upsampled addend), LeakyReLU -> the E image; rows outside the frame are zero ----
  using E = typename G::E;
  float* sE = reinterpret_cast<float*>(ring);
  e_zero_columns<E, G::RIN, G::NQ, HW>(sE, tid);
  {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(b1 + n0 + 16 * wn + 4 * q);
#pragma unroll
    for (int i = 0; i < G::MTW; ++i) {
      const int t = wm + G::WPN * i;
      const int px = 16 * t + l15;
      if (t < G::MT && px < G::M) {
        const int yl = px / HW, x = px - yl * HW, y = y0 + yl;
        const bool inside = y >= 0 && y < HW;
        f32x4 v = acc[i] + bias;
        if (ups && inside) {
          v += ups_at_lds<HW>(sG, gy0, y, x, 4 * wn + q);
        }
        *reinterpret_cast<f32x4*>(sE + E::at(yl, x, 4 * wn + q)) = inside ? lrelu4(v) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  __syncthreads();

  // ---- epilogue 2: depthwise 3x3 over the strip, + bd, LeakyReLU -> D ----
  {
    constexpr int HO = G::HO, NITEM = G::NQ * HO * G::RS;
    const int oy_first = st * SR, rows_out = HO - oy_first < SR ? HO - oy_first : SR;
    f32x4 wt[9];
    int cq_of = -1;
    for (int id = tid; id < NITEM; id += 256) {
      const int cq = id % G::NQ, rest = id / G::NQ, ox = rest % HO, r0 = (rest / HO) * G::RPS;
      if (r0 >= rows_out) continue;
      if (cq != cq_of) {
        const float* wq = wd + n0 + 4 * cq;
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(wq + (size_t)t * N);
        cq_of = cq;
      }
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bd + n0 + 4 * cq);
      const int nrows = rows_out - r0 < G::RPS ? rows_out - r0 : G::RPS;
      dw_run<E, STRIDE, HW, false>(sE, r0 * STRIDE, ox, cq, 0, wt, bv,
                                   D + (((size_t)fr * HO + oy_first + r0) * HO + ox) * ldd + n0 + 4 * cq, (size_t)HO * ldd, nrows);
    }
  }
}

template <int HW, int SR, int STRIDE, int BN, int KF>
int launch_fs(const float* a, int lda, const float* w1, const float* b1, const float* wd, const float* bd, float* d, int ldd,
              int frames, int k, int n, const float* ups, int ld_ups, hipStream_t stream) {
  using G = FSGeom<HW, SR, STRIDE, BN, KF>;
  auto kern = pw_dw_strip_kernel<HW, SR, STRIDE, BN, KF>;
  static unsigned long long attr_once = 0;
  static_assert(BN == 32 && G::lds % 16 == 0 && G::lds + kUpsTileBytes <= 160 * 1024 && (HW / 2) * 128 * 6 <= kUpsTileBytes, "upsample tile");
  if (int st = casync_ensure_dyn_lds(&attr_once, reinterpret_cast<const void*>(kern), (int)G::lds + kUpsTileBytes)) return st;
  const int n_nt = n / BN;
  const long long nwg = (long long)frames * G::NS * n_nt;
  const unsigned long long ab = ((unsigned long long)((long long)frames * G::P - 1) * lda + k) * 4, wb = (unsigned long long)n * k * 4;
  CASYNC_REQUIRE(nwg < (1ll << 31) && ab < (1ull << 31) && wb < (1ull << 31), "pw_dw: operand larger than 2 GiB");
  CASYNC_REQUIRE((long long)frames * G::P < (1ll << 24) && (long long)lda * 4 < (1ll << 24),
                 "pw_dw: %lld rows of %d floats exceed the kernel's 24-bit row arithmetic", (long long)frames * G::P, lda);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), G::lds + (ups ? kUpsTileBytes : 0), stream, a, lda, w1, b1, wd, bd, d, ldd, frames, k, n, n_nt,
                     (int)nwg, (unsigned)ab, (unsigned)wb, ups, ld_ups);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

template <int HW, int F, int BN, int KF, int S, int NST = 2>
int launch_ft(const float* a, int lda, const float* w1, const float* b1, const float* wd, const float* bd, float* d, int ldd,
              int frames, int k, int n, const float* ups, int ld_ups, hipStream_t stream) {
  using G = FTGeom<HW, F, BN, KF, S, NST>;
  auto kern = pw_dw_kernel<HW, F, BN, KF, S, NST>;
  static unsigned long long attr_once = 0;
  constexpr bool UPS_OK = F == 1 && BN == 32 && (HW / 2) * (HW / 2) * 128 <= kUpsTileBytes && G::lds % 16 == 0;   // (20x20: 12.8 KB)
  CASYNC_REQUIRE(!ups || UPS_OK, "pw_dw: no upsampled addend for %dx%d tiles of %d frames", HW, HW, F);
  if (int st = casync_ensure_dyn_lds(&attr_once, reinterpret_cast<const void*>(kern), (int)G::lds + (UPS_OK ? kUpsTileBytes : 0))) return st;
  const int n_ft = (frames + F - 1) / F, n_nt = n / BN;
  const long long nwg = (long long)n_ft * n_nt;
  const unsigned long long ab = ((unsigned long long)((long long)frames * G::P - 1) * lda + k) * 4, wb = (unsigned long long)n * k * 4;
  CASYNC_REQUIRE(nwg < (1ll << 31) && ab < (1ull << 31) && wb < (1ull << 31), "pw_dw: operand larger than 2 GiB");
  CASYNC_REQUIRE((long long)frames * G::P < (1ll << 24) && (long long)lda * 4 < (1ll << 24),
                 "pw_dw: %lld rows of %d floats exceed the kernel's 24-bit row arithmetic", (long long)frames * G::P, lda);
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), G::lds + (ups ? kUpsTileBytes : 0), stream, a, lda, w1, b1, wd, bd, d, ldd, frames, k, n, n_nt,
                     (int)nwg, (unsigned)ab, (unsigned)wb, ups, ld_ups);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

}  // namespace

bool pw_dw_supported(int hw, int cin, int cexp, int stride) {
  if (cin % 16 || cexp % 32) return false;
  if (hw == 10 || hw == 16) return stride == 1;
  return (hw == 20 || hw == 40) && (stride == 1 || stride == 2);
}

// the deep-ring instances serve launches of 2 .. `fuse_dw_deep` - 1 frames (10x10 / 16x16, stride 1, no upsampled addend):
// measured B=4 -1.2 %, B=8 -2.3 %; B=1 +1.1 %, B=11 +1.4 % (profiles/r5_ab_small_batch.txt)
// (their k-tiles are 32 input channels wide: cin % 32, or the last 16 channels would be dropped -- ADVICE r5)
bool pw_dw_deep(int hw, int frames, int stride, bool ups, int cin) {
  return hw <= 16 && stride == 1 && !ups && cin % 32 == 0 && frames >= 2 && frames < casync_opts().fuse_dw_deep;
}

const char* pw_dw_kernel_name(int hw, int cin, int frames, int stride) {
  static thread_local char buf[64];
  if (hw == 40) snprintf(buf, sizeof(buf), "pw_dw_strip_kernel<40, %d, %d, 32, 16>", stride == 1 ? 8 : 4, stride);
  else if (pw_dw_deep(hw, frames, stride, false, cin)) snprintf(buf, sizeof(buf), "pw_dw_kernel<%d, 1, 32, 32, 1, %d>", hw, hw == 10 ? 4 : 3);
  else snprintf(buf, sizeof(buf), "pw_dw_kernel<%d, %d, 32, 16, %d, 2>", hw, hw == 10 ? 2 : 1, stride);
  return buf;
}

int launch_pw_dw(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd, void* d, int ldd,
                 int frames, int hw, int stride, int cin, int cexp, hipStream_t stream, const void* ups, int ld_ups) {
  CASYNC_REQUIRE(a && w1 && b1 && wd && bd && d && frames > 0, "pw_dw: bad args");
  CASYNC_REQUIRE(pw_dw_supported(hw, cin, cexp, stride), "pw_dw: no instance for %dx%d cin=%d cexp=%d stride=%d", hw, hw, cin, cexp,
                 stride);
  CASYNC_REQUIRE(lda >= cin && lda % 4 == 0 && ldd >= cexp && ldd % 4 == 0, "pw_dw: bad leading dimensions");
  CASYNC_REQUIRE(!ups || (ld_ups >= cexp && ld_ups % 4 == 0 && (uintptr_t)ups % 16 == 0 && hw % 2 == 0), "pw_dw: bad upsampled addend");
  const float* uf = static_cast<const float*>(ups);
  CASYNC_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)d % 16) == 0 && ((uintptr_t)b1 % 16) == 0 &&
                     ((uintptr_t)wd % 16) == 0 && ((uintptr_t)bd % 16) == 0,
                 "pw_dw: pointers must be 16-B aligned");
  const float* af = static_cast<const float*>(a);
  const float* wf = static_cast<const float*>(w1);
  float* df = static_cast<float*>(d);
  // 32-channel tiles, 64-B k-tile rows: 31 KB (10x10 frame pairs), 37 KB (16x16), 55 KB (20x20) of LDS per workgroup
  if (hw == 40)   // strips of 8 (stride 2: 4) output rows: 57 KB of LDS (10 / 5 rows: 65 KB, measured equal)
    return stride == 1 ? launch_fs<40, 8, 1, 32, 16>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                       : launch_fs<40, 4, 2, 32, 16>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  // small launches (round 5): one frame per tile, 128-B k-tile rows and a four-stage (16x16: three-stage) ring -- twice
  // the workgroups, half the k-iterations, three k-tiles in flight
  const bool deep = pw_dw_deep(hw, frames, stride, ups != nullptr, cin);
  if (hw == 10)
    return deep ? launch_ft<10, 1, 32, 32, 1, 4>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                : launch_ft<10, 2, 32, 16, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  if (hw == 16)
    return deep ? launch_ft<16, 1, 32, 32, 1, 3>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream)
                : launch_ft<16, 1, 32, 16, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  if (stride == 1) return launch_ft<20, 1, 32, 16, 1>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
  return launch_ft<20, 1, 32, 16, 2>(af, lda, wf, b1, wd, bd, df, ldd, frames, cin, cexp, uf, ld_ups, stream);
}
