// Row-streaming fused inverted residual (fp32) for the plan's high-resolution blocks:
//
//   out = [x +] lrelu(W2 * lrelu(dw3x3(lrelu(W1 * x + b1)) + bd) + b2)          (reference module/unet.py:16-40)
//
// Same arithmetic and the same three phases per 16-channel chunk of the expanded tensor as ir_fused.hip
// (P1 expand GEMM over the halo -> E in LDS, P2 depthwise 3x3 -> D in LDS, P3 project GEMM into registers), but a
// workgroup does not own ONE 8x16 tile: it walks DOWN a 16-pixel-wide strip, eight output rows per step, and keeps
// the last two rows of the expanded tile (all CE channels of them, 18 KB for CE = 128) in LDS between steps.  A step
// then expands only its eight NEW halo rows -- 9 MFMA tiles of 16 pixels instead of the 12 a free-standing tile
// needs (the 1-pixel halo above and below is what the tile kernel recomputes: 25 % of its expand MFMAs).
//
// What round 2's counters said about the tile kernel (profiles/r2_mfma_busy.json, r2_ir_ablation.txt): its loop is
// issue bound -- on gfx950 an fp32 MFMA and a VALU instruction of ANY wave of a SIMD do not overlap, so a
// workgroup-tile costs 32 cycles x 512 MFMAs + ~5 cycles x 1,670 VALU instructions per wave, and the three waves a
// SIMD holds fill it completely.  A third of that VALU count was prologue / epilogue address arithmetic, masks and
// staging.  Hence, here:
//   * work is a linear list of steps (frame, strip, step in strip) cut into equal contiguous runs, one per
//     workgroup, grid = what the chip holds at once; per-lane addressing is computed ONCE per workgroup:
//     every global access of a step is   uniform 64-bit base (SALU) + per-lane constant 32-bit offset + immediate;
//   * weights arrive by LDS-DMA (buffer_load ... lds, per-lane constant offsets, the chunk offset in an SGPR):
//     no staging registers, no ds_write, no address VALU;
//   * no LDS staging of the output: the MFMA C layout (lane = pixel, four consecutive channels) stores 64-B row
//     pieces directly, the residual input is loaded in the same layout;
//   * image borders are a wave-uniform branch (only strips / steps that touch the border compute masks).
//
// Shapes: fp32, stride 1 (8 x 16 output pixels per step) or stride 2 (4 x 16), H a multiple of 8, W of 16 * stride.
// Everything else (ragged shapes, bf16) stays with ir_fused.hip.
//
// Balance.  The waves of a workgroup meet at two barriers per chunk, so what counts is the busiest wave of each
// barrier interval, not the sum.  (1) The interval [barrier 2 of chunk c, barrier 1 of chunk c+1] holds P3(c) and
// P1(c+1).  Wave 3 expands one tile more than the others (the row tails), so it takes correspondingly fewer of the
// project GEMM's (pixel tile, channel tile) units: P3Map below splits the units so that P1 + P3 MFMAs are equal.
// (2) The rows a step hands to the next one are never copied: P1 writes them straight into a carry slot and this
// step's own P2 reads them from there.  Slots rotate: chunk c of a step reads slot (c - rot) and writes slot
// (c - 1 - rot) mod (NCH + 1) -- the slot chunk c-1 consumed a whole chunk earlier -- and the next step (rot + 1)
// finds its rows where this one put them.  One slot more than chunks, no double buffer.
//
// P1 tile slots of a wave (16 pixels each, a tile never straddles a halo row, see ir_fused.hip e_off()):
//   slots 0 .. NB-1   body tiles of the NEW rows (rows KEEP .. IH-1), always
//   slot NB           wave 3: the row tails (hx >= 16 * BT) of the new rows, always
//                     waves 0, 1: the body tiles of the KEEP carried rows, wave 2: their tails -- only in a FRESH
//                     step (the first of a run or of a strip), which has nothing carried to start from
//
// STATUS (round 3): measured and NOT adopted -- compiled only with CASYNC_EXPERIMENTAL=1, like gemm_experimental.inc.
// It is bit-identical to the tile kernel (tests/test_ops_gpu.py::test_ir_stream_equals_tile_kernel), issues 19 % fewer
// MFMAs and 30 % fewer VALU instructions per output pixel, and is 3 % faster on up4's first block alone (0.267 vs
// 0.277 ms at 32 frames) -- but 1.3 % SLOWER end to end in the two-lane engine (11.25 k vs 11.40 k frames/s, three
// alternating rounds each, profiles/r3_ab_end_to_end.txt): its 52 KB of LDS and run-long workgroups pack worse with the
// other lane's kernels than the tile kernel's 6,400 short workgroups.  What the timeline says about it is in DESIGN.md.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "ir_common.h"

#ifdef CASYNC_EXPERIMENTAL

namespace {

template <int CIN, int COUT, int STRIDE>
struct SGeom {
  static constexpr int CC = 16, CE = 2 * CIN, NCH = CE / CC;
  static constexpr int TH = STRIDE == 1 ? 8 : 4, OP = TH * TW;
  static constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
  static constexpr int NEW = TH * STRIDE, KEEP = IH - NEW;   // E rows expanded per step / carried from the step above
  static constexpr int BT = IW / 16, TAIL = IW - 16 * BT;
  static constexpr int EROW = IW * CC + 4;                    // == IRGeom::EROW (e_off() assumes it)
  static constexpr int NB = NEW * BT / 4, MT1 = NB + 1;
  static constexpr int KG = CIN / 16, MT3 = OP / 64, NT3 = COUT / 16;
  static constexpr int NPX = OP / 64, NROW = (NPX - 1) * STRIDE + 3;   // P2: pixels stacked in y per thread, tap rows
  // one weight buffer: W1c [CC][CIN], W2c [COUT][CC], Wd [9][CC], b1 [CC], bd [CC]
  static constexpr int wW1 = 0, wW2 = CC * CIN, wWd = wW2 + COUT * CC, wB = wWd + 9 * CC, WBUF = wB + 2 * CC;
  // E holds rows 0 .. NEW-1 of the expanded tile only: rows NEW .. IH-1 live in a carry slot from the start
  static constexpr int oE = 0, oD = NEW * EROW, oW = oD + OP * CC, oC = oW + 2 * WBUF;
  static constexpr int CSTR = KEEP * EROW;                    // carried rows of one chunk
  static constexpr int NSLOT = NCH + 1;
  static constexpr int total = oC + NSLOT * CSTR;
  // project GEMM units (pixel tile t, channel tile n), u = t * NT3 + n, split over the waves: wave 3 carries one
  // P1 tile (4 * KG MFMAs = KG units' worth) more than the others in a carry step
  static constexpr int NTP = OP / 16, U3 = NTP * NT3;
  static constexpr int cnt3 = U3 >= 3 * KG ? (U3 - 3 * KG) / 4 : 0;
  static constexpr int rest = U3 - cnt3;
  static constexpr int ucount(int w) { return w == 3 ? cnt3 : rest / 3 + (w < rest % 3 ? 1 : 0); }
  static constexpr int ustart(int w) { return w == 0 ? 0 : ustart(w - 1) + ucount(w - 1); }
  static constexpr int MAXU = rest / 3 + (rest % 3 ? 1 : 0) > cnt3 ? rest / 3 + (rest % 3 ? 1 : 0) : cnt3;
  // LDS-DMA of one weight chunk, split EVENLY over the four waves (every wave issues the same number of
  // instructions, so the counted vmcnt waits are the same immediates for all): W1c = NA1 instructions of L1 lanes
  // per wave, W2c = one of L2 lanes, [Wd | b1 | bd] (44 x 16 B) = one of 11 lanes
  static constexpr int W1Q = CC * CIN / 4, W2Q = COUT * CC / 4;         // floats per wave
  static constexpr int NA1 = (W1Q + 255) / 256, L1 = W1Q >= 256 ? 64 : W1Q / 4, L2 = W2Q / 4;
  static constexpr int NA = NA1 + 1, NBI = 1;                           // instructions per wave: group A, group B
  static_assert(W1Q % 4 == 0 && (W1Q < 256 || W1Q % 256 == 0) && W2Q % 4 == 0 && L2 <= 64, "weight chunk pieces");
  // workgroups per CU: what LDS allows, at most 3 (a fourth wave per SIMD would need <= 128 registers: spills)
  static constexpr int occ = 160 * 1024 / (total * 4) >= 3 ? 3 : 160 * 1024 / (total * 4);
  static_assert(NEW * BT % 4 == 0 && NEW * TAIL <= 16 && KEEP * BT <= 2 && KEEP * TAIL <= 16, "tile slots");
  static_assert(NCH % 2 == 0 && NCH >= 4, "two weight buffers, parts issued up to two chunks ahead");
  static_assert(total * 4 <= 160 * 1024 && occ >= 1, "LDS budget");
  static_assert(oD % 4 == 0 && oW % 4 == 0 && oC % 4 == 0 && CSTR % 4 == 0 && WBUF % 4 == 0, "16-B aligned carve");
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16 B per lane from a buffer straight into LDS (wave-uniform LDS base + lane * 16); see gemm.hip
__device__ __forceinline__ void dma16(const void* base, unsigned bytes, float* lds, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000),
                                           (void __attribute__((address_space(3)))*)lds, 16, voff, soff, 0, 0);
#endif
}

template <int CIN, int COUT, int STRIDE, bool UPS>
__global__ __launch_bounds__(256, (SGeom<CIN, COUT, STRIDE>::occ)) void ir_stream_kernel(
    const float* __restrict__ lo, int ld_lo, int c_lo, const float* __restrict__ in, int ld_in,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ wd,
    const float* __restrict__ bd, const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ out, int ld_out, int B, int H, int W, int res, int stagger_a, int stagger_b, int prio,
    unsigned long long* __restrict__ stamps) {
  using G = SGeom<CIN, COUT, STRIDE>;
  constexpr bool RES = CIN == COUT && STRIDE == 1;   // the blocks with a residual connection (module/unet.py:14); the
  (void)res;                                         // launcher sends res != RES to the tile kernel
  // diagnostic only (null in every product call; tools/experiments/ir_timeline.py): shader cycles wave 0 of a
  // workgroup spends in the step prologues / P1 / P2 / P3 (each including the wait or barrier that ends it) / the
  // step epilogues, summed over its run; words 5 / 6 = the waits at barrier 1 / 2 (not part of P1 / P2 here);
  // word 7 = steps of the run
  unsigned long long t_mark = stamps ? __builtin_amdgcn_s_memtime() : 0, t_phase[7] = {0, 0, 0, 0, 0, 0, 0};
  auto mark = [&](int slot) __attribute__((always_inline)) {
    if (stamps) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_phase[slot] += t - t_mark;
      t_mark = t;
    }
  };
  constexpr int CC = G::CC, CE = G::CE, NCH = G::NCH, EROW = G::EROW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sE = smem + G::oE;
  float* sD = smem + G::oD;
  float* sW = smem + G::oW;
  float* sC = smem + G::oC;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int Ho = H / STRIDE, Wo = W / STRIDE;
  const int NSX = Wo / TW, NSY = Ho / G::TH;

  // ---- this workgroup's run of steps.  Workgroups b, b + 8, ... share an XCD (round-robin dispatch): give every
  //      XCD one contiguous eighth of the list, so strips that share halo columns share an L2 (speed only).
  const long long U = (long long)B * NSX * NSY;
  int g = blockIdx.x;
  const int Gn = gridDim.x;
  if ((Gn & 7) == 0) g = (g & 7) * (Gn >> 3) + (g >> 3);
  const long long u0 = U * g / Gn, u1 = U * (g + 1) / Gn;
  if (u0 >= u1) return;
  int sy = (int)(u0 % NSY);
  int sx = (int)((u0 / NSY) % NSX);
  int bi = (int)(u0 / ((long long)NSY * NSX));

  // ================= per-lane constants (once per workgroup) =================
  // P1 slots: halo pixel (ry, hx) of this lane in E-tile coordinates; slot NB depends on the wave (see header)
  int s_ry[G::MT1], s_hx[G::MT1];
  bool x_live;   // slot NB holds a real pixel for this lane
#pragma unroll
  for (int i = 0; i < G::NB; ++i) {
    const int bt = wave + 4 * i;                       // body tile of the new rows
    s_ry[i] = G::KEEP + bt / G::BT;
    s_hx[i] = 16 * (bt % G::BT) + l15;
  }
  if (wave == 3) {
    x_live = l15 < G::NEW * G::TAIL;
    s_ry[G::NB] = G::KEEP + l15 / G::TAIL;
    s_hx[G::NB] = 16 * G::BT + l15 % G::TAIL;
  } else if (wave == 2) {
    x_live = l15 < G::KEEP * G::TAIL;
    s_ry[G::NB] = l15 / G::TAIL;
    s_hx[G::NB] = 16 * G::BT + l15 % G::TAIL;
  } else {
    x_live = wave < G::KEEP * G::BT;
    s_ry[G::NB] = wave / G::BT;
    s_hx[G::NB] = 16 * (wave % G::BT) + l15;
  }
  if (!x_live) s_ry[G::NB] = 0, s_hx[G::NB] = 0;
  // float offsets: input pixel relative to the step's halo origin; where the expanded pixel goes -- relative to E for
  // rows < NEW, relative to the chunk's NEW carry slot for the rows the next step inherits (ecar)
  int voffA[G::MT1], ewr[G::MT1];
  int ecar = 0;                        // bit i: slot i of this lane goes to the carry slot
#pragma unroll
  for (int i = 0; i < G::MT1; ++i) {
    voffA[i] = (s_ry[i] * W + s_hx[i]) * ld_in + 4 * q;
    const bool car = s_ry[i] >= G::NEW;
    ecar |= car ? 1 << i : 0;
    ewr[i] = e_off<STRIDE, CC, G::IW>(car ? s_ry[i] - G::NEW : s_ry[i], s_hx[i], q) + (car ? 0 : G::oE);
  }
  // W1 fragment offsets (the MFMA A operand: row = channel l15 of the chunk, 16-B column 4g + q, swizzled)
  int w1fr[G::KG];
#pragma unroll
  for (int g4 = 0; g4 < G::KG; ++g4) w1fr[g4] = G::wW1 + xs<CIN>(l15, 16 * g4 + 4 * q);
  const int w2fr = G::wW2 + xs<CC>(l15, 4 * q);          // + 16 * n rows = + 256 * n floats (key is n-independent)
  // P2: thread = channel quad x NPX pixels stacked in y
  const int p2_c4 = (tid & 3) * 4, p2_px = (tid >> 2) & 15, p2_py0 = wave * G::NPX, p2_row0 = p2_py0 * STRIDE;
  int ebk[3];                           // tap-column offsets in this wave's first tap row of E
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) ebk[kx] = e_off<STRIDE, CC, G::IW>(p2_row0, p2_px * STRIDE + kx, p2_c4 >> 2);
  const int dwr0 = xs<CC>(p2_py0 * TW + p2_px, p2_c4);   // + 16 rows = + 256 floats per further pixel (same key)
  // P3 / epilogue: unit (t, n) = pixel tile t (output row t of the step: lane pixel l15), channels 16n + 4q .. +3
  const int dfr0 = xs<CC>(l15, 4 * q);                     // + 256 * t floats (the key does not depend on t)
  const int laneO = l15 * ld_out + 4 * q;                  // + t * Wo * ld_out + 16 * n

  // ---- weights by LDS-DMA.  Two buffers per part; every part of chunk c is requested one to two chunks before
  //      its first reader, as soon as the LAST reader of the chunk it overwrites is behind a barrier:
  //        group A(c), behind barrier 1 of chunk c:  W1[c+2] -> buffer c&1 (P1(c) was its last reader),
  //                                                   W2[c+1] -> buffer (c+1)&1 (P3(c-1) was)
  //        group B(c), behind barrier 2 of chunk c:  [Wd|b1|bd][c+2] -> buffer c&1 (P2(c) was)
  //      and awaited with COUNTED waits (loads retire in order): in front of barrier 2 of chunk c everything up to
  //      A(c-1) has landed (W2[c] for P3(c), W1[c+1] for P1(c+1)), in front of barrier 1 of chunk c everything up to
  //      B(c-2) (Wd[c] for P2(c)) -- both leave the NA + NBI youngest requests in flight.  A request has a whole chunk
  //      (~4 us) to arrive; round-3's first version gave it the length of P2 and stalled on every chunk.
  //      Other vector-memory operations (step prologue loads, residual loads, output stores) only make these waits
  //      stricter.  c runs over the chunks of ALL steps of the run; weights repeat with period NCH.
  int voffW1[G::NA1];
#pragma unroll
  for (int j = 0; j < G::NA1; ++j) {   // LDS float offset f inside W1c -> (row, 16-B slot); source column = slot ^ key
    const int f = wave * G::W1Q + j * 256 + lane * 4, row = f / CIN, slot = (f - row * CIN) >> 2;
    const int src_col4 = (xs<CIN>(row, 4 * slot) - row * CIN) >> 2;      // xs is an involution on the slot index
    voffW1[j] = (row * CIN + 4 * src_col4) * 4;
  }
  int voffW2;
  {
    const int f = wave * G::W2Q + lane * 4, row = f / CC, slot = (f - row * CC) >> 2;
    const int src_col4 = (xs<CC>(row, 4 * slot) - row * CC) >> 2;
    voffW2 = (row * CE + 4 * src_col4) * 4;
  }
  // [Wd | b1 | bd]: 11 rows of 16 floats = 44 lanes, 11 per wave: lane -> (row t, quad)
  const int wd_i = wave * 11 + (lane < 11 ? lane : 0), wd_t = wd_i >> 2;
  const float* wd_src = (wd_t < 9 ? wd + (size_t)wd_t * CE : (wd_t == 9 ? b1 : bd)) + (wd_i & 3) * 4;
  auto issue_a = [&](int c) __attribute__((always_inline)) {
    const int c2 = (c + 2) % NCH, c1 = (c + 1) % NCH;
    float* w1b = sW + (c & 1) * G::WBUF + G::wW1 + wave * G::W1Q;
    float* w2b = sW + ((c + 1) & 1) * G::WBUF + G::wW2 + wave * G::W2Q;
#pragma unroll
    for (int j = 0; j < G::NA1; ++j)
      if (G::L1 == 64 || lane < G::L1) dma16(w1, (unsigned)CE * CIN * 4, w1b + j * 256, voffW1[j], c2 * CC * CIN * 4);
    if (G::L2 == 64 || lane < G::L2) dma16(w2, (unsigned)COUT * CE * 4, w2b, voffW2, c1 * CC * 4);
  };
  auto issue_b = [&](int c) __attribute__((always_inline)) {
    const int c2 = (c + 2) % NCH;
    if (lane < 11)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wd_src + c2 * CC),
                                       (void __attribute__((address_space(3)))*)(sW + (c & 1) * G::WBUF + G::wWd + wave * 44), 16, 0, 0);
  };
  constexpr int FLIGHT = G::NA + G::NBI;       // requests of this wave that may stay in flight across a counted wait

  // UPS: per-lane x taps of the bilinear x2 upsample (align_corners=True) are step-invariant
  const int Hl = H >> 1, Wl = W >> 1;
  const float ups_sy = UPS ? (float)(Hl - 1) / (float)(H - 1) : 0.f, ups_sx = UPS ? (float)(Wl - 1) / (float)(W - 1) : 0.f;

  // prime: chunks 0 and 1 completely except W2[1] (A(-2), B(-2), A(-1), B(-1) of the scheme above)
  issue_a(-2 + NCH);   // W1[0] -> buffer 0, W2[NCH-1] -> buffer 1 (overwritten by A(0) before anyone reads it)
  issue_b(-2 + NCH);   // Wd[0] -> buffer 0
  issue_a(-1 + NCH);   // W1[1] -> buffer 1, W2[0] -> buffer 0
  issue_b(-1 + NCH);   // Wd[1] -> buffer 1
  // De-phase the workgroups.  The whole grid starts within a microsecond and every run has the same length, so
  // without this the co-resident workgroups of a CU execute the same phase at the same time for the whole kernel:
  // all in the MFMA-dense P1, then all in the LDS / VALU bound P2 with the matrix pipe idle (the tile kernel does not
  // have the problem: its 6,400 short workgroups start whenever a slot frees up).  Workgroups that share a CU
  // (blockIdx 256 apart under breadth-first dispatch) start a third of a chunk apart; a small per-workgroup skew
  // spreads the HBM bursts of the step prologues chip-wide.  Units: 64 shader cycles.
  {
    int nap = stagger_a * (int)(blockIdx.x >> 8) + stagger_b * (int)(blockIdx.x & 15);
    for (; nap > 0; nap -= 64) __builtin_amdgcn_s_sleep(64);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // the first chunk's weights have landed for everyone (later steps: issued and awaited one chunk ahead)
  bool first = true;
  int rot = 0;        // carry-slot rotation: chunk c of this step reads slot (c - rot), writes (c - 1 - rot) mod NSLOT

#pragma unroll 1
  for (long long u = u0; u < u1; ++u) {
    const bool fresh = first || sy == 0;
    first = false;
    const int y0 = sy * G::TH, x0 = sx * TW;                   // output origin of the step
    const int iy0 = y0 * STRIDE - 1, ix0 = x0 * STRIDE - 1;    // input pixel of halo (0, 0)
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + G::IH > H || ix0 + G::IW > W;
    const bool has_x = fresh || wave == 3;
    const float* inb = in + ((size_t)bi * H * W + (long long)iy0 * W + ix0) * ld_in;   // may point before the frame: only
                                                                                        // dereferenced at live pixels
    float* outb = out + ((size_t)bi * Ho * Wo + (size_t)y0 * Wo + x0) * ld_out;

    // ---- A fragments of this step's new halo rows: HBM -> registers ----
    f32x4 fa[G::MT1][G::KG];
    int okm = 0;        // bit i: slot i of this lane is a pixel inside the image
#pragma unroll
    for (int i = 0; i < G::MT1; ++i) {
      const bool slot_on = i < G::NB || has_x;
      bool ok = i < G::NB || x_live;
      if (border) {
        const int iy = iy0 + s_ry[i], ix = ix0 + s_hx[i];
        ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
      }
      okm |= ok ? 1 << i : 0;
      if (!slot_on) continue;      // wave-uniform
      if constexpr (UPS) {
        // same arithmetic as upsample2x_kernel / ATen: src = dst*(in-1)/(out-1), l1 = frac, l0 = 1-l1
        const int iy = ok ? iy0 + s_ry[i] : 0, ix = ok ? ix0 + s_hx[i] : 0;
        const UpsTap ty = ups_tap(ups_sy, iy, Hl), tx = ups_tap(ups_sx, ix, Wl);
        const float* lb = lo + (size_t)bi * Hl * Wl * ld_lo + 4 * q;
        const float* p00 = lb + ((size_t)ty.i0 * Wl + tx.i0) * ld_lo;
        const float* p01 = lb + ((size_t)ty.i0 * Wl + tx.i1) * ld_lo;
        const float* p10 = lb + ((size_t)ty.i1 * Wl + tx.i0) * ld_lo;
        const float* p11 = lb + ((size_t)ty.i1 * Wl + tx.i1) * ld_lo;
#pragma unroll
        for (int g4 = 0; g4 < G::KG; ++g4) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) {
            if (16 * g4 < c_lo) {
              const f32x4 v00 = *reinterpret_cast<const f32x4*>(p00 + 16 * g4);
              const f32x4 v01 = *reinterpret_cast<const f32x4*>(p01 + 16 * g4);
              const f32x4 v10 = *reinterpret_cast<const f32x4*>(p10 + 16 * g4);
              const f32x4 v11 = *reinterpret_cast<const f32x4*>(p11 + 16 * g4);
              v = ups_lerp(ty, tx, v00, v01, v10, v11);
            } else {
              v = *reinterpret_cast<const f32x4*>(inb + (unsigned)voffA[i] + 16 * g4);
            }
          }
          fa[i][g4] = v;
        }
      } else {
        if (border) {
#pragma unroll
          for (int g4 = 0; g4 < G::KG; ++g4)
            fa[i][g4] = ok ? *reinterpret_cast<const f32x4*>(inb + (unsigned)voffA[i] + 16 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else if (i < G::NB) {
#pragma unroll
          for (int g4 = 0; g4 < G::KG; ++g4) fa[i][g4] = *reinterpret_cast<const f32x4*>(inb + (unsigned)voffA[i] + 16 * g4);
        } else {
#pragma unroll
          for (int g4 = 0; g4 < G::KG; ++g4)
            fa[i][g4] = x_live ? *reinterpret_cast<const f32x4*>(inb + (unsigned)voffA[i] + 16 * g4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }

    f32x4 acc3[G::MAXU], xres[RES ? G::MAXU : 1];   // this wave's project-GEMM units (P3Map), the residual input of each
#pragma unroll
    for (int j = 0; j < G::MAXU; ++j) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (stamps) {
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(fa[0][0]) : "memory");   // the prologue ends when the fragments are in
      mark(0);
    }

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const float* wb = sW + (ch & 1) * G::WBUF;
      int rslot = ch - rot;
      rslot += rslot < 0 ? G::NSLOT : 0;
      const int wslot = rslot == 0 ? G::NSLOT - 1 : rslot - 1;
      // float offsets in smem (integers, so that every access below stays an LDS access for the compiler):
      const int oNew = G::oC + wslot * G::CSTR;       // rows NEW .. IH-1 of this chunk: written by P1, read by this P2
      const int oOld = G::oC + rslot * G::CSTR;       // rows 0 .. KEEP-1: what the step above wrote for this chunk
      // ---- P1: expand GEMM over the new halo rows (weights = MFMA A operand, pixels = B operand: a lane ends up
      //      with 4 consecutive channels of one pixel -> one 16-B LDS store per tile).  Bias = initial accumulator.
      {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(wb + G::wB + 4 * q);
        auto p1 = [&](auto ns_c) __attribute__((always_inline)) {
          constexpr int NS = decltype(ns_c)::value;
          f32x4 acc[NS];
#pragma unroll
          for (int i = 0; i < NS; ++i) acc[i] = bias;
#pragma unroll
          for (int g4 = 0; g4 < G::KG; ++g4) {
            const f32x4 fb = *reinterpret_cast<const f32x4*>(wb + w1fr[g4]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int i = 0; i < NS; ++i) acc[i] = mfma16(fb[s], fa[i][g4][s], acc[i]);
          }
          if (border) {   // the depthwise conv zero-pads E: halo pixels outside the image are 0, not lrelu(b1)
#pragma unroll
            for (int i = 0; i < NS; ++i)
              if (i < G::NB || x_live)
                *reinterpret_cast<f32x4*>(smem + ewr[i] + ((ecar >> i) & 1 ? oNew : 0)) =
                    (okm >> i) & 1 ? lrelu4(acc[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
#pragma unroll
            for (int i = 0; i < NS; ++i)
              if (i < G::NB || x_live) *reinterpret_cast<f32x4*>(smem + ewr[i] + ((ecar >> i) & 1 ? oNew : 0)) = lrelu4(acc[i]);
          }
        };
        if (has_x) p1(std::integral_constant<int, G::MT1>{});
        else p1(std::integral_constant<int, G::NB>{});
      }
      if (stamps) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mark(1);             // P1 compute (MFMAs, LeakyReLU, E stores)
      wait_vm<FLIGHT>();   // this wave's share of [Wd|b1|bd] of THIS chunk (group B, two chunks ago) has landed
      __syncthreads();     // E complete; every wave is done with the previous chunk's P3; Wd visible
      mark(5);             // ... and the wait for the other waves at barrier 1
      // P2 is ~60 VALU instructions; beside two other waves' MFMA streams each of them otherwise gets one issue slot
      // per 32-cycle MFMA and the phase stretches to ~2,000 cycles on the workgroup's critical path.  At raised
      // priority it issues back to back (the MFMAs it delays are delayed by the few cycles they would pay anyway).
      if (prio) __builtin_amdgcn_s_setprio(3);
      issue_a(ch);
      if (RES && ch == NCH - 1) {      // residual input in the accumulator layout, in flight under P2 / P3
        const int u_lo = wave == 0 ? G::ustart(0) : wave == 1 ? G::ustart(1) : wave == 2 ? G::ustart(2) : G::ustart(3);
        const int u_n = wave == 0 ? G::ucount(0) : wave == 1 ? G::ucount(1) : wave == 2 ? G::ucount(2) : G::ucount(3);
#pragma unroll
        for (int j = 0; j < G::MAXU; ++j)
          if (j < u_n) {
            const int t = (u_lo + j) / G::NT3, n = (u_lo + j) % G::NT3;
            // the block input at the output pixel: halo pixel (t + 1, l15 + 1)
            xres[j] = *reinterpret_cast<const f32x4*>(inb + (size_t)(t + 1) * W * ld_in + (unsigned)((l15 + 1) * ld_in + 4 * q) + 16 * n);
          }
      }

      // ---- P2: depthwise 3x3 over E -> D.  The tap rows of a wave are rows p2_row0 .. of the expanded tile: wave 0's
      //      first KEEP rows are the carried ones (the chunk's old slot; E itself in a fresh step), wave 3's last KEEP
      //      rows are the ones P1 has just put into the new slot, everything else is E.
      {
        const float* wq = wb + p2_c4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(wq + G::wB + CC);
        f32x4 a[G::NPX];
#pragma unroll
        for (int j = 0; j < G::NPX; ++j) a[j] = bv;
        auto taps = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {
          constexpr bool LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {   // one tap column at a time: only three weight vectors live
            f32x4 wt[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) wt[ky] = *reinterpret_cast<const f32x4*>(wq + G::wWd + (ky * 3 + kx) * CC);
            const float* e0 = smem + (G::oE + ebk[kx]);
            // (the same tap column, re-based from E row p2_row0 to row 0 of a slot)
            [[maybe_unused]] const float* l0 = smem + (oOld - p2_row0 * EROW + ebk[kx]);
            [[maybe_unused]] const float* h0 = smem + (oNew - p2_row0 * EROW + ebk[kx]);
            f32x4 e[G::NROW];      // all tap rows of the column in flight before the first multiply
#pragma unroll
            for (int r = 0; r < G::NROW; ++r) {
              if (LO && r < G::KEEP) e[r] = *reinterpret_cast<const f32x4*>(l0 + r * EROW);
              else if (HI && r >= G::NROW - G::KEEP) e[r] = *reinterpret_cast<const f32x4*>(h0 + (r - (G::NROW - G::KEEP)) * EROW);
              else e[r] = *reinterpret_cast<const f32x4*>(e0 + r * EROW);
            }
#pragma unroll
            for (int r = 0; r < G::NROW; ++r) {
#pragma unroll
              for (int j = 0; j < G::NPX; ++j) {
                const int ky = r - j * STRIDE;
                if (ky >= 0 && ky < 3) a[j] += e[r] * wt[ky];
              }
            }
          }
        };
        if (wave == 0 && !fresh) taps(std::true_type{}, std::false_type{});
        else if (wave == 3) taps(std::false_type{}, std::true_type{});
        else taps(std::false_type{}, std::false_type{});
#pragma unroll
        for (int j = 0; j < G::NPX; ++j) *reinterpret_cast<f32x4*>(sD + dwr0 + 256 * j) = lrelu4(a[j]);
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
      if (stamps) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mark(2);             // P2 compute
      wait_vm<FLIGHT>();   // this wave's share of group A of the previous chunk: W2 of this chunk, W1 of the next
      __syncthreads();     // D complete, those weights visible
      mark(6);             // ... and the wait at barrier 2
      issue_b(ch);

      // ---- P3: project GEMM, acc3[unit] += D[pixel tile][CC] x W2c[channel tile]^T (W2c = A operand, pixels = B
      //      operand); this wave's units, compile-time per wave (P3Map) ----
      {
        auto p3 = [&](auto w_c) __attribute__((always_inline)) {
          constexpr int WV = decltype(w_c)::value, U0 = G::ustart(WV), UN = G::ucount(WV);
          if constexpr (UN > 0) {
            constexpr int T0 = U0 / G::NT3, T1 = (U0 + UN - 1) / G::NT3;
            f32x4 fd[T1 - T0 + 1], fw[G::NT3];
#pragma unroll
            for (int t = T0; t <= T1; ++t) fd[t - T0] = *reinterpret_cast<const f32x4*>(sD + dfr0 + 256 * t);
#pragma unroll
            for (int n = 0; n < G::NT3; ++n)
              if (UN >= G::NT3 || (U0 % G::NT3 <= n && n <= (U0 + UN - 1) % G::NT3) || (U0 + UN - 1) / G::NT3 > U0 / G::NT3)
                fw[n] = *reinterpret_cast<const f32x4*>(wb + w2fr + 256 * n);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
              for (int j = 0; j < UN; ++j)
                acc3[j] = mfma16(fw[(U0 + j) % G::NT3][s4], fd[(U0 + j) / G::NT3 - T0][s4], acc3[j]);
          }
        };
        if (wave == 0) p3(std::integral_constant<int, 0>{});
        else if (wave == 1) p3(std::integral_constant<int, 1>{});
        else if (wave == 2) p3(std::integral_constant<int, 2>{});
        else p3(std::integral_constant<int, 3>{});
      }
      // no barrier here: the next P1 writes E only, which nobody reads until after its barrier
      if (stamps) asm volatile("s_nop 0" ::"v"(acc3[0]));   // keep P3's MFMAs in front of the stamp
      mark(3);
    }

    // ---- epilogue: + b2, LReLU (+ x), straight from the accumulator layout: 64-B row pieces per pixel ----
    {
      auto epi = [&](auto w_c) __attribute__((always_inline)) {
        constexpr int WV = decltype(w_c)::value, U0 = G::ustart(WV), UN = G::ucount(WV);
#pragma unroll
        for (int j = 0; j < UN; ++j) {
          constexpr int dummy = 0;
          (void)dummy;
          const int t = (U0 + j) / G::NT3, n = (U0 + j) % G::NT3;
          const f32x4 bias = *reinterpret_cast<const f32x4*>(b2 + 16 * n + 4 * q);
          f32x4 v = lrelu4(acc3[j] + bias);
          if constexpr (RES) v += xres[j];
          *reinterpret_cast<f32x4*>(outb + (size_t)t * Wo * ld_out + (unsigned)laneO + 16 * n) = v;
        }
      };
      if (wave == 0) epi(std::integral_constant<int, 0>{});
      else if (wave == 1) epi(std::integral_constant<int, 1>{});
      else if (wave == 2) epi(std::integral_constant<int, 2>{});
      else epi(std::integral_constant<int, 3>{});
    }
    rot = rot + 1 == G::NSLOT ? 0 : rot + 1;

    if (stamps) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      mark(4);
    }
    if (++sy == NSY) {
      sy = 0;
      if (++sx == NSX) sx = 0, ++bi;
    }
  }
  if (stamps && tid == 0 && blockIdx.x < 4096) {
    for (int k = 0; k < 7; ++k) stamps[(size_t)blockIdx.x * 8 + k] = t_phase[k];
    stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)(u1 - u0);
  }
  wait_vm<0>();   // the weight requests issued ahead of the (non-existent) next chunks still target this workgroup's LDS
}

template <int CIN, int COUT, int STRIDE, bool UPS>
int launch_stream(const float* lo, int ld_lo, int c_lo, const float* in, int ld_in, const float* w1, const float* b1,
                  const float* wd, const float* bd, const float* w2, const float* b2, float* out, int ld_out, int batch,
                  int h, int w, int res, hipStream_t stream) {
  using G = SGeom<CIN, COUT, STRIDE>;
  constexpr size_t lds = (size_t)G::total * sizeof(float);
  auto kern = ir_stream_kernel<CIN, COUT, STRIDE, UPS>;
  static unsigned long long attr_once = 0;
  if (int st = casync_ensure_dyn_lds(&attr_once, reinterpret_cast<const void*>(kern), (int)lds)) return st;
  const long long steps = (long long)batch * (w / STRIDE / TW) * (h / STRIDE / G::TH);
  // one run per workgroup slot of the chip; at least ir_stream_min steps per run (a run's first step has no carried
  // rows and expands all IH of them)
  const int min_steps = casync_opts().ir_stream_min > 0 ? casync_opts().ir_stream_min : 1;
  const int per_cu = casync_opts().ir_stream_wgs > 0 ? casync_opts().ir_stream_wgs : G::occ;   // > occ: several rounds
  long long grid = 256ll * per_cu;
  if (grid * min_steps > steps) grid = (steps + min_steps - 1) / min_steps;
  if (grid >= 16) grid &= ~7ll;      // whole multiples of the eight XCDs (the run -> XCD remap needs it)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, lo, ld_lo, c_lo, in, ld_in, w1, b1, wd, bd, w2, b2,
                     out, ld_out, batch, h, w, res, casync_opts().ir_stream_stagger, casync_opts().ir_stream_skew,
                     casync_opts().ir_stream_prio, casync_ir_stamps());
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

}  // namespace

// Shapes the streaming kernel takes (everything else: ir_fused.hip): fp32, stride 1, whole steps.  It exists for the two
// block shapes where it beats the tile kernel (measured, profiles/r3_ir_stream_ab.txt): 64 -> 32 (up4's first block,
// with or without the folded upsample) and 32 -> 32 at 160 x 160 (up4's second block).  For 128 -> 32 and 64 -> 64 the
// carry slots leave room for two workgroups per CU only and the tile kernel wins; those instances are not built.
// ir_stream = 2 sends every shape with an instance here (tests: small images).
bool ir_stream_supported(int cin, int cout, int stride, int h, int w, bool ups, int res) {
  if ((res != 0) != (cin == cout && stride == 1)) return false;   // the residual is compiled in or out per instance
  if (stride != 1 || h % 8 || w % 16 || cout != 32) return false;
  const bool all = casync_opts().ir_stream >= 2;
  if (cin == 64) return all || h >= 40;
  if (cin == 32 && !ups) return all || (long long)h * w >= 160 * 160;
  return false;
}

const char* ir_stream_kernel_name(int cin, int cout, int stride, bool ups) {
  static thread_local char buf[64];
  snprintf(buf, sizeof(buf), "ir_stream_kernel<%d, %d, %d, %s>", cin, cout, stride, ups ? "true" : "false");
  return buf;
}

int launch_ir_stream(const void* lo, int ld_lo, int c_lo, const void* in, int ld_in, const void* w1, const float* b1,
                     const float* wd, const float* bd, const void* w2, const float* b2, void* out, int ld_out,
                     int batch, int h, int w, int cin, int cout, int stride, int res, bool ups, hipStream_t stream) {
  CASYNC_REQUIRE(in && w1 && b1 && wd && bd && w2 && b2 && out && (!ups || lo), "ir_stream: null pointer");
  CASYNC_REQUIRE(ir_stream_supported(cin, cout, stride, h, w, ups, res), "ir_stream: no instance for cin=%d cout=%d stride=%d %dx%d",
                 cin, cout, stride, h, w);
  CASYNC_REQUIRE(batch > 0 && ld_in >= cin && ld_in % 4 == 0 && ld_out >= cout && ld_out % 4 == 0, "ir_stream: bad ld");
  CASYNC_REQUIRE(!res || (stride == 1 && cin == cout), "ir_stream: residual needs stride 1 and cin == cout");
  CASYNC_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)lo % 16) == 0 &&
                     ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)wd % 16) == 0 &&
                     ((uintptr_t)b1 % 16) == 0 && ((uintptr_t)bd % 16) == 0 && ((uintptr_t)b2 % 16) == 0,
                 "ir_stream: alignment");
  // per-lane offsets are 32-bit floats-in-frame: one frame of the widest buffer must stay below 2^31 bytes
  CASYNC_REQUIRE((long long)h * w * ld_in * 4 < (1ll << 31) && (long long)h * w * ld_out * 4 < (1ll << 31), "ir_stream: frame too large");
  if (ups) CASYNC_REQUIRE(c_lo > 0 && c_lo < cin && c_lo % 16 == 0 && ld_lo >= c_lo && ld_lo % 4 == 0, "ir_stream: bad c_lo/ld_lo");
#define S_CASE(CI, CO, S, U)                                                                                      \
  if (cin == CI && cout == CO && stride == S && ups == U)                                                         \
    return launch_stream<CI, CO, S, U>((const float*)lo, ld_lo, c_lo, (const float*)in, ld_in, (const float*)w1, b1, wd, bd, \
                                       (const float*)w2, b2, (float*)out, ld_out, batch, h, w, res, stream);
  S_CASE(64, 32, 1, true)     // up4.ir0
  S_CASE(64, 32, 1, false)
  S_CASE(32, 32, 1, false)    // up4.ir1
#undef S_CASE
  casync_set_error("ir_stream: no instance for cin=%d cout=%d stride=%d", cin, cout, stride);
  return CASYNC_ERR_ARG;
}

#else   // product build: the entry points exist, nothing routes to them

bool ir_stream_supported(int, int, int, int, int, bool, int) { return false; }
const char* ir_stream_kernel_name(int, int, int, bool) { return "ir_stream_kernel (not built)"; }
int launch_ir_stream(const void*, int, int, const void*, int, const void*, const float*, const float*, const float*, const void*,
                     const float*, void*, int, int, int, int, int, int, int, int, bool, hipStream_t) {
  casync_set_error("ir_stream: this library was built without CASYNC_EXPERIMENTAL");
  return CASYNC_ERR_STATE;
}

#endif
