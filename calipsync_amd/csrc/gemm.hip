// 1x1 convolution / linear layer as an fp32 GEMM on the gfx950 matrix cores.
//
//   C[M,N] = epilogue( A[M,K] * W[N,K]^T )
//
// Replaces nn.Conv2d(k=1) / nn.Linear + eval BatchNorm (folded into W and bias on the
// host) + LeakyReLU + residual adds of the reference (module/unet.py:17-20, 31-33, 38,
// 201-204, 227-229, 256-259, 267-269, 323-326).  Activations are NHWC so a 1x1 conv over
// B*H*W pixels IS a row-major GEMM with M = B*H*W; `lda`/`ldc` let a layer read from /
// write into a channel slice of a wider (concat) buffer, which is how torch.cat
// (module/unet.py:96, 323) disappears.
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain; 64 FLOP/clk/SIMD).
// Operand maps (cdna guide §3): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// Both tiles sit in LDS as [row][k] with k contiguous (W is stored [N][K], PyTorch's own
// conv-weight order), so one ds_read_b128 per operand feeds four MFMA k-steps: lane half
// `kh` takes k = 8g+4kh .. +3 of every group of 8 -- any k permutation is legal as long
// as A and B use the same one.  Row stride 36 floats (144 B) makes those b128 reads
// bank-conflict free (16 distinct 16-B slots per lane group).
//
// The same kernel serves bf16 activations/weights (BASELINE configs[2]): a k-tile is 128 B per
// row either way (32 floats or 64 bf16), the 16 B a lane reads per operand feed four
// v_mfma_f32_32x32x2_f32 (fp32) or one v_mfma_f32_32x32x16_bf16 (lane half kh holds k = 8kh..8kh+7
// of each 16), accumulation is fp32 and the C/D register layout is dtype-independent.
//
// Epilogue: accumulators are transposed through LDS (the A/B staging area is dead by then)
// so that bias / residual / activation run on float4 and every global access is a full
// coalesced row segment, instead of 4-B-per-lane scatter in the MFMA C layout.
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

constexpr int ROWB = 128;        // bytes of one k-tile row (32 floats / 64 bf16)
constexpr int LDSB = ROWB + 16;  // padded LDS row stride in bytes (conflict-free b128 reads)

template <int BM, int BN>
constexpr size_t gemm_lds_bytes() {
  const size_t stage = 2ull * (BM + BN) * LDSB, ctile = (size_t)BM * (BN + 4) * sizeof(float);
  return stage > ctile ? stage : ctile;
}

template <typename T> struct MfmaK;   // how one 16-B operand chunk per lane is consumed
template <> struct MfmaK<float> {
  static __device__ __forceinline__ f32x16 run(f32x4 a, f32x4 b, f32x16 c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
    return c;
  }
};
template <> struct MfmaK<bf16_t> {
  static __device__ __forceinline__ f32x16 run(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
  }
};

// Epilogue rows shared by both GEMM kernels: the fp32 C tile staged row-major in LDS ->
// bias, scaled pre-residual, LReLU, post-residual, second affine + LReLU, running-sum output,
// with one 16-B global access per thread and tensor (4 fp32 or 8 bf16 columns), rows coalesced.
// The per-column constants of the epilogue for the CPT columns a thread owns (loaded once per output tile).
template <typename T>
struct EpiCols {
  static constexpr int CPT = V16<T>::N;
  float bias[CPT], pscale[CPT], as[CPT], at[CPT];
  __device__ __forceinline__ void load(const GemmEpilogue& epi, int n) {
#pragma unroll
    for (int e = 0; e < CPT; e += 4) {
      const f32x4 b = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + n + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 p = epi.pre_scale ? *reinterpret_cast<const f32x4*>(epi.pre_scale + n + e) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 s = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_s + n + e) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 t = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_t + n + e) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) bias[e + q] = b[q], pscale[e + q] = p[q], as[e + q] = s[q], at[e + q] = t[q];
    }
  }
};

// column constant e of a thread: from the register copy, or (LITE) straight from memory / the default when absent
template <bool LITE>
__device__ __forceinline__ float col_const(const float* regs, const float* mem, float dflt, int n, int e) {
  if constexpr (LITE) return mem ? mem[n + e] : dflt;
  else return regs[e];
}

// PAD: floats of padding per staged row; SWZ: the staging tile has no padding and column bit 5 is flipped on rows
// with bit 2 set instead (the wide kernel's 32-row slabs: the two lane halves of an MFMA C register are 4 rows apart).
// LITE: only k.bias is held in registers; the rarely used per-column vectors (pre-residual scale, second affine) are
// fetched where they are used (the A-stationary kernel has 128 registers of A fragments live across its epilogue).
// EUPS: the epilogue can add a bilinearly upsampled low-resolution tensor (GemmEpilogue::ups_src).  Compiled only into
// pw_gemm_ups_kernel: its four extra row gathers per output row would set the register count -- and with it the
// occupancy -- of every other GEMM instance (measured: bf16 128x128 2 -> 1 waves/SIMD, -13 % at B=512).
template <typename T, int NT, int BM, int BN, int PAD = 4, bool SWZ = false, bool LITE = false, bool EUPS = false>
__device__ __forceinline__ void epilogue_rows_cols(const float* Cs, int m0, int n0, int M, T* __restrict__ C, int ldc,
                                                   const GemmEpilogue& epi, int tid, const EpiCols<T>& k) {
  constexpr int CPT = V16<T>::N, TPR = BN / CPT, RPP = NT / TPR, LDC_S = BN + PAD;
  static_assert(BN % CPT == 0 && NT % TPR == 0 && BM % RPP == 0, "epilogue thread map");
  const int c0 = (tid % TPR) * CPT, rr = tid / TPR, n = n0 + c0;
#pragma unroll 4
  for (int p = 0; p < BM / RPP; ++p) {
    const int row = rr + RPP * p, m = m0 + row;
    if (m >= M) break;
    V16<T> v;
#pragma unroll
    for (int e = 0; e < CPT; e += 4) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(Cs + row * LDC_S + ((c0 + e) ^ (SWZ ? ((row >> 2) & 1) << 5 : 0)));
#pragma unroll
      for (int q = 0; q < 4; ++q) v.v[e + q] = c[q] + k.bias[e + q];
    }
    if constexpr (EUPS) {
      const int hw = epi.ups_h * epi.ups_w, Hl = epi.ups_h >> 1, Wl = epi.ups_w >> 1;
      const int b = m / hw, rem = m - b * hw, y = rem / epi.ups_w, x = rem - y * epi.ups_w;
      // (the runtime division stays here: ups_scale()'s switch costs this epilogue 8 registers = its fourth wave per SIMD)
      const UpsTap ty = ups_tap((float)(Hl - 1) / (float)(epi.ups_h - 1), y, Hl);
      const UpsTap tx = ups_tap((float)(Wl - 1) / (float)(epi.ups_w - 1), x, Wl);
      const T* g = static_cast<const T*>(epi.ups_src) + (size_t)b * Hl * Wl * epi.ups_ld + n;
#pragma unroll
      for (int e = 0; e < CPT; e += 4) {
        const f32x4 r = ups_lerp(ty, tx, ld4(g + ((size_t)ty.i0 * Wl + tx.i0) * epi.ups_ld + e),
                                 ld4(g + ((size_t)ty.i0 * Wl + tx.i1) * epi.ups_ld + e),
                                 ld4(g + ((size_t)ty.i1 * Wl + tx.i0) * epi.ups_ld + e),
                                 ld4(g + ((size_t)ty.i1 * Wl + tx.i1) * epi.ups_ld + e));
#pragma unroll
        for (int q = 0; q < 4; ++q) v.v[e + q] += r[q];
      }
    }
    if (epi.pre_res) {
      const V16<T> r = ld16(static_cast<const T*>(epi.pre_res) + (size_t)m * epi.ld_pre + n);
#pragma unroll
      for (int e = 0; e < CPT; ++e) v.v[e] += col_const<LITE>(k.pscale, epi.pre_scale, 1.f, n, e) * r.v[e];
    }
    if (epi.act) {
#pragma unroll
      for (int e = 0; e < CPT; ++e) v.v[e] = lrelu(v.v[e]);
    }
    if (epi.post_res) {
      const V16<T> r = ld16(static_cast<const T*>(epi.post_res) + (size_t)m * epi.ld_post + n);
#pragma unroll
      for (int e = 0; e < CPT; ++e) v.v[e] += r.v[e];
    }
    if (epi.aff_s && !epi.aff_on_acc) {
#pragma unroll
      for (int e = 0; e < CPT; ++e)
        v.v[e] = lrelu(v.v[e] * col_const<LITE>(k.as, epi.aff_s, 1.f, n, e) + col_const<LITE>(k.at, epi.aff_t, 0.f, n, e));
    }
    st16(C + (size_t)m * ldc + n, v);
    if (epi.acc_out) {
      V16<T> sacc = ld16(static_cast<const T*>(epi.acc_in) + (size_t)m * epi.ld_acc + n);
#pragma unroll
      for (int e = 0; e < CPT; ++e) sacc.v[e] += v.v[e];
      if (epi.aff_s && epi.aff_on_acc) {
#pragma unroll
        for (int e = 0; e < CPT; ++e)
          sacc.v[e] = lrelu(sacc.v[e] * col_const<LITE>(k.as, epi.aff_s, 1.f, n, e) + col_const<LITE>(k.at, epi.aff_t, 0.f, n, e));
      }
      st16(static_cast<T*>(epi.acc_out) + (size_t)m * epi.ld_acc + n, sacc);
    }
  }
}

template <typename T, int NT, int BM, int BN, int PAD = 4, bool SWZ = false, bool EUPS = false>
__device__ __forceinline__ void epilogue_rows(const float* Cs, int m0, int n0, int M, T* __restrict__ C,
                                              int ldc, const GemmEpilogue& epi, int tid) {
  EpiCols<T> k;
  k.load(epi, n0 + (tid % (BN / V16<T>::N)) * V16<T>::N);
  epilogue_rows_cols<T, NT, BM, BN, PAD, SWZ, false, EUPS>(Cs, m0, n0, M, C, ldc, epi, tid, k);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void pw_gemm_kernel(const T* __restrict__ A, int lda,
                                                      const T* __restrict__ W, T* __restrict__ C,
                                                      int ldc, int M, int N, int K, int n_ntiles,
                                                      int nwg, GemmEpilogue epi) {
  constexpr int BK = ROWB / (int)sizeof(T);   // k elements per tile
  constexpr int E16 = 16 / (int)sizeof(T);    // elements per 16-B chunk
  constexpr int NT = 64 * WM * WN;     // threads per workgroup
  constexpr int RPS = NT / 8;          // rows staged per pass (8 lanes per 128-B row chunk)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_PASS = BM / RPS, B_PASS = BN / RPS;
  static_assert(BM % RPS == 0 && BN % RPS == 0, "tile rows must be a multiple of the staging pass");
  constexpr int LDC_S = BN + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  char* As = smem_raw;                      // [2][BM][LDSB] bytes
  char* Bs = smem_raw + 2 * BM * LDSB;      // [2][BN][LDSB]
  float* Cs = reinterpret_cast<float*>(smem_raw);  // [BM][BN+4] fp32, epilogue only

  const int tid = threadIdx.x;
  const int lrow = tid >> 3, lcb = (tid & 7) * 16;   // staging: 8 lanes cover one 128-B row (16 B each)
  const int lce = (tid & 7) * E16;                     // ... the same column in elements
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r32 = lane & 31, kh = lane >> 5;
  const int nk = K / BK;

  f32x4 ra[A_PASS], rb[B_PASS];
  const T* a_ptr[A_PASS];
  const T* b_ptr[B_PASS];

  // XCD-aware tile order: workgroups b, b+8, ... share an XCD (round-robin dispatch); give each
  // XCD a contiguous run of tiles so the N-tiles of one M-panel hit the same L2.  Bijective
  // for any tile count.  Speed only -- never correctness.
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = bid / n_ntiles;
    m0 = mt * BM;
    n0 = (bid - mt * n_ntiles) * BN;
  };
  auto set_ptrs = [&](int m0, int n0) {
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) {
      int row = m0 + lrow + RPS * p;
      row = row < M ? row : M - 1;  // tail rows re-read the last valid row; never stored
      a_ptr[p] = A + (size_t)row * lda + lce;
    }
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) b_ptr[p] = W + (size_t)(n0 + lrow + RPS * p) * K + lce;
  };
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < A_PASS; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + kt * BK);
#pragma unroll
    for (int p = 0; p < B_PASS; ++p) rb[p] = *reinterpret_cast<const f32x4*>(b_ptr[p] + kt * BK);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < A_PASS; ++p)
      *reinterpret_cast<f32x4*>(As + (buf * BM + lrow + RPS * p) * LDSB + lcb) = ra[p];
#pragma unroll
    for (int p = 0; p < B_PASS; ++p)
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + lrow + RPS * p) * LDSB + lcb) = rb[p];
  };

  // epilogue constants that do not depend on the tile row

  // Persistent over tiles: the first k-tile of the NEXT tile is fetched into registers while
  // the last k-tile of this one computes, so its HBM latency hides under the MFMAs and the
  // epilogue instead of being exposed once per tile.
  int tile = blockIdx.x;
  int m0, n0;
  tile_origin(tile, m0, n0);
  set_ptrs(m0, n0);
  gload(0);
  for (; tile < nwg; tile += gridDim.x) {
    sstore(0);
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int next = tile + gridDim.x;
    int m0n = 0, n0n = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) {
        gload(kt + 1);  // in flight under the MFMAs below
      } else if (next < nwg) {
        tile_origin(next, m0n, n0n);
        set_ptrs(m0n, n0n);
        gload(0);       // next tile's first k-tile
      }
      const char* a_base = As + (buf * BM + wm * (BM / WM) + r32) * LDSB + 16 * kh;
      const char* b_base = Bs + (buf * BN + wn * (BN / WN) + r32) * LDSB + 16 * kh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // four 32-B column pairs per 128-B row; lane half kh takes one 16-B chunk
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDSB + 32 * g);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDSB + 32 * g);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MfmaK<T>::run(fa[i], fb[j], acc[i][j]);
      }
      if (kt + 1 < nk) sstore(buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue 1: accumulators -> LDS tile (row-major) ----
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float* cbase = Cs + (wm * (BM / WM) + i * 32 + 4 * kh) * LDC_S + wn * (BN / WN) + j * 32 + r32;
#pragma unroll
        for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * LDC_S] = acc[i][j][r];
      }
    __syncthreads();

    // ---- epilogue 2: bias, residuals, activation, coalesced 16-B stores ----
    epilogue_rows<T, NT, BM, BN>(Cs, m0, n0, M, C, ldc, epi, tid);
    __syncthreads();  // the C tile aliases the staging buffers the next tile is about to fill
    m0 = m0n;
    n0 = n0n;
  }
}

// Stream-K split of a launch of `tiles` output tiles with `nk` k-iterations each (ring kernel).
// Whole rounds of 256 tiles stay data-parallel; the remainder's iterations are cut into equal runs,
// one per CU, so the last round costs remainder/256 of a tile time instead of a whole one
// (M = 6400 rows leave 78-89 % of a round empty otherwise; measured 83 -> 105 TFLOP/s class).
struct StreamKSplit { long long dp_tiles; int wgs, per; };
inline int stream_k_mode() { return casync_opts().gemm_streamk; }   // 0 gives batch-invariant bits (tests)
inline StreamKSplit stream_k_split(long long tiles, int nk, int tile_floats, bool have_scratch) {
  StreamKSplit s{tiles, 0, 0};
  const long long rem = tiles % kStreamKWgs;
  if (rem > 224) return s;   // a nearly full last round: the fix-up would cost more than it saves
  // needs scratch, a remainder worth balancing, tiles that fit the scratch slots and enough
  // k-iterations per run to keep the ring busy
  if (!have_scratch || !stream_k_mode() || rem == 0 || nk < 8 ||
      (long long)tile_floats * 2 * kStreamKWgs > kStreamKFloats)
    return s;
  const long long its = rem * nk;
  long long wgs = its / 4 < kStreamKWgs ? its / 4 : kStreamKWgs;   // >= 4 iterations per run
  if (wgs < rem) wgs = rem;                                        // per <= nk: at most two tiles per run
  if (wgs <= rem) return s;                                        // nothing to split
  s.dp_tiles = tiles - rem;
  s.per = (int)((its + wgs - 1) / wgs);
  s.wgs = (int)((its + s.per - 1) / s.per);
  return s;
}

// =====================================================================================
// Direct-to-LDS variant: the k-tiles arrive by global_load_lds_dwordx4 (no register staging, no
// ds_write) into an NST-deep ring, NST-1 tiles ahead of the MFMAs.  A bf16 k-tile is only ~512 MFMA
// cycles per wave, less than the L2/HBM latency, so the one-tile register prefetch of the kernel
// above cannot cover it; the ring can.  An LDS-DMA wave-instruction writes base + lane*16 linearly
// (8 rows x 128 B), so rows are unpadded and the 16-B columns are XOR-swizzled through the
// per-lane SOURCE address: LDS column p of row r holds global column p ^ ((r>>1)&7); readers
// apply the same key (16 consecutive rows of one column then hit 16 different bank slots).
// Sync per k-tile: counted s_waitcnt vmcnt (this wave's part of tile kt has landed) -> raw
// s_barrier (everyone's part has landed, and everyone is done reading the stage about to be
// refilled) -> issue tile kt+NST-1 -> MFMAs on tile kt.  One barrier per k-tile, loads stay in
// flight across it.
// =====================================================================================
// 16 B per lane from a buffer straight into LDS: `buffer_load_dwordx4 v, s[rsrc], s_off offen lds`.  Per-lane byte
// offset in a VGPR, a wave-uniform byte offset in an SGPR, LDS destination = wave-uniform base + lane * 16.  (The
// descriptor type only exists in the device pass, hence the guard: the host pass sees an empty body.)
__device__ __forceinline__ void buffer_load_lds16(const void* base, unsigned bytes, void __attribute__((address_space(3)))* lds,
                                                  int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000), lds,
                                           16, voff, soff, 0, 0);
#endif
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 256 B of zeros: where the implicit-GEMM convolution points its LDS-DMA loads for taps outside the image
__device__ __attribute__((aligned(256))) float g_zero_page[64];

// waves per SIMD the register allocator must leave room for: the 64x64 two-stage ring needs 32 KB of LDS, so five
// workgroups fit a CU -- if the kernel stays within 512 / 5 = 102 registers
template <typename T, int BM, int BN, int NST, bool CONV>
constexpr int glds_min_waves() {
  if (sizeof(T) == 2 && BM == 128 && BN == 128 && NST == 2 && !CONV) return 2;   // round-6 experiment: two 64-KB rings per CU
  return sizeof(T) == 4 && BM == 64 && BN == 64 && NST == 2 && !CONV ? 5 : 1;
}

template <typename T, int BM, int BN, int WM, int WN, int NST, bool CONV, bool EUPS>
__device__ __forceinline__ void glds_body(const T* __restrict__ A, int lda, const T* __restrict__ W, T* __restrict__ C,
                                          int ldc, int M, int N, int K, int n_ntiles, int nwg, int dp_tiles, int sk_wgs,
                                          int sk_per, const GemmEpilogue& epi) {
  // WM x WN waves tile the output; with two of them (the 64x32 tile of small-M launches) the other factor of two splits
  // the K of every k-tile: waves 0..1 take its first two 32-B column pairs, waves 2..3 the last two, and the epilogue
  // adds the halves in LDS.  A launch with M = 800 rows (B = 8 at 10x10) has 100-200 64x64 tiles for 256 CUs and one
  // workgroup per CU, so nothing hides the ~0.45 us an LDS-DMA k-tile takes to arrive behind a two-stage ring: every
  // tile shape ran at 0.45 us per k-iteration (tools/experiments/small_m_gemm.sh).  Half-size tiles fill the chip and
  // a four-stage ring keeps three k-tiles in flight.
  static_assert(WM * WN == 4 || WM * WN == 2, "4 waves");
  constexpr int WK = 4 / (WM * WN);
  static_assert(WK == 1 || (NST >= 3 && !CONV && !EUPS && sizeof(T) == 4), "the K split exists in the fp32 multi-stage loop only");
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int E16 = 16 / (int)sizeof(T);
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * ROWB;          // bytes per ring stage
  constexpr int LPT = ROWS / 32;              // LDS-DMA instructions per wave per k-tile (8 rows each, 4 waves)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDC_S = BN + 4;
  static_assert((size_t)NST * STAGE >= (size_t)(BM / WM) * LDC_S * sizeof(float), "ring must hold one wave-row block of C");
  static_assert(LPT * (NST - 1) < 64, "vmcnt range");   // 6-bit counter
  extern __shared__ __attribute__((aligned(16))) char ring[];
  float* Cs = reinterpret_cast<float*>(ring);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WM * WN), wmn = wave - wk * (WM * WN);
  const int wm = wmn / WN, wn = wmn - wm * WN;
  const int r32 = lane & 31, kh = lane >> 5;
  const int nk = K / BK;
  const int lrow8 = lane >> 3, lcol = lane & 7;

  // epilogue constants

  f32x16 acc[TM][TN];
  int m0 = 0, n0 = 0;
  auto tile_origin = [&](int tile) {   // XCD-aware bijection tile -> (m0, n0)
    const int q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = bid / n_ntiles;
    m0 = mt * BM;
    n0 = (bid - mt * n_ntiles) * BN;
  };
  auto zero_acc = [&] {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // acc = A[m0.., k-tiles k0 .. k0+cnt) . W[n0.., same k-tiles)^T; ends with the ring idle
  auto run_k = [&](int k0, int cnt) {
    // this lane's source pointer for each of the wave's LPT instructions (k-tile k0)
    const T* src[LPT];
    int tapmask[CONV ? LPT : 1];   // CONV: which of the 9 taps of this row's pixel are inside the image
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int r = (j * 4 + wave) * 8 + lrow8;             // row inside the stage: [0,BM) = A, [BM,ROWS) = W
      const int cs = lcol ^ ((r >> 1) & 7);                 // swizzled source column
      if (r < BM) {
        int row = m0 + r;
        row = row < M ? row : M - 1;
        if constexpr (CONV) {
          // row = output pixel (b, oy, ox); src = its tap (0,0) -- possibly outside the buffer, only
          // dereferenced for taps the mask marks valid
          const int hw = epi.conv_ho * epi.conv_wo;
          const int b = row / hw, rem = row - b * hw;
          const int oy = rem / epi.conv_wo, ox = rem - oy * epi.conv_wo;
          const int iy0 = oy * epi.conv_stride - epi.conv_pad, ix0 = ox * epi.conv_stride - epi.conv_pad;
          src[j] = A + (((long long)b * epi.conv_h + iy0) * epi.conv_w + ix0) * (long long)epi.conv_c + cs * E16;
          int mk = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int iy = iy0 + t / 3, ix = ix0 + t % 3;
            mk |= (iy >= 0 && iy < epi.conv_h && ix >= 0 && ix < epi.conv_w) ? 1 << t : 0;
          }
          tapmask[j] = mk;
        } else {
          src[j] = A + (size_t)row * lda + cs * E16 + (size_t)k0 * BK;
        }
      } else {
        src[j] = W + (size_t)(n0 + r - BM) * K + cs * E16 + (size_t)k0 * BK;
      }
    }
    // LDS-DMA pieces [LPT*part/4, LPT*(part+1)/4) of k-tile kt (part 0..3; issue() = all four)
    auto issue_part = [&](int kt, int part) {
      char* st = ring + (kt % NST) * STAGE;
      [[maybe_unused]] int tap = 0, tap_off = 0;
      if constexpr (CONV) {   // k-tile -> (tap, channel offset): C / BK k-tiles per tap
        const int kpt = epi.conv_c / BK, kg = k0 + kt;
        tap = kg / kpt;
        tap_off = ((tap / 3) * epi.conv_w + tap % 3) * epi.conv_c + (kg - tap * kpt) * BK;
      }
#pragma unroll
      for (int j = 0; j < LPT; ++j) {
        if (j < LPT * part / 4 || j >= LPT * (part + 1) / 4) continue;
        const T* p;
        if (CONV && j < BM / 32) {   // rows of the A part (compile-time per j: r < BM <=> j < BM/32)
          const int r = (j * 4 + wave) * 8 + lrow8;
          p = (tapmask[CONV ? j : 0] >> tap) & 1 ? src[j] + tap_off
                                                 : reinterpret_cast<const T*>(g_zero_page) + (lcol ^ ((r >> 1) & 7)) * E16;
        } else {
          p = src[j] + (size_t)kt * BK;
        }
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p,
                                         (void __attribute__((address_space(3)))*)(st + (j * 4 + wave) * 8 * ROWB),
                                         16, 0, 0);
      }
    };
    auto issue = [&](int kt) {
#pragma unroll
      for (int part = 0; part < 4; ++part) issue_part(kt, part);
    };
    zero_acc();
    // fragment addressing: row byte offset and swizzle key are loop-invariant per lane
    int a_off[TM], a_key[TM], b_off[TN], b_key[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * (BM / WM) + i * 32 + r32;
      a_off[i] = r * ROWB;
      a_key[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int r = BM + wn * (BN / WN) + j * 32 + r32;
      b_off[j] = r * ROWB;
      b_key[j] = (r >> 1) & 7;
    }
    // one 32-B column pair (g) of the tile in stage `st`: the 16 B this lane feeds to four MFMA k-steps
    auto frag = [&](const char* st, int g, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + (((2 * g + kh) ^ a_key[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + (((2 * g + kh) ^ b_key[j]) << 4));
    };
    auto mma = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MfmaK<T>::run(fa[i], fb[j], acc[i][j]);
    };

    if constexpr (NST >= 3) {
      // Software-pipelined loop over an NST-stage ring.  Invariant behind the barrier at the top of k-tile
      // kt: tiles kt AND kt+1 have landed for every wave (each wave waited for its own LDS-DMA first:
      // counted vmcnt, tiles kt+2 .. kt+NST-2 stay in flight), and every wave is done reading the stage of
      // tile kt-1.  So inside iteration kt
      //   * the loads of tile kt+NST-1 go into the stage of tile kt-1, issued BETWEEN the MFMA groups (an
      //     LDS-DMA instruction costs 60-180 issue cycles: in front of the MFMAs that was a bubble),
      //   * the fragments of the next column pair -- for the last pair: of tile kt+1's first pair, which
      //     this barrier already covers -- are read while the current pair's MFMAs run,
      // and the only thing left between two k-tiles' MFMAs is the barrier itself.  A tile has NST-2
      // iterations to land before it is waited for (an LDS-DMA load takes 1-2.5 us under load, one
      // iteration 0.4-1.7 us: tools/experiments/gemm_timeline.py).
      constexpr int AHEAD = NST - 1;            // tiles issued ahead of the one being computed
#pragma unroll
      for (int t = 0; t < AHEAD; ++t)
        if (t < cnt) issue(t);
      f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
      // cold start: tile 0 (and tile 1) landed, the first column pair of tile 0 in registers
      if (cnt >= AHEAD) wait_vmcnt<(NST - 3) * LPT>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int g0 = WK == 1 ? 0 : 2 * wk;   // K split: this wave's column pairs are g0, g0 + 1
      frag(ring, g0, fa0, fb0);
      // straight-line body (no branches: MORE / NEXT are compile-time), so the scheduler hints below hold
      // and the waitcnt pass sees one basic block per k-tile
      constexpr int NM = TM * TN * (sizeof(T) == 4 ? 4 : 1);   // MFMAs per column pair
      constexpr int ND = TM + TN;                               // ds_read_b128 per column pair
      auto body = [&](int kt, auto more_c, auto next_c, auto flight_c) {
        constexpr bool MORE = decltype(more_c)::value, NEXT = decltype(next_c)::value;
        // tiles <= kt+1 landed; FLIGHT later tiles (kt+2 ...) stay in flight
        wait_vmcnt<decltype(flight_c)::value * LPT>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = ring + (kt % NST) * STAGE;
        const char* st_next = ring + ((kt + 1) % NST) * STAGE;
        if constexpr (WK == 2) {
          frag(st, g0 + 1, fa1, fb1);
          if constexpr (MORE) issue_part(kt + AHEAD, 0), issue_part(kt + AHEAD, 1);
          mma(fa0, fb0);
          if constexpr (NEXT) frag(st_next, g0, fa0, fb0);
          if constexpr (MORE) issue_part(kt + AHEAD, 2), issue_part(kt + AHEAD, 3);
          mma(fa1, fb1);
          return;
        }
        frag(st, 1, fa1, fb1);
        if constexpr (MORE) issue_part(kt + AHEAD, 0);
        mma(fa0, fb0);
        frag(st, 2, fa0, fb0);
        if constexpr (MORE) issue_part(kt + AHEAD, 1);
        mma(fa1, fb1);
        frag(st, 3, fa1, fb1);
        if constexpr (MORE) issue_part(kt + AHEAD, 2);
        mma(fa0, fb0);
        if constexpr (NEXT) frag(st_next, 0, fa0, fb0);
        if constexpr (MORE) issue_part(kt + AHEAD, 3);
        mma(fa1, fb1);
        // per column pair: one MFMA (its operands were read a whole pair ago, so the wait in front of it
        // covers no fresh read), then the next pair's LDS reads, this quarter of the LDS-DMA pieces, the
        // rest of the MFMAs
#define CASYNC_GEMM_HINT(G)                                                                                    \
  do {                                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                           \
    if constexpr ((G) < 3 || NEXT) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                          \
    if constexpr (MORE && (LPT * ((G) + 1) / 4 - LPT * (G) / 4) > 0)                                           \
      __builtin_amdgcn_sched_group_barrier(0x20, LPT * ((G) + 1) / 4 - LPT * (G) / 4, 0);                      \
    if constexpr (NM > 1) __builtin_amdgcn_sched_group_barrier(0x8, NM - 1, 0);                                \
  } while (0)
        CASYNC_GEMM_HINT(0);
        CASYNC_GEMM_HINT(1);
        CASYNC_GEMM_HINT(2);
        CASYNC_GEMM_HINT(3);
#undef CASYNC_GEMM_HINT
      };
      using std::integral_constant;
      int kt = 0;
      for (; kt + AHEAD < cnt; ++kt) body(kt, std::true_type{}, std::true_type{}, integral_constant<int, NST - 3>{});
      // drain: `left` tiles follow kt, all issued; left - 1 of them may stay in flight
      for (; kt + 1 < cnt; ++kt) {
        const int left = cnt - 1 - kt;
        if (NST >= 6 && left >= 4) body(kt, std::false_type{}, std::true_type{}, integral_constant<int, NST >= 6 ? 3 : 0>{});
        else if (NST >= 5 && left == 3) body(kt, std::false_type{}, std::true_type{}, integral_constant<int, NST >= 5 ? 2 : 0>{});
        else if (NST >= 4 && left == 2) body(kt, std::false_type{}, std::true_type{}, integral_constant<int, NST >= 4 ? 1 : 0>{});
        else body(kt, std::false_type{}, std::true_type{}, integral_constant<int, 0>{});
      }
      body(kt, std::false_type{}, std::false_type{}, integral_constant<int, 0>{});
    } else if constexpr (!CONV) {
      // Two-stage loop with NO vector-ALU instruction in it.  On gfx950 an fp32 MFMA and a VALU instruction of ANY
      // wave on the same SIMD do not overlap (tools/experiments/mfma_valu_overlap.hip: times add, ~5 cycles per
      // wave-instruction), so the ~27 address instructions per k-tile of the plain loop (64-bit pointer bumps,
      // readfirstlane for M0, LDS address adds) cost 13 % of a 64x64 tile's 1,024 MFMA cycles whatever the
      // occupancy.  Here the loads are buffer_load ... lds: per-lane offset fixed for the whole tile, the k-tile
      // offset in an SGPR, the LDS destination wave-uniform (SALU); the fragment addresses are per-lane constants
      // and the stage offset is an instruction immediate (the loop is unrolled over the two stages).
      const int wave_s = __builtin_amdgcn_readfirstlane(wave);
      int voff[LPT];
#pragma unroll
      for (int j = 0; j < LPT; ++j) {
        const int r = (j * 4 + wave) * 8 + lrow8;
        const int cs = lcol ^ ((r >> 1) & 7);
        if (j < BM / 32) {
          int row = m0 + r;
          row = row < M ? row : M - 1;
          voff[j] = (int)(((long long)row * lda) * (int)sizeof(T)) + cs * 16;
        } else {
          voff[j] = (int)(((long long)(n0 + r - BM) * K) * (int)sizeof(T)) + cs * 16;
        }
      }
      auto issue2 = [&](int kt, int stage) {
        const int soff = (k0 + kt) * ROWB;
#pragma unroll
        for (int j = 0; j < LPT; ++j)
          buffer_load_lds16(j < BM / 32 ? static_cast<const void*>(A) : static_cast<const void*>(W),
                            j < BM / 32 ? epi.buf_a_bytes : epi.buf_w_bytes,
                            (void __attribute__((address_space(3)))*)(ring + stage * STAGE + (j * 4 + wave_s) * 8 * ROWB), voff[j], soff);
      };
      // per-lane fragment addresses of the four column pairs (stage 0); stage 1 = + STAGE, an immediate
      const char* a_ptr[TM][4];
      const char* b_ptr[TN][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_ptr[i][g] = ring + a_off[i] + (((2 * g + kh) ^ a_key[i]) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) b_ptr[j][g] = ring + b_off[j] + (((2 * g + kh) ^ b_key[j]) << 4);
      }
      auto tile_mma = [&](auto stage_c) {
        constexpr int SOFF = decltype(stage_c)::value * STAGE;
        // the next column pair's fragments are read while the current pair's MFMAs run (two register sets)
        f32x4 fa[2][TM], fb[2][TN];
        auto rd = [&](int g, f32x4 (&xa)[TM], f32x4 (&xb)[TN]) {
#pragma unroll
          for (int i = 0; i < TM; ++i) xa[i] = *reinterpret_cast<const f32x4*>(a_ptr[i][g] + SOFF);
#pragma unroll
          for (int j = 0; j < TN; ++j) xb[j] = *reinterpret_cast<const f32x4*>(b_ptr[j][g] + SOFF);
        };
        rd(0, fa[0], fb[0]);
        rd(1, fa[1], fb[1]);
        mma(fa[0], fb[0]);
        rd(2, fa[0], fb[0]);
        mma(fa[1], fb[1]);
        rd(3, fa[1], fb[1]);
        mma(fa[0], fb[0]);
        mma(fa[1], fb[1]);
        // keep that order (the scheduler otherwise sinks every read next to its use)
        constexpr int NM = TM * TN * (sizeof(T) == 4 ? 4 : 1), ND = TM + TN;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, NM, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 2 * NM, 0);
      };
      issue2(0, 0);
      for (int kt = 0; kt < cnt; kt += 2) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < cnt) issue2(kt + 1, 1);
        tile_mma(std::integral_constant<int, 0>{});
        if (kt + 1 < cnt) {
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (kt + 2 < cnt) issue2(kt + 2, 0);
          tile_mma(std::integral_constant<int, 1>{});
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NST - 1; ++s)
        if (s < cnt) issue(s);
      for (int kt = 0; kt < cnt; ++kt) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < cnt) issue(kt + NST - 1);
        const char* st = ring + (kt % NST) * STAGE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 fa[TM], fb[TN];
          frag(st, g, fa, fb);
          mma(fa, fb);
        }
      }
    }
    __syncthreads();   // everything landed and consumed: the ring is free (it becomes the C tile)
  };

  // acc -> C tile in LDS -> bias / residuals / activation -> coalesced store; ends with the ring idle
  // One wave-row block (BM / WM rows) at a time, so the staging area is BM / WM x (BN + 4) floats and a
  // two- or three-stage ring of any tile shape can hold it.
  constexpr bool PREF64 = WK == 1 && sizeof(T) == 4 && BM == 64 && BN == 64 && NST == 2 && !CONV && !EUPS;
  [[maybe_unused]] EpiCols<T> kcols;
  auto epilogue = [&] {
#pragma unroll
    for (int h = 0; h < WM; ++h) {
      if (wm == h && wk == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float* cbase = Cs + (i * 32 + 4 * kh) * LDC_S + wn * (BN / WN) + j * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * LDC_S] = acc[i][j][r];
          }
      }
      __syncthreads();
      if constexpr (WK == 2) {   // the other half of K
        if (wm == h && wk == 1) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              float* cbase = Cs + (i * 32 + 4 * kh) * LDC_S + wn * (BN / WN) + j * 32 + r32;
#pragma unroll
              for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * LDC_S] += acc[i][j][r];
            }
        }
        __syncthreads();
      }
      if constexpr (WK == 2) epilogue_rows_cols<T, 256, BM / WM, BN, 4, false, false, EUPS>(Cs, m0 + h * (BM / WM), n0, M, C, ldc, epi, tid, kcols);
      else if (PREF64) epilogue_rows_cols<T, 256, BM / WM, BN, 4, false, false, EUPS>(Cs, m0 + h * (BM / WM), n0, M, C, ldc, epi, tid, kcols);
      else epilogue_rows<T, 256, BM / WM, BN, 4, false, EUPS>(Cs, m0 + h * (BM / WM), n0, M, C, ldc, epi, tid);
      __syncthreads();   // staging consumed before the next block / the next loads overwrite the ring
    }
  };

  // ---- stream-K part: the k-iterations of the remaining tiles, cut into sk_wgs equal runs of
  //      sk_per (<= nk, so a run touches at most two tiles).  A run that does not cover its tile
  //      parks the partial accumulators in the scratch, adds its iteration count to the tile's
  //      counter, and the run that completes the count sums the partials IN RUN ORDER (the result
  //      does not depend on arrival order) and applies the epilogue.  Nobody waits on anybody.
  //      The runs go to the LAST workgroups of the grid (the ones with the fewest whole tiles) and
  //      are done BEFORE the whole tiles, so the parking / fix-up latency hides under the
  //      data-parallel work of the co-resident workgroups instead of forming the tail.
  const int g = (int)gridDim.x - 1 - (int)blockIdx.x;
  if (g < sk_wgs) {
  const long long it_total = (long long)(nwg - dp_tiles) * nk;
  long long it = (long long)g * sk_per;
  const long long it_end = it + sk_per < it_total ? it + sk_per : it_total;
  constexpr int SLOT = BM * BN;   // floats per parked tile = TM*TN*16 per thread
  for (int seg = 0; it < it_end; ++seg) {
    const int t = (int)(it / nk);
    const int k0 = (int)(it - (long long)t * nk);
    const int cnt = (int)((long long)(nk - k0) < it_end - it ? (long long)(nk - k0) : it_end - it);
    it += cnt;
    tile_origin(dp_tiles + t);
    if (PREF64) kcols.load(epi, n0 + (tid % (BN / V16<T>::N)) * V16<T>::N);   // (as in the data-parallel part below)
    run_k(k0, cnt);
    if (cnt != nk) {
      float* slot = epi.sk_ws + (size_t)(2 * g + seg) * SLOT + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(slot + ((i * TN + j) * 16 + r) * 256, acc[i][j][r], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      // The eight XCDs have private L2s.  A device-scope fence would write back and invalidate the
      // whole L2 (measured: +65 us per launch); instead the parked words themselves are agent-scope
      // accesses (write-through stores, L2-bypassing loads) ordered by vmcnt(0) + the barrier in
      // front of the counter update.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* flag = reinterpret_cast<int*>(ring);
      if (tid == 0)
        *flag = __hip_atomic_fetch_add(epi.sk_cnt + t, (unsigned)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
                    (unsigned)cnt == (unsigned)nk;
      __syncthreads();
      const bool last = *flag != 0;
      __syncthreads();   // the flag word is ring memory: read before anything is loaded over it
      if (!last) continue;
      const int g0 = (int)(((long long)t * nk) / sk_per), g1 = (int)(((long long)t * nk + nk - 1) / sk_per);
      zero_acc();
      for (int gg = g0; gg <= g1; ++gg) {
        const int sg = (int)(((long long)gg * sk_per) / nk) == t ? 0 : 1;   // t is run gg's first or second tile
        const float* part = epi.sk_ws + (size_t)(2 * gg + sg) * SLOT + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += __hip_atomic_load(part + ((i * TN + j) * 16 + r) * 256, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) __hip_atomic_store(epi.sk_cnt + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zeroed for the next launch
    }
    epilogue();
  }
  }

  // ---- data-parallel part: whole tiles, every workgroup strides over them ----
  // Diagnostic stamps (epi.stamps, null in every product call: tools/experiments/gemm_timeline.py sets
  // them): 100 MHz wall clock at entry / after the k loop / after the epilogue of each of the first
  // two tiles, and at exit.
  auto stamp = [&](int slot) {
    if (epi.stamps && tid == 0) epi.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  // slot 7: shader-clock cycles (s_memtime) spent between entry and exit -> clock = cycles / (exit - entry)
  const unsigned long long cyc0 = epi.stamps ? __builtin_amdgcn_s_memtime() : 0;
  stamp(0);
  int done = 0;
  for (int tile = blockIdx.x; tile < dp_tiles; tile += gridDim.x, ++done) {
    tile_origin(tile);
    // small-M instance: one workgroup per CU, nobody to hide the epilogue's column constants behind -- request them
    // before the k loop (the other instances keep their registers: five 64x64 workgroups per CU need <= 102)
    if constexpr (WK == 2) kcols.load(epi, n0 + (tid % (BN / V16<T>::N)) * V16<T>::N);
    // single-lane plan, 64x64 fp32: same reason (B=31: 2.86 -> 2.81 ms).  With these 16 registers live across the k loop the
    // compiler re-derives the eight fragment addresses of the two-stage loop once per pair of k-tiles (8 v_add_u32 per 32
    // MFMAs: the loop is no longer free of vector instructions); the two-lane B=64 schedule, which never takes this
    // branch, measured +0.7 % with that code (profiles/r4_ab_small_batch.txt), so it stays.
    else if (PREF64) kcols.load(epi, n0 + (tid % (BN / V16<T>::N)) * V16<T>::N);
    run_k(0, nk);
    if (done < 2) stamp(1 + 2 * done);
    epilogue();
    if (done < 2) stamp(2 + 2 * done);
  }
  stamp(5);
  if (epi.stamps && tid == 0) {
    epi.stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)done;
    epi.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - cyc0;
  }
}

template <typename T, int BM, int BN, int WM, int WN, int NST, bool CONV = false>
__global__ __launch_bounds__(256, (glds_min_waves<T, BM, BN, NST, CONV>())) void pw_gemm_glds_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, T* __restrict__ C, int ldc, int M, int N, int K, int n_ntiles,
    int nwg, int dp_tiles, int sk_wgs, int sk_per, GemmEpilogue epi) {
  glds_body<T, BM, BN, WM, WN, NST, CONV, false>(A, lda, W, C, ldc, M, N, K, n_ntiles, nwg, dp_tiles, sk_wgs, sk_per, epi);
}
// The same kernel with the upsampled addend in its epilogue (GemmEpilogue::ups_src): the expand conv of an Up block
// whose upsampled half was computed at the low resolution (engine.hip decode(), ups_commute).
template <typename T, int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(256) void pw_gemm_ups_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, T* __restrict__ C, int ldc, int M, int N, int K, int n_ntiles,
    int nwg, int dp_tiles, int sk_wgs, int sk_per, GemmEpilogue epi) {
  glds_body<T, BM, BN, WM, WN, NST, false, true>(A, lda, W, C, ldc, M, N, K, n_ntiles, nwg, dp_tiles, sk_wgs, sk_per, epi);
}

template <typename T, int BM, int BN, int WM, int WN, int NST, bool CONV = false, bool EUPS = false>
int launch_glds_t(const T* a, int lda, const T* w, T* c, int ldc, int m, int n, int k,
                  const GemmEpilogue& epi, hipStream_t stream, bool use_sk) {
  constexpr size_t lds = (size_t)NST * (BM + BN) * ROWB;
  static unsigned long long attr_once = 0;
  auto kern = [] {
    if constexpr (EUPS) return pw_gemm_ups_kernel<T, BM, BN, WM, WN, NST>;
    else return pw_gemm_glds_kernel<T, BM, BN, WM, WN, NST, CONV>;
  }();
  if (int st = casync_ensure_dyn_lds(&attr_once, reinterpret_cast<const void*>(kern), (int)lds)) return st;
  const int n_mtiles = (m + BM - 1) / BM, n_ntiles = n / BN;
  const long long nwg = (long long)n_mtiles * n_ntiles;
  CASYNC_REQUIRE(nwg < (1ll << 31), "gemm grid too large");
  constexpr int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
  const long long cap = 256ll * per_cu;
  const StreamKSplit sk = stream_k_split(nwg, k / (ROWB / (int)sizeof(T)), BM * BN, epi.sk_ws != nullptr && use_sk && WM * WN == 4);
  const long long want = sk.dp_tiles + sk.wgs;
  const unsigned grid = (unsigned)(want > cap ? cap : want);
  GemmEpilogue e2 = epi;
  if constexpr (!CONV) {   // extents for the buffer-addressed loads of the two-stage loop (32-bit offsets)
    const unsigned long long ab = ((unsigned long long)(m - 1) * lda + k) * sizeof(T), wb = (unsigned long long)n * k * sizeof(T);
    CASYNC_REQUIRE(ab < (1ull << 31) && wb < (1ull << 31), "gemm: operand larger than 2 GiB");
    e2.buf_a_bytes = (unsigned)ab;
    e2.buf_w_bytes = (unsigned)wb;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, lda, w, c, ldc, m, n, k, n_ntiles,
                     (int)nwg, (int)sk.dp_tiles, sk.wgs, sk.per, e2);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}


template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg_t(const T* a, int lda, const T* w, T* c, int ldc, int m, int n, int k,
                 const GemmEpilogue& epi, hipStream_t stream) {
  constexpr size_t lds = gemm_lds_bytes<BM, BN>();
  static unsigned long long attr_once = 0;
  auto kern = pw_gemm_kernel<T, BM, BN, WM, WN>;
  if (int st = casync_ensure_dyn_lds(&attr_once, reinterpret_cast<const void*>(kern), (int)lds)) return st;
  const int n_mtiles = (m + BM - 1) / BM, n_ntiles = n / BN;
  const long long nwg = (long long)n_mtiles * n_ntiles;
  CASYNC_REQUIRE(nwg < (1ll << 31), "gemm grid too large");
  // persistent grid: as many workgroups as can be resident (LDS-limited), each walks tiles
  const int persist = casync_opts().gemm_persist;
  constexpr int per_cu = (int)(160 * 1024 / lds) < 8 ? (int)(160 * 1024 / lds) : 8;
  const long long cap = 256ll * (per_cu < 1 ? 1 : per_cu);
  const unsigned grid = (unsigned)(persist && nwg > cap ? cap : nwg);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, a, lda, w, c, ldc, m, n, k,
                     n_ntiles, (int)nwg, epi);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

// 0 = register-staged kernel only, 1 = LDS-DMA ring for bf16, 2 = for both types (default)
inline int glds_mode() { return casync_opts().gemm_glds; }

template <int BM, int BN, int WM, int WN>
int launch_cfg(const void* a, int lda, const void* w, void* c, int ldc, int m, int n, int k,
               const GemmEpilogue& epi, hipStream_t stream, int dtype, bool use_sk) {

  // ring depth: 128x128 -> 3 stages (96 KB, one workgroup per CU), 128x64 -> 2 stages (48 KB, three
  // per CU), 64x64 -> 2 stages (32 KB, five per CU).  Measured every time: one more co-resident
  // workgroup beats one more stage of prefetch (128x64: +1.5 % fp32, +3.6 % bf16 end to end;
  // 64x64 two stages vs three: +0.8 % at B=64, +1.6 % at B=32 in the two-lane schedule).
  constexpr int NST2 = (BM + BN) >= 256 ? 3 : 2;
  // The 128x128 ring leaves room for one workgroup per CU only: use it when the launch has at most
  // one tile per CU anyway (then its deeper pipeline wins), else the register-staged kernel with
  // two co-resident workgroups.  The smaller tiles always take the ring.
  const long long tiles = (long long)((m + BM - 1) / BM) * (n / BN);
  const size_t esz = dtype == DT_BF16 ? 2 : 4;
  const bool fits32 = ((size_t)(m - 1) * lda + k) * esz < (1ull << 31) && (size_t)n * k * esz < (1ull << 31);
  const bool ring_ok = fits32 && ((BM + BN) < 256 || tiles <= 256);
  if constexpr (BM == 128 && BN == 128) {
    // round-6 experiment (`gemm_ring128`): the 128x128 bf16 tile on a TWO-stage ring, two workgroups per CU
    // (64 KB each), for the multi-round launches the register-staged kernel serves -- the configuration missing from
    // profiles/r5_bf16_gemm_variants.txt (the ring has no ds_write, which that table priced at 15-35 % of the staged kernel)
    if (fits32 && dtype == DT_BF16 && glds_mode() >= 1 && casync_opts().gemm_ring128 && tiles > 256)
      return launch_glds_t<bf16_t, 128, 128, WM, WN, 2>(static_cast<const bf16_t*>(a), lda, static_cast<const bf16_t*>(w),
                                                        static_cast<bf16_t*>(c), ldc, m, n, k, epi, stream, use_sk);
  }
  if constexpr (WM * WN == 4 && BN >= 64) {
    if (ring_ok && dtype == DT_BF16 && glds_mode() >= 1) {
      return launch_glds_t<bf16_t, BM, BN, WM, WN, NST2>(static_cast<const bf16_t*>(a), lda, static_cast<const bf16_t*>(w),
                                                         static_cast<bf16_t*>(c), ldc, m, n, k, epi, stream, use_sk);
    }
    if (ring_ok && dtype == DT_F32 && glds_mode() >= 2) {
      const float* af = static_cast<const float*>(a);
      const float* wf = static_cast<const float*>(w);
      float* cf = static_cast<float*>(c);
      return launch_glds_t<float, BM, BN, WM, WN, NST2>(af, lda, wf, cf, ldc, m, n, k, epi, stream, use_sk);
    }
  }
  if (dtype == DT_BF16)
    return launch_cfg_t<bf16_t, BM, BN, WM, WN>(static_cast<const bf16_t*>(a), lda, static_cast<const bf16_t*>(w),
                                                static_cast<bf16_t*>(c), ldc, m, n, k, epi, stream);
  return launch_cfg_t<float, BM, BN, WM, WN>(static_cast<const float*>(a), lda, static_cast<const float*>(w),
                                             static_cast<float*>(c), ldc, m, n, k, epi, stream);
}

// Tile choice.  On a 256-CU chip a launch of G workgroups finishes after ceil(G/256) "rounds" of
// one tile per CU (co-resident tiles share the CU's matrix pipes, so they add, not overlap);
// pick the tile that minimises rounds x tile area, preferring the larger tile on ties.
enum Cfg { C128x128 = 0, C128x64, C64x64, C128x32, C64x32, CFG_COUNT };

struct TileCfg { Cfg id; int bm, bn; };
constexpr TileCfg kTiles[] = {{C128x128, 128, 128}, {C128x64, 128, 64}, {C64x64, 64, 64}, {C128x32, 128, 32}, {C64x32, 64, 32}};

// does launch_cfg() send this config to the LDS-DMA ring kernel?
bool takes_ring(const TileCfg& t, long long tiles, int dtype) {
  if (t.id == C64x32) return dtype == DT_F32 && glds_mode() >= 2;   // fp32 only (see pick_cfg)
  return (t.id == C128x128 || t.id == C128x64 || t.id == C64x64) &&
         (t.bm + t.bn < 256 || tiles <= 256) &&
         ((dtype == DT_BF16 && glds_mode() >= 1) || (dtype == DT_F32 && glds_mode() >= 2));
}

int pick_cfg(int m, int n, int k, bool stream_k, int dtype, bool* use_sk = nullptr, bool concurrent = false, bool small_m_ok = true) {
  const int forced = casync_opts().gemm_cfg;
  const int nk = k / (ROWB / dtype_size(dtype));
  if (use_sk) *use_sk = false;
  if (forced >= 0 && forced < CFG_COUNT && n % kTiles[forced].bn == 0 && (forced != C64x32 || dtype == DT_F32)) {
    const TileCfg& t = kTiles[forced];
    const long long g = (long long)((m + t.bm - 1) / t.bm) * (n / t.bn);
    if (use_sk) *use_sk = takes_ring(t, g, dtype) && stream_k_split(g, nk, t.bm * t.bn, stream_k).wgs > 0;
    return forced;
  }
  // Cost model fitted to tools/experiments/streamk_sweep.sh: a launch costs a fixed ~17 us (not
  // modelled, equal for all) plus rounds x nk x (area + 600) / 8192 us, where a "round" is one
  // tile per CU (co-resident tiles share the CU's matrix pipes, so they add, not overlap) and the
  // +600 is the worse operand reuse of small tiles.  Stream-K makes the last round fractional
  // and adds ~5 us (64x64) to ~10 us (128x64) of short-run start-up, parking and fix-up (priced
  // a little higher here so that it is only chosen where it clearly wins).
  // bf16 tiles are bound by data movement, not by the MFMAs: above 2048 tiles the 128-wide tiles win there whatever the
  // option says (B=512: 33.7 k frames/s against 32.3 k with 64x64 tiles up to 8192)
  const int conc_mode = casync_opts().gemm_conc;
  const int conc_tiles = dtype == DT_BF16 && casync_opts().gemm_conc_tiles > 2048 ? 2048 : casync_opts().gemm_conc_tiles;
  const long long t64 = (long long)((m + 63) / 64) * (n / 64);
  // Launches that do not give a 64x64 grid one tile per CU (M = 100..800 rows: B = 1..8 at 10x10): the 64x32 tile with
  // the K split inside the workgroup and a four-stage ring.  Single-lane plan, fp32 only (its epilogue's thread map).
  // Measured against 64x64 + stream-K (tools/experiments/small_m_gemm.sh, us per launch): (800,576,1024) 14 / 17,
  // (800,1024,512) 14 / 16, (800,512,1024) 14 / 16, (800,512,256) 7 / 9, (100,1024,1024) 14 / 14.5; from 208 tiles with
  // 32 k-tiles on it loses: (800,1024,1024) 24 / 23, (800,2048,1024) 42 / 34.  A six-stage ring is slower than four.
  // (Like stream-K it splits K, i.e. changes the summation order with the batch size: gemm_streamk=0 switches both off and
  // gives batch-invariant bits.)
  if (small_m_ok && !concurrent && dtype == DT_F32 && glds_mode() >= 2 && casync_opts().gemm_small_m && stream_k_mode() && n % 32 == 0 &&
      (t64 <= 128 || (t64 <= 256 && nk <= 16)))
    return C64x32;
  int best = -1;
  double best_cost = 0;
  for (const TileCfg& t : kTiles) {
    if (n % t.bn) continue;
    if (t.id == C64x32) continue;   // chosen by the rule above only
    // single-lane fp32 plan (B < 32): 64x64 everywhere, as in the two-lane plan -- the 128-row tiles the cost model picks
    // for its 200-tile launches stream 1.5x the bytes through each CU's LDS-DMA path (B=16: 1.81 -> 1.76 ms)
    if (!concurrent && dtype == DT_F32 && casync_opts().gemm_single64 && n % 64 == 0 && t.id != C64x64 && t64 <= casync_opts().gemm_single64) continue;
    if (concurrent && n % 64 == 0) {
      if (conc_mode == 1 && t.id != C64x64) continue;
      if (conc_mode == 2 && t.id == C128x128) continue;
      if (conc_mode == 3 && t64 <= conc_tiles && t.id != C64x64) continue;
    }
    const long long g = (long long)((m + t.bm - 1) / t.bm) * (n / t.bn);
    // bf16 tiles are data-movement / latency bound (the MFMAs are 8x cheaper), so the fixed cost of a
    // tile weighs ~10x more against its area: larger tiles win (all-128x128 +3.4 % at B=512)
    const double per_round = nk * (t.bm * t.bn + (dtype == DT_BF16 ? 6000.0 : 600.0)) / 8192.0;
    double cost = (double)((g + 255) / 256) * per_round;
    bool sk_better = false;
    if (takes_ring(t, g, dtype) && stream_k_split(g, nk, t.bm * t.bn, stream_k).wgs) {
      const double sk_cost = (double)g / 256.0 * per_round + 2.5 + 9.5 * (t.bm * t.bn) / 8192.0;
      if (sk_cost < cost) cost = sk_cost, sk_better = true;
    }
    if (best < 0 || cost < best_cost) {
      best = t.id, best_cost = cost;
      if (use_sk) *use_sk = sk_better;
    }
  }
  return best;
}

}  // namespace

// Name of the kernel instance launch_pw_gemm() will pick (as rocprofv3 prints it).
const char* pw_gemm_kernel_name(int m, int n, int k, bool stream_k, int dtype, bool concurrent, bool ups) {
  static thread_local char buf[64];
  if (ups) return "pw_gemm_ups_kernel<float, 64, 64, 2, 2, 2>";
  const char* t = dtype == DT_BF16 ? "__bf16" : "float";
  const int id = pick_cfg(m, n, k, stream_k, dtype, nullptr, concurrent);
  const char* cfg;
  switch (id) {
    case C128x128: cfg = "128, 128, 2, 2"; break;
    case C128x64: cfg = "128, 64, 2, 2"; break;
    case C64x64: cfg = "64, 64, 2, 2"; break;
    case C64x32: cfg = "64, 32, 2, 1"; break;   // four stages, see the snprintf below
    default: cfg = "128, 32, 4, 1"; break;
  }
  const TileCfg& tc = kTiles[id];
  const long long tiles = (long long)((m + tc.bm - 1) / tc.bm) * (n / tc.bn);
  const size_t esz = dtype_size(dtype);
  if (id == C128x128 && dtype == DT_BF16 && glds_mode() >= 1 && casync_opts().gemm_ring128 && tiles > 256 &&
      (size_t)m * k * esz < (1ull << 31) && (size_t)n * k * esz < (1ull << 31))
    snprintf(buf, sizeof(buf), "pw_gemm_glds_kernel<__bf16, 128, 128, 2, 2, 2, false>");
  else if (takes_ring(tc, tiles, dtype))
    snprintf(buf, sizeof(buf), "pw_gemm_glds_kernel<%s, %s, %d, false>", t, cfg, id == C64x32 ? 4 : (tc.bm + tc.bn >= 256 ? 3 : 2));
  else
    snprintf(buf, sizeof(buf), "pw_gemm_kernel<%s, %s>", t, cfg);
  return buf;
}

int launch_pw_gemm(const void* a, int lda, const void* w, void* c, int ldc, int m, int n, int k,
                   const GemmEpilogue& epi, hipStream_t stream, int dtype) {
  const int bk = ROWB / dtype_size(dtype), e16 = 16 / dtype_size(dtype);
  CASYNC_REQUIRE(a && w && c, "pw_gemm: null pointer");
  CASYNC_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "pw_gemm: dtype %d", dtype);
  CASYNC_REQUIRE(m > 0 && n > 0 && k > 0, "pw_gemm: empty problem m=%d n=%d k=%d", m, n, k);
  CASYNC_REQUIRE(k % bk == 0, "pw_gemm: K=%d must be a multiple of %d", k, bk);
  CASYNC_REQUIRE(n % 32 == 0, "pw_gemm: N=%d must be a multiple of 32", n);
  CASYNC_REQUIRE(lda % e16 == 0 && lda >= k, "pw_gemm: lda=%d (K=%d) must be >= K and a multiple of %d", lda, k, e16);
  CASYNC_REQUIRE(ldc >= n && ldc % e16 == 0, "pw_gemm: ldc=%d must be >= N=%d and a multiple of %d", ldc, n, e16);
  CASYNC_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)c % 16) == 0,
                 "pw_gemm: A/W/C must be 16-B aligned");
  CASYNC_REQUIRE(!epi.acc_out || epi.acc_in, "pw_gemm: acc_out without acc_in");
  CASYNC_REQUIRE(!epi.ups_src || (epi.ups_h > 2 && epi.ups_w > 2 && epi.ups_h % 2 == 0 && epi.ups_w % 2 == 0 && epi.ups_ld >= n &&
                                  epi.ups_ld % e16 == 0 && (uintptr_t)epi.ups_src % 16 == 0 && m % (epi.ups_h * epi.ups_w) == 0),
                 "pw_gemm: bad upsampled addend (%dx%d, ld %d)", epi.ups_h, epi.ups_w, epi.ups_ld);
  CASYNC_REQUIRE((!epi.pre_res || epi.ld_pre % e16 == 0) && (!epi.post_res || epi.ld_post % e16 == 0) &&
                     (!epi.acc_out || epi.ld_acc % e16 == 0),
                 "pw_gemm: residual leading dimensions must be multiples of %d", e16);
  CASYNC_REQUIRE((!epi.pre_res || (uintptr_t)epi.pre_res % 16 == 0) && (!epi.post_res || (uintptr_t)epi.post_res % 16 == 0) &&
                     (!epi.acc_out || ((uintptr_t)epi.acc_out % 16 == 0 && (uintptr_t)epi.acc_in % 16 == 0)),
                 "pw_gemm: residual pointers must be 16-B aligned");
  bool sk = false;
  if (epi.ups_src) {   // one instance carries the upsampled addend: fp32, the 64x64 two-stage ring
    CASYNC_REQUIRE(dtype == DT_F32 && n % 64 == 0, "pw_gemm: the upsampled addend needs fp32 and N %% 64 == 0 (N=%d)", n);
    return launch_glds_t<float, 64, 64, 2, 2, 2, false, true>(static_cast<const float*>(a), lda, static_cast<const float*>(w),
                                                              static_cast<float*>(c), ldc, m, n, k, epi, stream, false);
  }
  switch (pick_cfg(m, n, k, epi.sk_ws != nullptr, dtype, &sk, epi.concurrent != 0)) {
    case C128x128: return launch_cfg<128, 128, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype, sk);
    case C128x64: return launch_cfg<128, 64, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype, sk);
    case C64x64: return launch_cfg<64, 64, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype, sk);
    case C64x32: {
      const bool fits32 = ((size_t)(m - 1) * lda + k) * 4 < (1ull << 31) && (size_t)n * k * 4 < (1ull << 31);
      CASYNC_REQUIRE(dtype == DT_F32, "pw_gemm: the 64x32 tile is fp32 only");
      if (!fits32) return launch_cfg<128, 32, 4, 1>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype, false);   // pointer-addressed kernel
      return launch_glds_t<float, 64, 32, 2, 1, 4>(static_cast<const float*>(a), lda, static_cast<const float*>(w),
                                                   static_cast<float*>(c), ldc, m, n, k, epi, stream, false);
    }
    default: return launch_cfg<128, 32, 4, 1>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype, sk);
  }
}

// ---------------------------------------------------------------- dense 3x3 as an implicit GEMM
namespace {
template <typename T>
int launch_conv_t(const T* in, const T* w, T* out, int ldc, int m, int n, int k, const GemmEpilogue& epi,
                  hipStream_t stream, int dtype) {
  bool sk = false;
  const int cfg = pick_cfg(m, n, k, epi.sk_ws != nullptr, dtype, &sk, epi.concurrent != 0, false);
  // the ring kernel's two 48-KB-class tiles; the A "leading dimension" is unused (rows are gathered)
  const bool small = cfg == C64x64 || cfg == C64x32 || n % 64 || m <= 4096;
  return small ? launch_glds_t<T, 64, 64, 2, 2, 2, true>(in, 0, w, out, ldc, m, n, k, epi, stream, sk && cfg == C64x64)
               : launch_glds_t<T, 128, 64, 2, 2, 2, true>(in, 0, w, out, ldc, m, n, k, epi, stream, sk && cfg == C128x64);
}
}  // namespace

const char* conv3x3_gemm_kernel_name(int batch, int h, int wdt, int cin, int cout, int stride, int pad, int dtype,
                                     bool concurrent, bool stream_k) {
  static thread_local char buf[64];
  const int ho = (h + 2 * pad - 3) / stride + 1, wo = (wdt + 2 * pad - 3) / stride + 1;
  const int m = batch * ho * wo;
  const int cfg = pick_cfg(m, cout, 9 * cin, stream_k, dtype, nullptr, concurrent, false);   // the launch's own choice
  const bool small = cfg == C64x64 || cfg == C64x32 || cout % 64 || m <= 4096;
  snprintf(buf, sizeof(buf), "pw_gemm_glds_kernel<%s, %s, 2, 2, %d, true>", dtype == DT_BF16 ? "__bf16" : "float",
             small ? "64, 64" : "128, 64", 2);
  return buf;
}

int launch_conv3x3_gemm(const void* in, const void* w, void* out, int ldc, int batch, int h, int wdt, int cin,
                        int cout, int stride, int pad, const GemmEpilogue& epi_in, hipStream_t stream, int dtype) {
  const int bk = ROWB / dtype_size(dtype), e16 = 16 / dtype_size(dtype);
  CASYNC_REQUIRE(in && w && out && batch > 0, "conv3x3: bad args");
  CASYNC_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "conv3x3: dtype %d", dtype);
  CASYNC_REQUIRE(cin % bk == 0, "conv3x3: cin=%d must be a multiple of %d", cin, bk);
  CASYNC_REQUIRE(cout % 64 == 0, "conv3x3: cout=%d must be a multiple of 64", cout);
  CASYNC_REQUIRE(stride >= 1 && pad >= 0 && h + 2 * pad >= 3 && wdt + 2 * pad >= 3, "conv3x3: geometry");
  CASYNC_REQUIRE(ldc >= cout && ldc % e16 == 0, "conv3x3: ldc=%d", ldc);
  CASYNC_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "conv3x3: in/w/out must be 16-B aligned");
  GemmEpilogue epi = epi_in;
  epi.conv_on = 1;
  epi.conv_h = h; epi.conv_w = wdt; epi.conv_c = cin; epi.conv_stride = stride; epi.conv_pad = pad;
  epi.conv_ho = (h + 2 * pad - 3) / stride + 1;
  epi.conv_wo = (wdt + 2 * pad - 3) / stride + 1;
  const long long m = (long long)batch * epi.conv_ho * epi.conv_wo;
  CASYNC_REQUIRE(m < (1ll << 31), "conv3x3: too many output pixels");
  if (dtype == DT_BF16)
    return launch_conv_t<bf16_t>(static_cast<const bf16_t*>(in), static_cast<const bf16_t*>(w), static_cast<bf16_t*>(out),
                                 ldc, (int)m, cout, 9 * cin, epi, stream, dtype);
  return launch_conv_t<float>(static_cast<const float*>(in), static_cast<const float*>(w), static_cast<float*>(out), ldc,
                              (int)m, cout, 9 * cin, epi, stream, dtype);
}
