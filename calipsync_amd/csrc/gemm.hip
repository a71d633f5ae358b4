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
  constexpr int TPR = BN / 4;          // threads per output row
  constexpr int RPP = NT / TPR;        // rows per pass
  const int c4 = (tid % TPR) * 4, rr = tid / TPR;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};

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

    // ---- epilogue 2: float4 rows: bias, residuals, activation, coalesced stores ----
    const int n = n0 + c4;
    const f32x4 bias = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + n) : zero;
    const f32x4 pscale = epi.pre_scale ? *reinterpret_cast<const f32x4*>(epi.pre_scale + n) : one;
    const f32x4 as = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_s + n) : one;
    const f32x4 at = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_t + n) : zero;
#pragma unroll 4
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = rr + RPP * p, m = m0 + row;
      if (m >= M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LDC_S + c4) + bias;
      if (epi.pre_res) v += pscale * ld4(static_cast<const T*>(epi.pre_res) + (size_t)m * epi.ld_pre + n);
      if (epi.act) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
      if (epi.post_res) v += ld4(static_cast<const T*>(epi.post_res) + (size_t)m * epi.ld_post + n);
      if (epi.aff_s && !epi.aff_on_acc) {
        v = v * as + at;
        v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w);
      }
      st4(C + (size_t)m * ldc + n, v);
      if (epi.acc_out) {
        f32x4 s = ld4(static_cast<const T*>(epi.acc_in) + (size_t)m * epi.ld_acc + n) + v;
        if (epi.aff_s && epi.aff_on_acc) {
          s = s * as + at;
          s.x = lrelu(s.x); s.y = lrelu(s.y); s.z = lrelu(s.z); s.w = lrelu(s.w);
        }
        st4(static_cast<T*>(epi.acc_out) + (size_t)m * epi.ld_acc + n, s);
      }
    }
    __syncthreads();  // the C tile aliases the staging buffers the next tile is about to fill
    m0 = m0n;
    n0 = n0n;
  }
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
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(256) void pw_gemm_glds_kernel(const T* __restrict__ A, int lda,
                                                           const T* __restrict__ W, T* __restrict__ C,
                                                           int ldc, int M, int N, int K, int n_ntiles,
                                                           int nwg, GemmEpilogue epi) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int E16 = 16 / (int)sizeof(T);
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * ROWB;          // bytes per ring stage
  constexpr int LPT = ROWS / 32;              // LDS-DMA instructions per wave per k-tile (8 rows each, 4 waves)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDC_S = BN + 4;
  static_assert((size_t)NST * STAGE >= (size_t)BM * LDC_S * sizeof(float), "ring must hold the C tile");
  static_assert(LPT * (NST - 1) < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char ring[];
  float* Cs = reinterpret_cast<float*>(ring);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r32 = lane & 31, kh = lane >> 5;
  const int nk = K / BK;
  const int lrow8 = lane >> 3, lcol = lane & 7;

  // epilogue constants
  constexpr int TPR = BN / 4, RPP = 256 / TPR;
  const int c4 = (tid % TPR) * 4, rr = tid / TPR;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};

  for (int tile = blockIdx.x; tile < nwg; tile += gridDim.x) {
    int m0, n0;
    {
      const int q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
      const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      const int mt = bid / n_ntiles;
      m0 = mt * BM;
      n0 = (bid - mt * n_ntiles) * BN;
    }
    // this lane's source pointer for each of the wave's LPT instructions (k-tile 0)
    const T* src[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int r = (j * 4 + wave) * 8 + lrow8;             // row inside the stage: [0,BM) = A, [BM,ROWS) = W
      const int cs = lcol ^ ((r >> 1) & 7);                 // swizzled source column
      if (r < BM) {
        int row = m0 + r;
        row = row < M ? row : M - 1;
        src[j] = A + (size_t)row * lda + cs * E16;
      } else {
        src[j] = W + (size_t)(n0 + r - BM) * K + cs * E16;
      }
    }
    auto issue = [&](int kt) {
      char* st = ring + (kt % NST) * STAGE;
#pragma unroll
      for (int j = 0; j < LPT; ++j)
        __builtin_amdgcn_global_load_lds(
            (const void __attribute__((address_space(1)))*)(src[j] + (size_t)kt * BK),
            (void __attribute__((address_space(3)))*)(st + (j * 4 + wave) * 8 * ROWB), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) issue(s);

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

    for (int kt = 0; kt < nk; ++kt) {
      // tiles kt+1 .. kt+NST-2 may stay in flight (fewer near the end of the k loop)
      const int ahead = nk - 1 - kt;
      if (NST >= 4 && ahead >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * LPT>();
      else if (NST >= 3 && ahead >= 1) wait_vmcnt<LPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NST - 1 < nk) issue(kt + NST - 1);
      const char* st = ring + (kt % NST) * STAGE;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[i] = *reinterpret_cast<const f32x4*>(st + a_off[i] + (((2 * g + kh) ^ a_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[j] = *reinterpret_cast<const f32x4*>(st + b_off[j] + (((2 * g + kh) ^ b_key[j]) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MfmaK<T>::run(fa[i], fb[j], acc[i][j]);
      }
    }
    __syncthreads();   // everything landed and consumed: the ring becomes the C tile

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float* cbase = Cs + (wm * (BM / WM) + i * 32 + 4 * kh) * LDC_S + wn * (BN / WN) + j * 32 + r32;
#pragma unroll
        for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * LDC_S] = acc[i][j][r];
      }
    __syncthreads();

    const int n = n0 + c4;
    const f32x4 bias = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + n) : zero;
    const f32x4 pscale = epi.pre_scale ? *reinterpret_cast<const f32x4*>(epi.pre_scale + n) : one;
    const f32x4 as = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_s + n) : one;
    const f32x4 at = epi.aff_s ? *reinterpret_cast<const f32x4*>(epi.aff_t + n) : zero;
#pragma unroll 4
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = rr + RPP * p, m = m0 + row;
      if (m >= M) break;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LDC_S + c4) + bias;
      if (epi.pre_res) v += pscale * ld4(static_cast<const T*>(epi.pre_res) + (size_t)m * epi.ld_pre + n);
      if (epi.act) { v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w); }
      if (epi.post_res) v += ld4(static_cast<const T*>(epi.post_res) + (size_t)m * epi.ld_post + n);
      if (epi.aff_s && !epi.aff_on_acc) {
        v = v * as + at;
        v.x = lrelu(v.x); v.y = lrelu(v.y); v.z = lrelu(v.z); v.w = lrelu(v.w);
      }
      st4(C + (size_t)m * ldc + n, v);
      if (epi.acc_out) {
        f32x4 s = ld4(static_cast<const T*>(epi.acc_in) + (size_t)m * epi.ld_acc + n) + v;
        if (epi.aff_s && epi.aff_on_acc) {
          s = s * as + at;
          s.x = lrelu(s.x); s.y = lrelu(s.y); s.z = lrelu(s.z); s.w = lrelu(s.w);
        }
        st4(static_cast<T*>(epi.acc_out) + (size_t)m * epi.ld_acc + n, s);
      }
    }
    __syncthreads();   // C tile consumed before the next tile's loads overwrite the ring
  }
}

template <typename T, int BM, int BN, int WM, int WN, int NST>
int launch_glds_t(const T* a, int lda, const T* w, T* c, int ldc, int m, int n, int k,
                  const GemmEpilogue& epi, hipStream_t stream) {
  constexpr size_t lds = (size_t)NST * (BM + BN) * ROWB;
  static bool attr_set = false;
  auto kern = pw_gemm_glds_kernel<T, BM, BN, WM, WN, NST>;
  if (!attr_set) {
    CASYNC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int n_mtiles = (m + BM - 1) / BM, n_ntiles = n / BN;
  const long long nwg = (long long)n_mtiles * n_ntiles;
  CASYNC_REQUIRE(nwg < (1ll << 31), "gemm grid too large");
  constexpr int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
  const long long cap = 256ll * per_cu;
  const unsigned grid = (unsigned)(nwg > cap ? cap : nwg);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, lda, w, c, ldc, m, n, k, n_ntiles,
                     (int)nwg, epi);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg_t(const T* a, int lda, const T* w, T* c, int ldc, int m, int n, int k,
                 const GemmEpilogue& epi, hipStream_t stream) {
  constexpr size_t lds = gemm_lds_bytes<BM, BN>();
  static bool attr_set = false;
  auto kern = pw_gemm_kernel<T, BM, BN, WM, WN>;
  if (!attr_set) {
    CASYNC_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int n_mtiles = (m + BM - 1) / BM, n_ntiles = n / BN;
  const long long nwg = (long long)n_mtiles * n_ntiles;
  CASYNC_REQUIRE(nwg < (1ll << 31), "gemm grid too large");
  // persistent grid: as many workgroups as can be resident (LDS-limited), each walks tiles
  static const int persist = [] { const char* v = getenv("CASYNC_GEMM_PERSIST"); return v ? atoi(v) : 1; }();
  constexpr int per_cu = (int)(160 * 1024 / lds) < 8 ? (int)(160 * 1024 / lds) : 8;
  const long long cap = 256ll * (per_cu < 1 ? 1 : per_cu);
  const unsigned grid = (unsigned)(persist && nwg > cap ? cap : nwg);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, stream, a, lda, w, c, ldc, m, n, k,
                     n_ntiles, (int)nwg, epi);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

inline int glds_mode() {
  static const int v = [] { const char* e = getenv("CASYNC_GEMM_GLDS"); return e ? atoi(e) : 2; }();
  return v;   // 0 = register-staged kernel only, 1 = LDS-DMA ring for bf16, 2 = for both types (default)
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const void* a, int lda, const void* w, void* c, int ldc, int m, int n, int k,
               const GemmEpilogue& epi, hipStream_t stream, int dtype) {
  // ring depth: 128x128 -> 3 stages (96 KB, one workgroup per CU), 128x64 -> 2 stages (48 KB, three
  // per CU), 64x64 -> 3 stages (48 KB, three per CU).  Measured: one more co-resident workgroup
  // beats one more stage of prefetch (+1.5 % fp32, +3.6 % bf16 end to end).
  constexpr int NST = (BM + BN) >= 256 ? 3 : ((BM + BN) >= 192 ? 2 : 3);
  // The 128x128 ring leaves room for one workgroup per CU only: use it when the launch has at most
  // one tile per CU anyway (then its deeper pipeline wins), else the register-staged kernel with
  // two co-resident workgroups.  The smaller tiles always take the ring.
  const long long tiles = (long long)((m + BM - 1) / BM) * (n / BN);
  const bool ring_ok = (BM + BN) < 256 || tiles <= 256;
  if constexpr (WM * WN == 4 && BN >= 64) {
    if (ring_ok && dtype == DT_BF16 && glds_mode() >= 1)
      return launch_glds_t<bf16_t, BM, BN, WM, WN, NST>(static_cast<const bf16_t*>(a), lda,
                                                        static_cast<const bf16_t*>(w), static_cast<bf16_t*>(c),
                                                        ldc, m, n, k, epi, stream);
    if (ring_ok && dtype == DT_F32 && glds_mode() >= 2)
      return launch_glds_t<float, BM, BN, WM, WN, NST>(static_cast<const float*>(a), lda,
                                                       static_cast<const float*>(w), static_cast<float*>(c), ldc,
                                                       m, n, k, epi, stream);
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
enum Cfg { C128x128 = 0, C128x64, C64x64, C128x32, C64x32, C64x64W2, CFG_COUNT };

int pick_cfg(int m, int n) {
  static const int forced = [] { const char* v = getenv("CASYNC_GEMM_CFG"); return v ? atoi(v) : -1; }();
  struct T { Cfg id; int bm, bn; };
  const T tiles[] = {{C128x128, 128, 128}, {C128x64, 128, 64}, {C64x64, 64, 64}, {C128x32, 128, 32},
                     {C64x32, 64, 32}, {C64x64W2, 64, 64}};
  if (forced >= 0 && forced < CFG_COUNT && n % tiles[forced].bn == 0) return forced;
  int best = -1;
  double best_cost = 0;
  for (const T& t : tiles) {
    if (n % t.bn || t.id >= C64x32) continue;   // experimental configs: only when forced
    const long long g = (long long)((m + t.bm - 1) / t.bm) * (n / t.bn);
    const double rounds = (double)((g + 255) / 256);
    // small per-round overhead so that, when rounds x area ties, fewer / larger tiles win
    const double cost = rounds * (t.bm * t.bn + 600.0);
    if (best < 0 || cost < best_cost) best = t.id, best_cost = cost;
  }
  return best;
}

}  // namespace

// Name of the kernel instance launch_pw_gemm() will pick (as rocprofv3 prints it).
const char* pw_gemm_kernel_name(int m, int n, int dtype) {
  static thread_local char buf[64];
  const char* t = dtype == DT_BF16 ? "__bf16" : "float";
  const char* cfg;
  switch (pick_cfg(m, n)) {
    case C128x128: cfg = "128, 128, 2, 2"; break;
    case C128x64: cfg = "128, 64, 2, 2"; break;
    case C64x64: cfg = "64, 64, 2, 2"; break;
    case C64x32: cfg = "64, 32, 2, 1"; break;
    case C64x64W2: cfg = "64, 64, 2, 1"; break;
    default: cfg = "128, 32, 4, 1"; break;
  }
  const int id = pick_cfg(m, n);
  const int bm = (id == C64x64 || id == C64x32 || id == C64x64W2) ? 64 : 128;
  const int bn = id == C128x128 ? 128 : (id == C128x64 || id == C64x64 || id == C64x64W2 ? 64 : 32);
  const long long tiles = (long long)((m + bm - 1) / bm) * (n / bn);
  const bool ring = (id == C128x128 || id == C128x64 || id == C64x64) && (bm + bn < 256 || tiles <= 256) &&
                    ((dtype == DT_BF16 && glds_mode() >= 1) || (dtype == DT_F32 && glds_mode() >= 2));
  if (ring)
    snprintf(buf, sizeof(buf), "pw_gemm_glds_kernel<%s, %s, %d>", t, cfg, bm + bn >= 256 ? 3 : (bm + bn >= 192 ? 2 : 3));
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
  CASYNC_REQUIRE(ldc >= n && ldc % 4 == 0, "pw_gemm: ldc=%d must be >= N=%d and a multiple of 4", ldc, n);
  CASYNC_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)c % 16) == 0,
                 "pw_gemm: A/W/C must be 16-B aligned");
  CASYNC_REQUIRE(!epi.acc_out || epi.acc_in, "pw_gemm: acc_out without acc_in");
  CASYNC_REQUIRE((!epi.pre_res || epi.ld_pre % 4 == 0) && (!epi.post_res || epi.ld_post % 4 == 0) &&
                     (!epi.acc_out || epi.ld_acc % 4 == 0),
                 "pw_gemm: residual leading dimensions must be multiples of 4");
  switch (pick_cfg(m, n)) {
    case C128x128: return launch_cfg<128, 128, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
    case C128x64: return launch_cfg<128, 64, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
    case C64x64: return launch_cfg<64, 64, 2, 2>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
    case C64x32: return launch_cfg<64, 32, 2, 1>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
    case C64x64W2: return launch_cfg<64, 64, 2, 1>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
    default: return launch_cfg<128, 32, 4, 1>(a, lda, w, c, ldc, m, n, k, epi, stream, dtype);
  }
}
