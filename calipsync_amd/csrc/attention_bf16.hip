// Cross-attention core of CASync (reference module/unet.py:207-218) for the bf16 engine (BASELINE configs[2]), per frame:
//
//   E[i][j] = sum_d Q[i][d] K[j][d]          i: 100 face positions, j: 100 audio positions
//   P       = softmax_j(E)                   (no 1/sqrt(d) scale, single head)
//   out[i][c] = gamma * sum_j P[i][j] V[j][c] + res[i][c]
//
// Round 5.  attention.hip serves both storage types with fp32 matrix instructions (v_mfma_f32_32x32x2_f32: 240 of them per
// workgroup-wave, 15 k cycles) behind a chain of dependent phases; at B = 512 its eight launches were 0.78 ms of a 13.2 ms
// step at 0.26 MFMA-busy and 0.6 of the time in waits (profiles/r4_mfma_busy_bf16_b512.json).  Q, K and V ARE bf16 in this
// engine, so both products run on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; QK^T is exact in its products either way, P is
// rounded to bf16 before PV -- 2^-9 relative on weights in [0, 1], the same rounding the output takes anyway):
//
//   one workgroup = one frame, wave w = queries 32 w .. 32 w + 31 (100 -> 4 tiles; the last holds four real queries).
//   S^T = K Q^T   (keys on the ROWS, queries on the lanes): 4 key tiles x 4 k-steps of 16 channels = 16 MFMAs.  A query's
//                 128 scores then sit in the 64 accumulator registers of TWO lanes (l, l + 32): the softmax is register
//                 arithmetic plus two exchanges with the partner lane -- no LDS score tile, no shuffles per row.
//   O^T = V^T P^T: an accumulator tile of S^T, converted pairwise to bf16, IS the B operand of this product (it sums over
//                 the tile's row index = the key; cdna_hip_programming.md section 3), so P never leaves the registers.  The A
//                 operand V^T[channel][key] is read from the row-major V image ([key][128 channels], as it arrives from
//                 HBM) with ds_read_b64_tr_b16, the hardware transpose read: element j of lane half h must be key
//                 16 s + 8 (j >> 2) + 4 h + (j & 3) of the k-step -- two transposed reads of four consecutive keys each.
//                 7 k-steps (112 keys) x 16 channel tiles = 112 MFMAs per wave.
//
// V travels in four 128-channel chunks by LDS-DMA (global_load_lds, 16 B per lane, source-side XOR swizzle for the
// conflict-free image (b) of the guide's T10), double buffered; the second buffer overlays K and Q, which are dead once S^T
// is done: 60 KB of LDS, two workgroups per CU.  gamma * O + residual is applied on the accumulators; a lane holds four
// runs of four consecutive channels of one query: 8-byte loads / stores.
#include "common.h"

namespace {

constexpr int NP = 100;   // positions per frame (10 x 10)
constexpr int DQK = 64;   // query / key channels
constexpr int CV = 512;   // value channels
constexpr int KROWS = 128, VROWS = 112;                        // staged rows of K / Q (4 tiles of 32) and of a V chunk (7 k-steps of 16)
constexpr int oK = 0, oQ = KROWS * 128, oV0 = 2 * KROWS * 128, oV1 = 0;   // byte offsets; V buffer 1 overlays K + Q
constexpr int VBUF = VROWS * 256;
constexpr int ATTB_LDS_BYTES = oV0 + VBUF;
static_assert(VBUF <= oV0, "the second V buffer overlays K and Q");

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32b(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
// key of row r of the V image (256-B rows, sixteen 16-B chunks): image (b) of cdna_hip_programming.md T10 -- conflict free for
// the transposed reads below
__device__ __forceinline__ int vkey(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

__global__ __launch_bounds__(256, 2) void cross_attention_bf16_kernel(
    const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ k, int ldk, const bf16_t* __restrict__ v, int ldv,
    const bf16_t* __restrict__ res, int ld_res, const float* __restrict__ gamma, bf16_t* __restrict__ out, int ld_out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const size_t row0 = (size_t)blockIdx.x * NP;

  // ---- staging by LDS-DMA.  K / Q image: 128-B rows, eight 16-B chunks XOR-keyed by (row >> 1) & 7 (the ring GEMM's
  //      swizzle: conflict free for the 32-row fragment reads); rows past the frame re-read its last row (finite; their
  //      scores are masked / their queries never stored). ----
  {
    const int lrow = lane >> 3, lch = lane & 7;
#pragma unroll
    for (int j = 0; j < KROWS / 32; ++j) {
      const int r = (j * 4 + wave) * 8 + lrow, src_r = r < NP ? r : NP - 1, sc = lch ^ ((r >> 1) & 7);
      dma16(k + (row0 + src_r) * ldk + sc * 8, lds + oK + (j * 4 + wave) * 1024);
      dma16(q + (row0 + src_r) * ldq + sc * 8, lds + oQ + (j * 4 + wave) * 1024);
    }
  }
  auto v_stage = [&](int cz, int buf) __attribute__((always_inline)) {   // V chunk cz -> buffer buf: 4 rows x 256 B per instruction
    const int lrow = lane >> 4, lch = lane & 15;
    char* base = lds + (buf ? oV1 : oV0);
#pragma unroll
    for (int j = 0; j < VROWS / 16; ++j) {
      const int r = (j * 4 + wave) * 4 + lrow, src_r = r < NP ? r : NP - 1, sc = lch ^ vkey(r);
      dma16(v + (row0 + src_r) * ldv + cz * 128 + sc * 8, base + (j * 4 + wave) * 1024);
    }
  };
  v_stage(0, 0);
  const float gam = gamma[0];
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VROWS / 16) : "memory");   // K and Q have landed (the V chunk may still fly)
  __syncthreads();

  // ---- S^T = K Q^T: tile t = keys 32 t .. + 31 on the rows, this wave's queries on the lanes ----
  f32x16 st[4];
  {
    const int kq_chunk = h;   // this lane's 16 B of k-step s: chunk 2 s + h of its row
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
    const int qrow = 32 * wave + r32;
    bf16x8 fq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      fq[s] = *reinterpret_cast<const bf16x8*>(lds + oQ + qrow * 128 + (((2 * s + kq_chunk) ^ ((qrow >> 1) & 7)) << 4));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int krow = 32 * t + r32;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 fk = *reinterpret_cast<const bf16x8*>(lds + oK + krow * 128 + (((2 * s + kq_chunk) ^ ((krow >> 1) & 7)) << 4));
        st[t] = mfma32b(fk, fq[s], st[t]);
      }
    }
  }
  __syncthreads();   // everyone is done with K / Q: the second V buffer may land on them
  v_stage(1, 1);

  // ---- softmax over the keys of each query: register r of tile t, lane half h is key 32 t + 8 (r >> 2) + 4 h + (r & 3);
  //      keys 100 .. 127 do not exist (tile 3: only r < 4 on h = 0) ----
  bf16x8 pf[4][2];   // P^T as the B operand of PV: k-step s of tile t = registers 8 s .. 8 s + 7
  {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r >= 4 || h) st[3][r] = -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[t][r] = __expf(st[t][r] - mx);   // exp(-inf) = 0: the missing keys become exact zeros
        sum += st[t][r];
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[t][s][j] = (bf16_t)(st[t][8 * s + j] * inv);
  }

  // ---- transposed-read addresses: 16-lane group g = lane >> 4 covers channels 16 (g & 1) .. + 15 of a 32-channel tile on
  //      lane half h = g >> 1; lane 4 qq + p of the group supplies row r0 + qq, chunk c0 + (p >> 1), byte 8 (p & 1).
  //      r0 = 32 t + 16 s + 8 jj + 4 h is a multiple of four, so the row key is (qq << 2) | ((2 jj + h) & 3): the chunk's
  //      two high bits (the channel tile) XOR qq, its two low bits XOR (2 jj + h) & 3 -- four bases per jj, the k-step as
  //      an instruction immediate ----
  int tr_base[2][4];
  {
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int lo = (2 * (g & 1) + (p >> 1)) ^ ((2 * jj + h) & 3);
        tr_base[jj][ct] = 256 * (8 * jj + 4 * h + qq) + 16 * (((ct ^ qq) << 2) | lo) + 8 * (p & 1);
      }
  }

  const int qi = 32 * wave + r32;          // this lane's query
  const bool q_ok = qi < NP;
  const bf16_t* res_q = res + (row0 + (q_ok ? qi : 0)) * ld_res + 4 * h;
  bf16_t* out_q = out + (row0 + (q_ok ? qi : 0)) * ld_out + 4 * h;

#pragma unroll 1
  for (int cz = 0; cz < 4; ++cz) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of chunk cz (and everything older) has landed
    __syncthreads();                                   // ... everyone's
    const char* vb = lds + ((cz & 1) ? oV1 : oV0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int c0 = cz * 128 + ct * 32;
      // residual values of this tile's outputs: channels c0 + 8 j + 4 h .. + 3 of the lane's query, requested ahead of the MFMAs
      bf16x4 rv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[j] = *reinterpret_cast<const bf16x4*>(res_q + c0 + 8 * j);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (t == 3 && s == 1) continue;   // keys 112 .. 127: not staged, their weights are zero
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(vb + tr_base[0][ct] + 256 * (32 * t + 16 * s)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(vb + tr_base[1][ct] + 256 * (32 * t + 16 * s)));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc = mfma32b(__builtin_bit_cast(bf16x8, av), pf[t][s], acc);
        }
      if (q_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(gam * acc[4 * j + e] + (float)rv[j][e]);
          *reinterpret_cast<bf16x4*>(out_q + c0 + 8 * j) = o;
        }
      }
    }
    if (cz + 2 < 4) {
      __syncthreads();            // everyone has read this buffer: the chunk after next may land in it
      v_stage(cz + 2, cz & 1);
    }
  }
}

}  // namespace

int launch_cross_attention_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* res, int ld_res,
                                const float* gamma_dev, void* out, int ld_out, int batch, hipStream_t stream) {
  CASYNC_REQUIRE(q && k && v && res && gamma_dev && out, "cross_attention (bf16): null pointer");
  CASYNC_REQUIRE(batch > 0, "cross_attention (bf16): batch %d", batch);
  CASYNC_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ld_res % 4 == 0 && ld_out % 4 == 0 && ldq >= DQK && ldk >= DQK &&
                     ldv >= CV && ld_res >= CV && ld_out >= CV,
                 "cross_attention (bf16): bad leading dimensions");
  CASYNC_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)res % 8) == 0 &&
                     ((uintptr_t)out % 8) == 0,
                 "cross_attention (bf16): alignment");
  static unsigned long long once = 0;
  if (int st = casync_ensure_dyn_lds(&once, reinterpret_cast<const void*>(cross_attention_bf16_kernel), ATTB_LDS_BYTES)) return st;
  hipLaunchKernelGGL(cross_attention_bf16_kernel, dim3(batch), dim3(256), ATTB_LDS_BYTES, stream, (const bf16_t*)q, ldq,
                     (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (const bf16_t*)res, ld_res, gamma_dev, (bf16_t*)out, ld_out);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}
