// RESULT (forms 19-24, profiles/r6_two_models.txt section 9): on gfx950 a packed fp32 instruction whose LOW result takes the HIGH half of its
// second source (op_sel:[x,1,..]) reads that half as zero on lanes 48..63 beside another wave's v_mfma_f32_16x16x32_bf16 -- 200 of 200 launches;
// every other form below: never.
//
// Round 6: which instruction pattern returns wrong values beside ANOTHER wave's matrix instructions?  (profiles/r6_two_models.txt:
// the fp32 fused Up block computed wrong 16-pixel tiles whenever a bf16 GEMM of a second model shared the chip; its P1 epilogue
// `v += w * g` as scalar FMAs instead of v_pk_fma_f32 made that go away.)  This is the pattern without the kernel around it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/ubench/pk_hazard.hip -o /tmp/pk_hazard && /tmp/pk_hazard
//
// VICTIM kernel, per iteration and lane: two values P (the form under test) and E (the expected value, scalar FMAs on inputs that
// never pass through the instruction under test), compared bit for bit; mismatches are counted and the first one per launch is
// recorded.  All instructions under test are the COMPILER's (vector-typed C++), so the wait states gfx950 wants between an MFMA
// and a vector instruction reading its result are inserted as in the shipped kernel.  Forms:
//   0  v_pk_fma_f32, weight broadcast (op_sel_hi:[0,1,1]), accumulating into plain registers
//   1  v_pk_fma_f32, weight in both halves of a register pair (no op_sel), plain registers
//   2  v_pk_fma_f32 broadcast, accumulating into the result of a v_mfma_f32_16x16x4_f32 issued just before (a second MFMA in flight)
//   3  as 2 without the broadcast
//   4  as 2 with scalar v_fma_f32 (the shipped fix)
//   5  as 2 with ~128 idle cycles (s_sleep 2) between the MFMAs and the packed FMAs
//   6  v_pk_add_f32 of a register pair to the fresh MFMA result (a GEMM epilogue's `acc + bias`)
//   7  as 2, the MFMA a v_mfma_f32_16x16x32_bf16 (what the bf16 kernels do to their own accumulators)
// The MFMA inputs are small integers, uniform over the lanes and changing every iteration, so its exact result is known without
// reading it: the expected value does not depend on the registers under test.
// CO-RUNNER on a second stream: a register-only loop of one matrix instruction (or packed FMAs only, or nothing).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Sample { unsigned it, lane, form, which; float p, e, w, g, c, c_prev; };

__device__ __forceinline__ float sfma(float a, float b, float c) {   // scalar FMA the compiler cannot pack
  float x = __builtin_fmaf(a, b, c);
  asm("" : "+v"(x));
  return x;
}

template <int FORM>
__global__ __launch_bounds__(256, 4) void victim(unsigned* counts, Sample* first, int iters, const float* seed) {
  const int lane = threadIdx.x & 63;
  float w = seed[threadIdx.x], wdup;                     // wdup: an opaque copy of w, so that (w, wdup) is a real register pair
  f32x2 g01 = {seed[256 + threadIdx.x], seed[512 + threadIdx.x]}, g23 = {seed[768 + threadIdx.x], seed[1024 + threadIdx.x]};
  f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};              // forms 0 / 1: running accumulators
  float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
  unsigned bad = 0;
  float c_prev = 0.f;
  const float w_base = w;
  for (int it = 0; it < iters; ++it) {
    f32x2 p01, p23;
    float c = 0.f;
    // the weight changes every iteration (as the bilinear weights of the kernel change per pixel): a loop-invariant splat would be
    // hoisted into a register pair and the broadcast form (op_sel_hi:[0,1,1]) never issued
    w = w_base + (float)(it & 3);
    wdup = w;
    asm("" : "+v"(wdup));
    if constexpr (FORM == 0 || FORM == 1) {
      const f32x2 ww = FORM == 0 ? f32x2{w, w} : f32x2{w, wdup};
      a01 = __builtin_elementwise_fma(ww, g01, a01);
      a23 = __builtin_elementwise_fma(ww, g23, a23);
      p01 = a01, p23 = a23;
      e0 = sfma(w, g01.x, e0), e1 = sfma(w, g01.y, e1), e2 = sfma(w, g23.x, e2), e3 = sfma(w, g23.y, e3);
    } else {
      const float al = (float)(1 + (it & 7));            // uniform: every element of the 16x16 result is 4 * al * 2 (bf16: 32 * al * 2)
      f32x4 r, r2;
      if constexpr (FORM == 7) {
        bf16x8 A, B;
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] = (__bf16)al, B[j] = (__bf16)2.f;
        r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        r2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B, A, f32x4{1.f, 1.f, 1.f, 1.f}, 0, 0, 0);
        c = 64.f * al;
      } else {
        r = __builtin_amdgcn_mfma_f32_16x16x4f32(al, 2.f, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        r2 = __builtin_amdgcn_mfma_f32_16x16x4f32(2.f, al, f32x4{1.f, 1.f, 1.f, 1.f}, 0, 0, 0);
        c = 8.f * al;
      }
      if constexpr (FORM == 5) __builtin_amdgcn_s_sleep(2);
      if constexpr (FORM == 2 || FORM == 5 || FORM == 7) {
        p01 = __builtin_elementwise_fma(f32x2{w, w}, g01, f32x2{r[0], r[1]});
        p23 = __builtin_elementwise_fma(f32x2{w, w}, g23, f32x2{r[2], r[3]});
      } else if constexpr (FORM == 3) {
        p01 = __builtin_elementwise_fma(f32x2{w, wdup}, g01, f32x2{r[0], r[1]});
        p23 = __builtin_elementwise_fma(f32x2{w, wdup}, g23, f32x2{r[2], r[3]});
      } else if constexpr (FORM == 4) {
        p01 = f32x2{sfma(w, g01.x, r[0]), sfma(w, g01.y, r[1])};
        p23 = f32x2{sfma(w, g23.x, r[2]), sfma(w, g23.y, r[3])};
      } else {   // 6
        p01 = f32x2{r[0], r[1]} + g01;
        p23 = f32x2{r[2], r[3]} + g23;
      }
      if constexpr (FORM == 6) {
        e0 = c + g01.x, e1 = c + g01.y, e2 = c + g23.x, e3 = c + g23.y;
        asm("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
      } else {
        e0 = sfma(w, g01.x, c), e1 = sfma(w, g01.y, c), e2 = sfma(w, g23.x, c), e3 = sfma(w, g23.y, c);
      }
      // the second MFMA's result is consumed too (kept live, checked against its own exact value)
      if (__float_as_uint(r2[0]) != __float_as_uint(c + 1.f)) bad += 1u << 16;
    }
    const bool m0 = __float_as_uint(p01.x) != __float_as_uint(e0), m1 = __float_as_uint(p01.y) != __float_as_uint(e1);
    const bool m2 = __float_as_uint(p23.x) != __float_as_uint(e2), m3 = __float_as_uint(p23.y) != __float_as_uint(e3);
    if (m0 | m1 | m2 | m3) {
      ++bad;
      if (atomicAdd(&counts[1], 1u) == 0) {
        const int which = m0 ? 0 : m1 ? 1 : m2 ? 2 : 3;
        const float pv[4] = {p01.x, p01.y, p23.x, p23.y}, ev[4] = {e0, e1, e2, e3}, gv[4] = {g01.x, g01.y, g23.x, g23.y};
        *first = Sample{(unsigned)it, (unsigned)(blockIdx.x * 256 + threadIdx.x), (unsigned)FORM, (unsigned)which, pv[which], ev[which], w, gv[which], c, c_prev};
      }
      if constexpr (FORM == 0 || FORM == 1) a01 = f32x2{e0, e1}, a23 = f32x2{e2, e3};   // resynchronise the running sums
    }
    c_prev = c;
  }
  if (bad) atomicAdd(&counts[0], bad & 0xffffu), atomicAdd(&counts[2], bad >> 16);
  (void)lane;
}

// ---- forms 8..15: write-after-read on an MFMA's SrcC registers, hand-placed (hard registers v38..v57) --------------------------
// The compiler's hazard recogniser keeps wait states between an XDL (bf16) MFMA and a later write to its SrcC registers, and NONE
// for the fp32 ones (not XDL on gfx940+: taken to have read SrcC at issue).  The kernel that went wrong had, in the failing build
// only, fp32 MFMAs with vDst != SrcC -- partly overlapping, v[30:33] <- v[32:35] -- whose SrcC registers the NEXT instruction wrote.
//   8   control: in place (vDst = SrcC), an unrelated register written next
//   9   vDst != SrcC, a vector instruction writes SrcC[2] at +1        10  at +2        11  at +4
//   12  vDst overlaps SrcC (v[38:41] <- v[40:43]), SrcC[2] written at +1                13  the same overlap, nothing written
//   14  as 9 behind another fp32 MFMA of this wave (the pipe busy with the wave's own work)
//   15  as 9 with v_mfma_f32_16x16x32_bf16 (an XDL instruction WITHOUT the wait states the compiler would insert: what a violated
//       hazard looks like in this harness)
//   16  behind a run of eight independent fp32 MFMAs of this wave (as the kernel issues them): the A operand register of the LAST one
//       written at +1                      17  its B operand                      18  its SrcC[2] (vDst != SrcC)
#define WAR_HEAD "v_mov_b32 v40, %[c0]\n\tv_mov_b32 v41, %[c0]\n\tv_mov_b32 v42, %[c0]\n\tv_mov_b32 v43, %[c0]\n\tv_mov_b32 v48, 0\n\ts_nop 7\n\t"
#define WAR_TAIL(D0, D1, D2, D3) "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_mov_b32 %[r0], " D0 "\n\tv_mov_b32 %[r1], " D1 "\n\tv_mov_b32 %[r2], " D2 "\n\tv_mov_b32 %[r3], " D3 "\n\t"
#define WAR_OPS : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3) : [a] "v"(al), [b] "v"(two), [c0] "v"(c0), [junk] "v"(junk)
#define WAR_CLOB : "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"
template <int FORM>
__global__ __launch_bounds__(256, 4) void victim_war(unsigned* counts, Sample* first, int iters, const float* seed) {
  const float c0 = (float)(1 + (threadIdx.x & 63)), junk = 1.0e6f, two = 2.f;   // integers: the MFMA's own rounding order cannot matter
  (void)seed;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float al = (float)(1 + (it & 7));
    float r0, r1, r2, r3, expect = c0 + 8.f * al;
    if constexpr (FORM == 8)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[40:43], %[a], %[b], v[40:43]\n\tv_mov_b32 v48, %[junk]\n\t" WAR_TAIL("v40", "v41", "v42", "v43") WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 9)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[44:47], %[a], %[b], v[40:43]\n\tv_mov_b32 v42, %[junk]\n\t" WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 10)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[44:47], %[a], %[b], v[40:43]\n\tv_mov_b32 v48, %[junk]\n\tv_mov_b32 v42, %[junk]\n\t" WAR_TAIL("v44", "v45", "v46", "v47")
                   WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 11)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[44:47], %[a], %[b], v[40:43]\n\tv_mov_b32 v48, %[junk]\n\tv_mov_b32 v49, %[junk]\n\tv_mov_b32 v50, %[junk]\n\t"
                            "v_mov_b32 v42, %[junk]\n\t" WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 12)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[38:41], %[a], %[b], v[40:43]\n\tv_mov_b32 v42, %[junk]\n\t" WAR_TAIL("v38", "v39", "v40", "v41") WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 13)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[38:41], %[a], %[b], v[40:43]\n\tv_mov_b32 v48, %[junk]\n\t" WAR_TAIL("v38", "v39", "v40", "v41") WAR_OPS WAR_CLOB);
    else if constexpr (FORM == 14)
      asm volatile(WAR_HEAD "v_mfma_f32_16x16x4_f32 v[50:53], %[b], %[a], 0\n\tv_mfma_f32_16x16x4_f32 v[44:47], %[a], %[b], v[40:43]\n\tv_mov_b32 v42, %[junk]\n\t"
                   WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB);
    else if constexpr (FORM >= 16) {
#define RUN8 "v_mfma_f32_16x16x4_f32 v[60:63], %[b], %[a], 0\n\tv_mfma_f32_16x16x4_f32 v[64:67], %[b], %[a], 0\n\tv_mfma_f32_16x16x4_f32 v[68:71], %[b], %[a], 0\n\t" \
             "v_mfma_f32_16x16x4_f32 v[72:75], %[b], %[a], 0\n\tv_mfma_f32_16x16x4_f32 v[60:63], %[a], %[b], v[60:63]\n\tv_mfma_f32_16x16x4_f32 v[64:67], %[a], %[b], v[64:67]\n\t" \
             "v_mfma_f32_16x16x4_f32 v[68:71], %[a], %[b], v[68:71]\n\tv_mfma_f32_16x16x4_f32 v[72:75], %[a], %[b], v[72:75]\n\t"
#define WAR_CLOB2 WAR_CLOB, "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75"
      if constexpr (FORM == 16)
        asm volatile(WAR_HEAD "v_mov_b32 v58, %[a]\n\tv_mov_b32 v59, %[b]\n\ts_nop 7\n\t" RUN8 "v_mfma_f32_16x16x4_f32 v[44:47], v58, v59, v[40:43]\n\tv_mov_b32 v58, %[junk]\n\t"
                     WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB2);
      else if constexpr (FORM == 17)
        asm volatile(WAR_HEAD "v_mov_b32 v58, %[a]\n\tv_mov_b32 v59, %[b]\n\ts_nop 7\n\t" RUN8 "v_mfma_f32_16x16x4_f32 v[44:47], v58, v59, v[40:43]\n\tv_mov_b32 v59, %[junk]\n\t"
                     WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB2);
      else
        asm volatile(WAR_HEAD "v_mov_b32 v58, %[a]\n\tv_mov_b32 v59, %[b]\n\ts_nop 7\n\t" RUN8 "v_mfma_f32_16x16x4_f32 v[44:47], v58, v59, v[40:43]\n\tv_mov_b32 v42, %[junk]\n\t"
                     WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB2);
    } else {
      // bf16: A = al in every element (al <= 8 is exact), B = 2 in the first four k of every lane: 4 k x 4 lane groups x al x 2 = 32 al ... kept
      // simple: all 8 elements al and 0.25 -> 32 x al x 0.25 = 8 al, the same expectation as the fp32 forms
      asm volatile(WAR_HEAD
                   "v_cvt_pk_bf16_f32 v50, %[a], %[a]\n\tv_mov_b32 v51, v50\n\tv_mov_b32 v52, v50\n\tv_mov_b32 v53, v50\n\t"
                   "v_mov_b32 v54, 0x3e803e80\n\tv_mov_b32 v55, v54\n\tv_mov_b32 v56, v54\n\tv_mov_b32 v57, v54\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x32_bf16 v[44:47], v[50:53], v[54:57], v[40:43]\n\tv_mov_b32 v42, %[junk]\n\t" WAR_TAIL("v44", "v45", "v46", "v47") WAR_OPS WAR_CLOB);
    }
    const bool m0 = __float_as_uint(r0) != __float_as_uint(expect), m1 = __float_as_uint(r1) != __float_as_uint(expect);
    const bool m2 = __float_as_uint(r2) != __float_as_uint(expect), m3 = __float_as_uint(r3) != __float_as_uint(expect);
    if (m0 | m1 | m2 | m3) {
      ++bad;
      if (atomicAdd(&counts[1], 1u) == 0) {
        const int which = m0 ? 0 : m1 ? 1 : m2 ? 2 : 3;
        const float rv[4] = {r0, r1, r2, r3};
        *first = Sample{(unsigned)it, (unsigned)(blockIdx.x * 256 + threadIdx.x), (unsigned)FORM, (unsigned)which, rv[which], expect, c0, junk, 8.f * al, 0.f};
      }
    }
  }
  if (bad) atomicAdd(&counts[0], bad);
}

// ---- forms 19..22: packed fp32 instructions whose LOW result takes the HIGH half of an operand (op_sel) -------------------------
// The debug build of the failing kernel (tools/experiments/upg_dump.py) showed which instruction goes wrong there: the one FMA of the
// four whose weight sits in the high half of its register pair, `v_pk_fma_f32 v[54:55], v[54:55], v[96:97], v[58:59] op_sel:[0,1,0]`,
// returns its ADDEND in the low half of the result on lanes 48..63 (the product is missing), nothing else.
//   19  v_pk_fma_f32 d, t, w, c op_sel:[0,1,0], d = t in place (that instruction)     20  the same, d a fresh pair
//   21  v_pk_fma_f32 op_sel:[1,0,0] (low result from the high half of src0)             22  v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 again)
//   23  v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (low result from the high half of src1: the ORIGINAL failing build's bilinear weights)
//   24  v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]                 25  v_pk_fma_f32 op_sel:[0,0,1] (the ADDEND's high half into the low result)
template <int FORM>
__global__ __launch_bounds__(256, 4) void victim_opsel(unsigned* counts, Sample* first, int iters, const float* seed) {
  const float s0 = seed[threadIdx.x], s1 = seed[256 + threadIdx.x], s2 = seed[512 + threadIdx.x], s3 = seed[768 + threadIdx.x];
  unsigned bad = 0;
  f32x2 c = {s2, s3};
  for (int it = 0; it < iters; ++it) {
    const float k = (float)(it & 7);
    f32x2 t = {s0 + k, s1 - k}, w = {s3 * 0.5f + k, s2 + 0.25f * k}, d;
    asm volatile("" : "+v"(t), "+v"(w), "+v"(c));
    float e_lo, e_hi;
    if constexpr (FORM == 19) {
      d = t;
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]" : "+v"(d) : "v"(w), "v"(c));
      e_lo = sfma(t.x, w.y, c.x), e_hi = sfma(t.y, w.y, c.y);
    } else if constexpr (FORM == 20) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(d) : "v"(t), "v"(w), "v"(c));
      e_lo = sfma(t.x, w.y, c.x), e_hi = sfma(t.y, w.y, c.y);
    } else if constexpr (FORM == 21) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=&v"(d) : "v"(w), "v"(t), "v"(c));
      e_lo = sfma(w.y, t.x, c.x), e_hi = sfma(w.y, t.y, c.y);
    } else if constexpr (FORM == 22) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(d) : "v"(t), "v"(w));
      e_lo = t.y * w.x, e_hi = t.x * w.y;
      asm volatile("" : "+v"(e_lo), "+v"(e_hi));
    } else if constexpr (FORM == 23) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(t), "v"(w));
      e_lo = t.x * w.y, e_hi = t.y * w.x;
      asm volatile("" : "+v"(e_lo), "+v"(e_hi));
    } else if constexpr (FORM == 24) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(d) : "v"(t), "v"(w));
      e_lo = t.x + w.y, e_hi = t.y + w.x;
      asm volatile("" : "+v"(e_lo), "+v"(e_hi));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(d) : "v"(t), "v"(w), "v"(c));
      e_lo = sfma(t.x, w.x, c.y), e_hi = sfma(t.y, w.y, c.y);
    }
    const bool m0 = __float_as_uint(d.x) != __float_as_uint(e_lo), m1 = __float_as_uint(d.y) != __float_as_uint(e_hi);
    if (m0 | m1) {
      ++bad;
      if (atomicAdd(&counts[1], 1u) == 0)
        *first = Sample{(unsigned)it, (unsigned)(blockIdx.x * 256 + threadIdx.x), (unsigned)FORM, m0 ? 0u : 1u, m0 ? d.x : d.y, m0 ? e_lo : e_hi, w.y, m0 ? t.x : t.y,
                        m0 ? c.x : c.y, 0.f};
    }
    c = f32x2{e_lo * 0.001f + s2, e_hi * 0.001f + s3};       // the addend changes every iteration
  }
  if (bad) atomicAdd(&counts[0], bad);
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

static void launch_victim(int form, hipStream_t s, unsigned* counts, Sample* first, int iters, const float* seed, int blocks) {
  switch (form) {
#define V(F) case F: hipLaunchKernelGGL(victim<F>, dim3(blocks), dim3(256), 0, s, counts, first, iters, seed); break;
    V(0) V(1) V(2) V(3) V(4) V(5) V(6) V(7)
#undef V
#define W(F) case F: hipLaunchKernelGGL(victim_war<F>, dim3(blocks), dim3(256), 0, s, counts, first, iters, seed); break;
    W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18)
#undef W
#define O(F) case F: hipLaunchKernelGGL(victim_opsel<F>, dim3(blocks), dim3(256), 0, s, counts, first, iters, seed); break;
    O(19) O(20) O(21) O(22) O(23) O(24) O(25)
#undef O
  }
}
static void launch_burn(int kind, hipStream_t s, float* sink, int iters, int blocks) {
  switch (kind) {
#define B(K) case K: hipLaunchKernelGGL(burn_kernel<K>, dim3(blocks), dim3(256), 0, s, sink, iters); break;
    B(0) B(1) B(2) B(3) B(4)
#undef B
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100, iters = argc > 2 ? atoi(argv[2]) : 4000, blocks = 1024;
  static const char* form_name[26] = {"pk_fma bcast, plain regs", "pk_fma pair, plain regs", "pk_fma bcast <- fresh f32 MFMA", "pk_fma pair  <- fresh f32 MFMA",
                                     "scalar fma   <- fresh f32 MFMA", "pk_fma bcast <- f32 MFMA + s_sleep 2", "pk_add       <- fresh f32 MFMA",
                                     "pk_fma bcast <- fresh bf16 MFMA",
                                      "f32 MFMA in place (control)", "f32 MFMA, SrcC[2] written at +1", "f32 MFMA, SrcC[2] written at +2", "f32 MFMA, SrcC[2] written at +4",
                                      "f32 MFMA vDst overlaps SrcC, write +1", "f32 MFMA vDst overlaps SrcC, no write", "f32 MFMA behind own MFMA, write +1",
                                      "bf16 MFMA (XDL), SrcC[2] written at +1",
                                      "f32 MFMA behind 8 MFMAs, A written +1", "f32 MFMA behind 8 MFMAs, B written +1", "f32 MFMA behind 8 MFMAs, SrcC[2] written +1",
                                      "pk_fma op_sel:[0,1,0] in place", "pk_fma op_sel:[0,1,0]", "pk_fma op_sel:[1,0,0]", "pk_mul op_sel:[1,0] op_sel_hi:[0,1]",
                                      "pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_add op_sel:[0,1] op_sel_hi:[1,0]", "pk_fma op_sel:[0,0,1]"};
  static const char* burn_name[6] = {"32x32x16_bf16", "16x16x32_bf16", "32x32x2_f32", "16x16x4_f32", "pk_fma only", "nothing"};
  float hseed[1280];
  for (int i = 0; i < 1280; ++i) hseed[i] = 0.37f + 0.0131f * (float)((i * 2654435761u) % 97);
  float *seed, *sink;
  unsigned* counts;
  Sample* first;
  CHECK(hipMalloc(&seed, sizeof hseed));
  CHECK(hipMemcpy(seed, hseed, sizeof hseed, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&sink, 1024));
  CHECK(hipMalloc(&counts, 16));
  CHECK(hipMalloc(&first, sizeof(Sample)));
  hipStream_t sv, sb;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%d launches of %d workgroups x 256 lanes x %d iterations per cell; cells: launches with a mismatch / mismatching (lane, iteration)s\n", launches,
         blocks, iters);
  printf("%-40s", "victim form \\ co-runner");
  for (int k = 0; k < 6; ++k) printf(" %16s", burn_name[k]);
  printf("\n");
  const int form_lo = argc > 3 ? atoi(argv[3]) : 0, form_hi = argc > 4 ? atoi(argv[4]) : 25;
  for (int form = form_lo; form <= form_hi; ++form) {
    printf("%-40s", form_name[form]);
    Sample keep{};
    bool have = false;
    for (int kind = 0; kind < 6; ++kind) {
      // how long does a victim launch take alone?  size the co-runner's queue to outlast the victim's
      launch_victim(form, sv, counts, first, iters, seed, blocks);
      CHECK(hipStreamSynchronize(sv));
      CHECK(hipEventRecord(e0, sv));
      launch_victim(form, sv, counts, first, iters, seed, blocks);
      CHECK(hipEventRecord(e1, sv));
      CHECK(hipStreamSynchronize(sv));
      float vms = 0;
      CHECK(hipEventElapsedTime(&vms, e0, e1));
      float bms = 0;
      if (kind < 5) {
        CHECK(hipEventRecord(e0, sb));
        launch_burn(kind, sb, sink, 20000, blocks);
        CHECK(hipEventRecord(e1, sb));
        CHECK(hipStreamSynchronize(sb));
        CHECK(hipEventElapsedTime(&bms, e0, e1));
      }
      int bad_launches = 0;
      unsigned long long bad_items = 0, bad_r2 = 0;
      for (int l = 0; l < launches; ++l) {
        CHECK(hipMemsetAsync(counts, 0, 16, sv));
        CHECK(hipStreamSynchronize(sv));
        if (kind < 5) {
          const int nb = (int)(3.0f * vms / (bms > 0.01f ? bms : 0.01f)) + 2;   // the victim runs ~2x slower beside it
          for (int b = 0; b < nb; ++b) launch_burn(kind, sb, sink, 20000, blocks);
        }
        launch_victim(form, sv, counts, first, iters, seed, blocks);
        CHECK(hipStreamSynchronize(sv));
        unsigned h[4];
        CHECK(hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost));
        if (h[0] || h[2]) {
          ++bad_launches;
          bad_items += h[0];
          bad_r2 += h[2];
          if (!have) {
            CHECK(hipMemcpy(&keep, first, sizeof keep, hipMemcpyDeviceToHost));
            have = h[0] != 0;
          }
        }
        CHECK(hipStreamSynchronize(sb));
      }
      char cell[64];
      snprintf(cell, sizeof cell, "%d / %llu%s", bad_launches, bad_items, bad_r2 ? "*" : "");
      printf(" %16s", cell);
      fflush(stdout);
    }
    printf("\n");
    if (have) {
      const float stale = fmaf(keep.w, keep.g, keep.c_prev), zero = fmaf(keep.w, keep.g, 0.f);
      if (form >= 19)
        printf("    first mismatch: iteration %u, lane %u (lane %u of its wave), %s half: got %.9g, expected %.9g; weight %.9g operand %.9g addend %.9g\n", keep.it, keep.lane,
               keep.lane & 63, keep.which ? "high" : "low", keep.p, keep.e, keep.w, keep.g, keep.c);
      else if (form >= 8)
        printf("    first mismatch: iteration %u, lane %u, element %u: got %.9g, expected %.9g (SrcC %.9g + product %.9g; the value written over SrcC[2]: %.9g)\n", keep.it,
               keep.lane, keep.which, keep.p, keep.e, keep.w, keep.c, keep.g);
      else
      printf("    first mismatch: iteration %u, lane %u, element %u: got %.9g (0x%08x), expected %.9g (0x%08x); w %.9g g %.9g MFMA result %.9g "
             "(previous iteration's %.9g)\n    candidates: fma on the previous MFMA result %.9g, on 0 %.9g, w*g alone %.9g, MFMA result alone %.9g\n",
             keep.it, keep.lane, keep.which, keep.p, *(unsigned*)&keep.p, keep.e, *(unsigned*)&keep.e, keep.w, keep.g, keep.c, keep.c_prev, stale, zero,
             keep.w * keep.g, keep.c);
    }
  }
  printf("(* = the second MFMA's own result, read by a scalar compare, was wrong at least once)\n");
  return 0;
}
