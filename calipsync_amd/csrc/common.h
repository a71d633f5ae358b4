// Shared declarations for the casync HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/casync_hip.h"

#define CASYNC_LRELU_SLOPE 0.01f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Activation storage type of an engine / operator call.  Arithmetic is always fp32
// (accumulate, bias, activation, softmax); DT_BF16 stores activations and feeds the matrix
// cores as bf16 (BASELINE configs[2]).
enum DType { DT_F32 = 0, DT_BF16 = 1 };
inline int dtype_size(int dt) { return dt == DT_BF16 ? 2 : 4; }

// 4 consecutive activation elements <-> float4 (the unit every HBM-bound kernel moves per lane)
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * CASYNC_LRELU_SLOPE; }

// Bilinear x2 upsample with align_corners=True (module/unet.py:86-91), the arithmetic of ATen's upsample_bilinear2d
// evaluated exactly as written: src = dst * (in-1)/(out-1), l1 = frac, l0 = 1 - l1, taps combined as
// l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11).  No fma contraction, so every kernel that folds the upsample
// (upsample2x_kernel, ir_fused) produces the SAME bits -- what the compiler fuses would otherwise differ
// from one surrounding code to the next.
struct UpsTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpsTap ups_tap(float scale, int dst, int n_in) {
#pragma clang fp contract(off)
  const float f = scale * (float)dst;
  UpsTap t;
  t.i0 = (int)f;
  t.i1 = t.i0 + (t.i0 < n_in - 1);
  t.l1 = f - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}
// (n_out / 2 - 1) / (n_out - 1): the scale of the x2 align_corners=True upsample.  The model's resolutions are folded at compile
// time (the same correctly rounded quotients the runtime division gives), other sizes divide.
__device__ __forceinline__ float ups_scale(int n_out) {
  switch (n_out) {
    case 20: return 9.f / 19.f;
    case 40: return 19.f / 39.f;
    case 80: return 39.f / 79.f;
    case 160: return 79.f / 159.f;
  }
  return (float)((n_out >> 1) - 1) / (float)(n_out - 1);
}
// (element by element, each value pinned in a register: the vector form multiplies by the scalar weights with packed instructions
//  that may take a weight from the HIGH half of a register pair -- the gfx950 erratum of round 6 (zero on lanes 48..63 beside
//  another wave's v_mfma_f32_16x16x32_bf16; tools/isa_pk_opsel.py); the un-fused Up path returned wrong patches in 40 of 40
//  forwards beside a bf16 model with it, in none with this form.  Same operations in the same order: the same bits.)
__device__ __forceinline__ f32x4 ups_lerp(const UpsTap& ty, const UpsTap& tx, f32x4 v00, f32x4 v01, f32x4 v10, f32x4 v11) {
#pragma clang fp contract(off)
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = tx.l0 * v00[e], b = tx.l1 * v01[e], c = tx.l0 * v10[e], d = tx.l1 * v11[e];
    asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    float top = a + b, bot = c + d;
    asm("" : "+v"(top), "+v"(bot));
    float t2 = ty.l0 * top, b2 = ty.l1 * bot;
    asm("" : "+v"(t2), "+v"(b2));
    r[e] = t2 + b2;
  }
  return r;
}

// thread-local error text behind casync_last_error()
void casync_set_error(const char* fmt, ...);

// ---- tuning options ---------------------------------------------------------------------------
// Every switch of the engine.  The process defaults are read from the CASYNC_* environment ONCE (first
// use); casync_create copies them into the handle; casync_set_option changes a handle's copy (or, with
// a null handle, the process defaults that the single-operator API and later handles see).  Nothing on
// a launch path calls getenv.
struct CasyncOptions {
  int lanes = 2;             // CASYNC_LANES: concurrent sub-batch lanes (1..4) from 2*16 frames up
  int trunk_lanes = 0;       // CASYNC_TRUNK_LANES: lanes of the 10x10 trunk (0 = same as `lanes`; 1 = the
                             //   lanes join before the fusion MLP and the trunk runs as one stream-K lane)
  int lane_skew = 0;         // CASYNC_LANE_SKEW: two lanes of batch/2 + skew and batch/2 - skew frames (experiment)
  int overlap = 1;           // CASYNC_OVERLAP: audio encoder on its own stream per lane
  int gemm_streamk = 1;      // CASYNC_GEMM_STREAMK: K may be split (stream-K remainders, the small-M tile) in single-lane runs; 0 = batch-invariant bits
  int gemm_glds = 2;         // CASYNC_GEMM_GLDS: LDS-DMA ring GEMM: 0 off, 1 bf16 only, 2 both types
  int gemm_cfg = -1;         // CASYNC_GEMM_CFG: force one tile configuration
  int gemm_ring128 = 0;      // CASYNC_GEMM_RING128: bf16 128x128 launches of more than 256 tiles on a two-stage LDS-DMA ring, two workgroups per CU
                             //   (round-6 experiment; 0 = the register-staged 128x128 kernel)
  int gemm_single64 = 4096;  // CASYNC_GEMM_SINGLE64: single-lane fp32 launches of at most this many 64x64 tiles take 64x64 tiles only (0 = cost model)
  int skip_early = 12;       // CASYNC_SKIP_EARLY: below this many frames (single lane, fp32) the skip half of up1.0 / up2.0's expand conv runs
                             //   on the second stream beside the trunk and the decoder only adds up(W1a . lo) inside the depthwise kernel (0 = off)
  int gemm_small_m = 1;      // CASYNC_GEMM_SMALL_M: 64x32 tiles with the K split inside the workgroup for small-M launches of the single-lane plan
  int gemm_persist = 1;      // CASYNC_GEMM_PERSIST: persistent grid of the register-staged GEMM
  int lane_streamk = 0;      // CASYNC_LANE_STREAMK: stream-K also when two or more lanes run side by side (the other lane fills tails otherwise)
  int gemm_conc = 3;         // CASYNC_GEMM_CONC: tile policy when lanes share the chip
  int gemm_conc_tiles = 8192;  // CASYNC_GEMM_CONC_TILES (2048 until the low-resolution W1a GEMMs of the Up blocks: +0.3 % with them on 64x64)
  int fuse_ir = 1;           // CASYNC_FUSE_IR: fused inverted-residual kernel
  int fuse_up = 1;           // CASYNC_FUSE_UP: bilinear upsample folded into up3/up4
  int fuse_min_hw = 32;      // CASYNC_FUSE_MIN_HW: lowest resolution the fused kernel is used at
  int fuse_q = 1;            // CASYNC_FUSE_Q: query projection as 64 extra columns of the p_1 GEMM
  int ups_commute = 2;       // CASYNC_UPS_COMMUTE: Up blocks run the upsampled half of their expand conv at the low resolution
                             //   (upsample and 1x1 conv commute), fp32: 1 = the unfused blocks up1.0 / up2.0, 2 = also inside
                             //   the fused kernel (up3.0 / up4.0); 0 = upsample first, as the reference writes it
  int fuse_dw = 2;           // CASYNC_FUSE_DW: expand GEMM + depthwise 3x3 in one kernel (pw_dw.hip), fp32: 1 = the 10x10 / 16x16 /
                             //   20x20 blocks (whole-frame tiles), 2 = also the 40x40 blocks (row strips)
  int fuse_dw_min = 12;      // CASYNC_FUSE_DW_MIN: frames per launch from which the whole-frame tiles (10x10 / 16x16 / 20x20) are used
                             //   (round 4, column-walking epilogue: B=12 1.609 -> 1.584 ms with them, B=8 1.261 -> 1.275 ms)
  int fuse_dw_min40 = 8;     // CASYNC_FUSE_DW_MIN40: frames per launch from which the 40x40 strips are used (5 strips per frame:
                             //   B=8 1.312 -> 1.296 ms, B=1 0.841 -> 0.848 ms)
  int fuse_dw_deep = 10;     // CASYNC_FUSE_DW_DEEP: launches of 2 .. this - 1 frames take one-frame tiles with 128-B k-tile rows and a four-stage
                             //   ring (10x10 / 16x16, stride 1): the small-batch form of the fused expand + depthwise kernel (0 = off)
  int fuse_dw_bf16 = 2;      // CASYNC_FUSE_DW_BF16: the same fusion in the bf16 engine (pw_dw_bf16.hip: 64-channel tiles, bf16 E image):
                             //   1 = 10x10 / 16x16 / 20x20 blocks, 2 = also the 40x40 strips, 0 = GEMM + depthwise launches
  int fuse_dw_bf16_bn = 128; // CASYNC_FUSE_DW_BF16_BN: channel tile of its 10x10 / 16x16 instances (64 or 128)
  int ups_commute_bf16 = 1;  // CASYNC_UPS_COMMUTE_BF16: up1.0 / up2.0 of the bf16 engine run the upsampled half of their expand conv at the low
                             //   resolution and pw_dw_bf16 adds its upsample (needs fuse_dw_bf16 >= 2 and >= fuse_dw_bf16_min frames per launch)
  int fuse_dw_bf16_min = 12; // CASYNC_FUSE_DW_BF16_MIN: frames per launch from which it is used
  int ir_dw_mfma = 1;        // CASYNC_IR_DW_MFMA: the bf16 fused inverted-residual block runs its depthwise 3x3 on the matrix pipe (block-diagonal
                             //   v_mfma_f32_16x16x32_bf16, taps rounded to bf16): 1 = in the instances where that measured faster (all but
                             //   up4.0's 64 -> 128 -> 32), 2 = in all, 0 = the VALU form with fp32 taps (round 5)
  int inc_mfma = 0;          // CASYNC_INC_MFMA: the bf16 engine's `inc` block projects 12 -> 32 channels on the matrix pipe (inc_bf16_kernel: the kernel
                             //   0.49 -> 0.37 ms per B = 512 step, end to end +0.5 %, the x1 tap's max error 2.4e-3 -> 3.7e-3: off by default)
  int bf16_plan = 256;       // CASYNC_BF16_PLAN: frames per forward from which the bf16 engine runs its large-batch plan -- three lanes (when `lanes`
                             //   is at its default 2), `gemm_ring128`, the audio encoder on the lane's own stream (`overlap` 0); 0 = never
  int dw_lds = 1;            // CASYNC_DW_LDS: LDS-slab depthwise kernel
  int dw_lds_bytes = 32768;  // CASYNC_DW_LDS_BYTES
  int att_nz = 0;            // CASYNC_ATT_NZ: channel split of the attention kernel (0 = by batch)
  int att_bf16 = 1;          // CASYNC_ATT_BF16: the bf16 engine's attention core on bf16 matrix instructions (attention_bf16.hip); 0 = the
                             //   fp32-MFMA kernel of attention.hip on bf16 storage
  int kv_early = 1;          // CASYNC_KV_EARLY: the attention K/V projection GEMM runs on the audio stream beside the face encoder:
                             //   1 = in single-lane runs (small batches), 2 = always, 0 = never (between fusion MLP and attention)
};
CasyncOptions& casync_default_options();      // process defaults (environment read once, thread-safe)
const CasyncOptions& casync_opts();           // options of the call in progress on this thread
int casync_option_ref(CasyncOptions& o, const char* name, int** slot);   // name -> field (CASYNC_ERR_ARG if unknown)
struct CasyncOptScope {                       // makes `o` the current options of this thread for its lifetime
  const CasyncOptions* prev;
  explicit CasyncOptScope(const CasyncOptions* o);
  ~CasyncOptScope();
};
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `once_mask` is a per-kernel-instance
// bit mask of devices already done (atomic; setting the attribute twice is harmless)
int casync_ensure_dyn_lds(unsigned long long* once_mask, const void* fn, int bytes);

#define CASYNC_CHECK_HIP(expr)                                                        \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                          \
      casync_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
      return CASYNC_ERR_HIP;                                                          \
    }                                                                                 \
  } while (0)

#define CASYNC_REQUIRE(cond, ...)                \
  do {                                           \
    if (!(cond)) {                               \
      casync_set_error(__VA_ARGS__);             \
      return CASYNC_ERR_ARG;                     \
    }                                            \
  } while (0)

// 16 bytes of activations (4 floats or 8 bf16) as fp32 lanes: the widest per-lane access
template <typename T> struct V16 {
  static constexpr int N = 16 / (int)sizeof(T);
  float v[N];
};
template <typename T> __device__ __forceinline__ V16<T> ld16(const T* p);
template <> __device__ __forceinline__ V16<float> ld16<float>(const float* p) {
  const f32x4 x = *reinterpret_cast<const f32x4*>(p);
  return V16<float>{{x[0], x[1], x[2], x[3]}};
}
template <> __device__ __forceinline__ V16<bf16_t> ld16<bf16_t>(const bf16_t* p) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
  V16<bf16_t> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = (float)x[i];
  return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const V16<T>& r);
template <> __device__ __forceinline__ void st16<float>(float* p, const V16<float>& r) {
  *reinterpret_cast<f32x4*>(p) = f32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
}
template <> __device__ __forceinline__ void st16<bf16_t>(bf16_t* p, const V16<bf16_t>& r) {
  bf16x8 x;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (bf16_t)r.v[i];
  *reinterpret_cast<bf16x8*>(p) = x;
}

// ---- GEMM (1x1 conv / linear) -------------------------------------------
// Pointers marked (T) have the activation storage type of the call; the rest are fp32.
struct GemmEpilogue {
  const float* bias = nullptr;       // [N]
  const void* pre_res = nullptr;     // (T) [M, ld_pre]  added before the activation ...
  const float* pre_scale = nullptr;  // [N]          ... scaled per column (null = 1)
  int ld_pre = 0;
  int act = 0;                       // LeakyReLU(0.01)
  const void* post_res = nullptr;    // (T) [M, ld_post] added after the activation
  int ld_post = 0;
  const float* aff_s = nullptr;      // [N] v = lrelu(v*aff_s + aff_t) on the OUTPUT (aff_on_acc=0)
  const float* aff_t = nullptr;      //     or on the running accumulator (aff_on_acc=1)
  int aff_on_acc = 0;
  // + bilinear x2 upsample (align_corners=True) of a LOW-resolution tensor, added before the activation: the rows of C
  // are the pixels (b, y, x) of ups_h x ups_w frames, ups_src is [B * ups_h/2 * ups_w/2, ups_ld].  This is how an Up
  // block's expand conv takes its upsampled half: W1 . cat(up(lo), skip) = up(W1a . lo) + W1b . skip (module/unet.py:90-96)
  const void* ups_src = nullptr;     // (T)
  int ups_ld = 0, ups_h = 0, ups_w = 0;
  const void* acc_in = nullptr;      // (T) running sum: acc_out = acc_in + v
  void* acc_out = nullptr;           // (T)
  int ld_acc = 0;
  // stream-K scratch (null = plain data-parallel tiles): kStreamKFloats floats of partial tiles and
  // kStreamKCounters zeroed counters, private to the stream the GEMM runs on
  float* sk_ws = nullptr;
  unsigned* sk_cnt = nullptr;
  unsigned long long* stamps = nullptr;   // diagnostic only: 8 words per workgroup (see pw_gemm_glds_kernel)
  unsigned buf_a_bytes = 0, buf_w_bytes = 0;   // extents of A and W for the buffer-addressed loads (set by the launcher)
  // the launch shares the chip with another lane's kernels (two-lane schedule): tile choice then
  // favours many small workgroups that interleave on a CU over few large ones (see pick_cfg)
  int concurrent = 0;
  // implicit-GEMM 3x3 convolution (conv_on): A is the NHWC input [B,H,W,C]; row m of the GEMM is output
  // pixel (b, oy, ox), column k = (ky*3 + kx)*C + c reads in[b, oy*stride+ky-pad, ox*stride+kx-pad, c]
  // (zero outside the image); W is [N][(ky,kx,c)].  K = 9*C, C a multiple of the 128-B k-tile.
  int conv_on = 0, conv_h = 0, conv_w = 0, conv_c = 0, conv_ho = 0, conv_wo = 0, conv_stride = 0, conv_pad = 0;
};
constexpr int kStreamKWgs = 256;                                  // stream-K workgroups (one per CU)
constexpr long long kStreamKFloats = 2ll * kStreamKWgs * 128 * 64;  // two partial tiles per workgroup, up to 128x64
constexpr int kStreamKCounters = 256;
constexpr long long kStreamKBytes = kStreamKFloats * 4 + kStreamKCounters * 4;

const char* pw_gemm_kernel_name(int m, int n, int k, bool stream_k, int dtype = DT_F32, bool concurrent = false,
                                bool ups = false);
// a, w, c (and the (T) epilogue pointers) are `dtype` elements; lda/ldc in elements
int launch_pw_gemm(const void* a, int lda, const void* w, void* c, int ldc, int m, int n, int k,
                   const GemmEpilogue& epi, hipStream_t stream, int dtype = DT_F32);

// dense 3x3 convolution + bias (+ LeakyReLU if epi.act) as an implicit GEMM on the ring kernel: no im2col
// buffer, the taps are gathered by the LDS-DMA loads themselves.  in: [B,H,W,cin] contiguous NHWC,
// w: [cout][(ky,kx,cin)], out: [B*Ho*Wo, ldc].
int launch_conv3x3_gemm(const void* in, const void* w, void* out, int ldc, int batch, int h, int wdt, int cin,
                        int cout, int stride, int pad, const GemmEpilogue& epi, hipStream_t stream,
                        int dtype = DT_F32);
const char* conv3x3_gemm_kernel_name(int batch, int h, int wdt, int cin, int cout, int stride, int pad,
                                     int dtype = DT_F32, bool concurrent = false, bool stream_k = false);

// ---- other operators -------------------------------------------------------
int launch_dw3x3(const void* in, const float* w, const float* bias, void* out, int batch, int h,
                 int wdt, int c, int stride, hipStream_t stream, int dtype = DT_F32);
const char* dw3x3_kernel_name(int h, int wdt, int c, int stride, int dtype = DT_F32);
int launch_dw3x3_ups(const float* pre, const float* g, int ldg, const float* w, const float* bias, float* out, int batch, int h,
                     int wdt, int c, hipStream_t stream);
const char* dw3x3_ups_kernel_name(int h, int wdt, int c);
bool ir_fused_supported(int cin, int cout, int stride);
// as rocprofv3 prints it; h, w > 0: the instance launch_ir_fused / launch_ir_fused_up picks for that shape
const char* ir_fused_kernel_name(int cin, int cout, int stride, int dtype = DT_F32, bool ups = false, int h = 0, int w = 0);
bool ir_fused_up_supported(int cin, int cout);
int launch_ir_fused_up(const void* lo, int ld_lo, int c_lo, const void* in, int ld_in, const void* w1,
                       const float* b1, const float* wd, const float* bd, const void* w2,
                       const float* b2, void* out, int ld_out, int batch, int h, int w, int cin,
                       int cout, hipStream_t stream, int dtype = DT_F32);
// the Up block with the upsample commuted behind the expand conv (fp32): g = W1a * lo at the low resolution
int launch_ir_fused_upg(const float* g, int ld_g, const float* in, int ld_in, const float* w1, const float* b1,
                        const float* wd, const float* bd, const float* w2, const float* b2, float* out, int ld_out,
                        int batch, int h, int w, int cin, int cout, hipStream_t stream);
const char* ir_fused_upg_kernel_name(int cin, int cout);
// w1 / w2 are in the call's storage type (fp32 or bf16); b1, wd, bd, b2 are always fp32
int launch_ir_fused(const void* in, int ld_in, const void* w1, const float* b1, const float* wd,
                    const float* bd, const void* w2, const float* b2, void* out, int ld_out,
                    int batch, int h, int w, int cin, int cout, int stride, int res,
                    hipStream_t stream, int dtype = DT_F32);
// expand 1x1 + depthwise 3x3 of a low-resolution inverted residual in one kernel (pw_dw.hip): fp32, 10x10 / 16x16 /
// 20x20 frames; a [frames*hw*hw, lda], w1 [cexp][cin], wd [9][cexp], d [frames*ho*ho, ldd]
bool pw_dw_supported(int hw, int cin, int cexp, int stride);
const char* pw_dw_kernel_name(int hw, int cin, int frames, int stride = 1);
bool pw_dw_deep(int hw, int frames, int stride, bool ups, int cin);   // the launch takes the deep-ring one-frame tiles
// ups (optional): low-resolution addend [frames*(hw/2)^2, ld_ups] whose bilinear x2 upsample is added before the first
// activation (an Up block's upsampled half, see GemmEpilogue::ups_src)
int launch_pw_dw(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd, void* d, int ldd,
                 int frames, int hw, int stride, int cin, int cexp, hipStream_t stream, const void* ups = nullptr, int ld_ups = 0);
// the bf16 engine's expand + depthwise kernel (pw_dw_bf16.hip): a, w1, d bf16; b1, wd, bd fp32; cin % 32 == 0, cexp % 64 == 0
bool pw_dw_bf16_supported(int hw, int cin, int cexp, int stride);
const char* pw_dw_bf16_kernel_name(int hw, int cexp, int frames, int stride = 1);
// ups (optional, 20x20 / 40x40 stride 1): bf16 low-resolution addend [frames*(hw/2)^2, ld_ups], see launch_pw_dw
bool pw_dw_bf16_takes_ups(int hw, int stride);
int launch_pw_dw_bf16(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd, void* d, int ldd,
                      int frames, int hw, int stride, int cin, int cexp, hipStream_t stream, const void* ups = nullptr, int ld_ups = 0);
int launch_upsample2x(const void* in, void* out, int ldc, int batch, int h, int wdt, int c,
                      hipStream_t stream, int dtype = DT_F32);
int launch_cross_attention(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                           const void* res, int ld_res, const float* gamma_dev, void* out,
                           int ld_out, int batch, hipStream_t stream, int dtype = DT_F32);
const char* cross_attention_kernel_name(int dtype);
// the bf16 engine's attention core (attention_bf16.hip): q, k, v, res, out bf16; one workgroup per frame
int launch_cross_attention_bf16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* res, int ld_res,
                                const float* gamma_dev, void* out, int ld_out, int batch, hipStream_t stream);
int launch_nchw_to_nhwc(const float* in, void* out, int batch, int c, int hw, hipStream_t stream,
                        int dtype = DT_F32);
int launch_crop_to_input(const unsigned char* crops, float* x, int batch, hipStream_t stream);
int launch_pred_to_u8(const float* pred, unsigned char* out, int batch, hipStream_t stream);
int launch_audio_window_gather(const float* features, int n_steps, const int* idx_dev, void* out, int batch,
                               hipStream_t stream, int dtype = DT_F32);
int launch_audio_windows_nchw(const float* features, int n_steps, const int* idx_dev, float* out, int batch,
                              hipStream_t stream);
int launch_inc(const float* x_nchw, const float* packed_inc, void* out, int ldc, int batch,
               hipStream_t stream, int dtype = DT_F32);
int launch_outc(const void* in, int ld_in, const float* w, const float* b, float* out_nchw,
                int batch, hipStream_t stream, int dtype = DT_F32);

// packed sub-offsets of the `inc` block inside its packed tensors (floats)
constexpr int INC_CIN = 6, INC_CEXP = 12, INC_COUT = 32;
