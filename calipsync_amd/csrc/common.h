// Shared declarations for the casync HIP engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/casync_hip.h"

#define CASYNC_LRELU_SLOPE 0.01f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * CASYNC_LRELU_SLOPE; }

// thread-local error text behind casync_last_error()
void casync_set_error(const char* fmt, ...);

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

// ---- GEMM (1x1 conv / linear) -------------------------------------------
struct GemmEpilogue {
  const float* bias = nullptr;       // [N]
  const float* pre_res = nullptr;    // [M, ld_pre]  added before the activation ...
  const float* pre_scale = nullptr;  // [N]          ... scaled per column (null = 1)
  int ld_pre = 0;
  int act = 0;                       // LeakyReLU(0.01)
  const float* post_res = nullptr;   // [M, ld_post] added after the activation
  int ld_post = 0;
  const float* aff_s = nullptr;      // [N] v = lrelu(v*aff_s + aff_t) on the OUTPUT (aff_on_acc=0)
  const float* aff_t = nullptr;      //     or on the running accumulator (aff_on_acc=1)
  int aff_on_acc = 0;
  const float* acc_in = nullptr;     // running sum: acc_out = acc_in + v
  float* acc_out = nullptr;
  int ld_acc = 0;
};

const char* pw_gemm_kernel_name(int m, int n);
int launch_pw_gemm(const float* a, int lda, const float* w, float* c, int ldc, int m, int n, int k,
                   const GemmEpilogue& epi, hipStream_t stream);

// ---- other operators -------------------------------------------------------
int launch_dw3x3(const float* in, const float* w, const float* bias, float* out, int batch, int h,
                 int wdt, int c, int stride, hipStream_t stream);
bool ir_fused_supported(int cin, int cout, int stride);
const char* ir_fused_kernel_name(int cin, int cout, int stride);
bool ir_fused_up_supported(int cin, int cout);
int launch_ir_fused_up(const float* lo, int ld_lo, int c_lo, const float* in, int ld_in, const float* w1,
                       const float* b1, const float* wd, const float* bd, const float* w2,
                       const float* b2, float* out, int ld_out, int batch, int h, int w, int cin,
                       int cout, hipStream_t stream);
int launch_ir_fused(const float* in, int ld_in, const float* w1, const float* b1, const float* wd,
                    const float* bd, const float* w2, const float* b2, float* out, int ld_out,
                    int batch, int h, int w, int cin, int cout, int stride, int res,
                    hipStream_t stream);
int launch_im2col3x3(const float* in, float* out, int batch, int h, int wdt, int c, int stride,
                     int pad, hipStream_t stream);
int launch_upsample2x(const float* in, float* out, int ldc, int batch, int h, int wdt, int c,
                      hipStream_t stream);
int launch_cross_attention(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           const float* res, int ld_res, const float* gamma_dev, float* out,
                           int ld_out, int batch, hipStream_t stream);
int launch_nchw_to_nhwc(const float* in, float* out, int batch, int c, int hw, hipStream_t stream);
int launch_inc(const float* x_nchw, const float* packed_inc, float* out, int ldc, int batch,
               hipStream_t stream);
int launch_outc(const float* in, int ld_in, const float* w, const float* b, float* out_nchw,
                int batch, hipStream_t stream);

// packed sub-offsets of the `inc` block inside its packed tensors (floats)
constexpr int INC_CIN = 6, INC_CEXP = 12, INC_COUT = 32;
