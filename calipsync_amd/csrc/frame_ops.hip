// Image arithmetic of FrameSynthesizer.process_batch on the device (reference
// image_infer_v1/tools/frame_synthesizer/infer_api.py:234-239 and 263-346): every per-frame cv2 / numpy call
// around the model becomes a batched kernel over ragged per-frame regions, so a batch costs one upload of the
// crop regions and ONE download of the blended regions instead of B x (cv2.resize x2, fillPoly, dilate,
// blend) on the host and B separate .cpu() copies.
//
//   prepare:    region (h x w x 3 u8) --cv2.resize(INTER_LINEAR)--> 168 x 168 x 3      infer_api.py:234-235
//   synth:      168-crop with [4:164,4:164] := uint8(pred*255) --cv2.resize--> width x width x 3   :265-277
//   polymask:   cv2.fillPoly of the 33 contour points (scan-line fill + every edge drawn)           :280-291
//   dilate:     (2e+1)^2 window maximum, e = max(1, int(sqrt(area/pi) * 0.15))                     :294-301
//   blend:      crop*m + frame*(1-m) in float64, truncated into the uint8 frame                    :314-345
//
// All of it is byte / integer work bound by HBM (a few MB per frame); the arithmetic restates OpenCV 4.x's
// published C++ paths (resize.cpp fixed-point INTER_LINEAR, drawing.cpp FillEdgeCollection / LineIterator /
// clipLine) operation for operation -- oracle/frame_ops_oracle.py is the same restatement on the CPU and the
// -m gpu tests demand bit equality with it.  cv2 itself is not available here: parity with the real library
// is UNPINNED.
//
// Floating point that must round like the host's (numpy, scalar C++): contraction is switched off in this file.
#pragma clang fp contract(off)
#include "common.h"

namespace {

constexpr int GEOM = 12;   // int32 words per frame: see casync_hip.h (casync_frame_geom)
enum { G_REG_OFF = 0, G_H, G_W, G_WIDTH, G_VALID, G_SYNTH_OFF, G_MASK_OFF, G_FMASK_KIND, G_FMASK_H, G_FMASK_W, G_FMASK_LO, G_FMASK_HI };
enum { FMASK_NONE = -1, FMASK_F32 = 0, FMASK_U8 = 1 };   // the optional frame mask: float32 in [0,1], or uint8 (= value / 255)
constexpr int NPTS = 33;
constexpr int XY_SHIFT = 16;
constexpr long long XY_ONE = 1ll << XY_SHIFT;
// Left end of a fillPoly span: x1 = (xa + FILL_LEFT_DELTA) >> XY_SHIFT.  This file (and oracle/frame_ops_oracle.py)
// restate FillEdgeCollection of OpenCV 4.x drawing.cpp in its ceil(xa) form (XY_ONE - 1; the form of 3.4 and early 4.x as recalled).  Later
// 4.x sources carry a `delta` that is 0 for line types below LINE_AA (fillPoly's default LINE_8), i.e. floor(xa): one
// pixel more on the left edge wherever the outline has not already drawn it.  cv2 is absent from the build image, so
// which one the reference's installed version uses is UNPINNED; the rule is this one constant, in both files, and
// tests/test_frame_ops.py::test_oracle_against_cv2 decides it wherever cv2 exists.
constexpr long long FILL_LEFT_DELTA = XY_ONE - 1;

// ---- cv::resize INTER_LINEAR tables, one destination index at a time (resize.cpp) ----
struct Tap { int s0, s1; float f; };
__device__ __forceinline__ Tap linear_tap_x(int d, int src, int dst) {
  const double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  return Tap{s, s + 1 < src ? s + 1 : src - 1, f};     // weight of s1 is 0 wherever s + 1 leaves the row
}
__device__ __forceinline__ Tap linear_tap_y(int d, int src, int dst) {   // rows are clipped, weights kept
  const double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  const int y0 = s < 0 ? 0 : (s > src - 1 ? src - 1 : s);
  const int y1 = s + 1 < 0 ? 0 : (s + 1 > src - 1 ? src - 1 : s + 1);
  return Tap{y0, y1, f};
}
__device__ __forceinline__ int coef(float c) { return __float2int_rn(c * 2048.f); }   // saturate_cast<short>(c * 2048): cvRound

// 8-bit bilinear sample of destination pixel (dy, dx), channel c, of a (sh x sw) -> (dh x dw) resize.
// fetch(y, x, c) returns the source byte.
template <class F>
__device__ __forceinline__ unsigned char resize_u8_at(F fetch, int sh, int sw, int dh, int dw, int dy, int dx, int c) {
  if (sh == dh && sw == dw) return fetch(dy, dx, c);
  if (sw == 2 * dw && sh == 2 * dh)   // exact 2x decimation: cv::resize switches INTER_LINEAR to INTER_AREA
    return (unsigned char)((fetch(2 * dy, 2 * dx, c) + fetch(2 * dy, 2 * dx + 1, c) + fetch(2 * dy + 1, 2 * dx, c) +
                            fetch(2 * dy + 1, 2 * dx + 1, c) + 2) >> 2);
  const Tap tx = linear_tap_x(dx, sw, dw), ty = linear_tap_y(dy, sh, dh);
  const int a0 = coef(1.f - tx.f), a1 = coef(tx.f), b0 = coef(1.f - ty.f), b1 = coef(ty.f);
  const int h0 = fetch(ty.s0, tx.s0, c) * a0 + fetch(ty.s0, tx.s1, c) * a1;     // HResizeLinear (int, unshifted)
  const int h1 = fetch(ty.s1, tx.s0, c) * a0 + fetch(ty.s1, tx.s1, c) * a1;
  const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;   // VResizeLinear 8u fixed point
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// ---------------------------------------------------------------- prepare: region -> 168 x 168
__global__ __launch_bounds__(256) void frame_resize168_kernel(const unsigned char* __restrict__ regions,
                                                              const int* __restrict__ geom,
                                                              unsigned char* __restrict__ crops) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= 168 * 168) return;
  const int* g = geom + b * GEOM;
  const int h = g[G_H], w = g[G_W];
  const unsigned char* src = regions + g[G_REG_OFF];
  const int dy = p / 168, dx = p - dy * 168;
  auto fetch = [&](int y, int x, int c) -> int { return src[((size_t)y * w + x) * 3 + c]; };
  unsigned char* dst = crops + ((size_t)b * 168 * 168 + p) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c] = resize_u8_at(fetch, h, w, 168, 168, dy, dx, c);
}

// ---------------------------------------------------------------- synth: patched 168-crop -> width x width
__global__ __launch_bounds__(256) void frame_synth_kernel(const unsigned char* __restrict__ crops,
                                                          const float* __restrict__ pred,
                                                          const int* __restrict__ geom,
                                                          unsigned char* __restrict__ synth) {
  const int b = blockIdx.y;
  const int* g = geom + b * GEOM;
  const int width = g[G_WIDTH];
  if (!g[G_VALID]) return;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)width * width) return;
  const unsigned char* crop = crops + (size_t)b * 168 * 168 * 3;
  const float* pr = pred + (size_t)b * 3 * 25600;
  // crop_img[4:164, 4:164] = uint8(pred * 255)  (infer_api.py:265-266, 276): never materialised
  auto fetch = [&](int y, int x, int c) -> int {
    if (y >= 4 && y < 164 && x >= 4 && x < 164) return (int)(unsigned char)(pr[c * 25600 + (y - 4) * 160 + (x - 4)] * 255.0f);
    return crop[(y * 168 + x) * 3 + c];
  };
  const int dy = (int)(p / width), dx = (int)(p - (long long)dy * width);
  unsigned char* dst = synth + g[G_SYNTH_OFF] + p * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c] = resize_u8_at(fetch, 168, 168, width, width, dy, dx, c);
}

// ---------------------------------------------------------------- fillPoly: scan-line spans
// One 64-thread workgroup per (frame, row).  Lane 0 builds the sorted crossing list of the row exactly as
// FillEdgeCollection sees it (edges with y0 <= y < y1, x = x0 + (y - y0) * dx in 16.16 fixed point), the
// wave fills the even-odd spans [ceil(xa), floor(xb)].
__global__ __launch_bounds__(64) void frame_polyfill_kernel(const int* __restrict__ geom, const int* __restrict__ pts,
                                                            unsigned char* __restrict__ mask) {
  const int b = blockIdx.y, y = blockIdx.x;
  const int* g = geom + b * GEOM;
  const int h = g[G_H], w = g[G_W];
  if (!g[G_VALID] || y >= h) return;
  __shared__ long long xs[NPTS + 1];
  __shared__ int n_cross;
  if (threadIdx.x == 0) {
    const int* pt = pts + b * NPTS * 2;
    int n = 0, n_edges = 0;
    int p0x = pt[(NPTS - 1) * 2], p0y = pt[(NPTS - 1) * 2 + 1];
    for (int i = 0; i < NPTS; ++i) {
      const int p1x = pt[i * 2], p1y = pt[i * 2 + 1];
      if (p0y != p1y) {
        ++n_edges;
        const long long x0f = (long long)p0x << XY_SHIFT, x1f = (long long)p1x << XY_SHIFT;
        const long long dx = (x1f - x0f) / (long long)(p1y - p0y);        // int64 division truncates (PolyEdge::dx)
        const int ey0 = p0y < p1y ? p0y : p1y, ey1 = p0y < p1y ? p1y : p0y;
        const long long ex = p0y < p1y ? x0f : x1f;
        if (ey0 <= y && y < ey1) {
          const long long x = ex + (long long)(y - ey0) * dx;
          int k = n++;
          while (k > 0 && xs[k - 1] > x) { xs[k] = xs[k - 1]; --k; }     // insertion sort by x
          xs[k] = x;
        }
      }
      p0x = p1x;
      p0y = p1y;
    }
    n_cross = n_edges < 2 ? 0 : n;       // FillEdgeCollection returns early with fewer than two edges
  }
  __syncthreads();
  unsigned char* row = mask + g[G_MASK_OFF] + (size_t)y * w;
  for (int k = 0; k + 1 < n_cross; k += 2) {
    long long x1 = (xs[k] + FILL_LEFT_DELTA) >> XY_SHIFT, x2 = xs[k + 1] >> XY_SHIFT;
    if (x1 < w && x2 >= 0) {
      if (x1 < 0) x1 = 0;
      if (x2 >= w) x2 = w - 1;
      for (long long x = x1 + threadIdx.x; x <= x2; x += 64) row[x] = 255;
    }
  }
}

// ---------------------------------------------------------------- fillPoly: the polygon's edges (cv::Line)
__device__ bool clip_line(long long w, long long h, long long& x1, long long& y1, long long& x2, long long& y2) {
  const long long right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  return (c1 | c2) == 0;
}

__global__ __launch_bounds__(64) void frame_polylines_kernel(const int* __restrict__ geom, const int* __restrict__ pts,
                                                             unsigned char* __restrict__ mask) {
  const int b = blockIdx.x, i = threadIdx.x;     // one lane per polygon edge (pts[i-1] -> pts[i])
  const int* g = geom + b * GEOM;
  const int h = g[G_H], w = g[G_W];
  if (!g[G_VALID] || i >= NPTS) return;
  const int* pt = pts + b * NPTS * 2;
  const int j = i == 0 ? NPTS - 1 : i - 1;
  long long x1 = pt[j * 2], y1 = pt[j * 2 + 1], x2 = pt[i * 2], y2 = pt[i * 2 + 1];
  if (!(x1 >= 0 && x1 < w && x2 >= 0 && x2 < w && y1 >= 0 && y1 < h && y2 >= 0 && y2 < h))
    if (!clip_line(w, h, x1, y1, x2, y2)) return;
  // LineIterator, 8-connected, left to right (drawing.cpp)
  long long dx = x2 - x1, dy = y2 - y1;
  int delta_x = 1, delta_y = 1;
  if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
  if (dy < 0) { dy = -dy; delta_y = -1; }
  const bool vert = dy > dx;
  if (vert) {
    const long long t = dx; dx = dy; dy = t;
    const int d = delta_x; delta_x = delta_y; delta_y = d;
  }
  long long err = dx - (dy + dy);
  const long long plus_delta = dx + dx, minus_delta = -(dy + dy);
  int minus_shift = delta_x, plus_shift = 0, minus_step = 0, plus_step = delta_y;   // x += shift, y += step
  if (vert) {
    int t = plus_step; plus_step = plus_shift; plus_shift = t;
    t = minus_step; minus_step = minus_shift; minus_shift = t;
  }
  unsigned char* m = mask + g[G_MASK_OFF];
  long long x = x1, y = y1;
  for (long long k = 0; k <= dx; ++k) {
    m[(size_t)y * w + x] = 255;
    const bool neg = err < 0;
    err += minus_delta + (neg ? plus_delta : 0);
    y += minus_step + (neg ? plus_step : 0);
    x += minus_shift + (neg ? plus_shift : 0);
  }
}

// ---------------------------------------------------------------- mask area (np.sum(face_mask > 0))
__global__ __launch_bounds__(256) void frame_area_kernel(const int* __restrict__ geom, const unsigned char* __restrict__ mask,
                                                         int* __restrict__ area) {
  const int b = blockIdx.y;
  const int* g = geom + b * GEOM;
  if (!g[G_VALID]) return;
  const long long n = (long long)g[G_H] * g[G_W];
  const unsigned char* m = mask + g[G_MASK_OFF];
  int cnt = 0;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) cnt += m[p] > 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(area + b, cnt);
}

__device__ __forceinline__ int expand_pixels(int area) {   // infer_api.py:294-298, float64 like numpy
  const double radius = __dsqrt_rn((double)area / 3.141592653589793);
  const int e = (int)(radius * 0.15);
  return e < 1 ? 1 : e;
}

// ---------------------------------------------------------------- dilate (separable window maximum)
template <bool ROWS>
__global__ __launch_bounds__(256) void frame_dilate_kernel(const int* __restrict__ geom, const int* __restrict__ area,
                                                           const unsigned char* __restrict__ in,
                                                           unsigned char* __restrict__ out) {
  const int b = blockIdx.y;
  const int* g = geom + b * GEOM;
  if (!g[G_VALID]) return;
  const int h = g[G_H], w = g[G_W];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)h * w) return;
  const int e = expand_pixels(area[b]);
  const int y = (int)(p / w), x = (int)(p - (long long)y * w);
  const unsigned char* src = in + g[G_MASK_OFF];
  int v = 0;
  if (ROWS) {
    const int lo = x - e < 0 ? 0 : x - e, hi = x + e > w - 1 ? w - 1 : x + e;     // the border does not count
    for (int k = lo; k <= hi && v < 255; ++k) v = max(v, (int)src[(size_t)y * w + k]);
  } else {
    const int lo = y - e < 0 ? 0 : y - e, hi = y + e > h - 1 ? h - 1 : y + e;
    for (int k = lo; k <= hi && v < 255; ++k) v = max(v, (int)src[(size_t)k * w + x]);
  }
  out[g[G_MASK_OFF] + p] = (unsigned char)v;
}

// ---------------------------------------------------------------- blend (infer_api.py:314-345)
__global__ __launch_bounds__(256) void frame_blend_kernel(const unsigned char* __restrict__ regions,
                                                          const int* __restrict__ geom,
                                                          const unsigned char* __restrict__ synth,
                                                          const unsigned char* __restrict__ mask,
                                                          unsigned char* __restrict__ out) {
  const int b = blockIdx.y;
  const int* g = geom + b * GEOM;
  const int h = g[G_H], w = g[G_W];
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)h * w) return;
  const unsigned char* img = regions + g[G_REG_OFF] + p * 3;
  unsigned char* dst = out + g[G_REG_OFF] + p * 3;
  if (!g[G_VALID]) {      // shapes differ: the reference returns the original frame (:320-324)
    dst[0] = img[0]; dst[1] = img[1]; dst[2] = img[2];
    return;
  }
  double comb = (double)mask[g[G_MASK_OFF] + p] / 255.0;            // final_face_mask / 255.0
  if (g[G_FMASK_KIND] != FMASK_NONE) {
    // resized_mask = cv2.resize(mask, (w, h)) on float32 (float tables, S0*b0 + S1*b1).  The mask is the frame's
    // own allocation (absolute address in the geometry record); a uint8 mask stands for value / 255 in float32,
    // exactly what infer_api.py:68-70 computes on the host (imread(...).astype(np.float32) / 255.0).
    const int mh = g[G_FMASK_H], mw = g[G_FMASK_W];
    const unsigned long long addr = (unsigned long long)(unsigned)g[G_FMASK_LO] | ((unsigned long long)(unsigned)g[G_FMASK_HI] << 32);
    const bool u8 = g[G_FMASK_KIND] == FMASK_U8;
    const float* fmf = reinterpret_cast<const float*>(addr);
    const unsigned char* fmu = reinterpret_cast<const unsigned char*>(addr);
    auto fm = [&](size_t i) { return u8 ? __fdiv_rn((float)fmu[i], 255.0f) : fmf[i]; };
    const int y = (int)(p / w), x = (int)(p - (long long)y * w);
    float rm;
    if (mh == h && mw == w) {
      rm = fm((size_t)y * mw + x);
    } else if (mw == 2 * w && mh == 2 * h) {
      rm = (fm((size_t)(2 * y) * mw + 2 * x) + fm((size_t)(2 * y) * mw + 2 * x + 1) + fm((size_t)(2 * y + 1) * mw + 2 * x) +
            fm((size_t)(2 * y + 1) * mw + 2 * x + 1)) * 0.25f;
    } else {
      const Tap tx = linear_tap_x(x, mw, w), ty = linear_tap_y(y, mh, h);
      const float a0 = 1.f - tx.f, a1 = tx.f, b0 = 1.f - ty.f, b1 = ty.f;
      const float h0 = fm((size_t)ty.s0 * mw + tx.s0) * a0 + fm((size_t)ty.s0 * mw + tx.s1) * a1;
      const float h1 = fm((size_t)ty.s1 * mw + tx.s0) * a0 + fm((size_t)ty.s1 * mw + tx.s1) * a1;
      rm = h0 * b0 + h1 * b1;
    }
    const float inverted = 1.0f - rm;                      // 1.0 - resized_mask_3ch        (float32)
    comb = comb * (double)(1.0f - inverted);               // final * (1.0 - inverted_mask)  (float64)
  }
  const unsigned char* syn = synth + g[G_SYNTH_OFF] + p * 3;
  const double keep = 1.0 - comb;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double r = (double)syn[c] * comb + (double)img[c] * keep;
    dst[c] = (unsigned char)r;                             // float64 -> uint8 store: truncation
  }
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int casync_frame_prepare(const uint8_t* regions_dev, const int32_t* geom_dev, int batch, uint8_t* crops168_dev,
                         float* x_dev, casync_stream stream) {
  CASYNC_REQUIRE(regions_dev && geom_dev && crops168_dev && batch > 0 && batch <= 65535, "frame_prepare: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(frame_resize168_kernel, dim3(blocks_for(168 * 168), batch), dim3(256), 0, s, regions_dev, geom_dev,
                     crops168_dev);
  CASYNC_CHECK_HIP(hipGetLastError());
  if (x_dev) return launch_crop_to_input(crops168_dev, x_dev, batch, s);
  return CASYNC_OK;
}

int casync_frame_paste_back(const uint8_t* regions_dev, const int32_t* geom_dev, const int32_t* pts_dev,
                            const uint8_t* crops168_dev, const float* pred_dev, int batch,
                            int max_h, int max_w, int max_width, int64_t mask_bytes, uint8_t* synth_dev,
                            uint8_t* mask_a_dev, uint8_t* mask_b_dev, int32_t* area_dev, uint8_t* out_regions_dev,
                            casync_stream stream) {
  CASYNC_REQUIRE(regions_dev && geom_dev && pts_dev && crops168_dev && pred_dev && synth_dev && mask_a_dev && mask_b_dev &&
                     area_dev && out_regions_dev,
                 "frame_paste_back: null pointer");
  CASYNC_REQUIRE(batch > 0 && batch <= 65535 && max_h > 0 && max_w > 0 && max_h <= 65535 && mask_bytes > 0,
                 "frame_paste_back: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const long long max_px = (long long)max_h * max_w;
  CASYNC_CHECK_HIP(hipMemsetAsync(mask_a_dev, 0, (size_t)mask_bytes, s));
  CASYNC_CHECK_HIP(hipMemsetAsync(area_dev, 0, (size_t)batch * sizeof(int32_t), s));
  if (max_width > 0)
    hipLaunchKernelGGL(frame_synth_kernel, dim3(blocks_for((long long)max_width * max_width), batch), dim3(256), 0, s,
                       crops168_dev, pred_dev, geom_dev, synth_dev);
  hipLaunchKernelGGL(frame_polyfill_kernel, dim3(max_h, batch), dim3(64), 0, s, geom_dev, pts_dev, mask_a_dev);
  hipLaunchKernelGGL(frame_polylines_kernel, dim3(batch), dim3(64), 0, s, geom_dev, pts_dev, mask_a_dev);
  hipLaunchKernelGGL(frame_area_kernel, dim3(64, batch), dim3(256), 0, s, geom_dev, mask_a_dev, area_dev);
  hipLaunchKernelGGL(frame_dilate_kernel<true>, dim3(blocks_for(max_px), batch), dim3(256), 0, s, geom_dev, area_dev,
                     mask_a_dev, mask_b_dev);
  hipLaunchKernelGGL(frame_dilate_kernel<false>, dim3(blocks_for(max_px), batch), dim3(256), 0, s, geom_dev, area_dev,
                     mask_b_dev, mask_a_dev);
  hipLaunchKernelGGL(frame_blend_kernel, dim3(blocks_for(max_px), batch), dim3(256), 0, s, regions_dev, geom_dev, synth_dev,
                     mask_a_dev, out_regions_dev);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

}  // extern "C"
