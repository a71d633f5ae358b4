// HBM-bound operators of the CASync U-Net on NHWC fp32 activations (gfx950).
//
//  dw3x3        depthwise 3x3 (+folded BN bias, LeakyReLU)      module/unet.py:21-30
//  im2col3x3    patch gather for the two dense stride-2 convs   module/unet.py:161-168
//  upsample2x   bilinear x2, align_corners=True, into a concat  module/unet.py:86-96
//  nchw_to_nhwc boundary layout change for the HuBERT window
//  inc          whole `inc` inverted-residual from the NCHW crop module/unet.py:58-67
//  outc         OutConv + outc_bn + sigmoid -> NCHW             module/unet.py:100-106,342-344
//
// All of them move 16 B per lane with the channel axis innermost (coalesced 128-B+ runs).
#include "common.h"

namespace {

// ---------------------------------------------------------------- depthwise 3x3
// One thread = 4 channels x PX consecutive output pixels of one row; the (PX-1)*S+3 input
// columns it needs are held in registers and slid over, so each input float4 is loaded once
// per row of taps instead of 3 times.
template <typename T, int STRIDE, int PX>
__global__ __launch_bounds__(256) void dw3x3_kernel(const T* __restrict__ in,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    T* __restrict__ out, int H, int W, int C,
                                                    int Ho, int Wo, int strips, long long total) {
  constexpr int V = V16<T>::N;   // channels per thread = one 16-B access (4 fp32 / 8 bf16)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cvn = C / V;
  const int c = (int)(idx % cvn) * V;
  long long t = idx / cvn;
  const int sx = (int)(t % strips);
  t /= strips;
  const int oy = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int ox0 = sx * PX;
  constexpr int NC = (PX - 1) * STRIDE + 3;
  const int iy0 = oy * STRIDE - 1, ix0 = ox0 * STRIDE - 1;

  float acc[PX][V];
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float bv = bias[c + e];
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[p][e] = bv;
  }

#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = iy0 + ky;
    if (iy < 0 || iy >= H) continue;  // zero padding
    float wt[3][V];                   // this row of taps only (keeps the register footprint small)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < V; e += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * C + c + e);
        wt[kx][e] = x[0]; wt[kx][e + 1] = x[1]; wt[kx][e + 2] = x[2]; wt[kx][e + 3] = x[3];
      }
    const T* row = in + ((size_t)b * H + iy) * (size_t)W * C + c;
    V16<T> v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int ix = ix0 + j;
      if (ix >= 0 && ix < W) {
        v[j] = ld16(row + (size_t)ix * C);
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) v[j].v[e] = 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[p][e] += v[p * STRIDE + kx].v[e] * wt[kx][e];
  }
  T* orow = out + ((size_t)b * Ho + oy) * (size_t)Wo * C + c;
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    if (ox0 + p < Wo) {
      V16<T> r;
#pragma unroll
      for (int e = 0; e < V; ++e) r.v[e] = lrelu(acc[p][e]);
      st16(orow + (size_t)(ox0 + p) * C, r);
    }
  }
}

// Low-resolution variant (W <= 48, stride 1): the kernel above fetches every input ~4.5x through
// L1 (3 tap rows x 1.5 column overlap), and at 10x10 .. 40x40 the vertical neighbours sit in other
// workgroups, so it is L2-bandwidth-bound (~3 TB/s algorithmic).  Here a workgroup stages a
// (TH+2) x (W+2) x CSV*16-B slab (zero borders included) in LDS with one coalesced pass -- every
// input byte crosses L2 once (x (TH+2)/TH when the rows are tiled) -- and all nine taps read LDS.
// One thread keeps one 16-B channel group (its tap weights stay in registers) and walks pixels.
template <typename T, int CSV>
__global__ __launch_bounds__(256) void dw3x3_lds_kernel(const T* __restrict__ in,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        T* __restrict__ out, int H, int W, int C, int TH) {
  constexpr int V = V16<T>::N;
  extern __shared__ __attribute__((aligned(16))) f32x4 dw_tile[];   // [(th+2)][W+2][CSV] raw 16-B chunks
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * CSV * V, y0 = blockIdx.y * TH, b = blockIdx.z;
  const int th = TH < H - y0 ? TH : H - y0;
  const int WP = W + 2;
  const T* inb = in + (size_t)b * H * W * C + c0;
  const int n_in = (th + 2) * WP * CSV;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // the taps are requested BEFORE the slab (they arrive under its loads: at small batch the launch is one latency chain)
  // 256 % CSV == 0: a thread's channel group never changes
  const int cv = tid % CSV, c = c0 + cv * V;
  float wt[9][V], bv[V];
#pragma unroll
  for (int e = 0; e < V; e += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(bias + c + e);
    bv[e] = x[0]; bv[e + 1] = x[1]; bv[e + 2] = x[2]; bv[e + 3] = x[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(w + k * C + c + e);
      wt[k][e] = y[0]; wt[k][e + 1] = y[1]; wt[k][e + 2] = y[2]; wt[k][e + 3] = y[3];
    }
  }
  for (int i = tid; i < n_in; i += 256) {
    const int cv = i % CSV, t = i / CSV;
    const int r = t / WP, xc = t - r * WP;
    const int iy = y0 - 1 + r, ix = xc - 1;
    f32x4 v = zero;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W)
      v = *reinterpret_cast<const f32x4*>(inb + ((size_t)iy * W + ix) * C + cv * V);
    dw_tile[i] = v;
  }
  __syncthreads();
  T* outb = out + ((size_t)b * H + y0) * W * C + c;
  const int n_out = th * W;
  for (int p = tid / CSV; p < n_out; p += 256 / CSV) {
    const int oy = p / W, ox = p - oy * W;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = bv[e];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const V16<T> x = ld16(reinterpret_cast<const T*>(dw_tile + ((oy + ky) * WP + ox + kx) * CSV + cv));
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += x.v[e] * wt[ky * 3 + kx][e];
      }
    V16<T> r;
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = lrelu(acc[e]);
    st16(outb + (size_t)p * C, r);
  }
}


// The same slab kernel behind an Up block's expand conv whose two halves were computed apart (engine.hip,
// skip_early): the slab value is LReLU(pre + up(g)) -- `pre` = W1b . skip + b at the full resolution, `g` = W1a . lo at
// half the resolution, bilinear x2 with align_corners=True (ups_tap / ups_lerp: the bits of every other consumer) --
// and zero outside the frame (the depthwise conv pads the EXPANDED tensor).  fp32 only.
template <int CSV>
__global__ __launch_bounds__(256) void dw3x3_ups_lds_kernel(const float* __restrict__ pre, const float* __restrict__ g, int ldg,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int H, int W, int C, int TH) {
  extern __shared__ __attribute__((aligned(16))) f32x4 dw_tile[];   // [(th+2)][W+2][CSV] 16-B chunks
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * CSV * 4, y0 = blockIdx.y * TH, b = blockIdx.z;
  const int th = TH < H - y0 ? TH : H - y0;
  const int WP = W + 2, Hl = H >> 1, Wl = W >> 1;
  const float* preb = pre + (size_t)b * H * W * C + c0;
  const float* gb = g + (size_t)b * Hl * Wl * ldg + c0;
  const float sy = ups_scale(H), sx = ups_scale(W);
  const int n_in = (th + 2) * WP * CSV;
  const int cv = tid % CSV, c = c0 + cv * 4;
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const f32x4*>(w + k * C + c);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
  for (int i = tid; i < n_in; i += 256) {
    const int cvi = i % CSV, t = i / CSV;
    const int r = t / WP, xc = t - r * WP;
    const int iy = y0 - 1 + r, ix = xc - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const UpsTap ty = ups_tap(sy, iy, Hl), tx = ups_tap(sx, ix, Wl);
      const float* gq = gb + cvi * 4;
      const f32x4 up = ups_lerp(ty, tx, *reinterpret_cast<const f32x4*>(gq + ((size_t)ty.i0 * Wl + tx.i0) * ldg),
                                *reinterpret_cast<const f32x4*>(gq + ((size_t)ty.i0 * Wl + tx.i1) * ldg),
                                *reinterpret_cast<const f32x4*>(gq + ((size_t)ty.i1 * Wl + tx.i0) * ldg),
                                *reinterpret_cast<const f32x4*>(gq + ((size_t)ty.i1 * Wl + tx.i1) * ldg));
      const f32x4 e = *reinterpret_cast<const f32x4*>(preb + ((size_t)iy * W + ix) * C + cvi * 4) + up;
      v = f32x4{lrelu(e[0]), lrelu(e[1]), lrelu(e[2]), lrelu(e[3])};
    }
    dw_tile[i] = v;
  }
  __syncthreads();
  float* outb = out + ((size_t)b * H + y0) * W * C + c;
  const int n_out = th * W;
  for (int p = tid / CSV; p < n_out; p += 256 / CSV) {
    const int oy = p / W, ox = p - oy * W;
    f32x4 acc = bv;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc += dw_tile[((oy + ky) * WP + ox + kx) * CSV + cv] * wt[ky * 3 + kx];
    *reinterpret_cast<f32x4*>(outb + (size_t)p * C) = f32x4{lrelu(acc[0]), lrelu(acc[1]), lrelu(acc[2]), lrelu(acc[3])};
  }
}

// ---------------------------------------------------------------- bilinear x2
// align_corners=True: src = dst * (in-1)/(out-1); weights as ATen computes them
// (lambda1 = src - floor(src), lambda0 = 1 - lambda1).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ in,
                                                         T* __restrict__ out, int ldc, int H,
                                                         int W, int C, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4n = C >> 2;
  const int c = (int)(idx % c4n) * 4;
  long long t = idx / c4n;
  const int Ho = 2 * H, Wo = 2 * W;
  const int ox = (int)(t % Wo);
  t /= Wo;
  const int oy = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const float sy = ups_scale(Ho), sx = ups_scale(Wo);
  const UpsTap ty = ups_tap(sy, oy, H), tx = ups_tap(sx, ox, W);
  const T* base = in + (size_t)b * H * W * C + c;
  const f32x4 v00 = ld4(base + ((size_t)ty.i0 * W + tx.i0) * C);
  const f32x4 v01 = ld4(base + ((size_t)ty.i0 * W + tx.i1) * C);
  const f32x4 v10 = ld4(base + ((size_t)ty.i1 * W + tx.i0) * C);
  const f32x4 v11 = ld4(base + ((size_t)ty.i1 * W + tx.i1) * C);
  const f32x4 r = ups_lerp(ty, tx, v00, v01, v10, v11);
  st4(out + (((size_t)b * Ho + oy) * Wo + ox) * ldc + c, r);
}

// ---------------------------------------------------------------- NCHW -> NHWC
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                           T* __restrict__ out, int C, int HW,
                                                           long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  long long t = idx / HW;
  const int c4n = C >> 2;
  const int c = (int)(t % c4n) * 4;
  const int b = (int)(t / c4n);
  const float* src = in + ((size_t)b * C + c) * HW + p;
  f32x4 v = {src[0], src[HW], src[2 * (size_t)HW], src[3 * (size_t)HW]};
  st4(out + ((size_t)b * HW + p) * C + c, v);
}

// ---------------------------------------------------------------- HuBERT window gather
// Reference: FrameSynthesizer._get_audio_features (infer_api.py:99-145): for video frame index i
// the window is features[i-8 : i+8] of the [T, 2, 1024] HuBERT array, zero-padded past either
// end, flattened and reshaped to (32, 32, 32) = [c][y][x].  Flat position f = c*1024 + y*32 + x
// lies in window row f/2048, half (f%2048)/1024, so channel c = 2*row + half and pixel
// p = f%1024:  audio[b][c][p] = window[c/2][c%2][p].  This writes the engine's NHWC audio input
// [B][1024][32] directly, so neither the B x 128 KB host windows nor the NCHW->NHWC pass exist.
//
// The reference's corner cases are reproduced exactly (infer_api.py:116-135):
//   left = i-8, right = i+8; pad_left = max(0,-left), left = max(left,0); pad_right = max(0,right-T),
//   right = min(right,T); auds = features[left:right]  (Python slice: a negative `right`, possible only
//   for i < -8, counts from the end); the pads are zeros_like(auds[:pad]), i.e. TRUNCATED to the rows
//   auds has at that point; the window is used only if it then has exactly 16 rows (numel >= 32768 and
//   the reshape succeeds), otherwise the frame gets the all-zero default window (:106,141-142).
// So row r of a valid window is: zero for r < pl, features[start + r - pl] for r < pl + n0, zero after.
struct AudioWindow { int start, n0, pl, ok; };
__host__ __device__ inline AudioWindow audio_window_plan(int idx, int T) {
  const int left0 = idx - 8, right0 = idx + 8;
  const int pad_left = left0 < 0 ? -left0 : 0, pad_right = right0 > T ? right0 - T : 0;
  const int left = left0 < 0 ? 0 : left0, right = right0 > T ? T : right0;
  const int start = left < T ? left : T;                             // slice start (left >= 0)
  int stop = right >= 0 ? right : (T + right > 0 ? T + right : 0);   // Python semantics of a negative stop
  stop = stop < T ? stop : T;
  const int n0 = stop > start ? stop - start : 0;
  const int pl = pad_left < n0 ? pad_left : n0;                      // zeros_like(auds[:pad_left])
  const int n1 = n0 + pl;
  const int pr = pad_right < n1 ? pad_right : n1;                    // zeros_like(auds[:pad_right])
  return AudioWindow{start, n0, pl, n1 + pr == 16};
}

// NCHW = false: the engine's NHWC image [B][1024 pixels][32 channels] in its storage type (thread = 4 channels of a pixel);
// NCHW = true : the reference's own array [B][32][32][32] fp32 (thread = the same 4 channels, stored to 4 planes).
template <typename T, bool NCHW = false>
__global__ __launch_bounds__(256) void audio_window_gather_kernel(const float* __restrict__ feat, int n_steps,
                                                                  const int* __restrict__ idx,
                                                                  T* __restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // thread = (b, p, 4 channels)
  if (i >= total) return;
  const int p = (int)(i % 1024);
  long long t = i / 1024;
  const int c = (int)(t % 8) * 4;
  const int b = (int)(t / 8);
  const AudioWindow w = audio_window_plan(idx[b], n_steps);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (w.ok) {
    const int r0 = c / 2 - w.pl, r1 = r0 + 1;   // c is a multiple of 4: window rows c/2 (c, c+1) and c/2+1 (c+2, c+3)
    if (r0 >= 0 && r0 < w.n0) {
      v.x = feat[(size_t)(w.start + r0) * 2048 + p];
      v.y = feat[(size_t)(w.start + r0) * 2048 + 1024 + p];
    }
    if (r1 >= 0 && r1 < w.n0) {
      v.z = feat[(size_t)(w.start + r1) * 2048 + p];
      v.w = feat[(size_t)(w.start + r1) * 2048 + 1024 + p];
    }
  }
  if constexpr (NCHW) {
    T* o = out + ((size_t)b * 32 + c) * 1024 + p;
    o[0] = (T)v.x;
    o[1024] = (T)v.y;
    o[2048] = (T)v.z;
    o[3072] = (T)v.w;
  } else {
    st4(out + ((size_t)b * 1024 + p) * 32 + c, v);
  }
}

// ---------------------------------------------------------------- frame-loop tensor glue
// FrameSynthesizer.process_batch around the model call (infer_api.py:238-245 and 265-266):
//   pre : crop168 (uint8 HWC BGR) -> inner [4:164,4:164]; masked copy = same with the filled black
//         rectangle (x,y,w,h) = (5,5,150,145), i.e. x in [5,154], y in [5,149]; both HWC->CHW, /255
//         (fp32 division), concat -> x[6,160,160]
//   post: pred[3,160,160] * 255 -> uint8 (truncation) -> HWC
// Pure indexing + one IEEE op each, so the kernels are bit-exact against the numpy restatement.
__global__ __launch_bounds__(256) void crop_to_input_kernel(const unsigned char* __restrict__ crops,
                                                            float* __restrict__ x, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one output pixel (all 6 channels)
  if (i >= total) return;
  const int px = (int)(i % 160), py = (int)((i / 160) % 160);
  const long long b = i / 25600;
  const unsigned char* src = crops + ((b * 168 + py + 4) * 168 + px + 4) * 3;
  const bool masked = px >= 5 && px < 155 && py >= 5 && py < 150;
  float* dst = x + b * 6 * 25600 + py * 160 + px;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)src[c] / 255.0f;
    dst[c * 25600] = v;
    dst[(3 + c) * 25600] = masked ? 0.f : v;
  }
}

__global__ __launch_bounds__(256) void pred_to_u8_kernel(const float* __restrict__ pred,
                                                         unsigned char* __restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one pixel
  if (i >= total) return;
  const long long b = i / 25600, p = i - b * 25600;
  const float* src = pred + b * 3 * 25600 + p;
  unsigned char* dst = out + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c] = (unsigned char)(src[c * 25600] * 255.0f);
}

// ---------------------------------------------------------------- inc (6 -> 12 -> dw -> 32)
// Block = 8 rows x 32 columns of one frame.  Phase A: 1x1 expand (+bias, LReLU) of the
// 10 x 34 halo into LDS, zero where the halo leaves the image (the depthwise conv pads the
// EXPANDED tensor with zeros).  Phase B: depthwise 3x3 from LDS, LReLU, 1x1 project,
// LReLU.  Phase C: coalesced NHWC write through LDS.
// packed = [w1T 6x12][b1 12][wd 9x12][bd 12][w2T 12x32][b2 32]: both 1x1 weights are stored INPUT-channel major, so the
// weights of two adjacent output channels are one aligned 8-byte pair.
// Round 5: the kernel is bound by vector-instruction issue (profiles/r4_mfma_busy_bf16_b512.json: 108 M instructions per
// 256-frame launch = 0.73 of its cycles; 2.2 TB/s), so all three stages work on channel PAIRS with packed fp32 FMAs
// (v_pk_fma_f32: the weight pair is a scalar register pair, the activation is broadcast by op_sel) -- 590 FMAs per pixel
// become ~300 instructions -- and E is stored as pairs (8-byte LDS accesses).  The tiles of a frame are handed to ONE
// XCD in order (xcd_run below): x-adjacent tiles share the 128-B lines their halo columns straddle, and with the default
// round-robin dispatch every one of them fetched those lines into a different L2 (fetch 3.2 x the input bytes).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int INC_TW = 32, INC_TH = 8, INC_HW = 160;
constexpr int INC_HALO_W = INC_TW + 2, INC_HALO_H = INC_TH + 2, INC_HALO = INC_HALO_W * INC_HALO_H;
constexpr int INC_TX = INC_HW / INC_TW, INC_TY = INC_HW / INC_TH, INC_TILES = INC_TX * INC_TY;

__device__ __forceinline__ f32x2 lrelu2(f32x2 v) {
  const f32x2 s = v * CASYNC_LRELU_SLOPE;
  return f32x2{fmaxf(v.x, s.x), fmaxf(v.y, s.y)};
}
// workgroups b, b + 8, ... share an XCD: give each XCD a contiguous run of the launch's tiles
__device__ __forceinline__ int xcd_run(int t, int nwg) {
  const int qn = nwg >> 3, r = nwg & 7, xcd = t & 7, idx = t >> 3;
  return (xcd < r ? xcd * (qn + 1) : r * (qn + 1) + (xcd - r) * qn) + idx;
}

template <typename T>
__global__ __launch_bounds__(256) void inc_kernel(const float* __restrict__ x,
                                                  const float* __restrict__ packed,
                                                  T* __restrict__ out, int ldc, int nwg) {
  __shared__ f32x2 E[INC_CEXP / 2][INC_HALO + 2];
  __shared__ float O[256][INC_COUT + 1];
  // the 620 weights are read with uniform addresses straight from memory: scalar loads into SGPR pairs that the packed
  // FMAs take as operands
  const int tid = threadIdx.x;
  const f32x2* w1 = reinterpret_cast<const f32x2*>(packed);          // [6 ci][6 pairs of ce]
  const f32x2* b1 = reinterpret_cast<const f32x2*>(packed + 72);
  const f32x2* wd = reinterpret_cast<const f32x2*>(packed + 84);     // [9 taps][6 pairs]
  const f32x2* bd = reinterpret_cast<const f32x2*>(packed + 192);
  const f32x2* w2 = reinterpret_cast<const f32x2*>(packed + 204);    // [12 ce][16 pairs of co]
  const f32x2* b2 = reinterpret_cast<const f32x2*>(packed + 588);
  const int bid = xcd_run(blockIdx.x, nwg);
  const int b = bid / INC_TILES, t = bid - b * INC_TILES;
  const int ty0 = (t / INC_TX) * INC_TH, tx0 = (t % INC_TX) * INC_TW;
  const float* xb = x + (size_t)b * INC_CIN * INC_HW * INC_HW;
  for (int p = tid; p < INC_HALO; p += 256) {
    const int hy = p / INC_HALO_W, hx = p - hy * INC_HALO_W;
    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
    const bool inside = iy >= 0 && iy < INC_HW && ix >= 0 && ix < INC_HW;
    float v[INC_CIN];
#pragma unroll
    for (int ci = 0; ci < INC_CIN; ++ci)
      v[ci] = inside ? xb[((size_t)ci * INC_HW + iy) * INC_HW + ix] : 0.f;
#pragma unroll
    for (int cp = 0; cp < INC_CEXP / 2; ++cp) {
      f32x2 s2 = b1[cp];
#pragma unroll
      for (int ci = 0; ci < INC_CIN; ++ci) s2 = __builtin_elementwise_fma(w1[ci * (INC_CEXP / 2) + cp], f32x2{v[ci], v[ci]}, s2);
      E[cp][p] = inside ? lrelu2(s2) : f32x2{0.f, 0.f};
    }
  }
  __syncthreads();
  const int ly = tid >> 5, lx = tid & 31;
  f32x2 d[INC_CEXP / 2];
#pragma unroll
  for (int cp = 0; cp < INC_CEXP / 2; ++cp) {
    f32x2 s2 = bd[cp];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        s2 = __builtin_elementwise_fma(wd[(ky * 3 + kx) * (INC_CEXP / 2) + cp], E[cp][(ly + ky) * INC_HALO_W + lx + kx], s2);
    d[cp] = lrelu2(s2);
  }
#pragma unroll
  for (int op = 0; op < INC_COUT / 2; ++op) {
    f32x2 s2 = b2[op];
#pragma unroll
    for (int cp = 0; cp < INC_CEXP / 2; ++cp) {
      s2 = __builtin_elementwise_fma(w2[(2 * cp) * (INC_COUT / 2) + op], f32x2{d[cp].x, d[cp].x}, s2);
      s2 = __builtin_elementwise_fma(w2[(2 * cp + 1) * (INC_COUT / 2) + op], f32x2{d[cp].y, d[cp].y}, s2);
    }
    s2 = lrelu2(s2);
    O[tid][2 * op] = s2.x;
    O[tid][2 * op + 1] = s2.y;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = j * 256 + tid;
    const int pix = i >> 3, c4 = (i & 7) * 4;
    const int py = pix >> 5, px = pix & 31;
    f32x4 v = {O[pix][c4], O[pix][c4 + 1], O[pix][c4 + 2], O[pix][c4 + 3]};
    st4(out + (((size_t)b * INC_HW + ty0 + py) * INC_HW + tx0 + px) * ldc + c4, v);
  }
}

// inc for the bf16 engine (round 6): the 12 -> 32 projection is two thirds of the block's multiply-adds (384 of 564 per pixel)
// and dense, so it runs on v_mfma_f32_16x16x32_bf16 (K = 12 of 32) instead of 192 packed fp32 FMAs per pixel.  Phases A / B
// (expand over the halo, depthwise 3x3) are inc_kernel's, in fp32.  Then a thread's twelve depthwise outputs go to LDS as
// bf16 (64 B per pixel: high and low halves of 12 channels, see below) -- read back by the same wave as the MFMA B operand (lane (q, n) = K-slots
// 8 q .. 8 q + 7 of pixel n: one 16-byte read), no workgroup barrier: LDS operations of a wave are in
// order -- W2's rows enter as the A operand permuted (row r of tile t = output channel 8 (r >> 2) + 4 t + (r & 3)) so that a
// lane's rows 4 q .. 4 q + 3 of both tiles are the eight consecutive channels 8 q .. 8 q + 7 of its pixel: bias is the
// accumulators' initial value, LeakyReLU + narrowing run on 8 values per lane and 16 pixels, and every pixel's 64-byte NHWC
// row leaves as four 16-byte stores of one instruction.  No staging of the output through LDS (inc_kernel: 34 KB), so six
// instead of three workgroups fit a CU.  W2 is rounded to bf16 before the product (fp32 accumulate), as in every other GEMM of
// the bf16 engine; the depthwise output enters as a bf16 high + low pair.
__global__ __launch_bounds__(256) void inc_bf16_kernel(const float* __restrict__ x, const float* __restrict__ packed,
                                                       bf16_t* __restrict__ out, int ldc, int nwg) {
  __shared__ f32x2 E[INC_CEXP / 2][INC_HALO + 2];
  __shared__ __attribute__((aligned(16))) bf16_t Dl[256][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const f32x2* w1 = reinterpret_cast<const f32x2*>(packed);          // [6 ci][6 pairs of ce]
  const f32x2* b1 = reinterpret_cast<const f32x2*>(packed + 72);
  const f32x2* wd = reinterpret_cast<const f32x2*>(packed + 84);     // [9 taps][6 pairs]
  const f32x2* bd = reinterpret_cast<const f32x2*>(packed + 192);
  const float* w2 = packed + 204;                                    // [12 ce][32 co]
  const float* b2 = packed + 588;
  // the two weight fragments (A operand) and the bias rows of this lane: once per workgroup, from L2
  bf16x8 fw[2];
  f32x4 bias[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int oc = 8 * (l15 >> 2) + 4 * t + (l15 & 3);
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // (every lane loads from a valid row, the K-slots past channel 11 are zeroed afterwards: no divergent code)
      const int ce = (8 * q + j) & 15;                    // K-slots 16 .. 27 repeat 0 .. 11: they meet the LOW halves of D
      const float v = w2[(ce < INC_CEXP ? ce : INC_CEXP - 1) * INC_COUT + oc];
      w[j] = ce < INC_CEXP ? v : 0.f;
    }
    const bf16x4 lo = __builtin_convertvector(f32x4{w[0], w[1], w[2], w[3]}, bf16x4), hi = __builtin_convertvector(f32x4{w[4], w[5], w[6], w[7]}, bf16x4);
    fw[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    bias[t] = *reinterpret_cast<const f32x4*>(b2 + 8 * q + 4 * t);
  }
  const int bid = xcd_run(blockIdx.x, nwg);
  const int b = bid / INC_TILES, t = bid - b * INC_TILES;
  const int ty0 = (t / INC_TX) * INC_TH, tx0 = (t % INC_TX) * INC_TW;
  const float* xb = x + (size_t)b * INC_CIN * INC_HW * INC_HW;
  for (int p = tid; p < INC_HALO; p += 256) {
    const int hy = p / INC_HALO_W, hx = p - hy * INC_HALO_W;
    const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
    const bool inside = iy >= 0 && iy < INC_HW && ix >= 0 && ix < INC_HW;
    float v[INC_CIN];
#pragma unroll
    for (int ci = 0; ci < INC_CIN; ++ci)
      v[ci] = inside ? xb[((size_t)ci * INC_HW + iy) * INC_HW + ix] : 0.f;
#pragma unroll
    for (int cp = 0; cp < INC_CEXP / 2; ++cp) {
      f32x2 s2 = b1[cp];
#pragma unroll
      for (int ci = 0; ci < INC_CIN; ++ci) s2 = __builtin_elementwise_fma(w1[ci * (INC_CEXP / 2) + cp], f32x2{v[ci], v[ci]}, s2);
      E[cp][p] = inside ? lrelu2(s2) : f32x2{0.f, 0.f};
    }
  }
  __syncthreads();
  const int ly = tid >> 5, lx = tid & 31;
  f32x2 d[INC_CEXP / 2];
#pragma unroll
  for (int cp = 0; cp < INC_CEXP / 2; ++cp) {
    f32x2 s2 = bd[cp];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        s2 = __builtin_elementwise_fma(wd[(ky * 3 + kx) * (INC_CEXP / 2) + cp], E[cp][(ly + ky) * INC_HALO_W + lx + kx], s2);
    d[cp] = lrelu2(s2);
  }
  {
    // D as bf16 HIGH + LOW halves (d = hi + lo to 16 mantissa bits): the K = 32 of the MFMA has room for both (slots 0 .. 11
    // and 16 .. 27 against the same weights), so the depthwise output enters the product almost unrounded at no extra
    // matrix instruction -- only W2 is rounded to bf16 (with hi alone the x1 tap's max error was 6.2e-3 against 2.4e-3)
    const f32x4 da = {d[0].x, d[0].y, d[1].x, d[1].y}, db = {d[2].x, d[2].y, d[3].x, d[3].y}, dc = {d[4].x, d[4].y, d[5].x, d[5].y};
    const bf16x4 h0 = __builtin_convertvector(da, bf16x4), h1 = __builtin_convertvector(db, bf16x4), h2 = __builtin_convertvector(dc, bf16x4);
    const bf16x4 l0 = __builtin_convertvector(da - __builtin_convertvector(h0, f32x4), bf16x4);
    const bf16x4 l1 = __builtin_convertvector(db - __builtin_convertvector(h1, f32x4), bf16x4);
    const bf16x4 l2 = __builtin_convertvector(dc - __builtin_convertvector(h2, f32x4), bf16x4);
    const bf16_t z = (bf16_t)0.f;
    *reinterpret_cast<bf16x8*>(&Dl[tid][0]) = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    *reinterpret_cast<bf16x8*>(&Dl[tid][8]) = bf16x8{h2[0], h2[1], h2[2], h2[3], z, z, z, z};
    *reinterpret_cast<bf16x8*>(&Dl[tid][16]) = bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    *reinterpret_cast<bf16x8*>(&Dl[tid][24]) = bf16x8{l2[0], l2[1], l2[2], l2[3], z, z, z, z};
  }
  // (same wave wrote what it reads: no barrier; the compiler's lgkmcnt wait orders the read behind the writes)
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * INC_HW * INC_HW * ldc, 0,
                                                                           (unsigned)INC_HW * INC_HW * (unsigned)ldc * 2u, 0x00020000);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int p = 64 * wave + 16 * g + l15;                 // this lane's pixel of group g (the tile's pixel index = its thread)
    const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&Dl[p][8 * q]);
    const f32x4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0], fb, bias[0], 0, 0, 0);
    const f32x4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[1], fb, bias[1], 0, 0, 0);
    const f32x4 s0 = a0 * CASYNC_LRELU_SLOPE, s1 = a1 * CASYNC_LRELU_SLOPE;
    const f32x4 v0 = {fmaxf(a0.x, s0.x), fmaxf(a0.y, s0.y), fmaxf(a0.z, s0.z), fmaxf(a0.w, s0.w)};
    const f32x4 v1 = {fmaxf(a1.x, s1.x), fmaxf(a1.y, s1.y), fmaxf(a1.z, s1.z), fmaxf(a1.w, s1.w)};
    const bf16x4 h0 = __builtin_convertvector(v0, bf16x4), h1 = __builtin_convertvector(v1, bf16x4);
    const bf16x8 o = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    const int py = p >> 5, px = p & 31;
    const unsigned off = (unsigned)((((ty0 + py) * INC_HW + tx0 + px) * ldc + 8 * q) * 2);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, (int)off, 0, 0);
  }
}

// ---------------------------------------------------------------- outc (32 -> 3, sigmoid)
template <typename T>
__global__ __launch_bounds__(256) void outc_kernel(const T* __restrict__ in, int ld_in,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ out) {
  __shared__ float Ts[256][33];
  __shared__ float Wl[3 * 32 + 3];
  const int tid = threadIdx.x;
  if (tid < 96) Wl[tid] = w[tid];
  if (tid < 3) Wl[96 + tid] = bias[tid];
  const size_t pix0 = (size_t)blockIdx.x * 256;  // 25600 % 256 == 0: a block never straddles frames
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = j * 256 + tid;
    const int pix = i >> 3, c4 = (i & 7) * 4;
    const f32x4 v = ld4(in + (pix0 + pix) * ld_in + c4);
    Ts[pix][c4] = v.x; Ts[pix][c4 + 1] = v.y; Ts[pix][c4 + 2] = v.z; Ts[pix][c4 + 3] = v.w;
  }
  __syncthreads();
  const size_t pix = pix0 + tid;
  const size_t b = pix / (INC_HW * INC_HW), p = pix - b * (INC_HW * INC_HW);
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    float s = Wl[96 + o];
#pragma unroll
    for (int c = 0; c < 32; ++c) s += Wl[o * 32 + c] * Ts[tid][c];
    out[(b * 3 + o) * (size_t)(INC_HW * INC_HW) + p] = 1.f / (1.f + expf(-s));
  }
}

inline unsigned blocks_for(long long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

#define DT_DISPATCH(dtype, CALL_F32, CALL_BF16) \
  do {                                        \
    if ((dtype) == DT_BF16) { CALL_BF16; } else { CALL_F32; } \
  } while (0)

// LDS-slab depthwise kernel for this shape?  Slab = 16 (narrow frames) or 8 16-B channel groups
// wide, as many rows as keep it <= 64 KB, rows split evenly.
static int lds_budget() { return casync_opts().dw_lds_bytes; }
static bool dw_lds_plan(int h, int wdt, int c, int stride, int dtype, int* csv, int* th, int* nt) {
  const int enabled = casync_opts().dw_lds;
  if (!enabled || stride != 1 || wdt > 48) return false;
  const int groups = c / (16 / dtype_size(dtype));
  *csv = (wdt <= 12 && groups % 16 == 0) ? 16 : 8;   // 64-B slabs (half an L2 line per pixel) measured 25 % slower
  // whole frames when they fit in 40 KB (no halo rows re-read), else <= 32 KB slabs: four or more
  // workgroups per CU overlap one's load phase with another's tap phase (measured sweet spot)
  const int whole = (h + 2) * (wdt + 2) * *csv * 16;
  const int rows_max = (whole <= 40960 ? whole : lds_budget()) / ((wdt + 2) * *csv * 16) - 2;
  if (groups % *csv || rows_max < 4) return false;
  *th = h < rows_max ? h : rows_max;
  *nt = (h + *th - 1) / *th;
  *th = (h + *nt - 1) / *nt;
  return true;
}

// depthwise 3x3 over LReLU(pre + up(g)) (dw3x3_ups_lds_kernel): fp32, stride 1, frames the slab plan takes
const char* dw3x3_ups_kernel_name(int h, int wdt, int c) {
  static thread_local char buf[64];
  int csv = 8, th, nt;
  dw_lds_plan(h, wdt, c, 1, DT_F32, &csv, &th, &nt);
  snprintf(buf, sizeof(buf), "dw3x3_ups_lds_kernel<%d>", csv);
  return buf;
}
int launch_dw3x3_ups(const float* pre, const float* g, int ldg, const float* w, const float* bias, float* out, int batch, int h,
                     int wdt, int c, hipStream_t stream) {
  CASYNC_REQUIRE(pre && g && w && bias && out, "dw3x3_ups: null pointer");
  CASYNC_REQUIRE(batch > 0 && h > 2 && wdt > 2 && h % 2 == 0 && wdt % 2 == 0 && c % 4 == 0 && ldg >= c && ldg % 4 == 0,
                 "dw3x3_ups: bad shape %dx%dx%d (ld %d)", h, wdt, c, ldg);
  int csv, th, nt;
  CASYNC_REQUIRE(dw_lds_plan(h, wdt, c, 1, DT_F32, &csv, &th, &nt), "dw3x3_ups: %dx%dx%d does not fit the slab kernel", h, wdt, c);
  const size_t lds = (size_t)(th + 2) * (wdt + 2) * csv * 16;
  const dim3 grid(c / 4 / csv, nt, batch);
  if (csv == 16) hipLaunchKernelGGL(dw3x3_ups_lds_kernel<16>, grid, dim3(256), lds, stream, pre, g, ldg, w, bias, out, h, wdt, c, th);
  else hipLaunchKernelGGL(dw3x3_ups_lds_kernel<8>, grid, dim3(256), lds, stream, pre, g, ldg, w, bias, out, h, wdt, c, th);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

const char* dw3x3_kernel_name(int h, int wdt, int c, int stride, int dtype) {
  static thread_local char buf[64];
  const char* t = dtype == DT_BF16 ? "__bf16" : "float";
  int csv, th, nt;
  if (dw_lds_plan(h, wdt, c, stride, dtype, &csv, &th, &nt))
    snprintf(buf, sizeof(buf), "dw3x3_lds_kernel<%s, %d>", t, csv);
  else
    snprintf(buf, sizeof(buf), "dw3x3_kernel<%s, %s>", t, stride == 1 ? "1, 4" : "2, 2");
  return buf;
}

int launch_dw3x3(const void* in, const float* w, const float* bias, void* out, int batch, int h,
                 int wdt, int c, int stride, hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(in && w && bias && out, "dw3x3: null pointer");
  const int vec = 16 / dtype_size(dtype);   // channels per thread
  CASYNC_REQUIRE(batch > 0 && h > 0 && wdt > 0 && c > 0 && c % vec == 0, "dw3x3: bad shape (C %% %d)", vec);
  CASYNC_REQUIRE(stride == 1 || stride == 2, "dw3x3: stride %d", stride);
  const int ho = (h + 2 - 3) / stride + 1, wo = (wdt + 2 - 3) / stride + 1;
  {
    int csv, th, nt;
    if (dw_lds_plan(h, wdt, c, stride, dtype, &csv, &th, &nt)) {
      const size_t lds = (size_t)(th + 2) * (wdt + 2) * csv * 16;
      const dim3 grid(c / vec / csv, nt, batch);
#define CASYNC_DW_LDS_LAUNCH(TT, CSVV)                                                                      \
  hipLaunchKernelGGL((dw3x3_lds_kernel<TT, CSVV>), grid, dim3(256), lds, stream, (const TT*)in, w, bias, \
                     (TT*)out, h, wdt, c, th)
      if (csv == 16) DT_DISPATCH(dtype, CASYNC_DW_LDS_LAUNCH(float, 16), CASYNC_DW_LDS_LAUNCH(bf16_t, 16));
      else DT_DISPATCH(dtype, CASYNC_DW_LDS_LAUNCH(float, 8), CASYNC_DW_LDS_LAUNCH(bf16_t, 8));
#undef CASYNC_DW_LDS_LAUNCH
      CASYNC_CHECK_HIP(hipGetLastError());
      return CASYNC_OK;
    }
  }
  if (stride == 1) {
    constexpr int PX = 4;
    const int strips = (wo + PX - 1) / PX;
    const long long total = (long long)batch * ho * strips * (c / vec);
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL((dw3x3_kernel<float, 1, PX>), dim3(blocks_for(total)), dim3(256), 0, stream,
                                   (const float*)in, w, bias, (float*)out, h, wdt, c, ho, wo, strips, total),
                hipLaunchKernelGGL((dw3x3_kernel<bf16_t, 1, PX>), dim3(blocks_for(total)), dim3(256), 0, stream,
                                   (const bf16_t*)in, w, bias, (bf16_t*)out, h, wdt, c, ho, wo, strips, total));
  } else {
    constexpr int PX = 2;
    const int strips = (wo + PX - 1) / PX;
    const long long total = (long long)batch * ho * strips * (c / vec);
    DT_DISPATCH(dtype,
                hipLaunchKernelGGL((dw3x3_kernel<float, 2, PX>), dim3(blocks_for(total)), dim3(256), 0, stream,
                                   (const float*)in, w, bias, (float*)out, h, wdt, c, ho, wo, strips, total),
                hipLaunchKernelGGL((dw3x3_kernel<bf16_t, 2, PX>), dim3(blocks_for(total)), dim3(256), 0, stream,
                                   (const bf16_t*)in, w, bias, (bf16_t*)out, h, wdt, c, ho, wo, strips, total));
  }
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_upsample2x(const void* in, void* out, int ldc, int batch, int h, int wdt, int c,
                      hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(in && out && batch > 0 && c % 4 == 0 && ldc >= c && ldc % 4 == 0 && h > 1 && wdt > 1,
                 "upsample2x: bad args");
  const long long total = (long long)batch * 4 * h * wdt * (c / 4);
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, stream,
                                 (const float*)in, (float*)out, ldc, h, wdt, c, total),
              hipLaunchKernelGGL(upsample2x_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, stream,
                                 (const bf16_t*)in, (bf16_t*)out, ldc, h, wdt, c, total));
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_nchw_to_nhwc(const float* in, void* out, int batch, int c, int hw, hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(in && out && batch > 0 && c % 4 == 0 && hw > 0, "nchw_to_nhwc: bad args");
  const long long total = (long long)batch * (c / 4) * hw;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, stream, in,
                                 (float*)out, c, hw, total),
              hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, stream, in,
                                 (bf16_t*)out, c, hw, total));
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_crop_to_input(const unsigned char* crops, float* x, int batch, hipStream_t stream) {
  CASYNC_REQUIRE(crops && x && batch > 0, "crop_to_input: bad args");
  const long long total = (long long)batch * 25600;
  hipLaunchKernelGGL(crop_to_input_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, crops, x, total);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_pred_to_u8(const float* pred, unsigned char* out, int batch, hipStream_t stream) {
  CASYNC_REQUIRE(pred && out && batch > 0, "pred_to_u8: bad args");
  const long long total = (long long)batch * 25600;
  hipLaunchKernelGGL(pred_to_u8_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, pred, out, total);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_audio_window_gather(const float* features, int n_steps, const int* idx_dev, void* out, int batch,
                               hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(features && idx_dev && out && batch > 0 && n_steps > 0, "audio_window_gather: bad args");
  const long long total = (long long)batch * 8 * 1024;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(audio_window_gather_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, stream,
                                 features, n_steps, idx_dev, (float*)out, total),
              hipLaunchKernelGGL(audio_window_gather_kernel<bf16_t>, dim3(blocks_for(total)), dim3(256), 0, stream,
                                 features, n_steps, idx_dev, (bf16_t*)out, total));
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_audio_windows_nchw(const float* features, int n_steps, const int* idx_dev, float* out, int batch,
                              hipStream_t stream) {
  CASYNC_REQUIRE(features && idx_dev && out && batch > 0 && n_steps > 0, "audio_windows: bad args");
  const long long total = (long long)batch * 8 * 1024;
  hipLaunchKernelGGL((audio_window_gather_kernel<float, true>), dim3(blocks_for(total)), dim3(256), 0, stream, features,
                     n_steps, idx_dev, out, total);
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_inc(const float* x_nchw, const float* packed_inc, void* out, int ldc, int batch,
               hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(x_nchw && packed_inc && out && batch > 0 && ldc >= INC_COUT && ldc % 4 == 0,
                 "inc: bad args");
  CASYNC_REQUIRE(batch <= (1 << 20), "inc: batch %d", batch);
  CASYNC_REQUIRE(((uintptr_t)packed_inc % 8) == 0, "inc: the packed weights must be 8-B aligned");
  const int nwg = batch * INC_TILES;
  if (dtype == DT_BF16 && casync_opts().inc_mfma) {
    CASYNC_REQUIRE(ldc % 8 == 0 && ((uintptr_t)out % 16) == 0 && (long long)INC_HW * INC_HW * ldc * 2 < (1ll << 31),
                   "inc (bf16): the output rows must be 16-B aligned and a frame smaller than 2 GiB");
    hipLaunchKernelGGL(inc_bf16_kernel, dim3(nwg), dim3(256), 0, stream, x_nchw, packed_inc, (bf16_t*)out, ldc, nwg);
    CASYNC_CHECK_HIP(hipGetLastError());
    return CASYNC_OK;
  }
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(inc_kernel<float>, dim3(nwg), dim3(256), 0, stream, x_nchw, packed_inc, (float*)out, ldc, nwg),
              hipLaunchKernelGGL(inc_kernel<bf16_t>, dim3(nwg), dim3(256), 0, stream, x_nchw, packed_inc, (bf16_t*)out, ldc, nwg));
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}

int launch_outc(const void* in, int ld_in, const float* w, const float* b, float* out_nchw,
                int batch, hipStream_t stream, int dtype) {
  CASYNC_REQUIRE(in && w && b && out_nchw && batch > 0 && ld_in >= 32 && ld_in % 4 == 0,
                 "outc: bad args");
  const long long blocks = (long long)batch * INC_HW * INC_HW / 256;
  DT_DISPATCH(dtype,
              hipLaunchKernelGGL(outc_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)in,
                                 ld_in, w, b, out_nchw),
              hipLaunchKernelGGL(outc_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream,
                                 (const bf16_t*)in, ld_in, w, b, out_nchw));
  CASYNC_CHECK_HIP(hipGetLastError());
  return CASYNC_OK;
}
