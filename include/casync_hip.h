/*
 * casync_hip.h -- C ABI of libcasync_hip.so, the MI355X (gfx950) engine for the
 * CASync lip-sync U-Net per-frame inference forward.
 *
 * The reference has no FFI: its boundary for this path is the Python call
 *     predictions = self.net(batch_tensor, hubert_tensor)
 * (reference image_infer_v1/tools/frame_synthesizer/infer_api.py:259-260) on an
 * nn.Module built by Model(6, "hubert") / load_state_dict / eval()
 * (infer_api.py:41-43; module/unet.py:273-345).  This library is what a
 * binding for that call binds: plain pointers and sizes, explicit stream, int
 * status codes, no C++ exceptions, no torch types.  calipsync_amd/unet.py is
 * the ctypes host that keeps the reference's Model.forward(x, audio_feat)
 * signature on top of it (INTEGRATION.md shows the stub).
 *
 * All device pointers are fp32, 16-byte aligned.  Tensors at the boundary are
 * the reference's own layouts (NCHW, contiguous); internally the engine works
 * in NHWC.  A handle is bound to one device and is not thread-safe: one
 * caller / one stream at a time (the reference has one caller at a time,
 * infer_api.py:259 under inference.py:80 or a worker thread).
 */
#ifndef CASYNC_HIP_H
#define CASYNC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct casync_engine* casync_handle;
typedef void* casync_stream;          /* a hipStream_t (NULL = default stream) */

/* status codes (0 = ok, negative = error; never throws) */
enum {
  CASYNC_OK = 0,
  CASYNC_ERR_ARG = -1,        /* null pointer / bad size / unsupported shape   */
  CASYNC_ERR_HIP = -2,        /* a HIP runtime call failed (see last_error)    */
  CASYNC_ERR_STATE = -3,      /* weights not loaded / workspace too small      */
  CASYNC_ERR_NO_DEVICE = -4   /* no gfx950 device visible                      */
};

/* ---- introspection (callable without a GPU) --------------------------- */
/* Bumped on every change of a prototype, struct layout or the packed-weight layout; the ctypes
 * host (calipsync_amd/_lib.py) refuses a library whose version differs from the one it binds.   */
#define CASYNC_ABI_VERSION 6
int         casync_abi_version(void);
const char* casync_last_error(void);           /* thread-local message         */

/* Packed-weight layout.  The host folds eval-mode BatchNorm into the conv /
 * linear weights (replaces nn.BatchNorm2d/1d + bias adds, module/unet.py:18,
 * 28,32,163,168,174,228,230,260,301,310,311) and writes each folded tensor at
 * the offset this table names.  Offsets/sizes are in floats.               */
int         casync_packed_count(void);
const char* casync_packed_name(int i);
int64_t     casync_packed_offset(int i);
int64_t     casync_packed_size(int i);
int64_t     casync_packed_total(void);          /* floats in the whole buffer  */

/* Workspace (activations, NHWC fp32) needed for a batch of B frames.        */
int64_t     casync_workspace_bytes(int batch);               /* fp32 engine            */
int64_t     casync_workspace_bytes_dt(int batch, int dtype); /* 0 = fp32, 1 = bf16     */
/* ... for THIS handle (its dtype and options: the arena is smaller when the fused kernels are on,
 * which is the default).  Any arena at least this large is accepted by casync_forward.          */
int64_t     casync_workspace_bytes_h(casync_handle h, int batch);

/* ---- tuning options ---------------------------------------------------- */
/* Every switch of the engine by name (the CASYNC_<NAME> environment variables, lower case without
 * the prefix: "lanes", "trunk_lanes", "overlap", "gemm_streamk", "fuse_ir", "fuse_q", ... -- DESIGN.md
 * lists them).  The environment is read ONCE per process; casync_create copies the process defaults
 * into the handle.  h != NULL changes that handle only; h == NULL changes the process defaults, which
 * the casync_op_* single-operator calls and handles created later use.  No counterpart in the
 * reference (it has no tuning surface).                                                         */
int  casync_set_option(casync_handle h, const char* name, int value);
int  casync_get_option(casync_handle h, const char* name, int* value);

/* ---- engine life cycle ------------------------------------------------- */
/* Replaces Model(6,"hubert").to(device) (infer_api.py:41).                   */
int  casync_create(int device_id, casync_handle* out);
/* dtype = activation storage type inside the engine: 0 = fp32 (parity path, < 1e-3 vs
 * the reference), 1 = bf16 activations + bf16 matrix-core operands with fp32 accumulate
 * (BASELINE configs[2]; ~1e-2 vs the reference, reported separately).  The boundary
 * tensors stay fp32 either way.                                                  */
int  casync_create_ex(int device_id, int dtype, casync_handle* out);
void casync_destroy(casync_handle h);

/* Replaces net.load_state_dict(...) (infer_api.py:42): the packed, BN-folded
 * buffer of casync_packed_total() floats.  _host copies from host memory into
 * an engine-owned device buffer; _device adopts a caller-owned device buffer
 * (e.g. the tensor an RCCL broadcast just filled) without copying -- the
 * caller keeps it alive for the life of the handle.                          */
int  casync_load_weights_host(casync_handle h, const float* packed, int64_t n_floats);
int  casync_load_weights_device(casync_handle h, const float* packed_dev, int64_t n_floats);

/* Replaces Model.forward(x, audio_feat) (module/unet.py:314-345).
 *   x_dev     [B,6,160,160]  NCHW fp32   (reference crop ch0-2, masked crop ch3-5)
 *   audio_dev [B,32,32,32]   NCHW fp32   (HuBERT window)
 *   out_dev   [B,3,160,160]  NCHW fp32   in (0,1)
 * Enqueues on `stream`, no host synchronisation, no allocation.             */
int  casync_forward(casync_handle h, const float* x_dev, const float* audio_dev,
                    float* out_dev, int batch, void* workspace_dev,
                    int64_t workspace_bytes, casync_stream stream);

/* Same forward, but the HuBERT windows are gathered on the device: `features_dev` is the whole
 * [n_steps, 2, 1024] fp32 feature array of the clip (uploaded once), frame_idx_dev[b] the video
 * frame index of batch entry b; entry b sees features[idx-8 : idx+8], zero-padded past both
 * ends, reshaped to (32,32,32) -- exactly FrameSynthesizer._get_audio_features
 * (infer_api.py:99-145), without the B x 128 KB host windows and their H2D copy.           */
int  casync_forward_windows(casync_handle h, const float* x_dev, const float* features_dev,
                            int n_steps, const int32_t* frame_idx_dev, float* out_dev, int batch,
                            void* workspace_dev, int64_t workspace_bytes, casync_stream stream);

/* Debug taps: copy a named NHWC intermediate of the LAST forward (same batch,
 * same workspace) into dst_dev; returns its per-frame float count or <0.
 * Names: x1 x2 x3 x4 x5 a tx kx fuse u1 u2 u3 u4 att0..att3 audio_conv2..5   */
int64_t casync_tap(casync_handle h, const char* name, int batch, void* workspace_dev,
                   void* dst_dev, int64_t dst_elems, casync_stream stream);  /* dst: engine dtype */

/* Per-kernel timing of one forward (HIP events around every launch on
 * `stream`; synchronises).  Writes up to `cap` entries; returns the count.  */
typedef struct {
  char  name[48];     /* plan step, e.g. "up4.conv.double_conv.0.fused"      */
  char  kernel[64];   /* HIP kernel instance as rocprofv3 names it           */
  float ms;           /* event-pair time minus the calibrated cost of an empty pair (clamped at 0) */
  float ms_raw;       /* the event-pair time as measured                     */
  double flops;       /* algorithmic flops of this launch                   */
  double bytes;       /* algorithmic bytes (inputs read once + outputs)     */
} casync_kernel_time;
int  casync_profile_forward(casync_handle h, const float* x_dev, const float* audio_dev,
                            float* out_dev, int batch, void* workspace_dev,
                            int64_t workspace_bytes, casync_stream stream,
                            casync_kernel_time* out, int cap);

/* ---- single operators (used by the parity tests and micro-benchmarks) ---- */
/* Activation storage type of the casync_op_* calls made by this thread afterwards:
 * 0 = fp32 (default), 1 = bf16 (activation / weight pointers are then bf16; bias,
 * scales and all arithmetic stay fp32).                                        */
int casync_op_set_dtype(int dtype);
/* 1x1 conv / linear as GEMM on NHWC rows: C[M,N] = epi(A[M,K] * W[N,K]^T).
 * Replaces nn.Conv2d(k=1)/nn.Linear + folded BN + LeakyReLU (+ residual)
 * (module/unet.py:17-20,31-33,201-204,227-229,256-259).
 * epilogue: v = acc + bias[n]; v += pre_scale[n]*pre_res[m,n]; v = lrelu(v) if
 * act; v += post_res[m,n]; v = lrelu(v*aff_s[n]+aff_t[n]) if aff_s.         */
int casync_op_pw_gemm(const void* a, int lda, const void* w, const float* bias,
                      void* c, int ldc, int m, int n, int k, int act,
                      const void* pre_res, int ld_pre, const float* pre_scale,
                      const void* post_res, int ld_post,
                      const float* aff_s, const float* aff_t, casync_stream stream);
/* Diagnostic (tools/experiments/gemm_timeline.py): the casync_op_pw_gemm calls this thread makes next
 * write 8 timestamp words per workgroup into dev_words (NULL = off).  Never used by the engine.  */
int casync_debug_gemm_stamps(void* dev_words);
/* Same for the fused inverted-residual kernel (fp32): 5 words per workgroup (first 4096 workgroups) = shader
 * cycles of wave 0 in prologue / P1 / P2 / P3 / epilogue.  tools/experiments/ir_timeline.py.              */
int casync_debug_ir_stamps(void* dev_words);
/* nn.Conv2d(k=3, bias) + folded BN + LeakyReLU of the audio encoder (conv3: stride 2 pad 1, conv5: stride 2
 * pad 3; module/unet.py:161-168) as an implicit GEMM: in [B,H,W,cin] NHWC, w [cout][(ky,kx,cin)],
 * out [B,Ho,Wo,cout].  cin % (128 B / elem) == 0, cout % 64 == 0. */
int casync_op_conv3x3(const void* in, const void* w, const float* bias, void* out, int batch, int h, int w_,
                      int cin, int cout, int stride, int pad, int act, casync_stream stream);

/* Depthwise 3x3, pad 1, stride 1|2, + bias + LeakyReLU on NHWC.
 * Replaces nn.Conv2d(groups=C,k=3)+BN+LeakyReLU (module/unet.py:21-30).
 * w is tap-major [9][C].                                                    */
int casync_op_dw3x3(const void* in, const float* w, const float* bias, void* out,
                    int batch, int h, int wdt, int c, int stride, casync_stream stream);
/* The same depthwise 3x3 (pad 1, stride 1) + bias + LeakyReLU behind an Up block's first expand conv whose two
 * input-channel halves were computed apart (engine plan `skip_early`, fp32): the conv's input is
 * LeakyReLU(pre + up(g)), pre = W1b . skip + b at h x wdt, g = W1a . lo at (h/2) x (wdt/2), up = bilinear x2 with
 * align_corners=True.  Replaces nn.Upsample + torch.cat + the expand conv's LeakyReLU + Conv2d(groups=C,k=3)+BN+
 * LeakyReLU (module/unet.py:90-97, 17-30).  pre [B,h,wdt,c] NHWC, g [B,h/2,wdt/2,ldg] (first c columns used),
 * w tap-major [9][C], out [B,h,wdt,c].  h, wdt even, c % 4 == 0, frames the LDS-slab plan takes (10..40).     */
int casync_op_dw3x3_ups(const float* pre, const float* g, int ldg, const float* w, const float* bias, float* out,
                        int batch, int h, int wdt, int c, casync_stream stream);
/* Expand 1x1 conv + BN + LeakyReLU + depthwise 3x3 (pad 1, stride 1|2) + BN + LeakyReLU in one kernel, fp32, for the
 * inverted residuals below 32x32 (module/unet.py:17-30): the GEMM's output tile is whole frames, the depthwise conv
 * runs on it in LDS.  a [frames*hw*hw, lda], w1 [cexp][cin], b1 [cexp], wd [9][cexp] tap-major, bd [cexp],
 * d [frames*ho*ho, ldd].  hw in {10, 16, 20} (stride 2 only at 20), cin % 32 == 0, cexp % 64 == 0.           */
int casync_op_pw_dw(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd,
                    void* d, int ldd, int frames, int hw, int stride, int cin, int cexp, const void* ups, int ld_ups,
                    casync_stream stream);
/* 1x1 conv + bias + [bilinear x2 upsample (align_corners=True) of a low-resolution tensor] + LeakyReLU:
 * c[m, :] = act(a[m, :] . w^T + bias + up2x(ups)[m, :]), rows m = pixels (b, y, x) of h x w frames, ups
 * [B*(h/2)*(w/2), ld_ups].  An Up block's expand conv (module/unet.py:90-96 + 17-20) with the upsample commuted behind
 * the conv: W1 . cat(up(lo), skip) = up(W1a . lo) + W1b . skip; `ups` = W1a . lo, a = skip, w = W1b.  casync_op_pw_dw
 * takes the same optional addend (ups may be NULL there).                                                   */
int casync_op_pw_gemm_ups(const void* a, int lda, const void* w, const float* bias, void* c, int ldc, int m, int n,
                          int k, int act, const void* ups, int ld_ups, int h, int w_, casync_stream stream);
/* Whole inverted-residual block in one kernel (expanded tensor stays in LDS); the
 * high-resolution stages use it.  Replaces InvertedResidual.forward
 * (module/unet.py:16-40) with BN folded: w1 [2cin][cin], wd [9][2cin], w2 [cout][2cin]
 * (w1/w2 in the op dtype, the rest fp32).
 * Returns CASYNC_ERR_ARG if (cin, cout, stride) has no instance.              */
int casync_op_ir_fused(const void* in, int ld_in, const void* w1, const float* b1,
                       const float* wd, const float* bd, const void* w2, const float* b2,
                       void* out, int ld_out, int batch, int h, int w, int cin, int cout,
                       int stride, int res, casync_stream stream);
/* Decoder variant: the block input is cat([bilinear_x2(lo)[0:c_lo], in[c_lo:cin]]) with the
 * upsample (align_corners=True) computed while loading -- Up.forward's interpolate + cat +
 * first InvertedResidual (module/unet.py:90-97) in one kernel.  lo: [B,h/2,w/2,ld_lo].     */
int casync_op_ir_fused_up(const void* lo, int ld_lo, int c_lo, const void* in, int ld_in,
                          const void* w1, const float* b1, const float* wd, const float* bd,
                          const void* w2, const float* b2, void* out, int ld_out, int batch,
                          int h, int w, int cin, int cout, casync_stream stream);
/* The same block with the upsample COMMUTED behind the expand conv (fp32 only): bilinear interpolation is linear
 * per channel and the 1x1 conv linear per pixel, so W1 . cat(up(lo), skip) = up(W1a . lo) + W1b . skip.
 * g = W1a . lo [B,h/2,w/2,ld_g >= 2*cin] (no bias), in = the skip half [B,h,w,ld_in >= cin/2], w1b [2*cin][cin/2];
 * cin is the block's logical input width (64 / 128).  Same result as casync_op_ir_fused_up up to fp32 rounding. */
int casync_op_ir_fused_upg(const float* g, int ld_g, const float* in, int ld_in, const float* w1b,
                           const float* b1, const float* wd, const float* bd, const float* w2,
                           const float* b2, float* out, int ld_out, int batch, int h, int w, int cin,
                           int cout, casync_stream stream);
/* Bilinear x2, align_corners=True (module/unet.py:86-87,91), NHWC, writing
 * into a wider row (ldc) so the concat with the skip is free.               */
int casync_op_upsample2x(const void* in, void* out, int ldc, int batch, int h, int wdt,
                         int c, casync_stream stream);
/* Cross attention core (module/unet.py:212-217): per frame
 * out = gamma * (softmax_j(Q K^T) V) + res, 100 face x 100 audio positions.  */
int casync_op_cross_attention(const void* q, int ldq, const void* k, int ldk,
                              const void* v, int ldv, const void* res, int ld_res,
                              const float* gamma_dev, void* out, int ld_out,
                              int batch, casync_stream stream);
/* FrameSynthesizer._get_audio_features (infer_api.py:99-145) alone, on the device: the gather casync_forward_windows
 * starts with, as an operator.  features_dev [n_steps,2,1024] fp32, frame_idx_dev [batch] int32 (any value: negative,
 * past the end).  nhwc = 0: windows_dev is the reference's own return value, [batch,32,32,32] fp32 (window row r of
 * features[idx-8 : idx+8] = channels 2r, 2r+1; all zeros where the reference falls back to its default window);
 * nhwc = 1: the engine's image [batch,1024,32] in the storage type of casync_op_set_dtype (what the first audio
 * kernel reads).  Bit-exact against the reference's own output (tests/golden/frame_windows.npz).                    */
int casync_op_audio_windows(const float* features_dev, int n_steps, const int32_t* frame_idx_dev, void* windows_dev,
                            int batch, int nhwc, casync_stream stream);
/* Tensor glue of FrameSynthesizer.process_batch around the model call, on the device:
 * crop_to_input: resized 168x168 BGR crops (uint8 HWC) -> the [B,6,160,160] fp32 model input
 *   (inner [4:164,4:164], masked copy with the black rectangle (5,5,150,145), HWC->CHW, /255,
 *   concat) -- infer_api.py:238-245;  pred_to_u8: pred*255 -> uint8 HWC [B,160,160,3]
 *   (truncation) -- infer_api.py:265-266.  Bit-exact; cv2.resize / blending stay on the host. */
int casync_op_crop_to_input(const uint8_t* crops168_dev, float* x_dev, int batch, casync_stream stream);
int casync_op_pred_to_u8(const float* pred_dev, uint8_t* out_dev, int batch, casync_stream stream);
/* The image arithmetic of FrameSynthesizer.process_batch on ragged per-frame regions (infer_api.py:234-235 and
 * 263-346).  The host slices every frame's crop box img[ymin:ymax, xmin:xmax] (infer_api.py:206-234) into ONE
 * byte buffer `regions` (h x w x 3 uint8 each, contiguous) and describes frame b in geom[b*12 .. b*12+11]
 * (int32): { region byte offset, h, w, width (xmax-xmin BEFORE the clamps: side of the synthesised square),
 * valid (1 when (width,width) == (h,w), else the frame is returned unchanged: infer_api.py:320-324), byte offset in
 * `synth`, byte offset in the two mask buffers, kind of the optional frame mask (-1 = none, 0 = float32 in [0,1],
 * 1 = uint8 standing for value / 255 -- what infer_api.py:68-70 computes on the host), its height, its width, and the
 * low / high 32 bits of its device address (masks live in allocations of their own, so a clip's masks can stay resident
 * on the device across batches instead of being uploaded with every batch) }.  pts: [B][33][2] int32 = the contour points already shifted / scaled / truncated
 * (infer_api.py:281-289).
 *   casync_frame_prepare: cv2.resize(region, (168,168)) -> crops168 [B,168,168,3] u8, and (x_dev != NULL) the
 *     [B,6,160,160] model input of casync_op_crop_to_input.
 *   casync_frame_paste_back: crop[4:164,4:164] = uint8(pred*255); cv2.resize to (width,width); cv2.fillPoly; area-
 *     scaled cv2.dilate; float64 blend with the original region (and the optional float mask, resized) -> out_regions
 *     (same layout as `regions`).  synth / mask_a / mask_b / area are scratch (sizes: sum of width*width*3, 2 x
 *     mask_bytes = sum of h*w, B int32).
 * OpenCV's arithmetic is restated from its published sources (resize.cpp, drawing.cpp, morph.cpp); cv2 is not
 * available in the build image, so parity with the real library is UNPINNED; parity with the CPU restatement
 * (oracle/frame_ops_oracle.py) is bit-exact.                                                                    */
int casync_frame_prepare(const uint8_t* regions_dev, const int32_t* geom_dev, int batch, uint8_t* crops168_dev,
                         float* x_dev, casync_stream stream);
int casync_frame_paste_back(const uint8_t* regions_dev, const int32_t* geom_dev, const int32_t* pts_dev,
                            const uint8_t* crops168_dev, const float* pred_dev, int batch,
                            int max_h, int max_w, int max_width, int64_t mask_bytes, uint8_t* synth_dev,
                            uint8_t* mask_a_dev, uint8_t* mask_b_dev, int32_t* area_dev, uint8_t* out_regions_dev,
                            casync_stream stream);
/* NCHW <-> NHWC helpers */
int casync_op_nchw_to_nhwc(const float* in, void* out, int batch, int c, int hw,
                           casync_stream stream);
/* inc block straight from the NCHW face crop (module/unet.py:58-67,290)      */
int casync_op_inc(const float* x_nchw, const float* packed_inc, void* out, int ldc,
                  int batch, casync_stream stream);
/* OutConv + outc_bn + sigmoid -> NCHW (module/unet.py:100-106,342-344)       */
int casync_op_outc(const void* in, int ld_in, const float* w, const float* b,
                   float* out_nchw, int batch, casync_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* CASYNC_HIP_H */
