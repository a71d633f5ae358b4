// The forward plan of the CASync U-Net on one MI355X: packed-weight layout, workspace
// arena and the launch sequence that replaces Model.forward (reference
// module/unet.py:314-345).  Mirrors calipsync_amd/arch.py (tests/test_abi.py checks the two
// agree).  Everything is NHWC fp32 inside; NCHW only at the boundary.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------ error text
static thread_local char g_err[512] = "";
void casync_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* casync_last_error(void) { return g_err; }

// diagnostic: device buffer of 8 x grid words that the GEMM launches of this thread stamp (null = off; see
// casync_debug_gemm_stamps).  Never set on the product path.
static thread_local unsigned long long* g_gemm_stamps = nullptr;

namespace {

// ------------------------------------------------------------------ architecture table
struct IR {  // inverted residual: PW expand -> DW3x3 -> PW project (module/unet.py:8-40)
  const char* prefix;
  int cin, cout, stride, res, hw_in;
  int cexp() const { return cin * 2; }
  int hw_out() const { return stride == 1 ? hw_in : (hw_in + 2 - 3) / 2 + 1; }
};

const IR kInc = {"inc.inconv.0", 6, 32, 1, 0, 160};
const IR kDown[4][2] = {
    {{"down1.maxpool_conv.0.double_conv.0", 32, 64, 2, 0, 160}, {"down1.maxpool_conv.0.double_conv.1", 64, 64, 1, 1, 80}},
    {{"down2.maxpool_conv.0.double_conv.0", 64, 128, 2, 0, 80}, {"down2.maxpool_conv.0.double_conv.1", 128, 128, 1, 1, 40}},
    {{"down3.maxpool_conv.0.double_conv.0", 128, 256, 2, 0, 40}, {"down3.maxpool_conv.0.double_conv.1", 256, 256, 1, 1, 20}},
    {{"down4.maxpool_conv.0.double_conv.0", 256, 512, 2, 0, 20}, {"down4.maxpool_conv.0.double_conv.1", 512, 512, 1, 1, 10}}};
const IR kAudio[5] = {{"audio_model.conv1", 32, 64, 1, 0, 32},
                      {"audio_model.conv2", 64, 128, 1, 0, 32},
                      {"audio_model.conv4", 256, 256, 1, 1, 16},
                      {"audio_model.conv6", 512, 512, 1, 1, 10},
                      {"audio_model.conv7", 512, 512, 1, 1, 10}};
const IR kFuse[4] = {{"fuse_conv.0.double_conv.0", 1024, 512, 1, 0, 10},
                     {"fuse_conv.0.double_conv.1", 512, 512, 1, 1, 10},
                     {"fuse_conv.1.double_conv.0", 512, 256, 1, 0, 10},
                     {"fuse_conv.1.double_conv.1", 256, 256, 1, 1, 10}};
const IR kUp[4][2] = {
    {{"up1.conv.double_conv.0", 512, 128, 1, 0, 20}, {"up1.conv.double_conv.1", 128, 128, 1, 1, 20}},
    {{"up2.conv.double_conv.0", 256, 64, 1, 0, 40}, {"up2.conv.double_conv.1", 64, 64, 1, 1, 40}},
    {{"up3.conv.double_conv.0", 128, 32, 1, 0, 80}, {"up3.conv.double_conv.1", 32, 32, 1, 1, 80}},
    {{"up4.conv.double_conv.0", 64, 32, 1, 0, 160}, {"up4.conv.double_conv.1", 32, 32, 1, 1, 160}}};
constexpr int kBlocks = 4;          // attention blocks
constexpr int kKV = 64 + 512;       // per-block [K | V] projection columns

// ------------------------------------------------------------------ packed weights
struct Packed {
  std::string name;
  int64_t offset, size;
};

struct Layout {
  std::vector<Packed> items;
  int64_t total = 0;
  void add(const std::string& name, int64_t size) {
    items.push_back({name, total, size});
    total += (size + 63) / 64 * 64;  // 256-B aligned tensors
  }
  void add_ir(const IR& b) {
    const std::string p = b.prefix;
    add(p + ".pw1.w", (int64_t)b.cexp() * b.cin);
    add(p + ".pw1.b", b.cexp());
    add(p + ".dw.w", 9 * (int64_t)b.cexp());
    add(p + ".dw.b", b.cexp());
    add(p + ".pw2.w", (int64_t)b.cout * b.cexp());
    add(p + ".pw2.b", b.cout);
  }
  int64_t off(const std::string& name) const {
    for (const auto& it : items)
      if (it.name == name) return it.offset;
    return -1;
  }
};

const Layout& layout() {
  static const Layout L = [] {
    Layout l;
    l.add("inc.inconv.0.fused", 620);  // [w1T 6x12][b1 12][wd 9x12][bd 12][w2T 12x32][b2 32]
    for (auto& st : kDown)
      for (auto& b : st) l.add_ir(b);
    l.add_ir(kAudio[0]);
    l.add_ir(kAudio[1]);
    l.add("audio_model.conv3.w", 256ll * 9 * 128);  // [N][(ky,kx,cin)]
    l.add("audio_model.conv3.b", 256);
    l.add_ir(kAudio[2]);
    l.add("audio_model.conv5.w", 512ll * 9 * 256);
    l.add("audio_model.conv5.b", 512);
    l.add_ir(kAudio[3]);
    l.add_ir(kAudio[4]);
    l.add("audio_model.bn7.s", 512);
    l.add("audio_model.bn7.t", 512);
    l.add("mlp_fusion.fc1.w", 1024ll * 1024);
    l.add("mlp_fusion.fc1.b", 1024);
    l.add("mlp_fusion.fc2.w", 1024ll * 1024);  // bn2 and bn_tx folded in
    l.add("mlp_fusion.fc2.b", 1024);
    l.add("mlp_fusion.fc2.rs", 1024);          // bn_tx scale on the cat(x5, a) residual
    l.add("att.kv.w", (int64_t)kBlocks * kKV * 512);  // rows: blk0 K(64) V(512), blk1 ...
    l.add("att.kv.b", kBlocks * kKV);
    for (int i = 0; i < kBlocks; ++i) {
      const std::string p = "attention_blocks." + std::to_string(i);
      l.add(p + ".p1.w", 512ll * 1024);
      l.add(p + ".p1.b", 512);
      l.add(p + ".q.w", 64ll * 512);
      l.add(p + ".q.b", 64);
      // rows 0..511 = p_1, rows 512..575 = query_conv o p_1 composed on the host in float64
      // (module/unet.py:201,209,256,264): q comes out of the p_1 GEMM as 64 extra columns
      l.add(p + ".p1q.w", 576ll * 1024);
      l.add(p + ".p1q.b", 576);
      l.add(p + ".gamma", 1);
      l.add(p + ".b1.w", 1024ll * 512);  // block bn folded in
      l.add(p + ".b1.b", 1024);
      l.add(p + ".b1.rs", 1024);         // block bn scale on the tx residual
    }
    l.add("bn_kx.s", 1024);
    l.add("bn_kx.t", 1024);
    for (auto& b : kFuse) l.add_ir(b);
    for (auto& st : kUp)
      for (auto& b : st) l.add_ir(b);
    // Up blocks: the two K-halves of the first block's expand matrix as matrices of their own -- [cexp][cin/2] for the
    // upsampled input (applied at the low resolution: upsample and 1x1 conv commute) and for the skip
    for (auto& st : kUp) {
      l.add(std::string(st[0].prefix) + ".pw1a.w", (int64_t)st[0].cexp() * (st[0].cin / 2));
      l.add(std::string(st[0].prefix) + ".pw1b.w", (int64_t)st[0].cexp() * (st[0].cin / 2));
    }
    l.add("outc.w", 96);  // [3][32], outc_bn folded in
    l.add("outc.b", 3);
    return l;
  }();
  return L;
}

// ------------------------------------------------------------------ workspace arena
// An activation pointer that knows its element size: `p + n` advances n ELEMENTS of the engine's
// storage type (fp32 or bf16) and it converts to void* at the kernel launchers.
struct Ptr {
  char* p = nullptr;
  int esz = 4;
  Ptr operator+(long long n) const { return Ptr{p + n * esz, esz}; }
  operator void*() const { return p; }
  operator const void*() const { return p; }
};

// Largest expanded (2x) tensor an UN-fused inverted residual of the plan writes, per frame: the E1/E2
// slots are sized for it.  With the fused kernels on (default) that is up2.0's 40x40x512, not the
// 160x160x128 of up4.0 the fused kernel keeps in LDS (19.6 MB per frame less workspace).
bool ir_is_fused(const CasyncOptions& o, const IR& b) {
  return o.fuse_ir && b.hw_in >= o.fuse_min_hw && ir_fused_supported(b.cin, b.cout, b.stride);
}
bool up_is_fused(const CasyncOptions& o, const IR& b0) {
  return o.fuse_ir && o.fuse_up && b0.hw_in >= o.fuse_min_hw && ir_fused_up_supported(b0.cin, b0.cout);
}
// How the first inverted residual of an Up stage takes its upsampled half.  ONE predicate for the plan (decode()) and for
// the workspace (max_unfused_expand()): round 3 had them apart and `fuse_up=0` sent up3.0 / up4.0 down the un-fused chain
// with E1 / E2 sized for up2.0.
enum class UpMode {
  CommuteUnfused,   // G = W1a.lo at the low resolution, then the un-fused chain (pw_dw / GEMM + depthwise) adds up(G)
  CommuteFused,     // G = W1a.lo, then ir_fused_kernel<.., UPS = 2>
  FusedLoad,        // ir_fused_kernel<.., UPS = 1>: bilinear taps while loading the A fragments
  Materialise       // upsample2x into the concat buffer, then the block as any other inverted residual
};
// does the bf16 engine's expand + depthwise kernel take this block with its upsampled addend? (`batch` frames per launch)
bool bf16_commutes(const CasyncOptions& o, const IR& b0, int batch) {
  return o.ups_commute_bf16 && o.fuse_dw_bf16 && (b0.hw_in < 40 || o.fuse_dw_bf16 >= 2) && batch >= o.fuse_dw_bf16_min &&
         pw_dw_bf16_takes_ups(b0.hw_in, b0.stride) && pw_dw_bf16_supported(b0.hw_in, b0.cin / 2, b0.cexp(), b0.stride);
}
UpMode up_mode(const CasyncOptions& o, const IR& b0, int dtype, int batch = 0) {
  const bool f32 = dtype == DT_F32;
  if (up_is_fused(o, b0)) return f32 && o.ups_commute >= 2 ? UpMode::CommuteFused : UpMode::FusedLoad;
  // a block the plain fused kernel takes (fuse_up = 0, fuse_ir = 1) keeps it: upsample first, E never leaves LDS
  if (!ir_is_fused(o, b0) && (f32 ? o.ups_commute != 0 : bf16_commutes(o, b0, batch))) return UpMode::CommuteUnfused;
  return UpMode::Materialise;
}
constexpr int kMinLaneBatch = 16;
// skip_early (decode()): small single-lane batches run the skip half of up1.0 / up2.0's expand conv beside the trunk
bool skip_early_batch(const CasyncOptions& o, int batch) {
  return o.skip_early > 0 && batch < o.skip_early && batch < kMinLaneBatch * 2;
}
int64_t max_unfused_expand(const CasyncOptions& o) {
  int64_t mx = 0;
  auto see = [&](const IR& b, bool fused) {
    if (!fused) mx = std::max<int64_t>(mx, (int64_t)b.hw_in * b.hw_in * b.cexp());
  };
  for (auto& st : kDown)
    for (auto& b : st) see(b, ir_is_fused(o, b));
  for (auto& b : kFuse) see(b, ir_is_fused(o, b));
  for (auto& st : kUp) {   // (dtype-independent: the fp32-only modes never un-fuse a block bf16 would fuse)
    see(st[0], up_is_fused(o, st[0]) || ir_is_fused(o, st[0]));
    see(st[1], ir_is_fused(o, st[1]));
  }
  return mx;
}

struct Arena {
  // per-frame float counts; pointers are filled by bind()
  struct Buf {
    const char* name;
    int64_t per_frame;
    char* p;
  };
  int esz = 4;
  enum Id {
    CAT4, CAT3, CAT2, CAT1, CATA, E1, E2, T0, U4, F, FM, U1, U2, U3, UG,
    A0, AC1, AC2, AC3, AC4, AC5, AC6, AE1, AE2,
    H, TX, OX0, OX1, OX2, OX3, KX, KXF, P1Q, AO, Q, KV, EP1, EP2, COUNT
  };
  Buf b[COUNT] = {
      {"cat4", 160 * 160 * 64, 0},  {"cat3", 80 * 80 * 128, 0},  {"cat2", 40 * 40 * 256, 0},
      {"cat1", 20 * 20 * 512, 0},   {"catA", 100 * 1024, 0},     {"E1", 160 * 160 * 128, 0},
      {"E2", 160 * 160 * 128, 0},   {"T0", 160 * 160 * 32, 0},   {"U4", 160 * 160 * 32, 0},
      {"F", 100 * 256, 0},          {"FM", 100 * 512, 0},        {"U1", 400 * 128, 0},
      {"U2", 1600 * 64, 0},         {"U3", 6400 * 32, 0},        {"UG", 6400 * 128, 0},   // W1a . lo of an Up block, low res
      {"A0", 1024 * 32, 0},
      {"AC1", 1024 * 64, 0},        {"AC2", 1024 * 128, 0},
      {"AC3", 256 * 256, 0},        {"AC4", 256 * 256, 0},       {"AC5", 100 * 512, 0},
      {"AC6", 100 * 512, 0},        {"AE1", 131072, 0},          {"AE2", 131072, 0},
      {"H", 100 * 1024, 0},         {"TX", 100 * 1024, 0},       {"OX0", 100 * 1024, 0},
      {"OX1", 100 * 1024, 0},       {"OX2", 100 * 1024, 0},      {"OX3", 100 * 1024, 0},
      {"KX", 100 * 1024, 0},        {"KXF", 100 * 1024, 0},      {"P1Q", 100 * 576, 0},
      {"AO", 100 * 512, 0},         {"Q", 100 * 64, 0},          {"KV", 100 * kBlocks * kKV, 0},
      // skip_early: W1b . skip + b of up1.0 (20x20 x 1024) and up2.0 (40x40 x 512), small batches only
      {"EP1", 400 * 1024, 0},       {"EP2", 1600 * 512, 0}};
  // (the two skip_early buffers exist only for the batches that use them: 4.9 MB per frame)
  Arena(const CasyncOptions& o, int batch) {
    b[E1].per_frame = b[E2].per_frame = max_unfused_expand(o);
    if (!skip_early_batch(o, batch)) b[EP1].per_frame = b[EP2].per_frame = 0;
  }
  static int64_t bytes(const CasyncOptions& o, int batch, int esz = 4) {
    Arena a(o, batch);
    int64_t tot = 0;
    for (auto& x : a.b) tot += (x.per_frame * batch + 63) / 64 * 64;
    return tot * (int64_t)esz;
  }
  void bind(void* base, int batch, int elem_size) {
    esz = elem_size;
    char* p = (char*)base;
    for (auto& x : b) {
      x.p = p;
      p += ((x.per_frame * batch + 63) / 64 * 64) * (int64_t)esz;
    }
  }
  // view of frames [b0, ...) of an arena bound for the whole batch (every buffer is frame-major)
  void slice(int b0) {
    for (auto& x : b) x.p += x.per_frame * (int64_t)b0 * esz;
  }
  Ptr operator[](Id i) const { return Ptr{b[i].p, esz}; }
  // elements per frame of the buffer that starts at `p` (-1: not the start of a buffer)
  int64_t per_frame_of(const Ptr& p) const {
    for (auto& x : b)
      if (x.p == p.p) return x.per_frame;
    return -1;
  }
};

// hipSetDevice for the life of a scope (the caller's device is restored on exit)
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    else if (err == hipSuccess) prev = -1;   // nothing to restore
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct casync_engine {
  int device = 0;
  int dtype = DT_F32;        // activation storage type (DT_BF16: bf16 activations + bf16 GEMM weights)
  const float* w = nullptr;  // packed weights on the device (fp32: biases, scales, DW / fused-IR weights)
  float* owned = nullptr;
  bf16_t* w16 = nullptr;     // bf16 image of the whole packed buffer (same element offsets), DT_BF16 only
  // stream-K scratch of the GEMMs (gemm.hip): one private region per stream that can run a GEMM
  // (lane x {main, audio}), counters zeroed once here and left zeroed by every launch
  char* sk = nullptr;
  char* sk_ctx(int lane, int aux_stream) const { return sk ? sk + (size_t)(lane * 2 + aux_stream) * kStreamKBytes : nullptr; }
  // second stream per lane for the audio encoder, which is independent of the face encoder until
  // the fusion MLP (module/unet.py:315-321): forked/joined with events inside casync_forward
  // Lanes: the batch is cut into kMaxLanes contiguous sub-batches that run concurrently, lane 0
  // on the caller's stream, the others on engine-owned streams, so one lane's memory-bound
  // kernels and kernel tails overlap another lane's MFMA-bound GEMMs.  Every lane also forks
  // its audio encoder onto its own second stream.
  static constexpr int kMaxLanes = 4;
  hipStream_t lane_s[kMaxLanes] = {};   // [0] unused (caller's stream)
  hipStream_t aux[kMaxLanes] = {};
  hipEvent_t ev_fork[kMaxLanes] = {}, ev_join[kMaxLanes] = {}, ev_done[kMaxLanes] = {}, ev_mid[kMaxLanes] = {};
  hipEvent_t ev_start = nullptr, ev_start2 = nullptr;
  hipEvent_t ev_fwd = nullptr;   // recorded behind this handle's last forward when another handle's forward follows it (FwdGate)
  bool streams_ready = false;
  CasyncOptions opt;         // this handle's switches: process defaults at create, casync_set_option afterwards
  CasyncOptions eff;         // what the forward in progress runs under (run_forward: `opt` + the bf16 large-batch plan)
  const float* W(const std::string& name) const { return w + layout().off(name); }
  // GEMM weight matrix in the engine's storage type
  const void* WG(const std::string& name) const {
    return dtype == DT_BF16 ? (const void*)(w16 + layout().off(name)) : (const void*)(w + layout().off(name));
  }
};

namespace {

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ in,
                                                          bf16_t* __restrict__ out, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i < n) st4(out + i, ld4(in + i));
}

// ------------------------------------------------------------------ launch recorder
struct Runner {
  hipStream_t s;
  bool profile = false;
  std::vector<casync_kernel_time> rec;
  std::vector<hipEvent_t> ev;
  int status = CASYNC_OK;

  template <class F>
  void run(const char* name, const char* kernel, double flops, double bytes, F&& f) {
    if (status != CASYNC_OK) return;
    if (profile) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, s);
      status = f();
      (void)hipEventRecord(b, s);
      ev.push_back(a);
      ev.push_back(b);
      casync_kernel_time t;
      memset(&t, 0, sizeof(t));
      strncpy(t.name, name, sizeof(t.name) - 1);
      strncpy(t.kernel, kernel, sizeof(t.kernel) - 1);
      t.flops = flops;
      t.bytes = bytes;
      rec.push_back(t);
    } else {
      status = f();
    }
  }
  void finish() {
    if (!profile) return;
    // An event pair around a launch also times the command processor's own work between two timestamps (~3-4 us on
    // MI355X), which is not kernel time: rocprofv3's begin/end of the same launches were 10 % shorter than the raw
    // pairs.  Calibrate it on this stream -- the median of 16 pairs with NOTHING between them -- and
    // take it off every figure, so that the HIP-event averages and the rocprofv3 averages of profiles/ agree.
    constexpr int NCAL = 16;
    hipEvent_t cal[2 * NCAL];
    for (int i = 0; i < 2 * NCAL; ++i) {
      (void)hipEventCreate(&cal[i]);
      (void)hipEventRecord(cal[i], s);
    }
    (void)hipStreamSynchronize(s);
    float gaps[NCAL];
    for (int i = 0; i < NCAL; ++i) {
      gaps[i] = 0.f;
      (void)hipEventElapsedTime(&gaps[i], cal[2 * i], cal[2 * i + 1]);
    }
    for (int i = 0; i < 2 * NCAL; ++i) (void)hipEventDestroy(cal[i]);
    std::sort(gaps, gaps + NCAL);
    const float overhead = gaps[NCAL / 2];
    for (size_t i = 0; i < rec.size(); ++i) {
      (void)hipEventElapsedTime(&rec[i].ms, ev[2 * i], ev[2 * i + 1]);
      rec[i].ms_raw = rec[i].ms;
      rec[i].ms = rec[i].ms > overhead ? rec[i].ms - overhead : 0.f;
      (void)hipEventDestroy(ev[2 * i]);
      (void)hipEventDestroy(ev[2 * i + 1]);
    }
  }
};

struct Plan {
  const casync_engine& e;
  Arena ar;
  Runner& r;
  int B;
  hipStream_t aux = nullptr;          // engine's second stream (null = no overlap)
  int lane = 0;                       // selects the stream-K scratch of this lane's streams
  // Stream-K GEMM remainders only when the batch runs as ONE lane: it cuts small-batch latency
  // (B=1 -23 %, B=4 -14 %, B=8 -6 %) and gains 2.6 % at B=64 single-lane, but with two lanes the
  // other lane's kernels already fill a launch's idle CUs and the split only adds traffic
  // (measured -1.1 % fp32 B=64, -1.9 % bf16 B=512).
  bool stream_k = false;
  bool concurrent = false;            // another lane runs beside this one
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_fork2 = nullptr, ev_skip = nullptr;   // skip_early: fork after the encoder, join in front of up1.0's depthwise
  bool skip_early = false;            // set by run_forward for the trunk and the decoder of a forward alike
  // audio source: either the NCHW windows tensor (`audio`) or, when win_feat is set, the whole
  // HuBERT feature array + per-frame indices, gathered on the device (infer_api.py:99-145)
  const float* win_feat = nullptr;
  int win_steps = 0;
  const int* win_idx = nullptr;
  const CasyncOptions& o = e.eff;     // the options of the forward in progress (run_forward)

  // GEMM wrapper with work accounting (algorithmic bytes: A + C once, W once)
  int dt() const { return e.dtype; }
  const char* tn() const { return e.dtype == DT_BF16 ? "__bf16" : "float"; }
  std::string kname(const char* base, const char* rest = "") const {
    return std::string(base) + "<" + tn() + rest + ">";
  }

  void gemm(const std::string& tag, const void* a, int lda, const std::string& wname, void* c,
            int ldc, long long m, int n, int k, GemmEpilogue epi, const std::string& bname = "",
            double alg_flops = 0) {
    const void* w = e.WG(wname);
    epi.bias = bname == "-" ? nullptr : e.W(bname.empty() ? wname.substr(0, wname.size() - 1) + "b" : bname);   // "-": none
    const double es = dtype_size(dt());
    double bytes = es * ((double)m * k + (double)m * n + (double)n * k);
    if (epi.pre_res) bytes += es * m * n;
    if (epi.post_res) bytes += es * m * n;
    if (epi.acc_out) bytes += 2 * es * m * n;
    if (char* ctx = stream_k ? e.sk_ctx(lane, aux && r.s == aux ? 1 : 0) : nullptr) {
      epi.sk_ws = reinterpret_cast<float*>(ctx);
      epi.sk_cnt = reinterpret_cast<unsigned*>(ctx + kStreamKFloats * 4);
    }
    epi.concurrent = concurrent ? 1 : 0;
    epi.stamps = g_gemm_stamps;
    r.run(tag.c_str(), pw_gemm_kernel_name((int)m, n, k, epi.sk_ws != nullptr, dt(), concurrent, epi.ups_src != nullptr),
          alg_flops > 0 ? alg_flops : 2.0 * m * n * k, bytes,
          [&] { return launch_pw_gemm(a, lda, w, c, ldc, (int)m, n, k, epi, r.s, dt()); });
  }

  // Dense 3x3 conv + bias + LReLU (audio conv3 / conv5, module/unet.py:161-168) as an implicit GEMM:
  // the ring kernel gathers the taps itself (no im2col buffer).
  void conv3x3(const std::string& tag, Ptr in, const std::string& wname, Ptr out, int hw, int cin, int cout,
               int stride, int pad) {
    const int ho = (hw + 2 * pad - 3) / stride + 1;
    const long long m = (long long)B * ho * ho;
    GemmEpilogue ep;
    ep.act = 1;
    ep.bias = e.W(wname.substr(0, wname.size() - 1) + "b");
    ep.concurrent = concurrent ? 1 : 0;
    if (char* ctx = stream_k ? e.sk_ctx(lane, aux && r.s == aux ? 1 : 0) : nullptr) {
      ep.sk_ws = reinterpret_cast<float*>(ctx);
      ep.sk_cnt = reinterpret_cast<unsigned*>(ctx + kStreamKFloats * 4);
    }
    const double es = dtype_size(dt());
    r.run(tag.c_str(), conv3x3_gemm_kernel_name(B, hw, hw, cin, cout, stride, pad, dt(), concurrent, ep.sk_ws != nullptr),
          2.0 * m * cout * 9 * cin, es * ((double)B * hw * hw * cin + (double)m * cout + 9.0 * cin * cout), [&] {
      return launch_conv3x3_gemm(in, e.WG(wname), out, cout, B, hw, hw, cin, cout, stride, pad, ep, r.s, dt());
    });
  }

  // One inverted-residual block.  in: [B*hw_in^2, cin] (ld_in); out: [B*hw_out^2, cout] (ld_out).
  // `ups` (an Up block's first inverted residual with ups_commute): `in` is only the SKIP half of the concatenated
  // input (k_in = cin / 2 channels, expand matrix pw1b), and ups = W1a . lo at the low resolution, whose bilinear x2
  // upsample the expand conv adds before its activation.
  void ir(const IR& b, Ptr in, int ld_in, Ptr out, int ld_out, Ptr e1, Ptr e2,
          const GemmEpilogue* extra = nullptr, Ptr ups = Ptr{}) {
    const std::string p = b.prefix;
    const long long m_in = (long long)B * b.hw_in * b.hw_in, m_out = (long long)B * b.hw_out() * b.hw_out();
    const int k_in = ups.p ? b.cin / 2 : b.cin;
    const std::string w1name = p + (ups.p ? ".pw1b.w" : ".pw1.w");
    if (!extra && !ups.p && ir_is_fused(o, b)) {
      const double flops = 2.0 * (m_in * (double)b.cin * b.cexp() + 9.0 * m_out * b.cexp() +
                                  (double)m_out * b.cexp() * b.cout);
      r.run((p + ".fused").c_str(), ir_fused_kernel_name(b.cin, b.cout, b.stride, dt(), false, b.hw_in, b.hw_in), flops,
            dtype_size(dt()) * (m_in * (double)b.cin + (double)m_out * b.cout), [&] {
        return launch_ir_fused(in, ld_in, e.WG(p + ".pw1.w"), e.W(p + ".pw1.b"), e.W(p + ".dw.w"),
                               e.W(p + ".dw.b"), e.WG(p + ".pw2.w"), e.W(p + ".pw2.b"), out, ld_out, B,
                               b.hw_in, b.hw_in, b.cin, b.cout, b.stride, b.res, r.s, dt());
      });
      return;
    }
    // the un-fused chain parks the expanded tensor in e1 / e2: they must have been sized for this block (the arena's
    // predicate and the plan's are the same functions; this catches the day they are not)
    if (ar.per_frame_of(e2) < (int64_t)b.hw_in * b.hw_in * b.cexp()) {
      casync_set_error("plan: %s un-fused needs %lld expanded elements per frame, the workspace slot holds %lld", b.prefix,
                       (long long)b.hw_in * b.hw_in * b.cexp(), (long long)ar.per_frame_of(e2));
      r.status = CASYNC_ERR_STATE;
      return;
    }
    // (from fuse_dw_min = 12 frames per launch: below that its whole-frame tiles are too few to fill the chip -- B=1 0.92 vs 0.82 ms)
    // (below fuse_dw_deep frames per launch -- round 5 -- the stride-1 blocks take its one-frame tiles with a four-stage ring)
    const bool deep = pw_dw_deep(b.hw_in, B, b.stride, ups.p != nullptr, k_in);
    if (dt() == DT_F32 && o.fuse_dw && (b.hw_in < 40 || o.fuse_dw >= 2) && (deep || B >= (b.hw_in == 40 ? o.fuse_dw_min40 : o.fuse_dw_min)) &&
        pw_dw_supported(b.hw_in, k_in, b.cexp(), b.stride)) {
      // expand GEMM whose output tile is whole frames: the depthwise conv runs on the tile in LDS, E never exists
      // (flops: what this launch executes -- with `ups` the upsampled half was a GEMM at the low resolution)
      r.run((p + ".pw1dw").c_str(), pw_dw_kernel_name(b.hw_in, k_in, B, b.stride),
            2.0 * (m_in * (double)k_in * b.cexp() + 9.0 * m_out * b.cexp()),
            4.0 * (m_in * (double)k_in + (double)b.cexp() * k_in + (double)m_out * b.cexp()), [&] {
        return launch_pw_dw(in, ld_in, e.W(w1name), e.W(p + ".pw1.b"), e.W(p + ".dw.w"), e.W(p + ".dw.b"), e2,
                            b.cexp(), B, b.hw_in, b.stride, k_in, b.cexp(), r.s, ups.p, b.cexp());
      });
    } else if (dt() == DT_BF16 && o.fuse_dw_bf16 && (b.hw_in < 40 || o.fuse_dw_bf16 >= 2) && B >= o.fuse_dw_bf16_min &&
               pw_dw_bf16_supported(b.hw_in, k_in, b.cexp(), b.stride) && (!ups.p || pw_dw_bf16_takes_ups(b.hw_in, b.stride))) {
      // the bf16 engine's counterpart (round 5): 64-channel tiles, bf16 E image in LDS; E never reaches HBM
      r.run((p + ".pw1dw").c_str(), pw_dw_bf16_kernel_name(b.hw_in, b.cexp(), B, b.stride),
            2.0 * (m_in * (double)k_in * b.cexp() + 9.0 * m_out * b.cexp()),
            2.0 * (m_in * (double)k_in + (double)b.cexp() * k_in + (double)m_out * b.cexp()), [&] {
        return launch_pw_dw_bf16(in, ld_in, e.WG(w1name), e.W(p + ".pw1.b"), e.W(p + ".dw.w"), e.W(p + ".dw.b"), e2, b.cexp(), B,
                                 b.hw_in, b.stride, k_in, b.cexp(), r.s, ups.p, b.cexp());
      });
    } else if (dt() == DT_BF16 && ups.p) {
      casync_set_error("plan: %s was planned with the commuted upsample but the bf16 expand + depthwise kernel does not take it", b.prefix);
      r.status = CASYNC_ERR_STATE;
      return;
    } else {
      GemmEpilogue ep1;
      ep1.act = 1;
      if (ups.p) {
        ep1.ups_src = ups;
        ep1.ups_ld = b.cexp();
        ep1.ups_h = ep1.ups_w = b.hw_in;
      }
      gemm(p + ".pw1", in, ld_in, w1name, e1, b.cexp(), m_in, b.cexp(), k_in, ep1, p + ".pw1.b");
      r.run((p + ".dw").c_str(), dw3x3_kernel_name(b.hw_in, b.hw_in, b.cexp(), b.stride, dt()),
            2.0 * 9 * m_out * b.cexp(), dtype_size(dt()) * (double)(m_in + m_out) * b.cexp(), [&] {
        return launch_dw3x3(e1, e.W(p + ".dw.w"), e.W(p + ".dw.b"), e2, B, b.hw_in, b.hw_in, b.cexp(),
                            b.stride, r.s, dt());
      });
    }
    GemmEpilogue ep2;
    if (extra) ep2 = *extra;
    ep2.act = 1;
    if (b.res) {
      ep2.post_res = in;
      ep2.ld_post = ld_in;
    }
    gemm(p + ".pw2", e2, b.cexp(), p + ".pw2.w", out, ld_out, m_out, b.cout, b.cexp(), ep2);
  }

  // att.kv beside the face encoder (kv_early: 1 = in single-lane runs, where the launch chain is the bound; 2 = always;
  // with two lanes the other lane already fills the chip and the extra stream only competes: -0.5 % at B=64)
  bool kv_in_encode = false;          // set by run_forward for all three phases of a forward alike
  bool kv_early() const { return kv_in_encode; }
  void kv_projection() {
    gemm("att.kv", ar[Arena::CATA] + 512, 1024, "att.kv.w", ar[Arena::KV], kBlocks * kKV, (long long)B * 100, kBlocks * kKV, 512,
         GemmEpilogue());
  }

  // Phase 1 of Model.forward: audio encoder || face encoder down to x5 (module/unet.py:315-321).
  void encode(const float* x, const float* audio) {
    using A = Arena;
    Ptr E1 = ar[A::E1], E2 = ar[A::E2], T0 = ar[A::T0];
    // The audio encoder and the face encoder are independent until the fusion MLP; with a
    // second stream they run concurrently and the many small 10x10 / 16x16 audio launches fill
    // CUs the face encoder leaves idle.  Profiling mode keeps everything on one stream.
    const bool overlap = aux && !r.profile;
    hipStream_t main_s = r.s;
    if (overlap) {
      if (hipEventRecord(ev_fork, main_s) != hipSuccess || hipStreamWaitEvent(aux, ev_fork, 0) != hipSuccess) {
        casync_set_error("forward: fork onto the audio stream failed");
        r.status = CASYNC_ERR_HIP;
        return;
      }
      r.s = aux;
    }
    // ---------------- audio encoder (module/unet.py:177-194)
    Ptr AE1 = ar[A::AE1], AE2 = ar[A::AE2];
    if (win_feat)
      r.run("audio.window_gather", kname("audio_window_gather_kernel").c_str(), 0, (4.0 + dtype_size(dt())) * B * 32768,
            [&] { return launch_audio_window_gather(win_feat, win_steps, win_idx, ar[A::A0], B, r.s, dt()); });
    else
      r.run("audio.nchw_to_nhwc", kname("nchw_to_nhwc_kernel").c_str(), 0, (4.0 + dtype_size(dt())) * B * 32768,
            [&] { return launch_nchw_to_nhwc(audio, ar[A::A0], B, 32, 1024, r.s, dt()); });
    ir(kAudio[0], ar[A::A0], 32, ar[A::AC1], 64, AE1, AE2);
    ir(kAudio[1], ar[A::AC1], 64, ar[A::AC2], 128, AE1, AE2);
    conv3x3("audio.conv3", ar[A::AC2], "audio_model.conv3.w", ar[A::AC3], 32, 128, 256, 2, 1);
    ir(kAudio[2], ar[A::AC3], 256, ar[A::AC4], 256, AE1, AE2);
    conv3x3("audio.conv5", ar[A::AC4], "audio_model.conv5.w", ar[A::AC5], 16, 256, 512, 2, 3);
    ir(kAudio[3], ar[A::AC5], 512, ar[A::AC6], 512, AE1, AE2);
    {
      GemmEpilogue bn7;  // relu7(bn7(x + conv(x))) fused behind conv7's residual add
      bn7.aff_s = e.W("audio_model.bn7.s");
      bn7.aff_t = e.W("audio_model.bn7.t");
      ir(kAudio[4], ar[A::AC6], 512, ar[A::CATA] + 512, 1024, AE1, AE2, &bn7);
    }
    // K and V projections of the audio features for all four attention blocks in one GEMM (module/unet.py:202-203,
    // 210, 214).  They depend on the audio branch alone, so they run HERE, on the audio stream beside the face encoder,
    // instead of between the fusion MLP and the first attention block (one launch less on the serial chain: at B=8,
    // the reference's own batch, 24 us of 1.26 ms).
    if (kv_early()) kv_projection();
    if (overlap) {
      if (r.status == CASYNC_OK && hipEventRecord(ev_join, aux) != hipSuccess) r.status = CASYNC_ERR_HIP;
      r.s = main_s;
    }
    // ---------------- face encoder (module/unet.py:315-319)
    r.run("inc", dt() == DT_BF16 && o.inc_mfma ? "inc_bf16_kernel" : kname("inc_kernel").c_str(), 2.0 * B * 25600 * (72 + 108 + 384),
          (double)B * 25600 * (6 * 4 + 32 * dtype_size(dt())), [&] {
      return launch_inc(x, e.W("inc.inconv.0.fused"), ar[A::CAT4] + 32, 64, B, r.s, dt());
    });
    struct Skip { Ptr p; int ld; };
    const Skip sk[5] = {{ar[A::CAT4] + 32, 64},   {ar[A::CAT3] + 64, 128}, {ar[A::CAT2] + 128, 256},
                        {ar[A::CAT1] + 256, 512}, {ar[A::CATA], 1024}};
    for (int i = 0; i < 4; ++i) {
      ir(kDown[i][0], sk[i].p, sk[i].ld, T0, kDown[i][0].cout, E1, E2);
      ir(kDown[i][1], T0, kDown[i][1].cin, sk[i + 1].p, sk[i + 1].ld, E1, E2);
    }
    if (overlap && r.status == CASYNC_OK && hipStreamWaitEvent(main_s, ev_join, 0) != hipSuccess) {
      casync_set_error("forward: join of the audio stream failed");
      r.status = CASYNC_ERR_HIP;
    }
  }

  // Phase 2: the 10x10 trunk -- fusion MLP, four attention blocks, fuse_conv (module/unet.py:323-337).
  // All of it is [B*100, C] GEMMs (+ the attention core and four small depthwise convs).
  void trunk() {
    using A = Arena;
    Ptr E1 = ar[A::E1], E2 = ar[A::E2], T0 = ar[A::T0];
    // ---------------- skip_early: W1b . skip + b of up1.0 / up2.0 depend on the encoder alone.  At small batch the
    // forward is one launch chain, so they leave it: they run on the second stream beside the
    // fusion MLP and the attention blocks (forked behind the MLP instead: B=8 1.169 against 1.165 ms), and the decoder's depthwise kernel adds up(W1a . lo) while it stages its slab
    // (dw3x3_ups_lds_kernel) -- at B=8 27 + 9 us become 12 us on the chain at up1.0 and 44 become 22 at up2.0, for ~20 us that
    // the two GEMMs cost the kernels they run beside: 1.182 -> 1.165 ms, B=1 0.760 -> 0.751 ms.
    if (skip_early) {
      const bool fork = aux && !r.profile;
      hipStream_t main_s = r.s;
      if (fork) {
        if (hipEventRecord(ev_fork2, main_s) != hipSuccess || hipStreamWaitEvent(aux, ev_fork2, 0) != hipSuccess) {
          casync_set_error("forward: fork onto the second stream failed");
          r.status = CASYNC_ERR_HIP;
          return;
        }
        r.s = aux;
      }
      const Arena::Id ep[2] = {A::EP1, A::EP2};
      const Arena::Id cat[2] = {A::CAT1, A::CAT2};
      int hw = 20, c = 256;
      for (int i = 0; i < 2; ++i) {
        const IR& b0 = kUp[i][0];
        const std::string p = b0.prefix;
        gemm(p + ".pw1b", ar[cat[i]] + c, 2 * c, p + ".pw1b.w", ar[ep[i]], b0.cexp(), (long long)B * hw * hw, b0.cexp(), c, GemmEpilogue(),
             p + ".pw1.b");
        hw *= 2;
        c = kUp[i][1].cout;
      }
      if (fork) {
        if (r.status == CASYNC_OK && hipEventRecord(ev_skip, aux) != hipSuccess) r.status = CASYNC_ERR_HIP;
        r.s = main_s;
      }
    }
    // ---------------- fusion (module/unet.py:323-326): tx = bn_tx(cat + mlp(cat))
    const long long M10 = (long long)B * 100;
    Ptr CATA = ar[A::CATA];
    {
      GemmEpilogue ep;
      ep.act = 1;
      gemm("mlp.fc1", CATA, 1024, "mlp_fusion.fc1.w", ar[A::H], 1024, M10, 1024, 1024, ep);
      GemmEpilogue ep2;
      ep2.pre_res = CATA;
      ep2.ld_pre = 1024;
      ep2.pre_scale = e.W("mlp_fusion.fc2.rs");
      gemm("mlp.fc2", ar[A::H], 1024, "mlp_fusion.fc2.w", ar[A::TX], 1024, M10, 1024, 1024, ep2);
    }
    if (!kv_early()) kv_projection();
    // ---------------- attention blocks (module/unet.py:331-336)
    Ptr ox[4] = {ar[A::OX0], ar[A::OX1], ar[A::OX2], ar[A::OX3]};
    Ptr prev = ar[A::TX];
    for (int i = 0; i < kBlocks; ++i) {
      const std::string p = "attention_blocks." + std::to_string(i);
      // ox = p_1(x) lands in columns 0..511 of P1Q (ld 576); q = query_conv(ox) in columns 512..575:
      // either as 64 extra output columns of the same GEMM (weights composed on the host, default) or
      // by the reference's own second GEMM on ox
      Ptr P1 = ar[A::P1Q], Qp = ar[A::P1Q] + 512;
      int ldq = 576;
      if (o.fuse_q) {
        gemm(p + ".p1q", prev, 1024, p + ".p1q.w", P1, 576, M10, 576, 1024, GemmEpilogue(), "",
             2.0 * M10 * (1024.0 * 512 + 512.0 * 64));
      } else {
        gemm(p + ".p1", prev, 1024, p + ".p1.w", P1, 576, M10, 512, 1024, GemmEpilogue());
        gemm(p + ".q", P1, 576, p + ".q.w", ar[A::Q], 64, M10, 64, 512, GemmEpilogue());
        Qp = ar[A::Q];
        ldq = 64;
      }
      Ptr kv = ar[A::KV] + i * kKV;
      r.run((p + ".attn").c_str(), cross_attention_kernel_name(dt()), 2.0 * M10 * 100 * (64 + 512),
            dtype_size(dt()) * (double)M10 * (64 + kKV + 1024), [&] {
        return launch_cross_attention(Qp, ldq, kv, kBlocks * kKV, kv + 64, kBlocks * kKV, P1, 576,
                                      e.W(p + ".gamma"), ar[A::AO], 512, B, r.s, dt());
      });
      GemmEpilogue ep;  // lrelu(bn(b_1(ox) + tx)); kx += ox; last block also lrelu(bn_kx(kx))
      ep.pre_res = ar[A::TX];
      ep.ld_pre = 1024;
      ep.pre_scale = e.W(p + ".b1.rs");
      ep.act = 1;
      ep.acc_in = i == 0 ? (const void*)ar[A::TX] : (const void*)ar[A::KX];
      ep.acc_out = i == kBlocks - 1 ? (void*)ar[A::KXF] : (void*)ar[A::KX];
      ep.ld_acc = 1024;
      if (i == kBlocks - 1) {
        ep.aff_s = e.W("bn_kx.s");
        ep.aff_t = e.W("bn_kx.t");
        ep.aff_on_acc = 1;
      }
      gemm(p + ".b1", ar[A::AO], 512, p + ".b1.w", ox[i], 1024, M10, 1024, 512, ep);
      prev = ox[i];
    }
    // ---------------- fuse_conv (module/unet.py:337)
    ir(kFuse[0], ar[A::KXF], 1024, T0, 512, E1, E2);
    ir(kFuse[1], T0, 512, ar[A::FM], 512, E1, E2);
    ir(kFuse[2], ar[A::FM], 512, T0, 256, E1, E2);
    ir(kFuse[3], T0, 256, ar[A::F], 256, E1, E2);
  }

  // Phase 3: decoder + head (module/unet.py:338-344).
  void decode(float* out) {
    using A = Arena;
    Ptr E1 = ar[A::E1], E2 = ar[A::E2], T0 = ar[A::T0];
    // ---------------- decoder (module/unet.py:338-341)
    Ptr lo = ar[A::F];
    Ptr cat[4] = {ar[A::CAT1], ar[A::CAT2], ar[A::CAT3], ar[A::CAT4]};
    Ptr uo[4] = {ar[A::U1], ar[A::U2], ar[A::U3], ar[A::U4]};
    int hw = 10, c = 256;
    for (int i = 0; i < 4; ++i) {
      const int cc = 2 * c;  // concat width
      const IR& b0 = kUp[i][0];
      const UpMode mode = up_mode(o, b0, dt(), B);
      if (mode == UpMode::CommuteUnfused) {
        // upsample and 1x1 conv commute: the upsampled half of the expand conv runs on the LOW-resolution tensor
        // (a quarter of the pixels: 37.5 % of this GEMM's multiply-adds never happen), the consumer adds its bilinear
        // x2 upsample before the activation.  Every launch books the flops it executes.
        gemm(std::string(b0.prefix) + ".pw1a", lo, c, std::string(b0.prefix) + ".pw1a.w", ar[A::UG], b0.cexp(),
             (long long)B * hw * hw, b0.cexp(), c, GemmEpilogue(), "-", 2.0 * B * hw * hw * (double)c * b0.cexp());
        if (skip_early && i < 2) {
          // the skip half is already there (trunk()): depthwise over LReLU(pre + up(G)), then the project GEMM
          const std::string p = b0.prefix;
          const long long m = (long long)B * 4 * hw * hw;
          if (i == 0 && aux && !r.profile && hipStreamWaitEvent(r.s, ev_skip, 0) != hipSuccess) {
            casync_set_error("forward: join of the second stream failed");
            r.status = CASYNC_ERR_HIP;
            return;
          }
          const float* pre = (const float*)ar[i == 0 ? A::EP1 : A::EP2].p;
          r.run((p + ".dwups").c_str(), dw3x3_ups_kernel_name(2 * hw, 2 * hw, b0.cexp()), 2.0 * 9 * m * b0.cexp(),
                4.0 * ((double)m * b0.cexp() * 2 + (double)m / 4 * b0.cexp()), [&] {
            return launch_dw3x3_ups(pre, (const float*)ar[A::UG].p, b0.cexp(), e.W(p + ".dw.w"), e.W(p + ".dw.b"), (float*)E2.p, B,
                                    2 * hw, 2 * hw, b0.cexp(), r.s);
          });
          GemmEpilogue ep2;
          ep2.act = 1;
          gemm(p + ".pw2", E2, b0.cexp(), p + ".pw2.w", T0, b0.cout, m, b0.cout, b0.cexp(), ep2);
        } else {
          ir(b0, cat[i] + c, cc, T0, b0.cout, E1, E2, nullptr, ar[A::UG]);
        }
      } else if (mode == UpMode::CommuteFused) {
        // the same commutation inside the fused kernel: G = W1a * lo by a GEMM at the low resolution, the fused
        // block runs its expand over the skip half only and adds up(G) chunk by chunk from LDS
        const std::string p = b0.prefix;
        const double m = (double)B * 4 * hw * hw;
        gemm(p + ".pw1a", lo, c, p + ".pw1a.w", ar[A::UG], b0.cexp(), (long long)B * hw * hw, b0.cexp(), c, GemmEpilogue(), "-",
             2.0 * B * hw * hw * (double)c * b0.cexp());
        r.run((p + ".upfused").c_str(), ir_fused_upg_kernel_name(b0.cin, b0.cout),
              2.0 * m * (c * b0.cexp() + 9.0 * b0.cexp() + b0.cexp() * b0.cout),
              4.0 * (m / 4 * b0.cexp() + m * c + m * b0.cout), [&] {
                return launch_ir_fused_upg((const float*)ar[A::UG].p, b0.cexp(), (const float*)(cat[i] + c).p, cc,
                                           e.W(p + ".pw1b.w"), e.W(p + ".pw1.b"), e.W(p + ".dw.w"), e.W(p + ".dw.b"),
                                           e.W(p + ".pw2.w"), e.W(p + ".pw2.b"), (float*)T0.p, b0.cout, B, 2 * hw,
                                           2 * hw, b0.cin, b0.cout, r.s);
              });
      } else if (mode == UpMode::FusedLoad) {
        // bilinear x2 folded into the fused block's input load: up(x) is never materialised
        const std::string p = b0.prefix;
        const double m = (double)B * 4 * hw * hw;
        r.run((p + ".upfused").c_str(), ir_fused_kernel_name(b0.cin, b0.cout, 1, dt(), true, 2 * hw, 2 * hw),
              2.0 * m * (b0.cin * b0.cexp() + 9.0 * b0.cexp() + b0.cexp() * b0.cout),
              dtype_size(dt()) * (m / 4 * c + m * c + m * b0.cout), [&] {
                return launch_ir_fused_up(lo, c, c, cat[i], cc, e.WG(p + ".pw1.w"), e.W(p + ".pw1.b"),
                                          e.W(p + ".dw.w"), e.W(p + ".dw.b"), e.WG(p + ".pw2.w"),
                                          e.W(p + ".pw2.b"), T0, b0.cout, B, 2 * hw, 2 * hw, b0.cin, b0.cout, r.s,
                                          dt());
              });
      } else {
        r.run(("up" + std::to_string(i + 1) + ".bilinear").c_str(), kname("upsample2x_kernel").c_str(), 0,
              dtype_size(dt()) * (double)B * hw * hw * c * 5, [&] {
          return launch_upsample2x(lo, cat[i], cc, B, hw, hw, c, r.s, dt());
        });
        ir(b0, cat[i], cc, T0, b0.cout, E1, E2);
      }
      ir(kUp[i][1], T0, kUp[i][1].cin, uo[i], kUp[i][1].cout, E1, E2);
      lo = uo[i];
      hw *= 2;
      c = kUp[i][1].cout;
    }
    // ---------------- head (module/unet.py:342-344)
    r.run("outc", kname("outc_kernel").c_str(), 2.0 * B * 25600 * 96, (double)B * 25600 * (32 * dtype_size(dt()) + 12), [&] {
      return launch_outc(ar[A::U4], 32, e.W("outc.w"), e.W("outc.b"), out, B, r.s, dt());
    });
  }
};

int check_forward_args(casync_handle h, const float* x, const float* a, float* out, int batch,
                       void* ws, int64_t ws_bytes) {
  CASYNC_REQUIRE(h, "null handle");
  CASYNC_REQUIRE(x && a && out && ws, "forward: null pointer");
  CASYNC_REQUIRE(batch > 0 && batch <= 8192, "forward: batch %d out of range [1, 8192]", batch);
  if (!h->w) {
    casync_set_error("forward: weights not loaded");
    return CASYNC_ERR_STATE;
  }
  if (ws_bytes < Arena::bytes(h->opt, batch, dtype_size(h->dtype))) {
    casync_set_error("forward: workspace %lld B < required %lld B", (long long)ws_bytes,
                     (long long)Arena::bytes(h->opt, batch, dtype_size(h->dtype)));
    return CASYNC_ERR_STATE;
  }
  CASYNC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                     ((uintptr_t)ws % 256) == 0,
                 "forward: pointers must be 16-B aligned (workspace 256-B)");
  return CASYNC_OK;
}

}  // namespace

// ====================================================================== C ABI
namespace {
// Forwards of DIFFERENT handles on one device do not overlap on the GPU (round 6).  Two models forwarding at the same time
// from two host threads -- an fp32 and a bf16 one -- produced sporadic wrong 2 x 16-pixel patches in the fp32 model's up3 / up4
// outputs (tools/experiments/two_models.py: up to 30 of 30 forwards; profiles/r6_two_models.txt).  Traced to packed fp32
// instructions at two sites (the fused Up block's G accumulation, ups_lerp()) returning wrong values while another wave of the
// SIMD executes bf16 matrix instructions; both sites are scalar now and the pair is clean with this gate switched off.  The
// gate stays as a second line -- the operand pattern the hardware trips on was not characterised: a forward starts behind
// the previous forward of any OTHER handle on the device: the gate is held while a forward is enqueued; a forward that finds another
// handle's forward in front of it records an event at the end of THAT forward's stream (everything of it is enqueued by then)
// and waits for it.  A process with one model never records or waits: one uncontended lock per forward (an event behind every
// forward cost 4 us per forward at B = 8).
struct FwdGate {
  std::mutex m;
  casync_handle last = nullptr;     // the handle whose forward was enqueued last on this device ...
  hipStream_t last_stream = nullptr;   // ... and the caller's stream it ended on (its lanes join there before it returns)
};
FwdGate& fwd_gate(int device) {
  static FwdGate gates[64];
  return gates[device < 0 || device >= 64 ? 0 : device];
}
struct StreamPool {
  std::mutex m;
  hipStream_t lane[casync_engine::kMaxLanes] = {};   // [0] unused (caller's stream)
  hipStream_t aux[casync_engine::kMaxLanes] = {};
};
StreamPool& stream_pool(int device) {
  static StreamPool pools[64];
  return pools[device < 0 || device >= 64 ? 0 : device];
}
}  // namespace

extern "C" {

int casync_abi_version(void) { return CASYNC_ABI_VERSION; }
int casync_packed_count(void) { return (int)layout().items.size(); }
const char* casync_packed_name(int i) {
  return (i >= 0 && i < casync_packed_count()) ? layout().items[i].name.c_str() : nullptr;
}
int64_t casync_packed_offset(int i) { return (i >= 0 && i < casync_packed_count()) ? layout().items[i].offset : -1; }
int64_t casync_packed_size(int i) { return (i >= 0 && i < casync_packed_count()) ? layout().items[i].size : -1; }
int64_t casync_packed_total(void) { return layout().total; }
int64_t casync_workspace_bytes(int batch) {
  return batch > 0 ? Arena::bytes(casync_default_options(), batch, 4) : -1;
}
int64_t casync_workspace_bytes_dt(int batch, int dtype) {
  return batch > 0 && (dtype == DT_F32 || dtype == DT_BF16)
             ? Arena::bytes(casync_default_options(), batch, dtype_size(dtype)) : -1;
}
int64_t casync_workspace_bytes_h(casync_handle h, int batch) {
  return h && batch > 0 ? Arena::bytes(h->opt, batch, dtype_size(h->dtype)) : -1;
}

int casync_set_option(casync_handle h, const char* name, int value) {
  int* slot = nullptr;
  const int st = casync_option_ref(h ? h->opt : casync_default_options(), name, &slot);
  if (st != CASYNC_OK) return st;
  if (!strcmp(name, "gemm_cfg") && value >= 5) {
    casync_set_error("option gemm_cfg=%d: tile configurations are 0..4 (-1 = cost model)", value);
    return CASYNC_ERR_ARG;
  }
  *slot = value;
  return CASYNC_OK;
}
int casync_get_option(casync_handle h, const char* name, int* value) {
  CASYNC_REQUIRE(value, "get_option: null out");
  int* slot = nullptr;
  const int st = casync_option_ref(h ? h->opt : casync_default_options(), name, &slot);
  if (st == CASYNC_OK) *value = *slot;
  return st;
}

int casync_create(int device_id, casync_handle* out) { return casync_create_ex(device_id, DT_F32, out); }

int casync_create_ex(int device_id, int dtype, casync_handle* out) {
  CASYNC_REQUIRE(out, "create: null out");
  CASYNC_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "create: dtype %d (0 = fp32, 1 = bf16)", dtype);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    casync_set_error("create: no HIP device visible");
    return CASYNC_ERR_NO_DEVICE;
  }
  CASYNC_REQUIRE(device_id >= 0 && device_id < n, "create: device %d of %d", device_id, n);
  hipDeviceProp_t prop;
  CASYNC_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    casync_set_error("create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    return CASYNC_ERR_NO_DEVICE;
  }
  casync_engine* e = new casync_engine();
  e->device = device_id;
  e->dtype = dtype;
  e->opt = casync_default_options();   // the environment was read once; this handle keeps its own copy
  {
    DeviceGuard guard(device_id);
    const size_t bytes = (size_t)casync_engine::kMaxLanes * 2 * kStreamKBytes;
    hipError_t err = guard.err;
    if (err == hipSuccess) err = hipMalloc((void**)&e->sk, bytes);
    if (err == hipSuccess) err = hipMemset(e->sk, 0, bytes);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) {
      casync_set_error("create: stream-K scratch (%zu bytes): %s", bytes, hipGetErrorString(err));
      if (e->sk) (void)hipFree(e->sk);
      delete e;
      return CASYNC_ERR_HIP;
    }
  }
  *out = e;
  return CASYNC_OK;
}

void casync_destroy(casync_handle h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  {
    FwdGate& gate = fwd_gate(h->device);
    std::lock_guard<std::mutex> lock(gate.m);
    if (gate.last == h) gate.last = nullptr;
    if (h->ev_fwd) { (void)hipEventSynchronize(h->ev_fwd); (void)hipEventDestroy(h->ev_fwd); h->ev_fwd = nullptr; }
  }
  if (h->streams_ready) {
    for (int l = 0; l < casync_engine::kMaxLanes; ++l) {
      if (h->lane_s[l]) (void)hipStreamSynchronize(h->lane_s[l]);   // (the streams belong to the process-wide pool: drained, not destroyed)
      if (h->aux[l]) (void)hipStreamSynchronize(h->aux[l]);
      if (h->ev_fork[l]) (void)hipEventDestroy(h->ev_fork[l]);
      if (h->ev_join[l]) (void)hipEventDestroy(h->ev_join[l]);
      if (h->ev_done[l]) (void)hipEventDestroy(h->ev_done[l]);
      if (h->ev_mid[l]) (void)hipEventDestroy(h->ev_mid[l]);
    }
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_start2) (void)hipEventDestroy(h->ev_start2);
  }
  if (h->owned) (void)hipFree(h->owned);
  if (h->w16) (void)hipFree(h->w16);
  if (h->sk) (void)hipFree(h->sk);
  delete h;
}

// bf16 image of the packed buffer; runs on the engine's device whatever the caller's current device is
static int refresh_bf16_weights(casync_handle h, int64_t n) {
  if (h->dtype != DT_BF16) return CASYNC_OK;
  DeviceGuard guard(h->device);
  CASYNC_CHECK_HIP(guard.err);
  if (!h->w16) CASYNC_CHECK_HIP(hipMalloc((void**)&h->w16, n * sizeof(bf16_t)));
  const long long blocks = (n / 4 + 255) / 256;   // packed total is a multiple of 64 floats
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, h->w, h->w16, (long long)n);
  CASYNC_CHECK_HIP(hipGetLastError());
  CASYNC_CHECK_HIP(hipDeviceSynchronize());   // one-time, at weight load (not on the forward path)
  return CASYNC_OK;
}

int casync_load_weights_host(casync_handle h, const float* packed, int64_t n_floats) {
  CASYNC_REQUIRE(h && packed, "load_weights: null");
  CASYNC_REQUIRE(n_floats == layout().total, "load_weights: %lld floats, layout needs %lld",
                 (long long)n_floats, (long long)layout().total);
  {
    DeviceGuard guard(h->device);
    CASYNC_CHECK_HIP(guard.err);
    if (!h->owned) CASYNC_CHECK_HIP(hipMalloc((void**)&h->owned, n_floats * sizeof(float)));
    CASYNC_CHECK_HIP(hipMemcpy(h->owned, packed, n_floats * sizeof(float), hipMemcpyHostToDevice));
  }
  h->w = h->owned;
  return refresh_bf16_weights(h, n_floats);
}

int casync_load_weights_device(casync_handle h, const float* packed_dev, int64_t n_floats) {
  CASYNC_REQUIRE(h && packed_dev, "load_weights_device: null");
  CASYNC_REQUIRE(n_floats == layout().total, "load_weights_device: %lld floats, layout needs %lld",
                 (long long)n_floats, (long long)layout().total);
  CASYNC_REQUIRE(((uintptr_t)packed_dev % 256) == 0, "load_weights_device: buffer must be 256-B aligned");
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, packed_dev) == hipSuccess && attr.type == hipMemoryTypeDevice)
    CASYNC_REQUIRE(attr.device == h->device, "load_weights_device: buffer lives on device %d, the engine on %d",
                   attr.device, h->device);
  else
    (void)hipGetLastError();
  h->w = packed_dev;
  return refresh_bf16_weights(h, n_floats);
}

// The engine's own streams are PROCESS-WIDE, one set per device, created as they are first needed (lane streams before
// audio streams), and every handle borrows them.  Round 6: with a set per handle a second model in the process ran 7 %
// slower than the first (bf16 B = 512: 45.7 k against 42.8 k frames/s, tools/experiments/bf16_after_fp32.py) -- the runtime
// multiplexes a process's streams onto a few hardware queues, so the second model's lanes landed on queues that already
// carried the first model's and serialised behind each other; `bench.py`'s bf16 leg inside the fp32 run read 42 k for the
// same reason.  Streams only order work: two handles that share them stay correct (every forward orders itself with its
// handle's own events), two forwards running at the same time share the queues -- as they did before.

static int ensure_streams(casync_handle h, int lanes, bool need_aux) {   // caller holds a DeviceGuard for h->device
  if (!h->streams_ready) {
    CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_start, hipEventDisableTiming));
    CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_start2, hipEventDisableTiming));
    for (int l = 0; l < casync_engine::kMaxLanes; ++l) {
      CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_fork[l], hipEventDisableTiming));
      CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_join[l], hipEventDisableTiming));
      CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_done[l], hipEventDisableTiming));
      CASYNC_CHECK_HIP(hipEventCreateWithFlags(&h->ev_mid[l], hipEventDisableTiming));
    }
    h->streams_ready = true;
  }
  bool missing = false;
  for (int l = 0; l < lanes; ++l) missing |= (l && !h->lane_s[l]) || (need_aux && !h->aux[l]);
  if (!missing) return CASYNC_OK;
  StreamPool& pool = stream_pool(h->device);
  std::lock_guard<std::mutex> lock(pool.m);
  for (int l = 1; l < lanes; ++l) {
    if (!pool.lane[l]) CASYNC_CHECK_HIP(hipStreamCreateWithFlags(&pool.lane[l], hipStreamNonBlocking));
    h->lane_s[l] = pool.lane[l];
  }
  if (need_aux)
    for (int l = 0; l < lanes; ++l) {
      if (!pool.aux[l]) CASYNC_CHECK_HIP(hipStreamCreateWithFlags(&pool.aux[l], hipStreamNonBlocking));
      h->aux[l] = pool.aux[l];
    }
  return CASYNC_OK;
}

// Lanes: from 2 x 16 frames up the batch is cut into `lanes` contiguous sub-batches that run
// concurrently (lane 0 on the caller's stream), so one lane's memory-bound kernels and kernel tails
// overlap another lane's MFMA-bound GEMMs; below that one lane with stream-K GEMM remainders
// (tools/latency_sweep.py, one lane vs two: B=8 +19 %, B=24 +4 %, B=32 -3.5 %, B=96 -3.4 %).
// trunk_lanes = 1 ("hybrid"): the lanes join before the fusion MLP, the 10x10 trunk (all GEMMs with
// M = B*100 rows) runs once over the whole batch with stream-K remainders, and the lanes fork again
// for the decoder.

struct FwdArgs {
  const float* x;
  const float* a;        // NCHW windows (null when `feat` is set)
  const float* feat;     // whole HuBERT array + per-frame indices, gathered on the device
  int n_steps;
  const int* idx;
  float* out;
  int batch;
  void* ws;
};

// The launch sequence of one forward.  prof == null: enqueue on the caller's stream + engine streams,
// no host sync.  prof != null: the SAME launches (same lanes, sub-batches, kernels, grids) serialised
// on the caller's stream with an event pair around each; synchronises.
static int run_forward(casync_handle h, const FwdArgs& A, hipStream_t caller, std::vector<casync_kernel_time>* prof) {
  // the options this forward runs under: the handle's, plus -- bf16 engine, from `bf16_plan` frames -- the large-batch plan
  // measured in round 6 (profiles/r6_ab_bf16_plan.txt: B = 512 43.65 -> 45.94 k frames/s, B = 128 +0.8 %): three lanes, the
  // 128x128 GEMM tile on the two-stage ring, the audio encoder on its lane's own stream
  h->eff = h->opt;
  if (h->dtype == DT_BF16 && h->opt.bf16_plan > 0 && A.batch >= h->opt.bf16_plan) {
    h->eff.lanes = h->opt.lanes == 2 ? 3 : h->opt.lanes;     // (a caller who set another lane count keeps it)
    h->eff.gemm_ring128 = 1;
    h->eff.overlap = 0;
  }
  const CasyncOptions& o = h->eff;
  CasyncOptScope scope(&h->eff);
  DeviceGuard guard(h->device);
  CASYNC_CHECK_HIP(guard.err);
  const bool serial = prof != nullptr;
  FwdGate& gate = fwd_gate(h->device);
  std::unique_lock<std::mutex> gate_lock(gate.m);          // held until this forward is enqueued
  static const bool gate_off = getenv("CASYNC_NO_FWD_GATE") != nullptr;   // diagnosis only (tools/experiments/two_models.py)
  if (!gate_off && gate.last && gate.last != h) {
    if (!gate.last->ev_fwd) CASYNC_CHECK_HIP(hipEventCreateWithFlags(&gate.last->ev_fwd, hipEventDisableTiming));
    if (hipEventRecord(gate.last->ev_fwd, gate.last_stream) == hipSuccess) {
      CASYNC_CHECK_HIP(hipStreamWaitEvent(caller, gate.last->ev_fwd, 0));
    } else {                          // (the other forward's stream is gone: its work is done or the device will tell)
      (void)hipGetLastError();
      CASYNC_CHECK_HIP(hipDeviceSynchronize());
    }
  }
  gate.last = h;
  gate.last_stream = caller;
  const bool overlap = o.overlap != 0 && !serial;
  int lanes = o.lanes < 1 ? 1 : (o.lanes > casync_engine::kMaxLanes ? casync_engine::kMaxLanes : o.lanes);
  if (A.batch < kMinLaneBatch * lanes) lanes = 1;  // small batches are latency-bound: cutting them only adds launches
  const bool hybrid = lanes > 1 && o.trunk_lanes == 1;
  if (overlap || lanes > 1) {
    const int st = ensure_streams(h, lanes, overlap);
    if (st != CASYNC_OK) return st;
  }
  const int esz = dtype_size(h->dtype);
  bool forked = false;
  // On an error after work went to the engine's own streams, drain them before returning: the caller
  // will drop the workspace, and torch's allocator only knows about the caller's stream.
  auto fail = [&](int st) {
    if (forked)
      for (int l = 0; l < casync_engine::kMaxLanes; ++l) {
        if (h->lane_s[l]) (void)hipStreamSynchronize(h->lane_s[l]);
        if (h->aux[l]) (void)hipStreamSynchronize(h->aux[l]);
      }
    return st;
  };
#define FWD_HIP(expr)                                                                           \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      casync_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));   \
      return fail(CASYNC_ERR_HIP);                                                              \
    }                                                                                           \
  } while (0)

  int b0s[casync_engine::kMaxLanes], bls[casync_engine::kMaxLanes];
  for (int l = 0, b0 = 0; l < lanes; ++l) {
    bls[l] = A.batch / lanes + (l < A.batch % lanes ? 1 : 0);
    if (lanes == 2 && o.lane_skew > 0 && o.lane_skew < A.batch / 2) bls[l] += l == 0 ? o.lane_skew : -o.lane_skew;
    b0s[l] = b0;
    b0 += bls[l];
  }
  auto lane_stream = [&](int l) { return serial || l == 0 ? caller : h->lane_s[l]; };
  // one phase of one lane (or of the whole batch: l = 0, b0 = 0, bl = batch)
  auto run_phase = [&](int phase, int l, int b0, int bl, bool stream_k, bool concurrent) -> int {
    Runner r;
    r.s = lane_stream(l);
    r.profile = serial;
    Plan p{*h, Arena(o, A.batch), r, bl};
    p.lane = l;
    p.kv_in_encode = o.kv_early >= 2 || (o.kv_early == 1 && lanes == 1);
    p.skip_early = lanes == 1 && h->dtype == DT_F32 && skip_early_batch(o, A.batch) &&
                   up_mode(o, kUp[0][0], DT_F32) == UpMode::CommuteUnfused && up_mode(o, kUp[1][0], DT_F32) == UpMode::CommuteUnfused;
    p.stream_k = stream_k;
    p.concurrent = concurrent;
    p.ar.bind(A.ws, A.batch, esz);
    p.ar.slice(b0);
    if (overlap) {
      p.aux = h->aux[l];
      p.ev_fork = h->ev_fork[l];
      p.ev_join = h->ev_join[l];
      p.ev_fork2 = h->ev_mid[l];     // (lane 0's ev_mid / ev_done are free in a single-lane forward)
      p.ev_skip = h->ev_done[l];
      forked = true;
    }
    if (A.feat) {
      p.win_feat = A.feat;
      p.win_steps = A.n_steps;
      p.win_idx = A.idx + b0;
    }
    if (phase == 0) p.encode(A.x + (size_t)b0 * 6 * 160 * 160, A.feat ? nullptr : A.a + (size_t)b0 * 32 * 32 * 32);
    else if (phase == 1) p.trunk();
    else p.decode(A.out + (size_t)b0 * 3 * 160 * 160);
    r.finish();
    if (prof) prof->insert(prof->end(), r.rec.begin(), r.rec.end());
    return r.status;
  };

  if (lanes > 1 && !serial) {
    FWD_HIP(hipEventRecord(h->ev_start, caller));
    for (int l = 1; l < lanes; ++l) FWD_HIP(hipStreamWaitEvent(h->lane_s[l], h->ev_start, 0));
    forked = true;
  }
  const bool conc = lanes > 1;
  const bool lane_sk = !conc || o.lane_streamk != 0;
  for (int l = 0; l < lanes; ++l)
    if (int st = run_phase(0, l, b0s[l], bls[l], lane_sk, conc)) return fail(st);
  if (hybrid) {
    if (!serial)
      for (int l = 1; l < lanes; ++l) {
        FWD_HIP(hipEventRecord(h->ev_mid[l], h->lane_s[l]));
        FWD_HIP(hipStreamWaitEvent(caller, h->ev_mid[l], 0));
      }
    if (int st = run_phase(1, 0, 0, A.batch, true, false)) return fail(st);
    if (!serial) {
      FWD_HIP(hipEventRecord(h->ev_start2, caller));
      for (int l = 1; l < lanes; ++l) FWD_HIP(hipStreamWaitEvent(h->lane_s[l], h->ev_start2, 0));
    }
  } else {
    for (int l = 0; l < lanes; ++l)
      if (int st = run_phase(1, l, b0s[l], bls[l], lane_sk, conc)) return fail(st);
  }
  for (int l = 0; l < lanes; ++l)
    if (int st = run_phase(2, l, b0s[l], bls[l], lane_sk, conc)) return fail(st);
  if (lanes > 1 && !serial)
    for (int l = 1; l < lanes; ++l) {
      FWD_HIP(hipEventRecord(h->ev_done[l], h->lane_s[l]));
      FWD_HIP(hipStreamWaitEvent(caller, h->ev_done[l], 0));
    }
#undef FWD_HIP
  return CASYNC_OK;
}

int casync_forward(casync_handle h, const float* x, const float* a, float* out, int batch, void* ws,
                   int64_t ws_bytes, casync_stream stream) {
  int st = check_forward_args(h, x, a, out, batch, ws, ws_bytes);
  if (st != CASYNC_OK) return st;
  return run_forward(h, FwdArgs{x, a, nullptr, 0, nullptr, out, batch, ws}, (hipStream_t)stream, nullptr);
}

int casync_forward_windows(casync_handle h, const float* x, const float* features, int n_steps,
                           const int32_t* frame_idx, float* out, int batch, void* ws, int64_t ws_bytes,
                           casync_stream stream) {
  CASYNC_REQUIRE(features && frame_idx && n_steps > 0, "forward_windows: null features / indices");
  CASYNC_REQUIRE(((uintptr_t)features % 16) == 0, "forward_windows: features must be 16-B aligned");
  int st = check_forward_args(h, x, features, out, batch, ws, ws_bytes);
  if (st != CASYNC_OK) return st;
  return run_forward(h, FwdArgs{x, nullptr, features, n_steps, frame_idx, out, batch, ws}, (hipStream_t)stream, nullptr);
}

int casync_profile_forward(casync_handle h, const float* x, const float* a, float* out, int batch,
                           void* ws, int64_t ws_bytes, casync_stream stream, casync_kernel_time* res,
                           int cap) {
  int st = check_forward_args(h, x, a, out, batch, ws, ws_bytes);
  if (st != CASYNC_OK) return st;
  CASYNC_REQUIRE(res && cap > 0, "profile_forward: null result buffer");
  std::vector<casync_kernel_time> rec;
  st = run_forward(h, FwdArgs{x, a, nullptr, 0, nullptr, out, batch, ws}, (hipStream_t)stream, &rec);
  if (st != CASYNC_OK) return st;
  int n = 0;
  for (size_t i = 0; i < rec.size() && n < cap; ++i) res[n++] = rec[i];
  return n;
}

int64_t casync_tap(casync_handle h, const char* name, int batch, void* ws, void* dst, int64_t dst_floats,
                   casync_stream stream) {
  CASYNC_REQUIRE(h && name && ws && dst && batch > 0, "tap: bad args");
  Arena ar(h->opt, batch);
  ar.bind(ws, batch, dtype_size(h->dtype));
  using A = Arena;
  struct T { const char* n; Ptr p; int ld, c, rows; };
  const T taps[] = {
      {"x1", ar[A::CAT4] + 32, 64, 32, 25600},  {"x2", ar[A::CAT3] + 64, 128, 64, 6400},
      {"x3", ar[A::CAT2] + 128, 256, 128, 1600}, {"x4", ar[A::CAT1] + 256, 512, 256, 400},
      {"x5", ar[A::CATA], 1024, 512, 100},      {"a", ar[A::CATA] + 512, 1024, 512, 100},
      {"tx", ar[A::TX], 1024, 1024, 100},       {"kx", ar[A::KXF], 1024, 1024, 100},
      {"att0", ar[A::OX0], 1024, 1024, 100},    {"att1", ar[A::OX1], 1024, 1024, 100},
      {"att2", ar[A::OX2], 1024, 1024, 100},    {"att3", ar[A::OX3], 1024, 1024, 100},
      {"fuse", ar[A::F], 256, 256, 100},        {"u1", ar[A::U1], 128, 128, 400},
      {"u2", ar[A::U2], 64, 64, 1600},          {"u3", ar[A::U3], 32, 32, 6400},
      {"u4", ar[A::U4], 32, 32, 25600},         {"audio_conv2", ar[A::AC2], 128, 128, 1024},
      {"audio_conv3", ar[A::AC3], 256, 256, 256}, {"audio_conv4", ar[A::AC4], 256, 256, 256},
      {"audio_conv5", ar[A::AC5], 512, 512, 100}};
  for (const T& t : taps) {
    if (strcmp(t.n, name) != 0) continue;
    const int64_t per_frame = (int64_t)t.rows * t.c;
    CASYNC_REQUIRE(dst_floats >= per_frame * batch, "tap %s: dst holds %lld floats, need %lld", name,
                   (long long)dst_floats, (long long)(per_frame * batch));
    const size_t es = dtype_size(h->dtype);   // dst receives elements of the engine's storage type
    CASYNC_CHECK_HIP(hipMemcpy2DAsync(dst, (size_t)t.c * es, t.p.p, (size_t)t.ld * es, (size_t)t.c * es,
                                      (size_t)t.rows * batch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return per_frame;
  }
  casync_set_error("tap: unknown name %s", name);
  return CASYNC_ERR_ARG;
}

// ---- single operators ------------------------------------------------------
int casync_debug_gemm_stamps(void* dev_words) {
  g_gemm_stamps = static_cast<unsigned long long*>(dev_words);
  return CASYNC_OK;
}
static thread_local int g_op_dtype = DT_F32;
int casync_op_set_dtype(int dtype) {
  CASYNC_REQUIRE(dtype == DT_F32 || dtype == DT_BF16, "op_set_dtype: %d", dtype);
  g_op_dtype = dtype;
  return CASYNC_OK;
}

int casync_op_pw_gemm(const void* a, int lda, const void* w, const float* bias, void* c, int ldc,
                      int m, int n, int k, int act, const void* pre_res, int ld_pre,
                      const float* pre_scale, const void* post_res, int ld_post, const float* aff_s,
                      const float* aff_t, casync_stream stream) {
  GemmEpilogue e;
  e.bias = bias;
  e.act = act;
  e.pre_res = pre_res;
  e.ld_pre = ld_pre;
  e.pre_scale = pre_scale;
  e.post_res = post_res;
  e.ld_post = ld_post;
  e.aff_s = aff_s;
  e.aff_t = aff_t;
  CASYNC_REQUIRE(!aff_s || aff_t, "pw_gemm: aff_s without aff_t");
  // stream-K scratch for the standalone operator: one region per device, created on first use
  // (callers of the op API run one GEMM at a time per device)
  static char* scratch[64] = {};
  static std::mutex scratch_mu;
  int dev = 0;
  CASYNC_CHECK_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64) {
    std::lock_guard<std::mutex> lock(scratch_mu);
    if (!scratch[dev]) {
      CASYNC_CHECK_HIP(hipMalloc((void**)&scratch[dev], kStreamKBytes));
      CASYNC_CHECK_HIP(hipMemset(scratch[dev], 0, kStreamKBytes));
      CASYNC_CHECK_HIP(hipDeviceSynchronize());
    }
    e.sk_ws = reinterpret_cast<float*>(scratch[dev]);
    e.sk_cnt = reinterpret_cast<unsigned*>(scratch[dev] + kStreamKFloats * 4);
  }
  e.stamps = g_gemm_stamps;
  return launch_pw_gemm(a, lda, w, c, ldc, m, n, k, e, (hipStream_t)stream, g_op_dtype);
}
int casync_op_conv3x3(const void* in, const void* w, const float* bias, void* out, int batch, int h, int wdt,
                      int cin, int cout, int stride, int pad, int act, casync_stream stream) {
  GemmEpilogue e;
  e.bias = bias;
  e.act = act;
  return launch_conv3x3_gemm(in, w, out, cout, batch, h, wdt, cin, cout, stride, pad, e, (hipStream_t)stream, g_op_dtype);
}
int casync_op_dw3x3(const void* in, const float* w, const float* bias, void* out, int batch, int h,
                    int wdt, int c, int stride, casync_stream stream) {
  return launch_dw3x3(in, w, bias, out, batch, h, wdt, c, stride, (hipStream_t)stream, g_op_dtype);
}
int casync_op_dw3x3_ups(const float* pre, const float* g, int ldg, const float* w, const float* bias, float* out, int batch, int h,
                        int wdt, int c, casync_stream stream) {
  CASYNC_REQUIRE(g_op_dtype == DT_F32, "dw3x3_ups: fp32 only");
  return launch_dw3x3_ups(pre, g, ldg, w, bias, out, batch, h, wdt, c, (hipStream_t)stream);
}
int casync_op_pw_dw(const void* a, int lda, const void* w1, const float* b1, const float* wd, const float* bd, void* d, int ldd,
                    int frames, int hw, int stride, int cin, int cexp, const void* ups, int ld_ups, casync_stream stream) {
  if (g_op_dtype == DT_BF16) {
    return launch_pw_dw_bf16(a, lda, w1, b1, wd, bd, d, ldd, frames, hw, stride, cin, cexp, (hipStream_t)stream, ups, ld_ups);
  }
  return launch_pw_dw(a, lda, w1, b1, wd, bd, d, ldd, frames, hw, stride, cin, cexp, (hipStream_t)stream, ups, ld_ups);
}
int casync_op_pw_gemm_ups(const void* a, int lda, const void* w, const float* bias, void* c, int ldc, int m, int n, int k, int act,
                          const void* ups, int ld_ups, int h, int w_, casync_stream stream) {
  GemmEpilogue e;
  e.bias = bias;
  e.act = act;
  e.ups_src = ups;
  e.ups_ld = ld_ups;
  e.ups_h = h;
  e.ups_w = w_;
  return launch_pw_gemm(a, lda, w, c, ldc, m, n, k, e, (hipStream_t)stream, g_op_dtype);
}
int casync_op_ir_fused(const void* in, int ld_in, const void* w1, const float* b1, const float* wd,
                       const float* bd, const void* w2, const float* b2, void* out, int ld_out,
                       int batch, int h, int w, int cin, int cout, int stride, int res,
                       casync_stream stream) {
  return launch_ir_fused(in, ld_in, w1, b1, wd, bd, w2, b2, out, ld_out, batch, h, w, cin, cout, stride,
                         res, (hipStream_t)stream, g_op_dtype);
}
int casync_op_ir_fused_up(const void* lo, int ld_lo, int c_lo, const void* in, int ld_in, const void* w1,
                          const float* b1, const float* wd, const float* bd, const void* w2,
                          const float* b2, void* out, int ld_out, int batch, int h, int w, int cin,
                          int cout, casync_stream stream) {
  return launch_ir_fused_up(lo, ld_lo, c_lo, in, ld_in, w1, b1, wd, bd, w2, b2, out, ld_out, batch, h, w, cin,
                            cout, (hipStream_t)stream, g_op_dtype);
}
int casync_op_ir_fused_upg(const float* g, int ld_g, const float* in, int ld_in, const float* w1b, const float* b1,
                           const float* wd, const float* bd, const float* w2, const float* b2, float* out, int ld_out,
                           int batch, int h, int w, int cin, int cout, casync_stream stream) {
  return launch_ir_fused_upg(g, ld_g, in, ld_in, w1b, b1, wd, bd, w2, b2, out, ld_out, batch, h, w, cin, cout,
                             (hipStream_t)stream);
}
int casync_op_upsample2x(const void* in, void* out, int ldc, int batch, int h, int wdt, int c,
                         casync_stream stream) {
  return launch_upsample2x(in, out, ldc, batch, h, wdt, c, (hipStream_t)stream, g_op_dtype);
}
int casync_op_cross_attention(const void* q, int ldq, const void* k, int ldk, const void* v,
                              int ldv, const void* res, int ld_res, const float* gamma_dev,
                              void* out, int ld_out, int batch, casync_stream stream) {
  return launch_cross_attention(q, ldq, k, ldk, v, ldv, res, ld_res, gamma_dev, out, ld_out, batch,
                                (hipStream_t)stream, g_op_dtype);
}
int casync_op_nchw_to_nhwc(const float* in, void* out, int batch, int c, int hw, casync_stream stream) {
  return launch_nchw_to_nhwc(in, out, batch, c, hw, (hipStream_t)stream, g_op_dtype);
}
int casync_op_inc(const float* x_nchw, const float* packed_inc, void* out, int ldc, int batch,
                  casync_stream stream) {
  return launch_inc(x_nchw, packed_inc, out, ldc, batch, (hipStream_t)stream, g_op_dtype);
}
int casync_op_audio_windows(const float* features_dev, int n_steps, const int32_t* frame_idx_dev, void* windows_dev,
                            int batch, int nhwc, casync_stream stream) {
  if (nhwc) return launch_audio_window_gather(features_dev, n_steps, frame_idx_dev, windows_dev, batch, (hipStream_t)stream,
                                              g_op_dtype);
  return launch_audio_windows_nchw(features_dev, n_steps, frame_idx_dev, (float*)windows_dev, batch, (hipStream_t)stream);
}
int casync_op_crop_to_input(const uint8_t* crops168_dev, float* x_dev, int batch, casync_stream stream) {
  return launch_crop_to_input(crops168_dev, x_dev, batch, (hipStream_t)stream);
}
int casync_op_pred_to_u8(const float* pred_dev, uint8_t* out_dev, int batch, casync_stream stream) {
  return launch_pred_to_u8(pred_dev, out_dev, batch, (hipStream_t)stream);
}
int casync_op_outc(const void* in, int ld_in, const float* w, const float* b, float* out_nchw,
                   int batch, casync_stream stream) {
  return launch_outc(in, ld_in, w, b, out_nchw, batch, (hipStream_t)stream, g_op_dtype);
}

}  // extern "C"
