"""Device-side frame loop around the model, host API.

The reference's ``FrameSynthesizer.process_batch`` (image_infer_v1/tools/frame_synthesizer/
infer_api.py:192-357) does, per frame and on the CPU: crop + ``cv2.resize`` to 168x168, slice /
mask / normalise / transpose / concat into the model input, one ``.cpu()`` per prediction, scale
to uint8, resize back, ``fillPoly`` + ``dilate`` of the jaw polygon and a float64 blend into the frame.
Here the host only computes the crop boxes and landmark transforms; everything between the crop box
and the pasted-back region runs on the GPU (``casync_frame_prepare`` / ``casync_frame_paste_back``,
csrc/frame_ops.hip), with one upload and one download per batch:

* ``submit_batch_device`` / ``PendingBatch.result`` -- the asynchronous pair (one batch can be in flight
  while the next is prepared), ``process_batch_device`` -- both in one call;
* ``crops_to_model_input`` / ``predictions_to_uint8`` / ``audio_windows_device`` -- the pure-indexing pieces on their own;
* ``audio_windows_host`` / ``crop_box`` -- the host-side restatements the device path is tested against.

The OpenCV arithmetic is restated bit for bit from its published algorithms (oracle/frame_ops_oracle.py);
cv2 itself is absent from the build image, so that restatement is "parity unpinned" (tests/test_frame_ops.py)."""
from __future__ import annotations

import os
import threading

import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def crops_to_model_input(crops168: torch.Tensor) -> torch.Tensor:
    """[B,168,168,3] uint8 BGR (the resized crops, infer_api.py:235) on a ROCm device ->
    [B,6,160,160] fp32: channels 0-2 the inner crop / 255, channels 3-5 the same with the mouth
    rectangle blacked out (infer_api.py:238-245)."""
    if crops168.dtype != torch.uint8 or crops168.dim() != 4 or tuple(crops168.shape[1:]) != (168, 168, 3):
        raise RuntimeError(f"crops must be uint8 [B,168,168,3], got {crops168.dtype} {tuple(crops168.shape)}")
    if crops168.device.type != "cuda":
        raise RuntimeError("crops must be on a ROCm device (no CPU fallback)")
    crops168 = crops168.contiguous()
    b = crops168.shape[0]
    x = torch.empty((b, 6, 160, 160), dtype=torch.float32, device=crops168.device)
    if b:
        with torch.cuda.device(crops168.device):
            _lib.check(_lib.load().casync_op_crop_to_input(crops168.data_ptr(), x.data_ptr(), b,
                                                           _stream(crops168.device)), "crop_to_input")
    return x


def predictions_to_uint8(pred: torch.Tensor) -> torch.Tensor:
    """[B,3,160,160] fp32 in (0,1) -> [B,160,160,3] uint8 BGR = ``np.array(pred.transpose(1,2,0) *
    255, dtype=np.uint8)`` (infer_api.py:265-266), so one D2H copy of 77 KB per frame replaces a
    307 KB float copy per frame."""
    if pred.dtype != torch.float32 or pred.dim() != 4 or tuple(pred.shape[1:]) != (3, 160, 160):
        raise RuntimeError(f"pred must be float32 [B,3,160,160], got {pred.dtype} {tuple(pred.shape)}")
    if pred.device.type != "cuda":
        raise RuntimeError("pred must be on a ROCm device (no CPU fallback)")
    pred = pred.contiguous()
    b = pred.shape[0]
    out = torch.empty((b, 160, 160, 3), dtype=torch.uint8, device=pred.device)
    if b:
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().casync_op_pred_to_u8(pred.data_ptr(), out.data_ptr(), b, _stream(pred.device)),
                       "pred_to_u8")
    return out


def audio_windows_device(features: torch.Tensor, frame_indices) -> torch.Tensor:
    """``FrameSynthesizer._get_audio_features`` (infer_api.py:99-145) on the device: ``features`` [T,2,1024] fp32 on a
    ROCm device (the clip's HuBERT features, uploaded once), ``frame_indices`` a sequence of ints or an int32 device
    tensor -> the reference's own return value [B,32,32,32] fp32, on the device (``casync_op_audio_windows``).
    ``Model.forward_windows`` runs the same gather inside the forward; this is the operator on its own."""
    if features.dtype != torch.float32 or features.dim() != 3 or tuple(features.shape[1:]) != (2, 1024):
        raise RuntimeError(f"features must be float32 [T,2,1024], got {features.dtype} {tuple(features.shape)}")
    if features.device.type != "cuda":
        raise RuntimeError("features must be on a ROCm device (no CPU fallback)")
    if features.shape[0] < 1:
        raise RuntimeError("features holds no step")
    features = features.contiguous()
    if torch.is_tensor(frame_indices):
        idx = frame_indices.to(device=features.device, dtype=torch.int32).contiguous()
    else:
        idx = torch.tensor([int(i) for i in frame_indices], dtype=torch.int32).to(features.device)
    b = idx.numel()
    out = torch.empty((b, 32, 32, 32), dtype=torch.float32, device=features.device)
    if b:
        with torch.cuda.device(features.device):
            _lib.check(_lib.load().casync_op_audio_windows(features.data_ptr(), features.shape[0], idx.data_ptr(),
                                                           out.data_ptr(), b, 0, _stream(features.device)),
                       "audio_windows")
    return out


# ---------------------------------------------------------------------------------------------------------------
# process_batch on the device (reference infer_api.py:192-357)
# ---------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402

GEOM_WORDS = 12


def _split64(addr: int):
    """A device address as two int32 words (low, high) of a geometry record."""
    lo, hi = addr & 0xFFFFFFFF, (addr >> 32) & 0xFFFFFFFF
    return (lo - (1 << 32) if lo >= (1 << 31) else lo), (hi - (1 << 32) if hi >= (1 << 31) else hi)


def audio_windows_host(features: np.ndarray, indices) -> np.ndarray:
    """``FrameSynthesizer._get_audio_features`` (infer_api.py:99-145) on the host: [len(indices),32,32,32] fp32.
    Same rule as ``audio_window_gather_kernel`` (closed form of the reference's truncated zero pads: a window
    is ``pl`` zero rows + features[start:start+n0] + zero rows, or all zeros when that is not 16 rows)."""
    n = features.shape[0]
    out = np.zeros((len(indices), 16, 2, 1024), dtype=np.float32)
    for k, idx in enumerate(indices):
        left0, right0 = idx - 8, idx + 8
        pad_left, pad_right = max(0, -left0), max(0, right0 - n)
        left, right = max(left0, 0), min(right0, n)
        start = min(left, n)
        stop = min(right if right >= 0 else max(n + right, 0), n)
        n0 = max(0, stop - start)
        pl = min(pad_left, n0)
        pr = min(pad_right, n0 + pl)
        if n0 + pl + pr == 16:
            out[k, pl:pl + n0] = features[start:start + n0]
    return out.reshape(len(indices), 32, 32, 32)


def crop_box(lms, height: int, img_width: int):
    """The crop box of infer_api.py:206-231: (ymin, ymax, xmin, xmax, width); `width` is xmax - xmin BEFORE the
    border clamps (it is the size the synthesised crop is resized to, :277)."""
    xmin, ymin, xmax = int(lms[1][0]), int(lms[52][1]), int(lms[31][0])
    width = xmax - xmin
    ymax = ymin + width
    if ymax > height:
        diff = ymax - height
        ymax = height
        ymin = max(0, ymin - diff)
    if ymin < 0:
        ymax = min(height, ymax - ymin)
        ymin = 0
    xmin = max(xmin, 0)
    xmax = min(xmax, img_width)
    return ymin, ymax, xmin, xmax, width


_POOL = None


def _host_pool():
    global _POOL
    with _PINNED_LOCK:
        return _host_pool_locked()


def _host_pool_locked():
    global _POOL
    if _POOL is None:
        import concurrent.futures
        n = min(8, max(2, (os.cpu_count() or 2) // 2))
        try:
            n = min(n, max(2, len(os.sched_getaffinity(0))))
        except AttributeError:
            pass
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=n, thread_name_prefix="casync-frames")
    return _POOL


class _Ready:
    """A finished future (copy_frames=False: the frame itself stands in for its copy)."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class PendingBatch:
    """A batch whose GPU work, frame copies and download are in flight (``submit_batch_device``); ``result()``
    waits for them and pastes the blended regions into the copied frames."""

    def __init__(self, copies, boxes, geom, host, done, keep):
        self._copies, self._boxes, self._geom, self._host, self._done, self._keep = copies, boxes, geom, host, done, keep
        self._frames = None

    def result(self):
        if self._frames is None:
            self._done.synchronize()
            host = self._host.numpy()

            def paste(i):
                out = self._copies[i].result()
                if self._geom[i, 4]:
                    ymin, ymax, xmin, xmax, _ = self._boxes[i]
                    h, w = ymax - ymin, xmax - xmin
                    out[ymin:ymax, xmin:xmax] = host[self._geom[i, 0]:self._geom[i, 0] + h * w * 3].reshape(h, w, 3)
                return out

            # (the copies were queued on the pool before these, so a paste never waits for a task behind it)
            frames = [f.result() for f in [_host_pool().submit(paste, i) for i in range(len(self._copies))]]
            _release_pinned(self._host)
            if self._keep is not None:
                _release_pinned(self._keep)
            self._frames, self._host, self._keep = frames, None, None
        return self._frames


# Pinned staging / download buffers.  Crop boxes differ from frame to frame, so the byte count of a batch differs
# from batch to batch: a request is served by the SMALLEST free buffer that is large enough, new buffers come in
# power-of-two size classes (>= 1 MiB; above 64 MiB in 16 MiB steps), and the free list is capped in bytes
# (CASYNC_PINNED_CAP_MB, default 1024): the excess is dropped, largest first, instead of growing page-locked host
# memory (and paying a pin_memory() stall) for every new size a long clip happens to produce.
_PINNED = []          # free pinned uint8 host buffers, any sizes
_PINNED_LOCK = threading.Lock()
_PINNED_CAP = int(os.environ.get("CASYNC_PINNED_CAP_MB", "1024")) << 20


def _size_class(nbytes: int) -> int:
    n = max(nbytes, 1 << 20)
    if n <= (64 << 20):
        return 1 << (n - 1).bit_length()
    return (n + (16 << 20) - 1) // (16 << 20) * (16 << 20)


def _acquire_pinned(nbytes: int) -> torch.Tensor:
    with _PINNED_LOCK:
        best = None
        for i, b in enumerate(_PINNED):
            if b.numel() >= nbytes and (best is None or b.numel() < _PINNED[best].numel()):
                best = i
        if best is not None:
            return _PINNED.pop(best)
    return torch.empty(_size_class(nbytes), dtype=torch.uint8).pin_memory()


def _release_pinned(buf: torch.Tensor) -> None:
    with _PINNED_LOCK:
        _PINNED.append(buf)
        total = sum(b.numel() for b in _PINNED)
        while total > _PINNED_CAP and _PINNED:
            k = max(range(len(_PINNED)), key=lambda i: _PINNED[i].numel())
            total -= _PINNED.pop(k).numel()


def pinned_pool_bytes() -> int:
    """Bytes of page-locked host memory the free list holds right now (tests, diagnostics)."""
    with _PINNED_LOCK:
        return sum(b.numel() for b in _PINNED)


# Frame masks resident on the device.  The reference reads the clip's masks again for every batch and the frame
# loop walks the same frames over and over (ping-pong order), so a caller that names its masks (`mask_keys`, e.g.
# the frame numbers) uploads each of them once; later batches only reference the device copies.  uint8 masks
# (= imread values, standing for value / 255) are a quarter of the float32 the reference holds on the host.
_MASK_CACHE_CAP = int(os.environ.get("CASYNC_MASK_CACHE_MB", "4096")) << 20


def _mask_cache(net):
    c = getattr(net, "_mask_cache", None)
    if c is None:
        import collections
        c = net._mask_cache = collections.OrderedDict()
        net._mask_cache_bytes = 0
    return c


def _mask_kind(mask: np.ndarray) -> int:
    if mask.ndim != 2:
        raise ValueError("a frame mask must be a 2-D array")
    if mask.dtype == np.uint8:
        return 1
    return 0


def process_batch_device(net, batch_images, batch_landmarks, batch_masks, *, windows=None, features=None,
                         frame_indices=None, copy_frames=True, mask_keys=None):
    """``FrameSynthesizer.process_batch`` with everything between the crop box and the pasted-back frame on the
    GPU: ONE upload (the crop regions of all frames, concatenated), cv2.resize -> model input -> ``net`` ->
    uint8 -> resize back -> polygon mask -> dilate -> blend, ONE download (the blended regions).

    ``windows``: device tensor [B,32,32,32] (host-built HuBERT windows, the reference's own calling form), or
    ``features`` [T,2,1024] on the device + ``frame_indices`` (windows gathered on the device).
    Returns the list of synthesised frames (copies; the inputs are not modified, like infer_api.py:201)."""
    return submit_batch_device(net, batch_images, batch_landmarks, batch_masks, windows=windows, features=features,
                               frame_indices=frame_indices, copy_frames=copy_frames, mask_keys=mask_keys).result()


def submit_batch_device(net, batch_images, batch_landmarks, batch_masks, *, windows=None, features=None,
                        frame_indices=None, copy_frames=True, mask_keys=None) -> PendingBatch:
    """The asynchronous half of ``process_batch_device``: host geometry, upload, every launch and the download
    are enqueued, the per-frame copies run on the host pool; nothing here waits for the GPU.  A caller that
    submits batch k+1 before taking ``result()`` of batch k overlaps its host work with the GPU.

    ``copy_frames=False`` gives up the reference's "the inputs are not modified" (infer_api.py:201): the blended
    regions are pasted into ``batch_images`` themselves, which saves a 6 MB copy per 1080p frame -- the frame copies
    are what bounds the end-to-end rate (400 MB per 64 frames against 57 MB of regions up and down).

    Masks: float32 in [0, 1] (the reference's ``imread(...).astype(np.float32) / 255.0``, infer_api.py:68-70) or the
    uint8 image itself (the device divides by 255 in float32: same bits, a quarter of the bytes).  ``mask_keys``: one
    hashable per frame (e.g. the frame number) naming its mask; a named mask is uploaded once and stays on the device
    (``net``'s cache, CASYNC_MASK_CACHE_MB), so a clip's masks do not travel with every batch -- a whole-frame float32
    mask is 8.3 MB per 1080p frame against 0.5-0.9 MB of crop region."""
    lib = _lib.load()
    dev = net._device()
    if dev.type != "cuda":
        raise RuntimeError("process_batch_device needs the model on a ROCm device (no CPU fallback)")
    B = len(batch_images)
    if B == 0:
        done = torch.cuda.Event()
        return PendingBatch([], [], np.zeros((0, GEOM_WORDS), dtype=np.int32), _acquire_pinned(1), done, None)
    geom = np.zeros((B, GEOM_WORDS), dtype=np.int32)
    pts = np.zeros((B, 33, 2), dtype=np.int32)
    boxes, regions, fmasks = [], [], []      # fmasks: (frame, array) to upload with this batch
    held_masks: list = []                    # cached device masks this batch reads (kept alive until it is enqueued)
    reg_off = synth_off = mask_off = fmask_bytes = 0
    cache = _mask_cache(net) if mask_keys is not None else None
    if mask_keys is not None and len(mask_keys) != B:
        raise ValueError("mask_keys must name every frame of the batch")
    for i, (img, lms, mask) in enumerate(zip(batch_images, batch_landmarks, batch_masks)):
        if img is None or img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise ValueError(f"frame {i}: expected a uint8 HxWx3 image")
        ymin, ymax, xmin, xmax, width = crop_box(lms, img.shape[0], img.shape[1])
        h, w = ymax - ymin, xmax - xmin
        if h <= 0 or w <= 0 or width <= 0:
            raise ValueError(f"frame {i}: empty crop box {(ymin, ymax, xmin, xmax)} (cv2.resize would fail)")
        boxes.append((ymin, ymax, xmin, xmax, width))
        regions.append(img[ymin:ymax, xmin:xmax])
        fp = np.asarray(lms[:33], dtype=np.float64).copy()          # infer_api.py:281-289
        fp[:, 0] -= xmin
        fp[:, 1] -= ymin
        fp[:, 0] *= width / (xmax - xmin)
        fp[:, 1] *= width / (ymax - ymin)
        pts[i] = fp.astype(np.int32)
        valid = int(width == h and width == w)
        g = geom[i]
        g[0], g[1], g[2], g[3], g[4], g[5], g[6] = reg_off, h, w, width, valid, synth_off, mask_off
        if mask is not None:
            kind = _mask_kind(mask)
            g[7], g[8], g[9] = kind, mask.shape[0], mask.shape[1]
            key = (mask_keys[i], kind, mask.shape) if cache is not None and mask_keys[i] is not None else None
            hit = cache.get(key) if key is not None else None
            if hit is not None:
                cache.move_to_end(key)
                held_masks.append(hit)           # the geometry record holds its raw address: see the eviction below
                g[10], g[11] = _split64(hit.data_ptr())
            else:
                m = np.ascontiguousarray(mask, dtype=np.uint8 if kind else np.float32)
                fmasks.append((i, key, m, fmask_bytes))
                fmask_bytes += (m.nbytes + 15) & ~15
        else:
            g[7] = -1
        reg_off += h * w * 3
        mask_off += h * w
        if valid:
            synth_off += width * width * 3
        if reg_off >= 2 ** 31:
            raise ValueError("batch too large for 32-bit region offsets")
    max_h, max_w = int(geom[:, 1].max()), int(geom[:, 2].max())
    max_width = int((geom[:, 3] * geom[:, 4]).max())
    stream = _stream(dev)
    # ONE pinned staging buffer, ONE upload: [crop regions | geom | pts | face masks], 16-B aligned parts
    al = lambda n: (n + 15) & ~15
    o_geom = al(reg_off)
    o_pts = o_geom + al(geom.nbytes)
    o_fm = o_pts + al(pts.nbytes)
    total = o_fm + fmask_bytes
    stage = _acquire_pinned(total)
    st = stage.numpy()
    pool = _host_pool()
    with torch.cuda.device(dev):
        staged = torch.empty(total, dtype=torch.uint8, device=dev)   # allocated first: the masks' device addresses
    for i, key, m, off in fmasks:                                    # go into the geometry records
        g = geom[i]
        g[10], g[11] = _split64(staged.data_ptr() + o_fm + off)

    def fill(dst_off, r):                       # one frame's crop region into its place in the pinned buffer
        n = r.shape[0] * r.shape[1] * 3
        st[dst_off:dst_off + n].reshape(r.shape)[...] = r

    fills = [pool.submit(fill, int(geom[i, 0]), r) for i, r in enumerate(regions)]
    # The reference returns NEW frames (frame = img.copy(), infer_api.py:201).  64 copies of a 1080p frame are
    # ~30 ms on one core, four times the GPU work of the batch, so they run on the same small thread pool (NumPy
    # copies release the GIL) behind the region copies, while the GPU is busy; only the paste waits for them.
    copies = [pool.submit(np.copy, img) for img in batch_images] if copy_frames else [_Ready(img) for img in batch_images]
    st[o_geom:o_geom + geom.nbytes] = geom.reshape(-1).view(np.uint8)
    st[o_pts:o_pts + pts.nbytes] = pts.reshape(-1).view(np.uint8)
    for i, key, m, off in fmasks:
        st[o_fm + off:o_fm + off + m.nbytes] = m.reshape(-1).view(np.uint8)
    for f in fills:
        f.result()
    with torch.cuda.device(dev):
        staged.copy_(stage[:total], non_blocking=True)
        base = staged.data_ptr()
        p_regions, p_geom, p_pts = base, base + o_geom, base + o_pts
        for i, key, m, off in fmasks:     # named masks move into allocations of their own and stay (device-side copy)
            if key is not None:
                keep = staged[o_fm + off:o_fm + off + m.nbytes].clone()
                old = cache.pop(key, None)
                if old is not None:
                    net._mask_cache_bytes -= old.numel()
                cache[key] = keep
                net._mask_cache_bytes += keep.numel()
        # Eviction only drops the CACHE's reference.  A mask this batch hit stays alive through `held_masks` until every
        # kernel of the batch is enqueued; torch's allocator is stream-ordered, so a block freed after that can only be
        # reused by work that runs behind those kernels (evicting first let `crops` / `x` land on a mask the
        # blend kernel was still going to read).
        while cache is not None and net._mask_cache_bytes > _MASK_CACHE_CAP and len(cache) > 1:
            net._mask_cache_bytes -= cache.popitem(last=False)[1].numel()
        crops = torch.empty((B, 168, 168, 3), dtype=torch.uint8, device=dev)
        x = torch.empty((B, 6, 160, 160), dtype=torch.float32, device=dev)
        _lib.check(lib.casync_frame_prepare(p_regions, p_geom, B, crops.data_ptr(), x.data_ptr(),
                                            stream), "casync_frame_prepare")
        if windows is not None:
            pred = net(x, windows)
        else:
            pred = net.forward_windows(x, features, frame_indices)
        synth = torch.empty(max(synth_off, 16), dtype=torch.uint8, device=dev)
        mask_a = torch.empty(mask_off, dtype=torch.uint8, device=dev)
        mask_b = torch.empty(mask_off, dtype=torch.uint8, device=dev)
        area = torch.empty(B, dtype=torch.int32, device=dev)
        out_regions = torch.empty(reg_off, dtype=torch.uint8, device=dev)
        _lib.check(lib.casync_frame_paste_back(
            p_regions, p_geom, p_pts, crops.data_ptr(), pred.data_ptr(), B, max_h, max_w,
            max_width, mask_off, synth.data_ptr(), mask_a.data_ptr(), mask_b.data_ptr(), area.data_ptr(),
            out_regions.data_ptr(), stream), "casync_frame_paste_back")
        host = _acquire_pinned(reg_off)                      # the ONE download of the batch, into pinned memory
        host[:reg_off].copy_(out_regions, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
    # device tensors may be dropped here (torch's allocator is stream-ordered on the current stream); the two
    # pinned buffers are ours until the event has passed, i.e. until result()
    del held_masks
    return PendingBatch(copies, boxes, geom, host, done, stage)
