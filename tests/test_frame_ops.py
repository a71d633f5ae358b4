"""Frame loop around the model: the OpenCV-restating oracle's own properties and the
frame-walk / iterator contract on the CPU; the device process_batch == oracle, bit for bit, on the GPU.

cv2 is absent from this image, so the oracle (oracle/frame_ops_oracle.py) is PARITY UNPINNED against
the real library; what these tests pin is (a) properties any correct restatement must have and (b) that
the HIP kernels compute exactly what the restatement computes."""
import os
import random

import numpy as np
import pytest
import torch

from frame_data import make_frames, write_dataset
from oracle import frame_loop_oracle, frame_ops_oracle as fo


# ------------------------------------------------------------------ the pin a maintainer WITH cv2 can run
def test_oracle_against_cv2():
    """Not runnable in this image (no cv2): compares every restated OpenCV call with the real one on random data.
    Until it has run somewhere, the cv2 half of rows f1 is "parity unpinned"."""
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(5)
    for (h, w) in [(211, 190), (384, 384), (97, 300), (168, 168), (336, 336), (84, 84)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(fo.resize_linear_u8(src, (168, 168)), cv2.resize(src, (168, 168)))
        small = rng.integers(0, 256, (160, 160, 3), dtype=np.uint8)
        assert np.array_equal(fo.resize_linear_u8(small, (w, w)), cv2.resize(small, (w, w)))
        m = rng.random((64, 48)).astype(np.float32)
        assert np.array_equal(fo.resize_linear_f32(m, (w, h)), cv2.resize(m, (w, h)))
        pts = rng.integers(-20, max(h, w) + 20, (33, 2)).astype(np.int32)
        want = np.zeros((h, w), np.uint8)
        cv2.fillPoly(want, [pts], 255)
        assert np.array_equal(fo.fill_poly((h, w), pts), want)
        e = int(rng.integers(1, 40))
        assert np.array_equal(fo.dilate_square(want, e), cv2.dilate(want, np.ones((2 * e + 1, 2 * e + 1), np.uint8), iterations=1))


# ------------------------------------------------------------------ oracle properties (CPU)
def test_resize_u8_basic_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(fo.resize_linear_u8(img, (53, 37)), img)                 # same size: copy
    const = np.full((20, 31, 3), 137, dtype=np.uint8)
    assert (fo.resize_linear_u8(const, (168, 168)) == 137).all()                   # weights sum to 2048 -> exact
    assert (fo.resize_linear_u8(const, (7, 5)) == 137).all()
    big = fo.resize_linear_u8(img, (168, 168))
    assert big.shape == (168, 168, 3) and big.dtype == np.uint8
    assert big.min() >= img.min() and big.max() <= img.max()                       # convex combination
    # exact 2x decimation is INTER_AREA in cv::resize: the rounded mean of each 2x2 block
    src = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    half = fo.resize_linear_u8(src, (32, 24))
    s = src.astype(np.int32)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # upscaling by an integer factor keeps the corner pixels and is monotone along a ramp
    ramp = np.tile(np.arange(0, 250, 10, dtype=np.uint8)[None, :, None], (4, 1, 3))
    up = fo.resize_linear_u8(ramp, (100, 16))
    assert up[0, 0, 0] == 0 and up[-1, -1, 0] == 240 and (np.diff(up[0, :, 0].astype(int)) >= 0).all()


def test_resize_f32_matches_direct_formula():
    rng = np.random.default_rng(1)
    m = rng.random((9, 13), dtype=np.float32)
    out = fo.resize_linear_f32(m, (40, 31))
    assert out.shape == (31, 40) and out.dtype == np.float32
    assert out.min() >= m.min() - 1e-6 and out.max() <= m.max() + 1e-6
    assert np.array_equal(fo.resize_linear_f32(m, (13, 9)), m)


def test_fill_poly_shapes():
    sq = fo.fill_poly((40, 50), np.array([[10, 5], [30, 5], [30, 25], [10, 25]], np.int32))
    want = np.zeros((40, 50), np.uint8)
    want[5:26, 10:31] = 255                                                          # edges included (cv2 draws them)
    assert np.array_equal(sq, want)
    tri = fo.fill_poly((60, 60), np.array([[5, 5], [50, 10], [20, 55]], np.int32))
    assert abs(int((tri > 0).sum()) - 0.5 * abs((50 - 5) * (55 - 5) - (20 - 5) * (10 - 5))) < 120   # shoelace area + boundary
    assert tri[5, 5] == 255 and tri[10, 50] == 255 and tri[55, 20] == 255             # vertices are drawn
    # partly outside the image: clipped, no wrap-around
    out = fo.fill_poly((30, 30), np.array([[-10, -5], [40, 10], [15, 45]], np.int32))
    assert out.shape == (30, 30) and out[29, 15] == 255 and out[0, 29] == 0
    # concave (a "U"): the notch stays empty
    u = fo.fill_poly((40, 40), np.array([[5, 5], [15, 5], [15, 25], [25, 25], [25, 5], [35, 5], [35, 35], [5, 35]], np.int32))
    assert u[10, 20] == 0 and u[30, 20] == 255 and u[10, 10] == 255


def test_dilate_is_a_window_maximum():
    rng = np.random.default_rng(2)
    m = (rng.random((23, 31)) > 0.93).astype(np.uint8) * 255
    for e in (1, 3, 7):
        d = fo.dilate_square(m, e)
        brute = np.zeros_like(m)
        for y in range(23):
            for x in range(31):
                brute[y, x] = m[max(0, y - e):y + e + 1, max(0, x - e):x + e + 1].max()
        assert np.array_equal(d, brute)
    assert fo.expand_pixels(0) == 1 and fo.expand_pixels(31416) == 15


def test_paste_back_leaves_the_frame_outside_the_mask_untouched():
    imgs, lms, _ = make_frames(2, 300, 400, seed=5)
    box, crop = fo.prepare_frame(imgs[0], lms[0])
    pred = np.random.default_rng(3).integers(0, 256, (160, 160, 3), dtype=np.uint8)
    out = fo.paste_back(imgs[0], lms[0], None, box, crop, pred)
    ymin, ymax, xmin, xmax, _ = box
    changed = np.any(out != imgs[0], axis=2)
    assert changed.any() and not changed[:ymin].any() and not changed[ymax:].any()
    assert not changed[:, :xmin].any() and not changed[:, xmax:].any()
    # a crop box pushed over the frame border changes its shape: the reference returns the frame as it was
    lm = lms[1].copy()
    lm[:, 0] += 400 - lm[31, 0] + 10                                # right edge beyond the image
    box2, crop2 = fo.prepare_frame(imgs[1], lm)
    assert np.array_equal(fo.paste_back(imgs[1], lm, None, box2, crop2, pred), imgs[1])


# ------------------------------------------------------------------ frame walk + iterator contract (CPU)
def _stub_net():
    class Stub:
        def eval(self):
            return self
    return Stub()


def test_frame_walk_matches_the_literal_restatement(tmp_path):
    """FrameSynthesizer's walk (frame_walk.PingPongWalk: one phase counter, whole requests by array arithmetic) against the
    literal restatement of the reference's loop (oracle FrameWalk): same frames for the same random draws, over clips of
    2..40 frames, ragged request sizes, many re-draws and turns; and the same run bookkeeping."""
    from calipsync_amd.frame_synth import FrameSynthesizer
    from calipsync_amd.frame_walk import PingPongWalk
    write_dataset(str(tmp_path), 40, 32, 48)
    for seed in (0, 1, 7, 123):
        fs = FrameSynthesizer(None, str(tmp_path), device="cpu", batch_size=8, seed=seed, net=_stub_net())
        walk = frame_loop_oracle.FrameWalk(40, random.Random(seed))
        for need in (8, 8, 3, 8, 8, 8, 40, 1, 8, 8, 5, 64, 0, 2, 200):
            got = fs._generate_frame_sequence(need)
            assert got == walk.generate(need)
            assert len(got) == need and all(0 <= p < 40 for p in got)
            assert all(abs(b - a) <= 1 for a, b in zip(got, got[1:]))          # a walk: +-1, or a bounce (+-1 too)
            assert (fs.current_frame_position, fs.processed_frame_count, fs.target_frame_count) == \
                   (walk.current_frame_position, walk.processed_frame_count, walk.target_frame_count)
    for total in (2, 3, 4, 5, 17, 40):
        for seed in range(6):
            mine, ref = PingPongWalk(total, random.Random(seed)), frame_loop_oracle.FrameWalk(total, random.Random(seed))
            sizes = random.Random(100 + seed)
            for _ in range(60):
                n = sizes.choice([1, 2, 3, 8, 8, 8, 16, 5 * total])
                assert mine.take(n) == ref.generate(n), (total, seed)
                assert mine.position == ref.current_frame_position
                if 0 < mine.position < total - 1:                              # at the two ends both directions are the same walk
                    assert mine.ascending == (ref.current_direction == 1)
    still = PingPongWalk(1, random.Random(0))                                  # the reference's loop never ends on a one-frame clip
    assert still.take(5) == [0] * 5 and still.position == 0


def test_iterator_contract_without_sync(tmp_path):
    """index / physical_index / variable last batch (infer_api.py:359-428), frames passed through."""
    from calipsync_amd.frame_synth import FrameSynthesizer
    imgs, _ = write_dataset(str(tmp_path), 12, 32, 48)
    fs = FrameSynthesizer(None, str(tmp_path), device="cpu", batch_size=5, seed=3, net=_stub_net())
    feats = np.zeros((13, 2, 1024), np.float32)                       # 13 = 5 + 5 + 3
    out = list(fs.iterate_synthesized_frames(feats, start_frame_idx=100, is_generate_sync_frame=False))
    assert [o["index"] for o in out] == list(range(100, 113))
    walk = frame_loop_oracle.FrameWalk(12, random.Random(3))
    want = walk.generate(5) + walk.generate(5) + walk.generate(3)
    assert [o["physical_index"] for o in out] == want
    for o in out:
        assert np.array_equal(o["frame"], imgs[o["physical_index"]])
    assert fs.total_frames == 12 and fs.last_logical_index == 112


def test_host_audio_windows_equal_the_literal_restatement():
    from calipsync_amd import frame_loop
    f = np.random.default_rng(0).standard_normal((19, 2, 1024)).astype(np.float32)
    idx = list(range(-12, 40))
    assert np.array_equal(frame_loop.audio_windows_host(f, idx), frame_loop_oracle.get_audio_features(f, idx))
    f5 = f[:5]
    assert np.array_equal(frame_loop.audio_windows_host(f5, idx), frame_loop_oracle.get_audio_features(f5, idx))


def test_crop_box_host_equals_oracle():
    from calipsync_amd import frame_loop
    rng = np.random.default_rng(4)
    for _ in range(200):
        lms = rng.uniform(-50, 500, (110, 2))
        assert frame_loop.crop_box(lms, 300, 400) == fo.crop_box(lms, 300, 400)


# ------------------------------------------------------------------ device == oracle (GPU)
@pytest.fixture(scope="module")
def gpu_net(recipe_sd):
    from calipsync_amd.unet import Model
    m = Model(6, "hubert").to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    return m.eval()


@pytest.mark.gpu
def test_device_resize168_is_bit_exact():
    from calipsync_amd import _lib, frame_loop
    lib = _lib.load()
    rng = np.random.default_rng(11)
    shapes = [(168, 168), (336, 336), (211, 211), (97, 97), (400, 400), (300, 123), (50, 333), (169, 167), (1, 1), (2, 700)]
    regions = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    geom = np.zeros((len(shapes), frame_loop.GEOM_WORDS), np.int32)
    off = 0
    for i, r in enumerate(regions):
        geom[i, :3] = (off, r.shape[0], r.shape[1])
        off += r.size
    dev = torch.device("cuda:0")
    reg_d = torch.from_numpy(np.concatenate([r.reshape(-1) for r in regions])).to(dev)
    geom_d = torch.from_numpy(geom.reshape(-1)).to(dev)
    crops = torch.empty((len(shapes), 168, 168, 3), dtype=torch.uint8, device=dev)
    _lib.check(lib.casync_frame_prepare(reg_d.data_ptr(), geom_d.data_ptr(), len(shapes), crops.data_ptr(), 0,
                                        torch.cuda.current_stream().cuda_stream), "prepare")
    got = crops.cpu().numpy()
    for i, r in enumerate(regions):
        want = fo.resize_linear_u8(r, (168, 168))
        assert np.array_equal(got[i], want), (shapes[i], int(np.abs(got[i].astype(int) - want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("with_masks", [False, True])
def test_device_process_batch_equals_oracle(gpu_net, with_masks):
    """Whole process_batch: device pipeline vs the CPU restatement around the SAME model predictions."""
    from calipsync_amd import frame_loop
    imgs, lms, masks = make_frames(7, 420, 560, seed=21, with_masks=with_masks)
    lms[3] = lms[3].copy()
    lms[3][:, 0] += 560 - lms[3][31, 0] + 25            # frame 3: crop box clamped at the right border -> unchanged frame
    lms[5] = lms[5].copy()
    lms[5][:, 1] += 420 - (lms[5][52, 1] + (lms[5][31, 0] - lms[5][1, 0])) + 12   # frame 5: box pushed past the bottom
    windows = np.random.default_rng(8).standard_normal((7, 32, 32, 32)).astype(np.float32)
    wd = torch.from_numpy(windows).cuda()

    def predict(x):
        return gpu_net(torch.from_numpy(x).cuda(), wd).cpu().numpy()
    want = fo.process_batch(imgs, lms, masks, predict)
    got = frame_loop.process_batch_device(gpu_net, imgs, lms, masks, windows=wd)
    assert len(got) == 7
    for i in range(7):
        assert got[i].shape == imgs[i].shape and got[i].dtype == np.uint8
        assert np.array_equal(got[i], want[i]), (i, int(np.abs(got[i].astype(int) - want[i].astype(int)).max()))
    assert np.array_equal(got[3], imgs[3])                                     # the shape-mismatch fallback
    assert any(not np.array_equal(g, im) for g, im in zip(got, imgs))          # and the others really changed
    assert all(np.array_equal(a, b) for a, b in zip(imgs, make_frames(7, 420, 560, seed=21, with_masks=with_masks)[0]))  # inputs intact
    # copy_frames=False: same pixels, pasted into the caller's own arrays
    mine = [im.copy() for im in imgs]
    got2 = frame_loop.process_batch_device(gpu_net, mine, lms, masks, windows=wd, copy_frames=False)
    assert all(g is m for g, m in zip(got2, mine)) and all(np.array_equal(g, w) for g, w in zip(got2, want))


@pytest.mark.gpu
def test_uint8_masks_named_and_resident_equal_float_masks(gpu_net):
    """uint8 masks (the imread image; the device divides by 255 in float32) give the pixels of the reference's float32
    masks, and named masks are uploaded once: the second batch finds them in the model's device cache."""
    from calipsync_amd import frame_loop
    rng = np.random.default_rng(31)
    imgs, lms, _ = make_frames(5, 300, 400, seed=13)
    u8 = [rng.integers(0, 256, (300, 400) if i % 2 else (100, 150), dtype=np.uint8) for i in range(5)]
    u8[2] = None
    f32 = [None if m is None else m.astype(np.float32) / 255.0 for m in u8]        # infer_api.py:68-70
    wd = torch.from_numpy(rng.standard_normal((5, 32, 32, 32)).astype(np.float32)).cuda()
    want = frame_loop.process_batch_device(gpu_net, imgs, lms, f32, windows=wd)
    gpu_net.__dict__.pop("_mask_cache", None)
    keys = [("clip", i) for i in range(5)]
    got = frame_loop.process_batch_device(gpu_net, imgs, lms, u8, windows=wd, mask_keys=keys)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    cache = gpu_net._mask_cache
    assert len(cache) == 4 and gpu_net._mask_cache_bytes == sum(m.size for m in u8 if m is not None)
    ptrs = {k: v.data_ptr() for k, v in cache.items()}
    again = frame_loop.process_batch_device(gpu_net, imgs, lms, u8, windows=wd, mask_keys=keys)
    assert all(np.array_equal(a, b) for a, b in zip(again, want))
    assert {k: v.data_ptr() for k, v in gpu_net._mask_cache.items()} == ptrs          # nothing was uploaded again


@pytest.mark.gpu
def test_mask_cache_eviction_keeps_this_batch_masks_alive(gpu_net, monkeypatch):
    """with a cap smaller than one batch's masks the LRU evicts masks the batch in hand still reads through raw
    addresses in its geometry records.  The batch must keep them alive: the pixels equal the un-cached run, batch after
    batch, and the byte count follows overwritten keys."""
    from calipsync_amd import frame_loop
    rng = np.random.default_rng(77)
    imgs, lms, _ = make_frames(6, 300, 400, seed=5)
    u8 = [rng.integers(0, 256, (300, 400), dtype=np.uint8) for _ in range(6)]
    wd = torch.from_numpy(rng.standard_normal((6, 32, 32, 32)).astype(np.float32)).cuda()
    want = frame_loop.process_batch_device(gpu_net, imgs, lms, u8, windows=wd)                 # masks travel with the batch
    gpu_net.__dict__.pop("_mask_cache", None)
    monkeypatch.setattr(frame_loop, "_MASK_CACHE_CAP", 2 * 300 * 400 + 1)                       # room for two of six
    keys = [("clip2", i) for i in range(6)]
    for _ in range(3):       # 1st: all misses, four evicted; 2nd / 3rd: the survivors hit and are evicted while in use
        got = frame_loop.process_batch_device(gpu_net, imgs, lms, u8, windows=wd, mask_keys=keys)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
        assert gpu_net._mask_cache_bytes == sum(v.numel() for v in gpu_net._mask_cache.values()) <= 2 * 300 * 400 + 1
    gpu_net.__dict__.pop("_mask_cache", None)


def test_pinned_pool_size_classes_and_cap(monkeypatch):
    """requests are served by the smallest free buffer that fits, new buffers come in a few size classes,
    and the free list is capped in bytes (no pinned allocation needed to check the policy)."""
    from calipsync_amd import frame_loop
    assert frame_loop._size_class(1) == 1 << 20 and frame_loop._size_class((1 << 20) + 1) == 2 << 20
    assert frame_loop._size_class(30 << 20) == 32 << 20 and frame_loop._size_class(33 << 20) == 64 << 20
    assert frame_loop._size_class(65 << 20) == 80 << 20 and frame_loop._size_class(530 << 20) == 544 << 20

    class Buf:          # stands in for a pinned tensor
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n
    monkeypatch.setattr(frame_loop, "_PINNED", [])
    monkeypatch.setattr(frame_loop, "_PINNED_CAP", 100 << 20)
    for n in (32, 64, 32, 16):
        frame_loop._release_pinned(Buf(n << 20))
    assert frame_loop.pinned_pool_bytes() == (32 + 32 + 16) << 20            # over the cap: the largest one was dropped
    got = frame_loop._acquire_pinned(20 << 20)
    assert got.numel() == 32 << 20                                           # smallest that fits, not the first
    assert frame_loop._acquire_pinned(10 << 20).numel() == 16 << 20
    assert frame_loop.pinned_pool_bytes() == 32 << 20


@pytest.mark.gpu
def test_frame_synthesizer_end_to_end(gpu_net, tmp_path):
    """FrameSynthesizer on a synthetic infer_data directory: device-gathered windows == host windows through
    process_batch, variable last batch, indices."""
    from calipsync_amd.frame_synth import FrameSynthesizer
    imgs, lms = write_dataset(str(tmp_path), 10, 270, 360, seed=4)
    feats = np.random.default_rng(5).standard_normal((11, 2, 1024)).astype(np.float32)
    fs = FrameSynthesizer(None, str(tmp_path), device="cuda:0", batch_size=4, seed=9, net=gpu_net)
    out = list(fs.iterate_synthesized_frames(feats, 0, True))
    assert [o["index"] for o in out] == list(range(11)) and len(out) == 11          # 4 + 4 + 3
    fs2 = FrameSynthesizer(None, str(tmp_path), device="cuda:0", batch_size=4, seed=9, net=gpu_net)
    k = 0
    for start in range(0, 11, 4):
        idx = list(range(start, min(start + 4, 11)))
        seq = fs2._generate_frame_sequence(len(idx))
        bi, bl, bm = fs2._load_batch_frames(seq)
        ref = fs2.process_batch(bi, bl, bm, fs2._get_audio_features(feats, idx))      # the reference's calling form
        for j, frame in enumerate(ref):
            assert out[k]["physical_index"] == seq[j]
            assert np.array_equal(out[k]["frame"], frame), (start, j)
            k += 1
    assert any(not np.array_equal(o["frame"], imgs[o["physical_index"]]) for o in out)


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [1, 0])
def test_video_stream_manager_writes_a_playable_file(gpu_net, tmp_path, in_flight):
    """The offline driver end to end (inference.py:47-110): features -> synthesised frames -> a video file.  Without cv2 /
    ffmpeg (this image) the container is Motion-JPEG AVI written with Pillow; its frames are the iterator's frames (to JPEG
    tolerance), in order, at the reference's 25 fps.  Both loop orders (one batch in flight / batch by batch) give the same
    frames."""
    from calipsync_amd import mjpeg_avi
    from calipsync_amd.frame_synth import FrameSynthesizer, VideoStreamManager
    data = tmp_path / "data"
    write_dataset(str(data), 10, 270, 360, seed=4)
    feats = np.random.default_rng(5).standard_normal((9, 2, 1024)).astype(np.float32)
    np.save(str(tmp_path / "feats.npy"), feats)
    vsm = VideoStreamManager(str(data), None, device="cuda:0", batch_size=4, seed=9, net=gpu_net, batches_in_flight=in_flight)
    out = vsm.process_single_file(str(tmp_path / "feats.npy"), str(tmp_path / "out.mp4"))
    try:
        import cv2  # noqa: F401
        assert out.endswith(".mp4") and os.path.getsize(out) > 0
        return
    except ImportError:
        pass
    assert out.endswith(".avi")
    fps, frames = mjpeg_avi.read_mjpeg_avi(out)
    want = [o["frame"] for o in FrameSynthesizer(None, str(data), device="cuda:0", batch_size=4, seed=9, net=gpu_net)
            .iterate_synthesized_frames(feats, 0, True)]
    assert fps == 25.0 and len(frames) == len(want) == 9
    for a, b in zip(want, frames):
        assert a.shape == b.shape and np.abs(a.astype(int) - b.astype(int)).mean() < 12.0     # random-noise frames: JPEG is lossy
