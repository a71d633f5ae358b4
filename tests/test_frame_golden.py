"""Rows f2 (HuBERT-window gather) and f3-walk (frame sequencing) against fixtures produced by the REFERENCE's own
``FrameSynthesizer._get_audio_features`` / ``_generate_frame_sequence`` (infer_api.py:99-145, 147-190), run in the build
container by tests/golden/make_frame_golden.py.  CPU tests pin the oracle restatements and the host logic; the
``gpu`` tests pin the device gather (C ABI ``casync_op_audio_windows`` and ``Model.forward_windows``) bit for bit."""
import hashlib
import os
import random

import numpy as np
import pytest
import torch

from frame_data import golden_features
from oracle import frame_loop_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def windows():
    return np.load(os.path.join(GOLDEN, "frame_windows.npz"))


@pytest.fixture(scope="module")
def walks():
    return np.load(os.path.join(GOLDEN, "frame_walk.npz"))


def _sha(a: np.ndarray) -> bytes:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()


def _check_windows(fx, t, got):
    """got [n_idx,32,32,32] fp32 for every index of the fixture's clip T=t: digests, zero flags, full windows."""
    idx = fx[f"T{t}.idx"].tolist()
    assert got.shape == (len(idx), 32, 32, 32) and got.dtype == np.float32
    want = fx[f"T{t}.window_sha256"]
    for k, i in enumerate(idx):
        assert _sha(got[k]) == want[k].tobytes(), (t, i)
        assert bool(got[k].any()) == bool(fx[f"T{t}.nonzero"][k]), (t, i)
    for name in fx.files:
        if name.startswith(f"T{t}.idx") and name.endswith(".full"):
            i = int(name[len(f"T{t}.idx"):-len(".full")])
            assert np.array_equal(got[idx.index(i)], fx[name]), (t, i)


# ---------------------------------------------------------------------------------------------- f2, CPU
def test_fixture_inputs_regenerate_bit_for_bit(windows):
    for t in windows["clips"].tolist():
        assert _sha(golden_features(t)) == windows[f"T{t}.features_sha256"].tobytes(), t
    # the fixture is not vacuous: windows of every kind are in it
    assert windows["T40.nonzero"].sum() == 41 and not windows["T1.nonzero"].any()      # idx 0..T are valid on a long clip
    assert windows["T7.nonzero"].sum() == 7     # a clip shorter than the pad: the truncated pads still reach 16 rows for idx 1..7
    nz16 = dict(zip(windows["T16.idx"].tolist(), windows["T16.nonzero"].tolist()))
    assert nz16[-16] and not nz16[-15] and not nz16[-1]   # idx = -T: `right` = -8 becomes a from-the-end Python slice that fits


def test_oracle_windows_equal_the_reference(windows):
    for t in windows["clips"].tolist():
        got = frame_loop_oracle.get_audio_features(golden_features(t), windows[f"T{t}.idx"].tolist())
        _check_windows(windows, t, got)


def test_host_windows_equal_the_reference(windows):
    """The product's host-side windows (``FrameSynthesizer._get_audio_features`` -> ``frame_loop.audio_windows_host``)."""
    from calipsync_amd import frame_loop
    for t in windows["clips"].tolist():
        _check_windows(windows, t, frame_loop.audio_windows_host(golden_features(t), windows[f"T{t}.idx"].tolist()))


def test_closed_form_plan_agrees_with_the_reference_zero_flags(windows):
    """``audio_window_plan`` (what the HIP kernel computes): a window is non-zero exactly when the plan is valid and
    copies at least one row (the features are non-zero almost everywhere)."""
    for t in windows["clips"].tolist():
        for i, nz in zip(windows[f"T{t}.idx"].tolist(), windows[f"T{t}.nonzero"].tolist()):
            start, n0, pl, ok = frame_loop_oracle.audio_window_plan(i, t)
            assert bool(ok and n0 > 0) == nz, (t, i)


# ---------------------------------------------------------------------------------------------- f3 walk, CPU
def _walk_cases(walks):
    for total in walks["clips"].tolist():
        for seed in walks["seeds"].tolist():
            for n in walks["requests"].tolist():
                yield total, seed, [n] * int(walks["calls"]), f"F{total}.n{n}.s{seed}"
            yield total, seed, walks["mixed_requests"].tolist(), f"F{total}.mixed.s{seed}"


def test_oracle_walk_equals_the_reference(walks):
    for total, seed, requests, key in _walk_cases(walks):
        walk = frame_loop_oracle.FrameWalk(total, random.Random(seed))
        frames, states = walks[key + ".frames"], walks[key + ".states"]
        at = 0
        for call, n in enumerate(requests):
            assert walk.generate(n) == frames[at:at + n].tolist(), (key, call)
            at += n
            assert [walk.current_direction, walk.target_frame_count, walk.processed_frame_count,
                    walk.current_frame_position] == states[call].tolist(), (key, call)
        assert at == len(frames)


def test_product_walk_equals_the_reference(walks):
    """``frame_walk.PingPongWalk`` (one phase counter, array arithmetic) under the reference's own random stream:
    ``random.Random(seed)`` and -- as the reference draws -- the module-level ``random`` after ``random.seed``."""
    from calipsync_amd.frame_walk import PingPongWalk
    for total, seed, requests, key in _walk_cases(walks):
        frames, states = walks[key + ".frames"], walks[key + ".states"]
        random.seed(seed)
        for walk in (PingPongWalk(total, random.Random(seed)), PingPongWalk(total)):     # the second draws from `random`
            at = 0
            for call, n in enumerate(requests):
                assert walk.take(n) == frames[at:at + n].tolist(), (key, call)
                at += n
                direction, target, processed, position = states[call].tolist()
                assert (walk.run_budget, walk.run_used, walk.position) == (target, processed, position), (key, call)
                if 0 < position < total - 1:          # at either end both directions are the same walk
                    assert walk.ascending == (direction == 1), (key, call)


def test_synthesizer_sequence_equals_the_reference(walks, tmp_path):
    """Through the drop-in class: ``FrameSynthesizer._generate_frame_sequence`` and the reference's state attributes."""
    from calipsync_amd.frame_synth import FrameSynthesizer
    from test_frame_ops import _stub_net, write_dataset
    write_dataset(str(tmp_path), 37, 32, 48)
    for seed in walks["seeds"].tolist():
        key = f"F37.mixed.s{seed}"
        random.seed(seed)
        fs = FrameSynthesizer(None, str(tmp_path), device="cpu", batch_size=8, net=_stub_net())   # seed=None: module `random`
        assert fs.total_frames == 37
        at = 0
        for call, n in enumerate(walks["mixed_requests"].tolist()):
            assert fs._generate_frame_sequence(n) == walks[key + ".frames"][at:at + n].tolist()
            at += n
            _, target, processed, position = walks[key + ".states"][call].tolist()
            assert (fs.target_frame_count, fs.processed_frame_count, fs.current_frame_position) == \
                   (target, processed, position)


# ---------------------------------------------------------------------------------------------- f2, device
@pytest.mark.gpu
def test_device_gather_equals_the_reference_windows(windows):
    """``casync_op_audio_windows`` (the kernel ``casync_forward_windows`` starts with) == the reference's return value
    for every (T, idx) of the fixture, bit for bit: the reference layout (nhwc=0) and the engine's NHWC image (nhwc=1)."""
    from calipsync_amd import _lib, frame_loop
    lib = _lib.load()
    for t in windows["clips"].tolist():
        idx = windows[f"T{t}.idx"].tolist()
        feats = torch.from_numpy(golden_features(t)).cuda()
        got = frame_loop.audio_windows_device(feats, idx)
        _check_windows(windows, t, got.cpu().numpy())
        # the engine's own layout: [B, 1024 pixels, 32 channels]
        it = torch.tensor(idx, dtype=torch.int32).cuda()
        nhwc = torch.full((len(idx), 1024, 32), float("nan"), device="cuda")
        _lib.check(lib.casync_op_set_dtype(0), "set_dtype")
        _lib.check(lib.casync_op_audio_windows(feats.data_ptr(), t, it.data_ptr(), nhwc.data_ptr(), len(idx), 1,
                                               torch.cuda.current_stream().cuda_stream), "audio_windows")
        assert torch.equal(nhwc.permute(0, 2, 1).reshape(len(idx), 32, 32, 32), got)
    # index tensors on the device, one window per call, an empty request
    feats = torch.from_numpy(golden_features(40)).cuda()
    one = frame_loop.audio_windows_device(feats, torch.tensor([20], dtype=torch.int32, device="cuda"))
    assert np.array_equal(one[0].cpu().numpy(), windows["T40.idx20.full"])
    assert frame_loop.audio_windows_device(feats, []).shape == (0, 32, 32, 32)
    with pytest.raises(RuntimeError):
        frame_loop.audio_windows_device(feats.cpu(), [0])                  # no CPU fallback


@pytest.mark.gpu
def test_forward_windows_equals_forward_on_the_reference_windows(windows, recipe_sd):
    """The model call of the reference's loop (``self.net(batch, hubert_tensor)`` on ``_get_audio_features``' windows,
    infer_api.py:247-260) == ``forward_windows`` on the raw features: bit for bit, with the REFERENCE's windows."""
    from calipsync_amd import recipe
    from calipsync_amd.unet import Model
    net = Model(6, "hubert").to("cuda:0")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    idx = [0, 3, 20, 39, 45]
    ref_windows = torch.from_numpy(np.stack([windows[f"T40.idx{i}.full"] for i in idx])).cuda()
    x, _ = recipe.make_inputs(len(idx))
    xt = torch.from_numpy(x).cuda()
    want = net(xt, ref_windows)
    got = net.forward_windows(xt, torch.from_numpy(golden_features(40)).cuda(), idx)
    assert torch.equal(got, want)
    assert (net(xt, ref_windows.flip(0).contiguous()) - want).abs().max() > 1e-3      # the windows matter
