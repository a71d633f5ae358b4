"""HuBERT window gather: oracle properties on the CPU, device gather == oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import frame_loop_oracle


def _features(t, seed=3):
    return np.random.default_rng(seed).standard_normal((t, 2, 1024)).astype(np.float32)


def test_window_layout_and_padding():
    f = _features(40)
    w = frame_loop_oracle.get_audio_features(f, [20, 0, 3, 39, 36])
    assert w.shape == (5, 32, 32, 32) and w.dtype == np.float32
    # interior: channel c is feature step idx-8+c//2, half c%2, the 1024 values as a 32x32 image
    assert np.array_equal(w[0, 5], f[20 - 8 + 2, 1].reshape(32, 32))
    # idx=0: the first 8 steps (16 channels) are zero padding, then features[0..7]
    assert not w[1, :16].any() and np.array_equal(w[1, 16], f[0, 0].reshape(32, 32))
    # idx=39 of 40: right end padded (steps 40..46 missing -> last 14 channels zero)
    assert np.array_equal(w[3, 17], f[39, 1].reshape(32, 32)) and not w[3, 18:].any()


def test_far_outside_is_zero():
    f = _features(10)
    w = frame_loop_oracle.get_audio_features(f, [30, -20])
    assert not w.any()


def test_closed_form_window_plan_equals_the_literal_restatement():
    """The HIP kernel implements `audio_window_plan`; the literal statement-by-statement restatement of
    infer_api.py:99-145 (truncated zeros_like pads, reshape-or-zeros, Python slice semantics for negative
    bounds) must give the same window for every (idx, T), including clips shorter than the pad."""
    for t in list(range(1, 20)) + [33, 40]:
        f = np.arange(1, t * 2048 + 1, dtype=np.float32).reshape(t, 2, 1024)      # all non-zero, row-identifying
        idxs = list(range(-30, t + 30))
        lit = frame_loop_oracle.get_audio_features(f, idxs)
        for k, idx in enumerate(idxs):
            start, n0, pl, ok = frame_loop_oracle.audio_window_plan(idx, t)
            want = np.zeros((16, 2, 1024), np.float32)
            if ok:
                want[pl:pl + n0] = f[start:start + n0]
            assert np.array_equal(lit[k].reshape(16, 2, 1024), want), (t, idx)


def test_short_clips_and_far_indices_follow_the_reference_truncation():
    """with T < 8 the reference's pad is zeros_like(auds[:pad]) -- at most len(auds) rows -- so
    the window misses 16 rows and the frame falls back to zeros; same for idx > T."""
    f = _features(5)
    w = frame_loop_oracle.get_audio_features(f, [0, 2, 9, 12, 4])
    assert not w[:4].any()                        # 5+5+3 = 13, 5+5+5 = 15, 4+4 = 8, 1+1 = 2 rows: never 16
    # idx=4: 5 rows + pad_left 4 + pad_right 7 (<= the 9 rows it has by then) = 16 -> a valid window
    assert not w[4, :8].any() and np.array_equal(w[4, 8], f[0, 0].reshape(32, 32)) and not w[4, 18:].any()
    f = _features(12)
    w = frame_loop_oracle.get_audio_features(f, [4, 12, 13, 20])
    assert np.array_equal(w[0, 8], f[0, 0].reshape(32, 32)) and not w[0, :8].any()   # pad 4 <= 12 rows: fine
    assert np.array_equal(w[1, 0], f[4, 0].reshape(32, 32)) and not w[1, 16:].any()  # idx == T: pad_right 8 <= 8 rows
    assert not w[2].any() and not w[3].any()      # idx > T: pad_right 9 > the 7 rows left -> zeros


@pytest.mark.gpu
def test_device_window_gather_corner_cases_match_the_literal_oracle(recipe_sd):
    """Short clips (T < 8), indices past the end and negative indices through the device gather."""
    from calipsync_amd import recipe
    from calipsync_amd.unet import Model
    net = Model(6, "hubert").to("cuda:0")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    for t, idx in ((5, [0, 1, 4, 7, 20]), (9, [-9, -8, -1, 0, 8, 9, 10, 17]), (12, [4, 12, 13, 20, -3]),
                   (16, [8, 0, 16, 17, 24, 25])):
        feats = _features(t, seed=t)
        x, _ = recipe.make_inputs(len(idx))
        xt = torch.from_numpy(x).cuda()
        host = torch.from_numpy(frame_loop_oracle.get_audio_features(feats, idx)).cuda()
        assert torch.equal(net.forward_windows(xt, torch.from_numpy(feats).cuda(), idx), net(xt, host)), (t, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_device_window_gather_matches_host_windows(recipe_sd, precision):
    """forward_windows(x, features, idx) == forward(x, host-built windows), bit for bit."""
    from calipsync_amd import recipe
    from calipsync_amd.unet import Model
    net = Model(6, "hubert", precision=precision).to("cuda:0")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    feats = _features(23)
    idx = [0, 1, 7, 8, 11, 15, 22, 21, 30]            # both ends padded, one fully outside on the right
    x, _ = recipe.make_inputs(len(idx))
    xt = torch.from_numpy(x).cuda()
    host = torch.from_numpy(frame_loop_oracle.get_audio_features(feats, idx)).cuda()
    ref = net(xt, host)
    got = net.forward_windows(xt, torch.from_numpy(feats).cuda(), idx)
    assert torch.equal(got, ref)
    assert (net(xt, host.flip(0).contiguous()) - ref).abs().max() > 1e-3   # the windows matter


def test_crop_oracle_layout():
    crops = np.random.default_rng(0).integers(0, 256, (2, 168, 168, 3), dtype=np.uint8)
    x = frame_loop_oracle.crops_to_model_input(crops)
    assert x.shape == (2, 6, 160, 160) and x.dtype == np.float32
    assert x[1, 2, 0, 0] == np.float32(crops[1, 4, 4, 2]) / np.float32(255.0)
    assert not x[:, 3:, 5:150, 5:155].any()                       # the blacked-out mouth rectangle
    assert np.array_equal(x[:, 3:, :5], x[:, :3, :5]) and np.array_equal(x[:, 3:, :, 155:], x[:, :3, :, 155:])


@pytest.mark.gpu
def test_device_crop_and_u8_are_bit_exact():
    from calipsync_amd import frame_loop
    rng = np.random.default_rng(1)
    crops = rng.integers(0, 256, (5, 168, 168, 3), dtype=np.uint8)
    x = frame_loop.crops_to_model_input(torch.from_numpy(crops).cuda())
    assert torch.equal(x.cpu(), torch.from_numpy(frame_loop_oracle.crops_to_model_input(crops)))
    pred = rng.random((5, 3, 160, 160), dtype=np.float32)
    pred[0, 0, 0, :4] = [0.0, 0.999999, 1.0 / 255, 0.5]            # truncation edge values
    u8 = frame_loop.predictions_to_uint8(torch.from_numpy(pred).cuda())
    assert torch.equal(u8.cpu(), torch.from_numpy(frame_loop_oracle.predictions_to_uint8(pred)))
    with pytest.raises(RuntimeError):
        frame_loop.crops_to_model_input(torch.from_numpy(crops))    # CPU tensor: no fallback
