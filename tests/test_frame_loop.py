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
