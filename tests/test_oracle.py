"""Pin the CPU oracle against golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from calipsync_amd import recipe
from oracle import unet_oracle
from conftest import sample_indices

TOL = 1e-5   # SURVEY §8c acceptance for the restatement vs the reference (fp32 CPU both)


@pytest.fixture(scope="module")
def oracle_run(recipe_sd):
    torch.set_num_threads(8)
    sd = unet_oracle.to_torch(recipe_sd)
    x, a = recipe.make_inputs(2)
    taps = {}
    out = unet_oracle.forward(sd, torch.from_numpy(x), torch.from_numpy(a), taps)
    return out, taps


def test_output_matches_reference(golden, oracle_run):
    out, _ = oracle_run
    ref = golden["out.full"]
    assert out.shape == ref.shape == (2, 3, 160, 160)
    d = np.abs(out.numpy() - ref).max()
    assert d <= TOL, d
    # and both sit on the reference's own fp64 result
    assert np.abs(out.numpy().astype(np.float64) - golden["out64.full"]).max() < 5e-6


TAPS = ["x1", "x2", "x3", "x4", "x5", "audio_conv2", "audio_conv3", "audio_conv4", "audio_conv5",
        "a", "tx", "att0", "att1", "att2", "att3", "kx", "fuse", "u1", "u2", "u3", "u4"]


@pytest.mark.parametrize("name", TAPS)
def test_intermediates_match_reference(golden, oracle_run, name):
    _, taps = oracle_run
    t = taps[name].contiguous().numpy()
    assert tuple(golden[f"{name}.shape"]) == t.shape
    if f"{name}.full" in golden:
        ref, got = golden[f"{name}.full"].reshape(-1), t.reshape(-1)
    else:
        ref, got = golden[f"{name}.samples"], t.reshape(-1)[sample_indices(t.size)]
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= TOL * scale
    s = golden[f"{name}.stats"]
    f64 = t.reshape(-1).astype(np.float64)
    assert abs(f64.sum() - s[0]) <= 1e-6 * max(1.0, s[1])
    assert abs((f64 * f64).sum() - s[2]) <= 1e-5 * max(1.0, s[2])


@pytest.fixture(scope="module")
def oracle_run_b(recipe_sd_b):
    torch.set_num_threads(8)
    sd = unet_oracle.to_torch(recipe_sd_b)
    x, a = recipe.make_inputs_b()
    taps = {}
    out = unet_oracle.forward(sd, torch.from_numpy(x), torch.from_numpy(a), taps)
    return out, taps


def test_recipe_b_is_bit_reproducible(golden_b, recipe_sd_b):
    import hashlib
    from calipsync_amd import arch
    h = hashlib.sha256()
    for k, _s, _d, _r in arch.manifest():
        h.update(np.ascontiguousarray(recipe_sd_b[k]).tobytes())
    assert bytes(golden_b["weights_sha256"]) == h.digest()
    x, a = recipe.make_inputs_b()
    assert x.shape[0] == int(golden_b["batch"][0]) == 3
    assert bytes(golden_b["inputs_sha256"]) == hashlib.sha256(x.tobytes() + a.tobytes()).digest()
    # the corners the fixture exists for
    gam = [float(recipe_sd_b[f"attention_blocks.{i}.cross_attention.gamma"][0]) for i in range(4)]
    assert min(gam) < 0 < max(gam)
    var = recipe_sd_b["inc.inconv.0.conv.1.running_var"]
    assert (var == np.float32(1e-3)).sum() >= 1
    assert float(np.abs(a).max()) > 12.0


def test_output_matches_reference_b(golden_b, oracle_run_b):
    """Same pin on the second fixture: negative / small / large attention gammas, BatchNorm channels with
    running_var = 1e-3 (eps matters in the fold), audio four times larger, an odd batch."""
    out, _ = oracle_run_b
    ref = golden_b["out.full"]
    assert out.shape == ref.shape == (3, 3, 160, 160)
    assert np.abs(out.numpy() - ref).max() <= TOL
    assert np.abs(out.numpy().astype(np.float64) - golden_b["out64.full"]).max() < 5e-6
    assert float(golden_b["audio_swap_maxdiff"][0]) > 1e-2


@pytest.mark.parametrize("name", TAPS)
def test_intermediates_match_reference_b(golden_b, oracle_run_b, name):
    _, taps = oracle_run_b
    t = taps[name].contiguous().numpy()
    assert tuple(golden_b[f"{name}.shape"]) == t.shape
    if f"{name}.full" in golden_b:
        ref, got = golden_b[f"{name}.full"].reshape(-1), t.reshape(-1)
    else:
        ref, got = golden_b[f"{name}.samples"], t.reshape(-1)[sample_indices(t.size)]
    assert np.abs(got - ref).max() <= TOL * max(1.0, float(np.abs(ref).max()))


def _close(got, ref):
    """|d| <= TOL relative to the tensor's magnitude (activations reach ~1e1)."""
    return np.abs(got - ref).max() <= TOL * max(1.0, float(np.abs(ref).max()))


def test_oracle_modules_match_reference_modules(golden, oracle_run, recipe_sd):
    """Module-level pins: MLP fusion, cross-attention with gamma != 0, bilinear x2."""
    _, taps = oracle_run
    sd = unet_oracle.to_torch(recipe_sd)
    mlp = unet_oracle.mlp_fusion(sd, taps["x5"], taps["a"]).numpy().reshape(-1)
    assert _close(mlp[sample_indices(mlp.size)], golden["mlp.samples"])
    up = torch.nn.functional.interpolate(taps["fuse"], scale_factor=2, mode="bilinear",
                                         align_corners=True).numpy().reshape(-1)
    assert _close(up[sample_indices(up.size)], golden["up1_bilinear.samples"])
    p = "attention_blocks.0"
    ox = torch.nn.functional.conv2d(taps["tx"], sd[f"{p}.attention_adjust_p_1.weight"],
                                    sd[f"{p}.attention_adjust_p_1.bias"])
    ca = unet_oracle.cross_attention(sd, f"{p}.cross_attention", ox, taps["a"]).numpy().reshape(-1)
    assert _close(ca[sample_indices(ca.size)], golden["ca0.samples"])


def test_frames_are_independent(recipe_sd):
    """Eval-mode forward is per-frame: batch of 2 == two batches of 1."""
    sd = unet_oracle.to_torch(recipe_sd)
    x, a = recipe.make_inputs(2)
    both = unet_oracle.forward(sd, torch.from_numpy(x), torch.from_numpy(a))
    one = unet_oracle.forward(sd, torch.from_numpy(x[1:]), torch.from_numpy(a[1:]))
    assert (both[1:] - one).abs().max() < 1e-6


def test_audio_branch_matters(golden, recipe_sd):
    """The golden recipe is not vacuous: swapping audio windows moves the output."""
    assert float(golden["audio_swap_maxdiff"][0]) > 1e-2
