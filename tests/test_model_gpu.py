"""-m gpu: the whole HIP forward through the drop-in Model vs (a) the golden vectors made
by the reference itself and (b) the CPU oracle on fresh seeded inputs.

Bar (BASELINE.json north_star): |delta| < 1e-3 per pixel, fp32."""
import numpy as np
import pytest
import torch

from calipsync_amd import recipe
from calipsync_amd.unet import Model
from conftest import sample_indices

pytestmark = pytest.mark.gpu
TOL = 1e-3          # north-star bar
EXPECT = 5e-5       # what fp32 MFMA + BN folding actually delivers; regressions show here first


@pytest.fixture(scope="module")
def net(recipe_sd):
    m = Model(6, "hubert").to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    return m.eval()


def test_golden_output(net, golden):
    x, a = recipe.make_inputs(2)
    out = net(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    assert out.shape == (2, 3, 160, 160) and out.dtype == torch.float32 and out.is_cuda
    d = np.abs(out.cpu().numpy() - golden["out.full"]).max()
    print("max|d| vs reference golden:", d)
    assert d < TOL and d < EXPECT


TAPS = ["x1", "x2", "x3", "x4", "x5", "audio_conv2", "audio_conv3", "audio_conv4", "audio_conv5", "a",
        "tx", "att0", "att1", "att2", "att3", "kx", "fuse", "u1", "u2", "u3", "u4"]


@pytest.mark.parametrize("name", TAPS)
def test_golden_intermediates(net, golden, name):
    x, a = recipe.make_inputs(2)
    net(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    t = net.tap(name, 2).cpu().numpy()
    assert tuple(golden[f"{name}.shape"]) == t.shape
    if f"{name}.full" in golden:
        ref, got = golden[f"{name}.full"].reshape(-1), t.reshape(-1)
    else:
        ref, got = golden[f"{name}.samples"], t.reshape(-1)[sample_indices(t.size)]
    scale = max(1.0, float(np.abs(ref).max()))
    d = np.abs(got - ref).max() / scale
    assert d < 2e-5, (name, d)


@pytest.mark.parametrize("batch", [1, 3, 64])
def test_against_oracle(net, recipe_sd, batch):
    """configs[0]/[1] of BASELINE.json: B=1 plumbing and B=64 fp32 vs the CPU path."""
    from oracle import unet_oracle
    torch.set_num_threads(16)
    sd = unet_oracle.to_torch(recipe_sd)
    x, a = recipe.make_inputs_range(100, batch)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    ref = unet_oracle.forward(sd, xt, at)
    out = net(xt.cuda(), at.cuda()).cpu()
    d = float((out - ref).abs().max())
    print(f"B={batch} max|d| vs oracle: {d:.3e}")
    assert d < TOL and d < EXPECT


def test_frames_independent_and_batch_invariant(net, monkeypatch):
    """A frame's output must not depend on its neighbours or its position in the batch."""
    x, a = recipe.make_inputs(5)
    xt, at = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    full = net(xt, at)
    part = net(xt[3:4].contiguous(), at[3:4].contiguous())
    # stream-K splits a GEMM's k range differently for different row counts: fp32 reassociation only
    assert (full[3:4] - part).abs().max() < 1e-5
    assert torch.equal(net(xt, at), full)            # and it is repeatable bit for bit
    monkeypatch.setenv("CASYNC_GEMM_STREAMK", "0")   # plain tiles: same order per frame -> bitwise
    assert torch.equal(net(xt, at)[3:4], net(xt[3:4].contiguous(), at[3:4].contiguous()))


def test_lanes_and_stream_overlap_do_not_change_bits(net, monkeypatch):
    """The batch is cut into concurrent lanes (ragged: 50 = 25 + 25 = 17 + 17 + 16) and the audio
    branch runs on a side stream; both are scheduling only, so with plain GEMM tiles the output is
    bit-identical (with stream-K the k-split depends on the lane's row count: checked to 1e-5)."""
    x, a = recipe.make_inputs(50)
    xt, at = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    sk_full = net(xt, at)
    monkeypatch.setenv("CASYNC_LANES", "1")
    assert (net(xt, at) - sk_full).abs().max() < 1e-5
    monkeypatch.setenv("CASYNC_GEMM_STREAMK", "0")
    monkeypatch.setenv("CASYNC_LANES", "1")
    monkeypatch.setenv("CASYNC_OVERLAP", "0")
    base = net(xt, at)
    for lanes, overlap in (("1", "1"), ("2", "1"), ("3", "1"), ("2", "0")):
        monkeypatch.setenv("CASYNC_LANES", lanes)
        monkeypatch.setenv("CASYNC_OVERLAP", overlap)
        assert torch.equal(net(xt, at), base), (lanes, overlap)


def test_audio_matters(net):
    x, a = recipe.make_inputs(2)
    xt, at = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    d = (net(xt, at) - net(xt, at.flip(0).contiguous())).abs().max()
    assert d > 1e-2


def test_reload_and_default_init(net, recipe_sd):
    """load_state_dict re-packs; a default-init model (gamma=0) also runs and stays in (0,1)."""
    m = Model(6, "hubert").to("cuda:0")
    x, a = recipe.make_inputs(1)
    out0 = m(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    assert torch.isfinite(out0).all() and out0.min() > 0 and out0.max() < 1
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    out1 = m(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    assert torch.equal(out1, net(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()))


def test_empty_and_bad_inputs(net):
    e = net(torch.zeros(0, 6, 160, 160).cuda(), torch.zeros(0, 32, 32, 32).cuda())
    assert e.shape == (0, 3, 160, 160)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 6, 160, 160), torch.zeros(1, 32, 32, 32).cuda())      # CPU input
    with pytest.raises(RuntimeError):
        net(torch.zeros(2, 6, 160, 160).cuda(), torch.zeros(1, 32, 32, 32).cuda())
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 6, 160, 160).cuda().half(), torch.zeros(1, 32, 32, 32).cuda())


# ------------------------------------------------------------------ bf16 engine (BASELINE configs[2])
@pytest.fixture(scope="module")
def net_bf16(recipe_sd):
    m = Model(6, "hubert", precision="bf16").to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    return m.eval()


def test_bf16_engine_error_is_reported_not_hidden(net_bf16, golden):
    """bf16 activations cannot meet the 1e-3 bar (the reference's own bf16 autocast is
    9e-3 max / 1.3e-3 mean off its fp32).  Pin the error level so regressions show, and keep
    it clearly labelled as NOT the parity path."""
    x, a = recipe.make_inputs(2)
    out = net_bf16(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    assert out.dtype == torch.float32 and out.shape == (2, 3, 160, 160)
    d = np.abs(out.cpu().numpy() - golden["out.full"])
    print(f"bf16 engine vs reference golden: max {d.max():.3e} mean {d.mean():.3e}")
    assert d.max() < 4e-2 and d.mean() < 4e-3
    assert d.max() > 1e-4          # it really is the bf16 path


@pytest.mark.parametrize("name", ["x1", "x5", "a", "tx", "kx", "fuse", "u4"])
def test_bf16_intermediates_track_fp32(net_bf16, golden, name):
    x, a = recipe.make_inputs(2)
    net_bf16(torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda())
    t = net_bf16.tap(name, 2).cpu().numpy()
    if f"{name}.full" in golden:
        ref, got = golden[f"{name}.full"].reshape(-1), t.reshape(-1)
    else:
        ref, got = golden[f"{name}.samples"], t.reshape(-1)[sample_indices(t.size)]
    rel = np.abs(got - ref).max() / max(1.0, float(np.abs(ref).max()))
    print(f"bf16 {name}: max rel err {rel:.3e}")
    assert rel < 6e-2


def test_bf16_frames_independent(net_bf16, monkeypatch):
    x, a = recipe.make_inputs(5)
    xt, at = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    full = net_bf16(xt, at)
    part = net_bf16(xt[3:4].contiguous(), at[3:4].contiguous())
    assert (full[3:4] - part).abs().max() < 2e-2     # stream-K k-split differs with the row count: bf16 ulps
    monkeypatch.setenv("CASYNC_GEMM_STREAMK", "0")
    assert torch.equal(net_bf16(xt, at)[3:4], net_bf16(xt[3:4].contiguous(), at[3:4].contiguous()))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_batch_properties(recipe_sd, precision, monkeypatch):
    """BASELINE configs[2]/[3] per-GPU size (512 frames): no oracle run at this size; instead the
    size-independent properties -- every frame equals its own single-frame forward (frames
    independent, lanes / tiling batch-invariant: bit for bit with plain GEMM tiles, to rounding
    with the stream-K k-split), duplicates agree, outputs in (0,1)."""
    m = Model(6, "hubert", precision=precision).to("cuda:0")
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    x16, a16 = recipe.make_inputs(16)
    x = torch.from_numpy(x16).cuda().repeat(32, 1, 1, 1)          # 512 frames, 16 distinct
    a = torch.from_numpy(a16).cuda().repeat(32, 1, 1, 1)
    tol = 1e-5 if precision == "fp32" else 2e-2
    out = m(x, a)
    assert out.shape == (512, 3, 160, 160) and torch.isfinite(out).all()
    assert out.min() > 0 and out.max() < 1
    assert torch.equal(out[:16], out[256:272])                    # same place in both lanes: same bits
    assert (out[5] - out[16 * 31 + 5]).abs().max() < tol          # duplicates anywhere
    for i in (0, 7, 15):
        single = m(x[i:i + 1].contiguous(), a[i:i + 1].contiguous())
        assert (single[0] - out[i]).abs().max() < tol, i
    monkeypatch.setenv("CASYNC_GEMM_STREAMK", "0")
    out = m(x, a)
    assert torch.equal(out[:16], out[256:272]) and torch.equal(out[5], out[16 * 31 + 5])
    for i in (0, 7, 15):
        single = m(x[i:i + 1].contiguous(), a[i:i + 1].contiguous())
        assert torch.equal(single[0], out[i]), i
