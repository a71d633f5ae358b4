"""C-ABI library: loads without a GPU, exports every declared symbol, and agrees with the
Python packer on the packed-weight layout.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from calipsync_amd import _lib, arch, pack
from calipsync_amd.unet import Model

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from calipsync_amd import build
    build.build()            # cross-compiles for gfx950 without a GPU
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(REPO, "include", "casync_hip.h")).read()
    declared = set(re.findall(r"\b(casync_[a-z0-9_]+)\s*\(", header))
    declared -= {"casync_engine"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_layout(lib):
    assert lib.casync_abi_version() == _lib.ABI_VERSION
    items, total = _lib.packed_layout()
    names = [n for n, _, _ in items]
    assert len(names) == len(set(names))
    end = 0
    for name, off, size in items:
        assert off % 64 == 0 and off >= end and size > 0, name
        end = off + size
    # 79 MB of folded fp32 weights + the composed p1q matrices (round 2) + the split expand matrices
    # of the four Up blocks (round 3)
    assert total >= end and total * 4 < 95e6


def test_packer_fills_the_engine_layout(lib, recipe_sd):
    buf = pack.pack(recipe_sd)
    items, total = _lib.packed_layout()
    assert buf.shape == (total,) and buf.dtype == np.float32
    used = sum(s for _, _, s in items)
    # folded weights == conv/linear weights + one bias/scale vector per layer
    assert used > 19_700_000
    off = dict((n, (o, s)) for n, o, s in items)
    o, s = off["attention_blocks.2.gamma"]
    assert buf[o] == np.float32(0.5) and s == 1


def test_bn_folding_is_exact_on_one_layer(recipe_sd):
    """Folded PW1 of down1.ir0 reproduces conv+BN of the oracle on random rows."""
    from oracle import unet_oracle
    import torch.nn.functional as F
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    p = "down1.maxpool_conv.0.double_conv.0"
    x = torch.randn(1, 32, 5, 7)
    ref = unet_oracle._bn(sd, f"{p}.conv.1", F.conv2d(x, sd[f"{p}.conv.0.weight"]))
    w = torch.from_numpy(f[f"{p}.pw1.w"].astype(np.float32))
    b = torch.from_numpy(f[f"{p}.pw1.b"].astype(np.float32))
    got = torch.einsum("nk,bkhw->bnhw", w, x) + b[None, :, None, None]
    assert (got - ref).abs().max() < 2e-6


def test_workspace_bytes_scale_with_batch(lib):
    one, b16, many = lib.casync_workspace_bytes(1), lib.casync_workspace_bytes(16), lib.casync_workspace_bytes(64)
    # 38.4 MB per frame with the fused kernels on (the 2 x 160x160x128 expanded slots are not needed);
    # 55.9 MB with CASYNC_FUSE_IR=0.  Below 12 frames the skip_early plan parks two more tensors (20x20x1024 and
    # 40x40x512 floats per frame: engine.hip Arena EP1 / EP2).
    assert 30e6 < many / 64 < 60e6 and abs(many / b16 - 4) < 0.01
    assert abs((one - many / 64) - 4 * (400 * 1024 + 1600 * 512)) < 1024
    assert lib.casync_workspace_bytes(0) < 0


def test_options_are_named_and_checked(lib):
    """casync_set_option / casync_get_option on the process defaults (no GPU needed); an unknown name is
    an error, and the workspace of the unfused plan is larger."""
    assert _lib.get_option("lanes") >= 1
    with pytest.raises(RuntimeError):
        _lib.set_option("no_such_switch", 1)
    fused = lib.casync_workspace_bytes(8)
    old = _lib.get_option("fuse_ir")
    try:
        _lib.set_option("fuse_ir", 0)
        assert lib.casync_workspace_bytes(8) > fused + 8 * 15e6
    finally:
        _lib.set_option("fuse_ir", old)
    assert lib.casync_workspace_bytes(8) == fused


def test_model_has_reference_checkpoint_format(recipe_sd):
    m = Model(6, "hubert")
    sd = m.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, s) for k, s, _, _ in arch.manifest()]
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})   # strict
    bad = dict(recipe_sd)
    bad.pop("bn_kx.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in bad.items()})
    with pytest.raises(NotImplementedError):
        Model(6, "wenet")


def test_no_cpu_fallback():
    m = Model(6, "hubert")
    with pytest.raises(RuntimeError, match="ROCm device only"):
        m(torch.zeros(1, 6, 160, 160), torch.zeros(1, 32, 32, 32))
    with pytest.raises(RuntimeError, match=r"\[B,6,160,160\]"):
        m(torch.zeros(1, 3, 160, 160), torch.zeros(1, 32, 32, 32))


def test_create_without_gpu_reports_no_device(lib):
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert lib.casync_create(0, ctypes.byref(h)) < 0
    assert b"device" in lib.casync_last_error()


def test_header_is_c99_and_a_plain_c_host_can_bind_it(lib, tmp_path):
    """The boundary is a C ABI, not a Python extension: tests/c/abi_host.c includes the header as
    C99 (-Wall -Werror), dlopens the library and walks the packed layout without Python."""
    import subprocess
    exe = tmp_path / "abi_host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                    os.path.join(REPO, "tests", "c", "abi_host.c"), "-ldl", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), _lib.lib_path()], check=True, capture_output=True, text=True).stdout
    items, total = _lib.packed_layout()
    assert f"abi {_lib.ABI_VERSION} tensors {len(items)} " in out and f"total {total} " in out and "gammas 4" in out


def test_every_engine_option_is_documented():
    """Every switch `casync_set_option` knows (the name table of csrc/runtime.hip) appears in DESIGN.md's option table, and the
    library answers for each of them (no GPU needed: the process defaults)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = re.findall(r'\{"([a-z0-9_]+)", &CasyncOptions::', open(os.path.join(root, "calipsync_amd", "csrc", "runtime.hip")).read())
    assert len(names) >= 30
    design = open(os.path.join(root, "DESIGN.md")).read()
    missing = [n for n in names if f"`{n}`" not in design]
    assert not missing, f"options missing from DESIGN.md section 4: {missing}"
    for n in names:
        assert isinstance(_lib.get_option(n), int)
