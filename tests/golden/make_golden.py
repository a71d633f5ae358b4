"""Generate the golden fixtures by running the REFERENCE itself (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py        # G1.2,   B=2 -> unet_g12_b2.npz
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py b      # G1.2-B, B=3 -> unet_g12b_b3.npz

Imports ``/root/reference/module/unet.py`` (read-only mount; it needs only
``torch``), loads the repo's deterministic "G1.2" state_dict
(``calipsync_amd/recipe.py``) into the reference ``Model(6, 'hubert')`` in
eval mode, runs seeded inputs through it on PyTorch-CPU fp32 and records the
output plus named intermediates captured with forward hooks.

Only data is written (``tests/golden/*.npz`` + a text manifest): inputs are
regenerated from the recipe, outputs are stored in full where small and as
(statistics, 4096 strided samples) otherwise.  No reference source, bytecode or
pickled module is copied anywhere.  ``/root/reference`` does not exist on the
GPU box, so nothing under ``tests/`` imports this script.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from calipsync_amd import arch, recipe          # noqa: E402
from module.unet import Model                   # noqa: E402  (the reference)

N_SAMPLES = 4096
BATCH = 2


def sample_indices(numel: int) -> np.ndarray:
    return (np.arange(N_SAMPLES, dtype=np.int64) * 2654435761) % numel


def summarize(name: str, t: torch.Tensor, store: dict, full: bool) -> None:
    a = t.detach().contiguous().numpy()
    flat = a.reshape(-1)
    store[f"{name}.shape"] = np.array(a.shape, dtype=np.int64)
    f64 = flat.astype(np.float64)
    store[f"{name}.stats"] = np.array([f64.sum(), np.abs(f64).sum(), (f64 * f64).sum(),
                                       f64.min(), f64.max()], dtype=np.float64)
    if full:
        store[f"{name}.full"] = a
    else:
        store[f"{name}.samples"] = flat[sample_indices(flat.size)]


def main(variant: str = "") -> None:
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd_np = recipe.make_state_dict_b() if variant == "b" else recipe.make_state_dict()
    net = Model(6, "hubert").eval()
    missing = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    print("load_state_dict:", missing)

    x_np, a_np = recipe.make_inputs_b() if variant == "b" else recipe.make_inputs(BATCH)
    batch = x_np.shape[0]
    x, a = torch.from_numpy(x_np), torch.from_numpy(a_np)

    taps: dict = {}
    relu_calls = []

    def hook(name):
        def fn(_m, _inp, out):
            taps[name] = out.detach().clone()
        return fn

    def relu_hook(_m, _inp, out):
        relu_calls.append(out.detach().clone())

    named = {
        "x1": net.inc, "x2": net.down1, "x3": net.down2, "x4": net.down3, "x5": net.down4,
        "audio_conv2": net.audio_model.conv2, "audio_conv4": net.audio_model.conv4,
        "a": net.audio_model, "mlp": net.mlp_fusion, "tx": net.bn_tx, "kx": net.lru_kx,
        "fuse": net.fuse_conv, "u1": net.up1, "u2": net.up2, "u3": net.up3, "u4": net.up4,
        "up1_bilinear": net.up1.up, "ca0": net.attention_blocks[0].cross_attention,
        "inc_pw1": net.inc.inconv[0].conv[2], "inc_dw": net.inc.inconv[0].conv[5],
        "down1_ir0": net.down1.maxpool_conv[0].double_conv[0],
    }
    for i, blk in enumerate(net.attention_blocks):
        named[f"att{i}"] = blk
    handles = [m.register_forward_hook(hook(n)) for n, m in named.items()]
    handles.append(net.audio_model.relu.register_forward_hook(relu_hook))

    with torch.no_grad():
        out = net(x, a)
        # fp64 run of the same reference: the clean floor for the fp32 tolerance
        net64 = Model(6, "hubert").double().eval()
        net64.load_state_dict({k: torch.from_numpy(v.copy()).double() if v.dtype != np.int64
                               else torch.from_numpy(v.copy()) for k, v in sd_np.items()})
        for h in handles:
            h.remove()
        out64 = net64(x.double(), a.double())
    taps["out"] = out
    taps["audio_conv3"], taps["audio_conv5"] = relu_calls[0], relu_calls[1]
    print("out range", float(out.min()), float(out.max()), "std", float(out.std()))
    print("fp32 vs fp64 max|d|", float((out.double() - out64).abs().max()))
    # sensitivity sanity: swapping the audio between the two frames must move the output
    with torch.no_grad():
        out_sw = net(x, a.flip(0))
    print("audio-swap max|d|", float((out - out_sw).abs().max()))

    store: dict = {}
    full_names = {"out", "x5", "a"}
    for name, t in taps.items():
        summarize(name, t, store, name in full_names)
    store["out64.full"] = out64.numpy().astype(np.float64)
    store["audio_swap_maxdiff"] = np.array([float((out - out_sw).abs().max())])

    # pin the recipe: hashes of the regenerated weights / inputs
    h = hashlib.sha256()
    for k, _s, _d, _r in arch.manifest():
        h.update(np.ascontiguousarray(sd_np[k]).tobytes())
    store["weights_sha256"] = np.frombuffer(h.digest(), dtype=np.uint8)
    store["inputs_sha256"] = np.frombuffer(
        hashlib.sha256(x_np.tobytes() + a_np.tobytes()).digest(), dtype=np.uint8)
    store["batch"] = np.array([batch])
    path = os.path.join(HERE, "unet_g12b_b3.npz" if variant == "b" else "unet_g12_b2.npz")
    np.savez(path, **store)

    if not variant:
        with open(os.path.join(HERE, "state_dict_manifest.txt"), "w") as f:
            for k, v in net.state_dict().items():
                f.write(f"{k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}\n")
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
