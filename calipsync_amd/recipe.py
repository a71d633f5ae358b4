"""Deterministic golden recipe "G1.2" for weights and inputs.

The reference ships no checkpoint and PyTorch's default init makes parity
vacuous (``gamma = 0`` switches all four cross-attention outputs off,
reference module/unet.py:205).  This recipe gives a trained-like, numerically
well-conditioned network whose output depends measurably on the audio branch
and on the attention blocks.

The random stream is the repo's own counter-based generator (splitmix64 hash
of ``(seed, crc32(key), index)`` -> Box-Muller), *not* ``torch.randn`` or a
NumPy ``Generator`` distribution, so the very same bits are produced in the
build container (where the golden fixtures are made from the reference) and on
the GPU box (where only this repo exists).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

from . import arch

WEIGHT_SEED = 1234
INPUT_SEED = 7
GAIN = 1.2
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _bits(seed: int, stream: int, n: int, lane: int) -> np.ndarray:
    """n 64-bit words of stream (seed, stream, lane); pure function of its args."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(seed << 32) ^ (stream & 0xFFFFFFFF)], dtype=np.uint64))
        base = _splitmix64(base + np.uint64(lane))
        idx = np.arange(n, dtype=np.uint64)
        return _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))


def uniform01(seed: int, stream: int, n: int, lane: int = 0) -> np.ndarray:
    """float64 uniforms in [0, 1) with 53 random bits."""
    return (_bits(seed, stream, n, lane) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal01(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 standard normals (Box-Muller on two independent lanes)."""
    u1 = uniform01(seed, stream, n, lane=1)
    u2 = uniform01(seed, stream, n, lane=2)
    r = np.sqrt(-2.0 * np.log1p(-u1))          # 1-u1 in (0, 1]
    return r * np.cos(2.0 * np.pi * u2)


def _stream(key: str) -> int:
    return zlib.crc32(key.encode("utf-8"))


def make_state_dict(seed: int = WEIGHT_SEED) -> Dict[str, np.ndarray]:
    """All 582 entries of the reference ``state_dict`` as NumPy arrays (G1.2)."""
    sd: Dict[str, np.ndarray] = {}
    for key, shape, dtype, role in arch.manifest():
        n = int(np.prod(shape)) if shape else 1
        s = _stream(key)
        if role in ("conv_weight", "linear_weight"):
            fan_in = int(np.prod(shape[1:]))
            v = normal01(seed, s, n) * (GAIN / np.sqrt(fan_in))
        elif role == "conv_bias":
            v = normal01(seed, s, n) * 0.05
        elif role == "bn_weight":
            v = 0.8 + 0.4 * uniform01(seed, s, n)
        elif role == "bn_bias":
            v = normal01(seed, s, n) * 0.1
        elif role == "bn_mean":
            v = normal01(seed, s, n) * 0.1
        elif role == "bn_var":
            v = 0.8 + 0.4 * uniform01(seed, s, n)
        elif role == "gamma":
            v = np.full(n, 0.5)
        elif role == "bn_count":
            sd[key] = np.zeros(shape, dtype=np.int64)
            continue
        else:  # pragma: no cover - manifest roles are closed
            raise ValueError(role)
        sd[key] = v.astype(np.float32).reshape(shape)
    return sd


# Second fixture, "G1.2-B" (tests/golden/unet_g12b_b3.npz): other seeds, and the corners the first recipe leaves
# out -- attention gammas of both signs and very different sizes, BatchNorm channels whose running variance
# (1e-3) is small enough for eps = 1e-5 to matter in the fold, audio features four times larger, an odd batch.
WEIGHT_SEED_B = 0x5EED0B
INPUT_SEED_B = 0x1B0B
GAMMAS_B = (0.5, -0.75, 1.25, 0.1)
SMALL_VAR_B = 1e-3
AUDIO_SCALE_B = 4.0
BATCH_B = 3


def make_state_dict_b() -> Dict[str, np.ndarray]:
    """The G1.2-B weights: ``make_state_dict(WEIGHT_SEED_B)`` with GAMMAS_B on the four cross-attention blocks
    and every 7th BatchNorm channel (offset by the layer) at running_var = 1e-3; the BatchNorm weight of those
    channels is scaled by sqrt(1e-3) so that activations keep their size through the 60-odd layers."""
    sd = make_state_dict(WEIGHT_SEED_B)
    g = 0
    for key, shape, _dtype, role in arch.manifest():
        if role == "gamma":
            sd[key] = np.full(shape, GAMMAS_B[g], dtype=np.float32)
            g += 1
        elif role == "bn_var":
            sel = np.arange(shape[0]) % 7 == _stream(key) % 7
            sd[key] = np.where(sel, np.float32(SMALL_VAR_B), sd[key]).astype(np.float32)
            wkey = key[: -len("running_var")] + "weight"
            sd[wkey] = np.where(sel, sd[wkey] * np.float32(np.sqrt(SMALL_VAR_B)), sd[wkey]).astype(np.float32)
    assert g == len(GAMMAS_B)
    return sd


def make_inputs_b(batch: int = BATCH_B) -> Tuple[np.ndarray, np.ndarray]:
    x, a = make_inputs(batch, INPUT_SEED_B)
    return x, (a * np.float32(AUDIO_SCALE_B)).astype(np.float32)


def make_inputs(batch: int, seed: int = INPUT_SEED) -> Tuple[np.ndarray, np.ndarray]:
    """Synthetic frames: x ~ U(0,1) [B,6,160,160], audio ~ N(0,1) [B,32,32,32].

    Mirrors the reference self-benchmark's shapes
    (image_infer_v1/models/unet.py:342-347).  Frame ``b`` depends only on
    ``(seed, b)``, so any shard of a larger batch reproduces the same frames."""
    x = np.empty((batch, 6, arch.FACE_HW, arch.FACE_HW), dtype=np.float32)
    a = np.empty((batch, 32, arch.AUDIO_HW, arch.AUDIO_HW), dtype=np.float32)
    for b in range(batch):
        x[b] = uniform01(seed, 0x1000000 + b, x[b].size).astype(np.float32).reshape(x[b].shape)
        a[b] = normal01(seed, 0x2000000 + b, a[b].size).astype(np.float32).reshape(a[b].shape)
    return x, a


def make_inputs_range(start: int, count: int, seed: int = INPUT_SEED):
    """Frames [start, start+count) of the infinite synthetic stream."""
    x = np.empty((count, 6, arch.FACE_HW, arch.FACE_HW), dtype=np.float32)
    a = np.empty((count, 32, arch.AUDIO_HW, arch.AUDIO_HW), dtype=np.float32)
    for i in range(count):
        b = start + i
        x[i] = uniform01(seed, 0x1000000 + b, x[i].size).astype(np.float32).reshape(x[i].shape)
        a[i] = normal01(seed, 0x2000000 + b, a[i].size).astype(np.float32).reshape(a[i].shape)
    return x, a
