"""Synthetic stand-ins for the reference's infer_data (frames / positions / masks): random frames with a
plausible 110-point landmark set whose first 33 points trace the jaw contour."""
import os

import numpy as np


def golden_features(t: int, seed: int = 3) -> np.ndarray:
    """The [t, 2, 1024] fp32 feature arrays behind tests/golden/frame_windows.npz, values in [-1, 1): 24 hashed bits
    per value and exact arithmetic only (integer hashing, one exact int->float conversion, one division by 2^23), so
    every box regenerates the same bits; the fixture carries their SHA-256."""
    from calipsync_amd import recipe
    bits = recipe._bits(seed, 0x77696E, t * 2048, lane=t)
    v = (bits >> np.uint64(40)).astype(np.int64) - (1 << 23)
    return (v.astype(np.float32) / np.float32(1 << 23)).reshape(t, 2, 1024)


def landmarks(cx: float, cy: float, r: float, rng, jitter: float = 1.5) -> np.ndarray:
    """110 x 2 float landmarks: points 0..32 run along the lower face contour from the left temple over the
    chin to the right temple; point 52 sits above the mouth (its y is the crop's top), points 1 and 31 give
    the crop's left / right edge (infer_api.py:206-208)."""
    lms = np.zeros((110, 2), dtype=np.float64)
    theta = np.linspace(np.pi * 1.02, np.pi * 1.98, 33)          # left -> bottom -> right (y grows downwards)
    lms[:33, 0] = cx + r * np.cos(theta)
    lms[:33, 1] = cy - 0.15 * r - 1.25 * r * np.sin(theta)
    lms[:33] += rng.normal(0, jitter, (33, 2))
    lms[33:] = np.stack([cx + rng.normal(0, r / 3, 77), cy + rng.normal(0, r / 3, 77)], 1)
    lms[52] = [cx, cy - 0.35 * r]
    return lms


def make_frames(n: int, h: int = 540, w: int = 720, seed: int = 0, with_masks: bool = False):
    rng = np.random.default_rng(seed)
    imgs, lms, masks = [], [], []
    for i in range(n):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[::7] //= 2                                            # some structure
        r = rng.uniform(0.18, 0.3) * min(h, w)
        cx = rng.uniform(0.35, 0.65) * w
        cy = rng.uniform(0.35, 0.55) * h
        imgs.append(img)
        lms.append(landmarks(cx, cy, r, rng))
        if with_masks and i % 2 == 0:
            mh, mw = (h, w) if i % 4 == 0 else (h // 3, w // 3)
            masks.append(rng.random((mh, mw), dtype=np.float32))
        else:
            masks.append(None)
    return imgs, lms, masks


def write_dataset(root: str, n: int, h: int = 270, w: int = 360, seed: int = 0):
    """frames/NNNNNN.npy + positions/NNNNNN.txt (+ every third masks/NNNNNN.npy)."""
    imgs, lms, _ = make_frames(n, h, w, seed)
    for d in ("frames", "positions", "masks"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rng = np.random.default_rng(seed + 1)
    for i in range(n):
        name = str(i).zfill(6)
        np.save(os.path.join(root, "frames", name + ".npy"), imgs[i])
        np.savetxt(os.path.join(root, "positions", name + ".txt"), lms[i])
        if i % 3 == 0:
            np.save(os.path.join(root, "masks", name + ".npy"), (rng.random((h, w)) * 255).astype(np.uint8))
    return imgs, lms
