"""CPU restatement of the frame loop's audio-window builder -- TEST INFRASTRUCTURE ONLY.

Reference: ``FrameSynthesizer._get_audio_features``
(image_infer_v1/tools/frame_synthesizer/infer_api.py:99-145).  Pure indexing, so parity
is bit-exact.  The reference module itself cannot be imported here (``import cv2`` at its top,
cv2 absent), so this restatement is pinned by reading the source only; the arithmetic it does
(slice, zero-pad, reshape) has no rounding, and the property tests in
``tests/test_frame_loop.py`` check it from both ends (padding, interior, ordering)."""
from __future__ import annotations

import numpy as np


def get_audio_features(features: np.ndarray, indices) -> np.ndarray:
    """features [T, 2, 1024] -> windows [len(indices), 32, 32, 32] (fp32)."""
    out = np.zeros((len(indices), 32, 32, 32), dtype=np.float32)
    n = features.shape[0]
    for k, idx in enumerate(indices):
        left, right = idx - 8, idx + 8                  # infer_api.py:111-112
        pad_left = -left if left < 0 else 0             # :116-118
        left = max(left, 0)
        pad_right = right - n if right > n else 0       # :119-121
        right = min(right, n)
        if right <= left:                               # window entirely outside: the reference's
            continue                                    # reshape fails -> default zeros (:106,141-142)
        win = features[left:right]
        if pad_left:
            win = np.concatenate([np.zeros((pad_left,) + win.shape[1:], win.dtype), win], 0)
        if pad_right:
            win = np.concatenate([win, np.zeros((pad_right,) + win.shape[1:], win.dtype)], 0)
        if win.size >= 32 * 32 * 32:                    # :131-135
            out[k] = win.reshape(32, 32, 32)
    return out


def crops_to_model_input(crops168: np.ndarray) -> np.ndarray:
    """[B,168,168,3] uint8 -> [B,6,160,160] fp32 (infer_api.py:238-245).

    ``cv2.rectangle(img, (5, 5, 150, 145), (0,0,0), -1)`` fills the rectangle x,y,w,h = 5,5,150,145:
    columns 5..154, rows 5..149 (restated with slicing; cv2 is absent here)."""
    out = np.empty((crops168.shape[0], 6, 160, 160), dtype=np.float32)
    for b, crop in enumerate(crops168):
        real = crop[4:164, 4:164].copy()                       # :238
        masked = real.copy()                                   # :239
        masked[5:150, 5:155] = 0
        out[b, :3] = real.transpose(2, 0, 1).astype(np.float32) / 255.0      # :241
        out[b, 3:] = masked.transpose(2, 0, 1).astype(np.float32) / 255.0    # :242
    return out


def predictions_to_uint8(pred: np.ndarray) -> np.ndarray:
    """[B,3,160,160] fp32 -> [B,160,160,3] uint8 (infer_api.py:265-266)."""
    return np.stack([np.array(p.transpose(1, 2, 0) * 255, dtype=np.uint8) for p in pred])
