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
