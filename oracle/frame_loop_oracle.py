"""CPU restatement of the frame loop's tensor glue around the model call -- TEST INFRASTRUCTURE ONLY.

Reference: ``FrameSynthesizer._get_audio_features`` and the model-input / prediction conversions of
``process_batch`` (image_infer_v1/tools/frame_synthesizer/infer_api.py:99-145, 238-245, 265-266).
Pure indexing (+ one IEEE division / multiplication), so parity is bit-exact.

PINNED (round 6) for the two functions the reference can run here: ``get_audio_features`` and ``FrameWalk`` equal
``tests/golden/frame_windows.npz`` / ``frame_walk.npz``, which ``tests/golden/make_frame_golden.py`` wrote by calling the
reference's own ``_get_audio_features`` / ``_generate_frame_sequence`` in the build container (the reference module's
``import cv2`` statement is satisfied by an attribute-less module object those two methods never enter; the script
asserts that).  ``tests/test_frame_golden.py`` holds the checks, ``tests/test_reference_live.py`` re-runs the reference
where it is mounted.  ``crops_to_model_input`` / ``predictions_to_uint8`` restate lines inside ``process_batch``, which
calls into OpenCV: those two stay pinned by source reading only.

``get_audio_features`` restates the reference statement by statement (same slicing, same truncated ``zeros_like`` pads,
same "reshape or fall back to zeros" rule) so that even its never-hit corners -- a clip shorter than the pad, an
index past the end, a negative index whose ``right`` becomes a from-the-end Python slice (idx = -T gives a NON-zero
window in the reference, and in the fixture) -- come out as the reference's own numpy/torch expressions do;
``audio_window_plan`` is the closed form of the same rule that the HIP kernel implements, and
``tests/test_frame_loop.py`` checks the two against each other over an exhaustive (idx, T) grid."""
from __future__ import annotations

import numpy as np


def get_audio_features(features: np.ndarray, indices) -> np.ndarray:
    """features [T, 2, 1024] -> windows [len(indices), 32, 32, 32] (fp32); literal restatement of
    infer_api.py:99-145 with numpy in place of torch (identical slicing semantics)."""
    batch = []
    for idx in indices:
        default = np.zeros((32, 32, 32), dtype=np.float32)          # :106
        added = False
        try:
            left, right = idx - 8, idx + 8                          # :111-112
            pad_left = pad_right = 0
            if left < 0:                                            # :116-118
                pad_left, left = -left, 0
            if right > features.shape[0]:                           # :119-121
                pad_right, right = right - features.shape[0], features.shape[0]
            auds = features[left:right]                             # :123
            if pad_left > 0:                                        # :125-126  zeros_like(auds[:pad]) is truncated
                auds = np.concatenate([np.zeros_like(auds[:pad_left]), auds], 0)
            if pad_right > 0:                                       # :127-128
                auds = np.concatenate([auds, np.zeros_like(auds[:pad_right])], 0)
            if auds.size >= 32 * 32 * 32:                           # :131-135
                batch.append(auds.reshape(32, 32, 32).astype(np.float32, copy=False))   # raises unless 16 rows
                added = True
        except Exception:                                           # :136-138
            pass
        if not added:                                               # :141-142
            batch.append(default)
    return np.array(batch, dtype=np.float32).reshape(len(batch), 32, 32, 32)


def audio_window_plan(idx: int, n_steps: int):
    """Closed form of the rule above (what audio_window_gather_kernel computes): (start, n0, pl, ok) --
    a valid window is ``pl`` zero rows, then features[start:start+n0], then zero rows up to 16."""
    left0, right0 = idx - 8, idx + 8
    pad_left, pad_right = max(0, -left0), max(0, right0 - n_steps)
    left, right = max(left0, 0), min(right0, n_steps)
    start = min(left, n_steps)
    stop = right if right >= 0 else max(n_steps + right, 0)
    stop = min(stop, n_steps)
    n0 = max(0, stop - start)
    pl = min(pad_left, n0)
    n1 = n0 + pl
    pr = min(pad_right, n1)
    return start, n0, pl, n1 + pr == 16


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


class FrameWalk:
    """Literal restatement of the frame-walk state machine of ``FrameSynthesizer._generate_frame_sequence``
    (infer_api.py:45-50, 147-190) with an injectable ``random``-like source -- the checker of
    ``calipsync_amd.frame_synth.FrameSynthesizer._generate_frame_sequence``."""

    def __init__(self, total_frames: int, rng):
        self.total_frames = total_frames
        self.rng = rng
        self.current_direction = None
        self.target_frame_count = 0
        self.processed_frame_count = 0
        self.current_frame_position = 0

    def generate(self, needed_frames: int) -> list:
        frame_sequence = []
        if self.processed_frame_count >= self.target_frame_count or self.current_direction is None:
            self.target_frame_count = self.total_frames * self.rng.randint(5, 15) // 100     # :160
            self.current_direction = self.rng.choice([1, -1])                                # :161
            self.processed_frame_count = 0
        while len(frame_sequence) < needed_frames:                                           # :165
            if self.current_direction == 1:
                available_frames = self.total_frames - self.current_frame_position
            else:
                available_frames = self.current_frame_position + 1
            seq_length = min(available_frames, needed_frames - len(frame_sequence))
            for _ in range(seq_length):
                frame_sequence.append(self.current_frame_position)
                self.current_frame_position += self.current_direction
                if self.current_frame_position >= self.total_frames:                         # :180-182
                    self.current_frame_position = self.total_frames - 2
                    self.current_direction = -1
                elif self.current_frame_position < 0:                                        # :183-185
                    self.current_frame_position = 1
                    self.current_direction = 1
        self.processed_frame_count += len(frame_sequence)                                    # :188
        return frame_sequence
