"""The order in which stored frames are replayed under new audio.

The reference walks back and forth over the clip's frames and re-draws, once per batch, how long it keeps going
before it may turn round (``FrameSynthesizer._generate_frame_sequence``,
image_infer_v1/tools/frame_synthesizer/infer_api.py:147-190; state at :45-50).  Seen from outside that is a
*triangle wave with random re-phasing*:

* the frames visited are ``tri(u) = u if u < T else P - u`` for a phase ``u`` that advances by one per frame
  modulo the period ``P = 2 (T - 1)`` (both ends are visited once per period, no frame is repeated at a turn);
* when the current run is used up (its budget is ``T * randint(5, 15) // 100`` frames, checked only at the start
  of a request) a direction is drawn, ``choice([1, -1])`` -- which, at frame ``p``, selects one of the two phases
  that sit on ``p``: ``u = p`` (ascending) or ``u = P - p`` (descending).

So the walk keeps ONE integer of position state and produces a whole request with array arithmetic; there is no
per-frame loop and no separate direction flag to keep consistent.  The random draws are made in the reference's
order (run length first, then direction) from a ``random``-like source, so a seeded run reproduces the reference's
sequence exactly; ``oracle/frame_loop_oracle.FrameWalk`` (a literal restatement of the reference's loop) is the
checker (tests/test_frame_ops.py).
"""
from __future__ import annotations

import random
from typing import List, Optional

import numpy as np


class PingPongWalk:
    def __init__(self, total_frames: int, rng=None):
        """``rng``: anything with ``randint(a, b)`` and ``choice(seq)`` (``random.Random(seed)`` for a reproducible
        walk; default: the module-level ``random``, as the reference uses)."""
        if total_frames < 1:
            raise ValueError("a clip needs at least one frame")
        self.total_frames = int(total_frames)
        self.period = max(2 * (self.total_frames - 1), 1)   # a one-frame clip stands still (the reference's loop never ends there)
        self.rng = rng if rng is not None else random
        self.phase = 0
        self.run_budget = 0          # frames the current run may last
        self.run_used = 0            # frames handed out since the last draw
        self._drawn = False

    # ---- position -------------------------------------------------------------------------------------------
    def _tri(self, u):
        return np.where(u < self.total_frames, u, self.period - u)

    @property
    def position(self) -> int:
        """The frame the next request starts with."""
        return int(self._tri(np.int64(self.phase))) if self.total_frames > 1 else 0

    @property
    def ascending(self) -> Optional[bool]:
        """Whether the frame after ``position`` has the larger index (None before the first request)."""
        if not self._drawn:
            return None
        return self.phase < self.total_frames - 1

    # ---- requests -------------------------------------------------------------------------------------------
    def _redraw(self) -> None:
        pos = self.position
        self.run_budget = self.total_frames * self.rng.randint(5, 15) // 100
        up = self.rng.choice([1, -1]) == 1
        self.phase = pos if up else (self.period - pos) % self.period
        self.run_used = 0
        self._drawn = True

    def take(self, count: int) -> List[int]:
        """The next ``count`` frame indices."""
        if not self._drawn or self.run_used >= self.run_budget:
            self._redraw()
        if count <= 0:
            return []
        if self.total_frames == 1:
            frames = [0] * count
        else:
            u = (self.phase + np.arange(count, dtype=np.int64)) % self.period
            frames = self._tri(u).tolist()
            self.phase = int((self.phase + count) % self.period)
        self.run_used += count
        return frames
