"""End-to-end frame-loop rate for bench.py --e2e: ``process_batch`` on synthetic 1080p frames (crop box ->
upload -> resize -> model -> paste-back blend -> ONE download -> host paste), device-gathered HuBERT windows.
Synthetic data only (there are no assets in the reference snapshot, .MISSING_LARGE_BLOBS)."""
from __future__ import annotations

import time

import numpy as np
import torch

from . import frame_loop


def synthetic_frames(n: int, h: int = 1080, w: int = 1920, seed: int = 0):
    """n random BGR frames with a 110-point landmark set (points 0..32 = jaw contour, 1 / 31 / 52 = crop box)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    imgs, lms = [], []
    theta = np.linspace(np.pi * 1.02, np.pi * 1.98, 33)
    for i in range(n):
        r = rng.uniform(0.16, 0.2) * h
        cx, cy = rng.uniform(0.4, 0.6) * w, rng.uniform(0.4, 0.5) * h
        p = np.zeros((110, 2))
        p[:33, 0] = cx + r * np.cos(theta)
        p[:33, 1] = cy - 0.15 * r - 1.25 * r * np.sin(theta)
        p[33:] = [cx, cy]
        p[52] = [cx, cy - 0.35 * r]
        imgs.append(np.roll(base, i * 7, axis=1))
        lms.append(p)
    return imgs, lms


def run(net, dev, batch: int = 64, batches: int = 8, warmup: int = 2) -> dict:
    imgs, lms = synthetic_frames(batch)
    no_masks = [None] * batch
    # the reference's masks: one whole-frame gray image per frame (infer_api.py:65-70); here uint8, named per frame
    rng = np.random.default_rng(3)
    u8_masks = [rng.integers(0, 256, imgs[0].shape[:2], dtype=np.uint8) for _ in range(4)]
    u8_masks = [u8_masks[i % 4] for i in range(batch)]
    feats = torch.from_numpy(np.random.default_rng(1).standard_normal((batch * (batches + warmup) + 16, 2, 1024))
                             .astype(np.float32)).to(dev)
    def timed(pipelined: bool, copy_frames: bool = True, with_masks: bool = False) -> float:
        masks = u8_masks if with_masks else no_masks
        keys = [("bench", i) for i in range(batch)] if with_masks else None
        prev = None
        n_out = 0
        for k in range(warmup + batches):
            if k == warmup:
                if prev is not None:
                    n_out += len(prev.result())
                    prev = None
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                n_out = 0
            idx = list(range(k * batch, (k + 1) * batch))
            cur = frame_loop.submit_batch_device(net, imgs, lms, masks, features=feats, frame_indices=idx,
                                                 copy_frames=copy_frames, mask_keys=keys)
            if pipelined:          # FrameSynthesizer.iterate_synthesized_frames: one batch in flight
                if prev is not None:
                    n_out += len(prev.result())
                prev = cur
            else:                  # process_batch: submit and wait, batch by batch
                n_out += len(cur.result())
        if prev is not None:
            n_out += len(prev.result())
        torch.cuda.synchronize(dev)
        assert n_out == batch * batches, n_out
        return time.perf_counter() - t0

    # host-bound (frame copies, slicing) and therefore noisy: every mode runs three times, alternating, and reports
    # its MEDIAN run
    med = lambda v: sorted(v)[len(v) // 2]
    seq, pipe, inplace, masked = [], [], [], []
    for _ in range(3):
        seq.append(timed(False))
        pipe.append(timed(True))
        inplace.append(timed(True, False))
        masked.append(timed(True, True, True))
    dt_seq, dt_pipe, dt_inplace, dt_masked = med(seq), med(pipe), med(inplace), med(masked)
    side = int(np.mean([int(l[31][0]) - int(l[1][0]) for l in lms]))
    return {"frames_per_s": round(batch * batches / dt_pipe, 1), "ms_per_batch": round(1e3 * dt_pipe / batches, 2),
            "frames_per_s_batch_by_batch": round(batch * batches / dt_seq, 1),
            "ms_per_batch_batch_by_batch": round(1e3 * dt_seq / batches, 2),
            "frames_per_s_in_place": round(batch * batches / dt_inplace, 1),     # copy_frames=False: not the reference's contract
            "frames_per_s_with_masks": round(batch * batches / dt_masked, 1),    # whole-frame uint8 masks, named: resident after
                                                                                 # their first batch
            "batch": batch,
            "frame": "1920x1080 BGR uint8, synthetic", "mean_crop_side_px": side,
            "pipeline": "host crop-box slices into one pinned buffer -> 1 H2D -> resize168 -> forward_windows -> uint8 -> "
                        "resize back -> fillPoly -> dilate -> blend -> 1 D2H (pinned) -> paste into threaded frame copies "
                        "(calipsync_amd.frame_loop.submit_batch_device / PendingBatch.result); frames_per_s = one batch "
                        "in flight as FrameSynthesizer.iterate_synthesized_frames runs it, batch_by_batch = process_batch",
            "frames_out": batch * batches, "repeats": "median of 3 alternating runs of 8 batches per mode"}
