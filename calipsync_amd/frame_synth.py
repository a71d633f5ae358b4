"""Drop-in host side of the reference frame loop on the MI355X engine.

``FrameSynthesizer`` keeps the reference's constructor, attributes and methods
(image_infer_v1/tools/frame_synthesizer/infer_api.py:12-455):

    FrameSynthesizer(unet_checkpoint, data_dir, device="cuda:0", batch_size=8)
    .iterate_synthesized_frames(features, start_frame_idx=0, is_generate_sync_frame=True)
        -> iterator of {'frame', 'index', 'physical_index'}
    .process_batch(batch_images, batch_landmarks, batch_masks, hubert_features) -> list of uint8 frames

What changes underneath: the model is ``calipsync_amd.unet.Model`` (HIP engine), and ``process_batch``
runs crop + resize + model + paste-back blend on the GPU (``calipsync_amd.frame_loop``) with one upload
and ONE download per batch instead of per-frame cv2 calls and B separate ``.cpu()`` copies
(infer_api.py:200-253, 263-346).  Inside ``iterate_synthesized_frames`` the clip's HuBERT array is
uploaded once and the 16-step windows are gathered on the device (``Model.forward_windows``) instead
of being materialised on the host per batch (infer_api.py:99-145, 257).

Frame sequencing (the ping-pong "motion generalisation" walk, infer_api.py:147-190) is
``frame_walk.PingPongWalk``: the same sequence for the same random draws, produced from one phase counter; the
only addition is an optional ``seed`` so a run can be reproduced (the reference draws from the unseeded
module-level ``random``).  File I/O (``cv2.imread`` /
``np.loadtxt``, infer_api.py:52-97) stays host code behind the same methods; where cv2 is not
installed JPEGs are decoded with Pillow (same libjpeg, BGR order restored).
"""
from __future__ import annotations

import os
import random
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import frame_loop
from .frame_walk import PingPongWalk
from .unet import Model


def _imread(path: str, gray: bool = False) -> Optional[np.ndarray]:
    """cv2.imread replacement: BGR uint8 HxWx3 (or HxW for gray); None when the file cannot be read."""
    try:
        import cv2  # noqa: WPS433  (optional: the reference's own decoder when it is installed)
        return cv2.imread(path, cv2.IMREAD_GRAYSCALE if gray else cv2.IMREAD_COLOR)
    except ImportError:
        pass
    if path.endswith(".npy"):
        return np.load(path)
    try:
        from PIL import Image
        with Image.open(path) as im:
            if gray:
                return np.asarray(im.convert("L"))
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except Exception:
        return None


class FrameSynthesizer:
    def __init__(self, unet_checkpoint: Optional[str], data_dir: str, device: str = "cuda:0", batch_size: int = 8, *,
                 seed: Optional[int] = None, precision: str = "fp32", net: Optional[Model] = None,
                 batches_in_flight: Optional[int] = None):
        """Same positional arguments as the reference (infer_api.py:13-14).  Keyword-only extensions:
        ``seed`` (reproducible frame walk), ``precision`` (engine storage type), ``net`` (an already
        loaded ``Model`` instead of a checkpoint path), ``batches_in_flight`` (default 1, or the
        ``CASYNC_BATCHES_IN_FLIGHT`` environment variable): how many enqueued batches ``iterate_synthesized_frames``
        leaves on the GPU while it LOADS the next one.  1 hides the per-batch file I/O (JPEG decode + ``np.loadtxt`` of
        ``batch_size`` frames, the reference's ``_load_batch_frames``) behind the device work; 0 is the reference's own
        batch-by-batch order.  With frames already in memory there is nothing to hide, and on a small host share the
        queued batch's host copies compete with the one being collected: ``bench.py``'s synthetic ``e2e`` block
        measures both (4.4-4.5 k with one in flight, 4.8-4.9 k batch by batch on 16 cores) -- pick 0 there."""
        if batches_in_flight is None:
            batches_in_flight = int(os.environ.get("CASYNC_BATCHES_IN_FLIGHT", "1"))
        if batches_in_flight < 0:
            raise ValueError("batches_in_flight must be >= 0")
        self.batches_in_flight = batches_in_flight
        self.device = device
        self.data_dir = data_dir
        self.batch_size = batch_size
        self.frames_dir = os.path.join(data_dir, "frames")
        self.positions_dir = os.path.join(data_dir, "positions")
        self.masks_dir = os.path.join(data_dir, "masks")
        # infer_api.py:34 counts the .jpg frames; .npy frames are accepted too (synthetic data sets)
        names = os.listdir(self.frames_dir)
        self._ext = ".jpg" if any(f.endswith(".jpg") for f in names) else ".npy"
        self.total_frames = len([f for f in names if f.endswith(self._ext)])
        self.executor = ThreadPoolExecutor(max_workers=self.batch_size)   # infer_api.py:38
        if net is None:                                                   # infer_api.py:41-43
            net = Model(6, "hubert", precision=precision).to(device)
            net.load_state_dict(torch.load(unet_checkpoint, map_location="cpu"))
        self.net = net.eval()
        # frame-walk state (infer_api.py:46-50); the reference draws from the module-level random
        self._walk = PingPongWalk(self.total_frames, random.Random(seed) if seed is not None else None)
        self.last_logical_index = -1
        self._features_dev = None        # (id of the host array, device copy) of the clip being synthesised

    # ------------------------------------------------------------------ file I/O (infer_api.py:52-97)
    def _load_single_frame(self, frame_idx: int, raw_mask: bool = False) -> tuple:
        """(image, landmarks, mask) of a frame, as the reference returns them (mask = gray image / 255 in float32).
        ``raw_mask`` (this loop's own calls): the mask stays the uint8 image -- the device divides by 255 in float32,
        the same bits at a quarter of the bytes, and the host skips an 8 MB conversion per 1080p frame."""
        stem = f"{frame_idx % self.total_frames:06d}"
        img = _imread(os.path.join(self.frames_dir, stem + self._ext))
        lms = np.loadtxt(os.path.join(self.positions_dir, stem + ".txt"))
        mask_path = next((p for p in (os.path.join(self.masks_dir, stem + ext) for ext in (".jpg", ".npy")) if os.path.exists(p)), None)
        mask = _imread(mask_path, gray=True) if mask_path else None
        if mask is not None and not (raw_mask and mask.dtype == np.uint8):
            mask = mask.astype(np.float32) / 255.0          # float32 division, as the reference (infer_api.py:68)
        return img, lms, mask

    def _load_batch_frames(self, frame_indices: list, raw_masks: bool = False) -> tuple:
        """(images, landmarks, masks) of the frames, decoded on the I/O pool (infer_api.py:80-97)."""
        loaded = list(self.executor.map(lambda i: self._load_single_frame(i, raw_masks), frame_indices))
        images, landmarks, masks = (list(column) for column in zip(*loaded)) if loaded else ([], [], [])
        return images, landmarks, masks

    # ------------------------------------------------------------------ audio windows (infer_api.py:99-145)
    def _get_audio_features(self, features: np.ndarray, indices: list) -> np.ndarray:
        """Host-side windows [len(indices), 32, 32, 32] with the reference's exact rule (truncated
        ``zeros_like`` pads, all-zero default window when the padded slice misses 16 rows).  The device
        path does not call this (``Model.forward_windows`` gathers on the GPU); it exists for callers of
        ``process_batch`` that hold host windows, as the reference's own loop does."""
        return frame_loop.audio_windows_host(features, indices)

    # ------------------------------------------------------------------ frame walk (infer_api.py:147-190)
    def _generate_frame_sequence(self, needed_frames: int) -> list:
        """The next ``needed_frames`` stored-frame indices of the ping-pong replay (``frame_walk.PingPongWalk``)."""
        return self._walk.take(needed_frames)

    @property
    def current_frame_position(self) -> int:      # the reference's attribute names, read-only views of the walk
        return self._walk.position

    @property
    def processed_frame_count(self) -> int:
        return self._walk.run_used

    @property
    def target_frame_count(self) -> int:
        return self._walk.run_budget

    # ------------------------------------------------------------------ the batch (infer_api.py:192-357)
    def process_batch(self, batch_images: list, batch_landmarks: list, batch_masks: list,
                      hubert_features: np.ndarray) -> list:
        """Reference signature.  ``hubert_features``: host windows [B,32,32,32].  Like the reference,
        any failure returns the un-synced originals (infer_api.py:352-357) -- after printing it."""
        try:
            windows = torch.from_numpy(np.ascontiguousarray(hubert_features, dtype=np.float32)).to(self.device)
            return frame_loop.process_batch_device(self.net, batch_images, batch_landmarks, batch_masks, windows=windows)
        except Exception as exc:   # the reference swallows everything here; keep the contract, say why
            print(f"process_batch failed, returning the original frames: {exc!r}")
            return batch_images

    def _submit_batch_indices(self, batch_images, batch_landmarks, batch_masks, features_dev, indices, mask_keys=None):
        """Enqueue a batch (frame_loop.submit_batch_device); a failure keeps the reference's contract of handing
        back the originals (infer_api.py:352-357) when the batch is collected."""
        try:
            return frame_loop.submit_batch_device(self.net, batch_images, batch_landmarks, batch_masks,
                                                  features=features_dev, frame_indices=indices,
                                                  mask_keys=mask_keys), batch_images
        except Exception as exc:
            print(f"process_batch failed, returning the original frames: {exc!r}")
            return None, batch_images

    @staticmethod
    def _collect(pending, originals) -> list:
        if pending is None:
            return originals
        try:
            return pending.result()
        except Exception as exc:
            print(f"process_batch failed, returning the original frames: {exc!r}")
            return originals

    # ------------------------------------------------------------------ the loop (infer_api.py:359-451)
    def iterate_synthesized_frames(self, features: np.ndarray, start_frame_idx: int = 0,
                                   is_generate_sync_frame: bool = True) -> Iterator[Dict]:
        self.last_logical_index = start_frame_idx - 1
        time_stats = {"load_frame": 0.0, "get_audio": 0.0, "process_batch": 0.0}
        total_frames = len(features)
        features_dev = None
        in_flight: list = []
        try:
            if is_generate_sync_frame and total_frames:
                t0 = time.time()   # one upload of the whole [T,2,1024] array replaces B x 128 KB per batch
                features_dev = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(self.device)
                time_stats["get_audio"] += time.time() - t0
            for batch_start in range(0, total_frames, self.batch_size):
                try:
                    batch_end = min(batch_start + self.batch_size, total_frames)    # variable last batch
                    frame_sequence = self._generate_frame_sequence(batch_end - batch_start)
                    t0 = time.time()
                    batch_images, batch_landmarks, batch_masks = self._load_batch_frames(
                        frame_sequence, raw_masks=is_generate_sync_frame)
                    time_stats["load_frame"] += time.time() - t0
                    if not is_generate_sync_frame:      # pass-through mode: the stored frames, numbered
                        yield from self._emit(batch_images, frame_sequence)
                        continue
                    # `batches_in_flight` (default 1): batch k+1 is loaded, cropped and enqueued while the GPU works on
                    # batch k, whose frames are yielded afterwards -- same frames, same order; 0 = collect each batch at once
                    t0 = time.time()
                    # masks are named by their file (frame number modulo the clip): each is uploaded once and stays
                    # on the device while the walk comes back to it
                    pending, originals = self._submit_batch_indices(
                        batch_images, batch_landmarks, batch_masks, features_dev, list(range(batch_start, batch_end)),
                        mask_keys=[(self.data_dir, f % self.total_frames) for f in frame_sequence])
                    in_flight.append((pending, originals, frame_sequence))
                    time_stats["process_batch"] += time.time() - t0
                    while len(in_flight) > self.batches_in_flight:
                        yield from self._drain_one(in_flight, time_stats)
                except Exception as exc:        # a failed batch is skipped, the iterator goes on (:429-436)
                    print(f"batch starting at {batch_start} failed: {exc!r}")
                    time.sleep(0.1)
                    continue
            while in_flight:
                yield from self._drain_one(in_flight, time_stats)
        except Exception as exc:                # fatal: one black frame so the consumer does not hang (:438-446)
            print(f"frame iterator failed: {exc!r}")
            yield from self._emit([np.zeros((480, 640, 3), dtype=np.uint8)], [0])
        finally:
            total_time = sum(time_stats.values())
            if total_time > 0:
                print(f"average frame rate: {total_frames / total_time:.2f} FPS "
                      f"(load {time_stats['load_frame']:.2f} s, audio {time_stats['get_audio']:.2f} s, "
                      f"batches {time_stats['process_batch']:.2f} s)")

    def _drain_one(self, in_flight: list, time_stats: dict):
        pending, originals, frame_sequence = in_flight.pop(0)
        t0 = time.time()
        processed = self._collect(pending, originals)
        time_stats["process_batch"] += time.time() - t0
        yield from self._emit(processed, frame_sequence)

    def _emit(self, frames, physical_indices):
        """The iterator's items (infer_api.py:400-405): consecutive logical indices over whatever is handed out."""
        for frame, physical in zip(frames, physical_indices):
            self.last_logical_index += 1
            yield {"frame": frame, "index": self.last_logical_index, "physical_index": physical}

    def __del__(self):
        if hasattr(self, "executor"):
            self.executor.shutdown()


class VideoStreamManager:
    """Offline driver with the reference's signature (inference.py:14-20, 47): features -> frames -> file.

    HuBERT extraction is outside the hot-path contract (the north star consumes
    pre-extracted windows; no HuBERT weights ship), so ``hubert_path`` may be a callable
    ``audio_path -> [T,2,1024] array`` or is ignored when ``audio_path`` is itself a ``.npy`` of
    features.  The mp4 writer / ffmpeg mux (inference.py:88-110) is used when cv2 / ffmpeg exist; otherwise
    the frames are written as a Motion-JPEG ``.avi`` with Pillow (``mjpeg_avi.py``; no audio track)."""

    def __init__(self, data_dir: str, unet_checkpoint: Optional[str], hubert_path=None, device: str = "cuda:0",
                 batch_size: int = 8, output_sample_rate: int = 24000, **synth_kwargs):
        self.synthesizer = FrameSynthesizer(unet_checkpoint=unet_checkpoint, data_dir=data_dir, device=device,
                                            batch_size=batch_size, **synth_kwargs)
        self.hubert_extractor = hubert_path if callable(hubert_path) else None
        self.feature_sample_rate = 16000
        self.output_sample_rate = output_sample_rate
        self.fps = 25

    def process_single_file(self, audio_path: str, output_path: str):
        if audio_path.endswith(".npy"):
            features = np.load(audio_path)
        elif self.hubert_extractor is not None:
            features = self.hubert_extractor(audio_path)
        else:
            raise RuntimeError("no HuBERT extractor configured: pass pre-extracted features (.npy) or a callable")
        frames = [info["frame"] for info in self.synthesizer.iterate_synthesized_frames(features, 0, True)]
        if not frames:
            raise ValueError("no video frame was generated")
        try:
            import cv2
        except ImportError:
            # no OpenCV in this environment: a Motion-JPEG AVI written with Pillow (calipsync_amd/mjpeg_avi.py; no audio
            # track -- muxing needs the ffmpeg binary), or the raw frames when Pillow is missing too
            try:
                from . import mjpeg_avi
                out = os.path.splitext(output_path)[0] + ".avi"
                print(f"VideoStreamManager: no OpenCV / ffmpeg here -- writing {out} (Motion-JPEG AVI, {self.fps} fps, NO audio track) "
                      f"instead of {output_path}")
                mjpeg_avi.write_mjpeg_avi(out, frames, fps=self.fps)
                return out
            except (ImportError, ValueError) as exc:      # no Pillow, or a clip past the 4 GiB AVI limit: the raw frames
                print(f"VideoStreamManager: {exc}; writing the raw frames to {output_path}.npy")
                np.save(output_path + ".npy", np.stack(frames))
                return output_path + ".npy"
        height, width = frames[0].shape[:2]
        temp = output_path.replace(".mp4", "_temp.mp4")
        writer = cv2.VideoWriter(temp, cv2.VideoWriter_fourcc(*"mp4v"), self.fps, (width, height))
        for frame in frames:
            writer.write(frame)
        writer.release()
        import shutil
        import subprocess
        if shutil.which("ffmpeg") and not audio_path.endswith(".npy"):
            subprocess.run(["ffmpeg", "-y", "-i", temp, "-i", audio_path, "-c:v", "copy", "-c:a", "aac", "-strict",
                            "experimental", "-map", "0:v:0", "-map", "1:a:0", output_path], capture_output=True)
            os.remove(temp)
        else:
            os.replace(temp, output_path)
        return output_path
