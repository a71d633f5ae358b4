"""Motion-JPEG AVI writer without OpenCV / ffmpeg: the offline driver's fallback container.

The reference saves its frames with ``cv2.VideoWriter(..., 'mp4v', fps, (w, h))`` and muxes the audio with the ``ffmpeg``
binary (inference.py:88-110).  Neither exists in this image; ``VideoStreamManager`` uses them when they do.  Without them
the frames used to go to a ``.npy`` file only; this module writes them as a playable video instead: every frame is
JPEG-encoded with Pillow and stored as one ``00dc`` chunk of a RIFF/AVI file ('MJPG' stream, an ``idx1`` index, fixed
frame rate).  No audio track (muxing needs an encoder this image does not have).  Host code only.
"""
from __future__ import annotations

import io
import struct
from typing import Iterable

import numpy as np


def _chunk(fourcc: bytes, data: bytes) -> bytes:
    return fourcc + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")


def _list(kind: bytes, data: bytes) -> bytes:
    return b"LIST" + struct.pack("<I", len(data) + 4) + kind + data


def encode_jpeg(frame_bgr: np.ndarray, quality: int = 95) -> bytes:
    """One BGR uint8 HxWx3 frame (the frame loop's layout, cv2 order) as a baseline JPEG."""
    from PIL import Image
    if frame_bgr.dtype != np.uint8 or frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3:
        raise ValueError(f"frames must be uint8 HxWx3 (BGR), got {frame_bgr.dtype} {frame_bgr.shape}")
    buf = io.BytesIO()
    # 4:4:4 (no chroma subsampling): the synthesised mouth region is small and colour edges matter more than file size
    Image.fromarray(np.ascontiguousarray(frame_bgr[:, :, ::-1])).save(buf, format="JPEG", quality=quality, subsampling=0)
    return buf.getvalue()


RIFF_LIMIT = (1 << 32) - (1 << 20)     # 32-bit RIFF sizes (plain AVI, no OpenDML segments): stop a MiB short of 4 GiB


def write_mjpeg_avi(path: str, frames: Iterable[np.ndarray], fps: float = 25.0, quality: int = 95) -> int:
    """Write `frames` (BGR uint8, all the same size) to `path` as Motion-JPEG AVI; returns the frame count.

    Frames are encoded and written one at a time (one JPEG in memory, not the clip); the header fields that depend on
    the whole clip (frame count, largest chunk, RIFF / LIST sizes) are patched when the last frame is on disk.  The
    container is plain RIFF/AVI with 32-bit sizes: a clip that would pass 4 GiB (about five minutes of 1080p at this
    quality) raises ValueError at the frame that would cross the limit, and the partial file is removed -- it fails while
    encoding, not after it."""
    usec = int(round(1e6 / fps))
    rate, scale = int(round(fps * 1000)), 1000
    n, biggest, width, height = 0, 0, None, None
    index = []
    fh = None
    try:
        for f in frames:
            if width is None:
                height, width = f.shape[:2]
                fh = open(path, "wb")
                # header with the clip-dependent fields zeroed: patched below (positions recorded as they are written)
                avih = struct.pack("<14I", usec, 0, 0, 0x10, 0, 0, 1, 0, width, height, 0, 0, 0, 0)   # 0x10 = AVIF_HASINDEX
                strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, 0, 0, 0xFFFFFFFF, 0, 0, 0, width, height)
                strf = struct.pack("<IiiHH4sIiiII", 40, width, height, 1, 24, b"MJPG", width * height * 3, 0, 0, 0, 0)
                hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", strf)))
                fh.write(b"RIFF" + struct.pack("<I", 0) + b"AVI " + hdrl)
                movi_at = fh.tell()
                fh.write(b"LIST" + struct.pack("<I", 0) + b"movi")
                off = 4                                              # idx1 offsets count from the 'movi' fourcc
            elif f.shape[:2] != (height, width):
                raise ValueError("all frames of a video must have the same size")
            j = encode_jpeg(f, quality)
            chunk = _chunk(b"00dc", j)
            if fh.tell() + len(chunk) + 16 * (n + 1) + 8 > RIFF_LIMIT:
                raise ValueError(f"{path}: the clip passes the 4 GiB limit of a plain AVI file at frame {n} "
                                 f"({fh.tell() / 2 ** 30:.2f} GiB written): write it in shorter segments")
            fh.write(chunk)
            index.append((off, len(j)))
            off += len(chunk)
            biggest = max(biggest, len(j))
            n += 1
        if n == 0:
            raise ValueError("no video frame was generated")
        movi_size = fh.tell() - movi_at - 8
        fh.write(_chunk(b"idx1", b"".join(b"00dc" + struct.pack("<III", 0x10, o, ln) for o, ln in index)))   # 0x10 = AVIIF_KEYFRAME
        riff_size = fh.tell() - 8
        # patches: RIFF size; avih (at 32): dwMaxBytesPerSec (+4), dwTotalFrames (+16), dwSuggestedBufferSize (+28);
        # strh (at 108): dwLength (+32), dwSuggestedBufferSize (+36); the movi LIST size
        for pos, value in ((4, riff_size), (32 + 4, int(biggest * fps)), (32 + 16, n), (32 + 28, biggest),
                           (108 + 32, n), (108 + 36, biggest), (movi_at + 4, movi_size)):
            fh.seek(pos)
            fh.write(struct.pack("<I", value))
        fh.close()
        fh = None
        return n
    finally:
        if fh is not None:              # an error part-way: no half-written file is left behind
            fh.close()
            try:
                import os
                os.remove(path)
            except OSError:
                pass


def read_mjpeg_avi(path: str):
    """(fps, [BGR frames]) of a file written by `write_mjpeg_avi` (tests; a minimal RIFF walk, not a general AVI reader)."""
    from PIL import Image
    data = open(path, "rb").read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("not a RIFF/AVI file")
    fps, frames = None, []

    def walk(lo: int, hi: int) -> None:
        nonlocal fps
        pos = lo
        while pos + 8 <= hi:
            cc, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            body = pos + 8
            if cc == b"LIST":
                walk(body + 4, body + size)
            elif cc == b"strh":
                scale, rate = struct.unpack("<II", data[body + 20:body + 28])
                fps = rate / scale
            elif cc == b"00dc":
                im = Image.open(io.BytesIO(data[body:body + size])).convert("RGB")
                frames.append(np.ascontiguousarray(np.asarray(im)[:, :, ::-1]))
            pos = body + size + (size & 1)

    walk(12, len(data))
    return fps, frames
