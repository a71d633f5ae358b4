"""The offline driver's container fallback (calipsync_amd/mjpeg_avi.py): where cv2 / ffmpeg are absent the frames of
inference.py:88-110 are written as Motion-JPEG AVI with Pillow.  Host code only."""
import os
import struct

import numpy as np
import pytest

from calipsync_amd import mjpeg_avi


def _frames(n, h=96, w=128):
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):      # smooth content (JPEG-friendly) that differs per frame and per channel
        f = np.stack([(xx * 2 + 5 * i) % 256, (yy * 2 + 3 * i) % 256, ((xx + yy) + 7 * i) % 256], -1).astype(np.uint8)
        out.append(np.ascontiguousarray(f))
    return out


def test_mjpeg_avi_round_trip(tmp_path):
    frames = _frames(7)
    path = str(tmp_path / "clip.avi")
    assert mjpeg_avi.write_mjpeg_avi(path, frames, fps=25) == 7
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    assert raw.count(b"00dc") >= 14 and b"idx1" in raw and b"MJPG" in raw          # 7 chunks + 7 index entries
    fps, back = mjpeg_avi.read_mjpeg_avi(path)
    assert fps == 25.0 and len(back) == 7
    for a, b in zip(frames, back):
        assert b.shape == a.shape and b.dtype == np.uint8
        assert np.abs(a.astype(int) - b.astype(int)).mean() < 4.0                  # JPEG at quality 95, BGR order kept
    assert np.abs(back[0].astype(int) - frames[3].astype(int)).mean() > 4.0        # frames are distinct and in order


def test_mjpeg_avi_rejects_bad_input(tmp_path):
    with pytest.raises(ValueError):
        mjpeg_avi.write_mjpeg_avi(str(tmp_path / "e.avi"), [])
    with pytest.raises(ValueError):
        mjpeg_avi.write_mjpeg_avi(str(tmp_path / "m.avi"), [_frames(1)[0], _frames(1, h=64)[0]])
    with pytest.raises(ValueError):
        mjpeg_avi.encode_jpeg(np.zeros((8, 8), np.uint8))


def test_mjpeg_avi_streams_and_fails_early_at_the_size_limit(tmp_path, monkeypatch):
    """frames are written as they are encoded (a generator is enough), the clip-dependent header fields are patched
    at the end, and a clip that would pass the 32-bit RIFF limit fails at THAT frame -- not after encoding everything -- and
    leaves no partial file behind."""
    path = str(tmp_path / "gen.avi")
    assert mjpeg_avi.write_mjpeg_avi(path, (f for f in _frames(5)), fps=25) == 5           # a generator: nothing is kept
    raw = open(path, "rb").read()
    total, largest = struct.unpack("<I", raw[48:52])[0], struct.unpack("<I", raw[60:64])[0]   # avih.dwTotalFrames, dwSuggestedBufferSize
    assert total == 5 and 0 < largest < len(raw)
    assert mjpeg_avi.read_mjpeg_avi(path)[0] == 25.0 and len(mjpeg_avi.read_mjpeg_avi(path)[1]) == 5
    seen = []

    def frames():
        for i, f in enumerate(_frames(50)):
            seen.append(i)
            yield f

    monkeypatch.setattr(mjpeg_avi, "RIFF_LIMIT", len(raw))                                  # room for about five frames
    big = str(tmp_path / "big.avi")
    with pytest.raises(ValueError, match="4 GiB"):
        mjpeg_avi.write_mjpeg_avi(big, frames(), fps=25)
    assert len(seen) < 10 and not os.path.exists(big)                                       # stopped at the limit, cleaned up
    assert not os.path.exists(str(tmp_path / "e.avi"))
