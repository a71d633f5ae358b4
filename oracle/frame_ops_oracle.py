"""CPU restatement of the image arithmetic of ``FrameSynthesizer.process_batch`` -- TEST INFRASTRUCTURE ONLY.

Reference: image_infer_v1/tools/frame_synthesizer/infer_api.py:200-253 (crop box, ``cv2.resize`` to
168x168) and :263-346 (write-back, ``cv2.resize`` to (width, width), ``cv2.fillPoly`` of the 33 contour
landmarks, area-scaled ``cv2.dilate``, mask blend).  Only ``tests/`` and ``bench.py`` may import this.

PARITY UNPINNED.  The arithmetic lives in OpenCV (``cv2``, unpinned version, absent from this image and
from /root/reference), so nothing here could be checked against the real library.  What is restated, from
the published OpenCV 4.x sources:
  * ``cv2.resize(..., INTER_LINEAR)`` on 8-bit images -- modules/imgproc/src/resize.cpp: coefficient
    tables of ``cv::resize`` (``fx = (float)((dx+0.5)*scale_x - 0.5)``, ``cvFloor``, border clamps,
    ``saturate_cast<short>(c * 2048)``), ``HResizeLinear`` (int accumulators), ``VResizeLinear<uchar,int,
    short,FixedPtCast<int,uchar,22>>``: ``dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``; and the
    exact-2x-downscale case, which ``cv::resize`` turns into INTER_AREA (``(a+b+c+d+2)>>2``).
  * ``cv2.resize`` on float32 (the optional frame mask): same tables, float coefficients, ``S0*b0 + S1*b1``.
  * ``cv2.fillPoly`` (one contour, LINE_8, shift 0) -- modules/imgproc/src/drawing.cpp: ``CollectPolyEdges``
    (16.16 fixed-point edges, every edge also drawn with ``Line``) and ``FillEdgeCollection`` (even-odd spans
    ``[ceil(xa), floor(xb)]`` per scan line); ``Line`` = ``LineIterator`` (8-connected Bresenham, left to
    right) behind ``clipLine``.
  * ``cv2.dilate`` with a (2e+1)^2 ones kernel, default border: the maximum over the in-image part of the window.
  * the numpy blend of infer_api.py:314-345 in float64 and its truncating store into the uint8 frame.
An IPP / OpenCL build of OpenCV can round differently from these C++ paths; that is part of "unpinned".
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT
# Left end of a fillPoly span, x1 = (xa + FILL_LEFT_DELTA) >> XY_SHIFT: ceil(xa) as in FillEdgeCollection of OpenCV 3.4 / early 4.x (as recalled)
# (the version restated here).  Later 4.x sources use delta = 0 for line types below LINE_AA, i.e. floor(xa).  UNPINNED
# (cv2 is absent); the same constant sits in calipsync_amd/csrc/frame_ops.hip, and test_oracle_against_cv2 decides it
# wherever cv2 is installed.
FILL_LEFT_DELTA = XY_ONE - 1
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


# ---------------------------------------------------------------------------------------------- crop box
def crop_box(lms: np.ndarray, height: int, img_width: int) -> Tuple[int, int, int, int, int]:
    """infer_api.py:206-231 -> (ymin, ymax, xmin, xmax, width); ``width`` is taken BEFORE the clamps."""
    xmin = int(lms[1][0])
    ymin = int(lms[52][1])
    xmax = int(lms[31][0])
    width = xmax - xmin
    ymax = ymin + width
    if ymax > height:
        diff = ymax - height
        ymax = height
        ymin = max(0, ymin - diff)
    if ymin < 0:
        ymax = min(height, ymax - ymin)
        ymin = 0
    if xmin < 0:
        xmin = 0
    if xmax > img_width:
        xmax = img_width
    return ymin, ymax, xmin, xmax, width


# ---------------------------------------------------------------------------------------------- cv2.resize
def _linear_tables(src: int, dst: int):
    """Source index and fractional weight per destination index (cv::resize, INTER_LINEAR)."""
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)          # fx = (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)         # fx -= sx  (float arithmetic)
    return s, f, scale


def _fixed_coefs(f: np.ndarray):
    """saturate_cast<short>((1-f)*2048), saturate_cast<short>(f*2048): cvRound = round half to even."""
    c0 = np.rint((np.float32(1.0) - f).astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int32)
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int32)
    return c0, c1


def resize_linear_u8(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """``cv2.resize(src, (dw, dh))`` for uint8 HxWxC (default INTER_LINEAR)."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    if dw == sw and dh == sh:
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:      # is_area_fast && iscale == 2: cv::resize switches to INTER_AREA
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx, _ = _linear_tables(sw, dw)
    lo = sx < 0                              # if (sx < 0) fx = 0, sx = 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1                        # if (sx >= ssize.width-1) fx = 0, sx = ssize.width-1
    fx = np.where(hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(hi, sw - 1, sx)
    a0, a1 = _fixed_coefs(fx)
    sx1 = np.minimum(sx + 1, sw - 1)         # a1 == 0 wherever sx+1 would leave the row (HResizeLinear's xmax split)
    sy, fy, _ = _linear_tables(sh, dh)
    b0, b1 = _fixed_coefs(fy)                # rows are clipped, the weights are not touched
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int32)
    if s.ndim == 2:
        s = s[:, :, None]
    # horizontal pass on the two source rows of every destination row (int, no shift)
    h0 = s[y0][:, sx] * a0[None, :, None] + s[y0][:, sx1] * a1[None, :, None]
    h1 = s[y1][:, sx] * a0[None, :, None] + s[y1][:, sx1] * a1[None, :, None]
    out = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if src.ndim == 2 else out


def resize_linear_f32(src: np.ndarray, dsize: Tuple[int, int]) -> np.ndarray:
    """``cv2.resize(src, (dw, dh))`` for float32 HxW: float tables, D = S[sx]*a0 + S[sx+1]*a1, dst = S0*b0 + S1*b1."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    if dw == sw and dh == sh:
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:
        s = src.astype(np.float32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2]) * np.float32(0.25)).astype(np.float32)
    sx, fx, _ = _linear_tables(sw, dw)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(hi, sw - 1, sx)
    a0, a1 = (np.float32(1.0) - fx).astype(np.float32), fx
    sx1 = np.minimum(sx + 1, sw - 1)
    sy, fy, _ = _linear_tables(sh, dh)
    b0, b1 = (np.float32(1.0) - fy).astype(np.float32), fy
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.float32)
    h0 = (s[y0][:, sx] * a0[None, :]).astype(np.float32) + (s[y0][:, sx1] * a1[None, :]).astype(np.float32)
    h1 = (s[y1][:, sx] * a0[None, :]).astype(np.float32) + (s[y1][:, sx1] * a1[None, :]).astype(np.float32)
    return ((h0.astype(np.float32) * b0[:, None]).astype(np.float32)
            + (h1.astype(np.float32) * b1[:, None]).astype(np.float32)).astype(np.float32)


# ---------------------------------------------------------------------------------------------- cv2.fillPoly
def _clip_line(w: int, h: int, x1: int, y1: int, x2: int, y2: int):
    """cv::clipLine(Size2l, Point2l&, Point2l&) -> (inside, x1, y1, x2, y2)."""
    right, bottom = w - 1, h - 1
    if w <= 0 or h <= 0:
        return False, x1, y1, x2, y2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, x1, y1, x2, y2


def _line(img: np.ndarray, p1, p2, color: int) -> None:
    """cv::Line (LineIterator, 8-connected, left to right)."""
    h, w = img.shape
    x1, y1, x2, y2 = int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1])
    if not (0 <= x1 < w and 0 <= x2 < w and 0 <= y1 < h and 0 <= y2 < h):
        ok, x1, y1, x2, y2 = _clip_line(w, h, x1, y1, x2, y2)
        if not ok:
            return
    dx, dy = x2 - x1, y2 - y1
    delta_x = delta_y = 1
    if dx < 0:                       # leftToRight: start from the left end point
        dx, dy = -dx, -dy
        x1, y1 = x2, y2
    if dy < 0:
        dy, delta_y = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
        delta_x, delta_y = delta_y, delta_x
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    minus_shift, plus_shift, minus_step, plus_step = delta_x, 0, 0, delta_y      # x += shift, y += step
    if vert:
        plus_step, plus_shift = plus_shift, plus_step
        minus_step, minus_shift = minus_shift, minus_step
    x, y = x1, y1
    for _ in range(dx + 1):
        img[y, x] = color
        m = err < 0
        err += minus_delta + (plus_delta if m else 0)
        y += minus_step + (plus_step if m else 0)
        x += minus_shift + (plus_shift if m else 0)


def fill_poly(shape: Tuple[int, int], pts: np.ndarray, color: int = 255) -> np.ndarray:
    """``cv2.fillPoly(zeros(shape, uint8), [pts], color)`` for one int32 contour."""
    h, w = shape
    img = np.zeros((h, w), dtype=np.uint8)
    pts = [(int(p[0]), int(p[1])) for p in pts]
    n = len(pts)
    edges = []        # [y0, y1, x, dx]  (x, dx in 16.16 fixed point, Python ints = int64 range here)
    p0 = pts[-1]
    for i in range(n):
        p1 = pts[i]
        _line(img, p0, p1, color)                      # CollectPolyEdges draws every edge
        if p0[1] != p1[1]:
            x0f, x1f = p0[0] << XY_SHIFT, p1[0] << XY_SHIFT
            num, den = x1f - x0f, p1[1] - p0[1]
            dx = abs(num) // abs(den) * (1 if (num < 0) == (den < 0) else -1)    # C++ int64 division truncates
            if p0[1] < p1[1]:
                edges.append([p0[1], p1[1], x0f, dx])
            else:
                edges.append([p1[1], p0[1], x1f, dx])
        p0 = p1
    if len(edges) < 2:
        return img
    y_min = min(e[0] for e in edges)
    y_max = min(max(e[1] for e in edges), h)
    for y in range(max(y_min, 0), y_max):
        xs = sorted(e[2] + (y - e[0]) * e[3] for e in edges if e[0] <= y < e[1])
        for k in range(0, len(xs) - 1, 2):
            xa, xb = xs[k], xs[k + 1]
            x1 = (xa + FILL_LEFT_DELTA) >> XY_SHIFT
            x2 = xb >> XY_SHIFT
            if x1 < w and x2 >= 0:
                x1, x2 = max(x1, 0), min(x2, w - 1)
                if x1 <= x2:
                    img[y, x1:x2 + 1] = color
    return img


# ---------------------------------------------------------------------------------------------- cv2.dilate
def dilate_square(mask: np.ndarray, e: int) -> np.ndarray:
    """``cv2.dilate(mask, ones((2e+1, 2e+1)), iterations=1)``: window maximum, the border does not count."""
    h, w = mask.shape
    out = mask.copy()
    for d in range(1, e + 1):          # rows
        out[:, d:] = np.maximum(out[:, d:], mask[:, :w - d]) if d < w else out[:, d:]
        out[:, :w - d] = np.maximum(out[:, :w - d], mask[:, d:]) if d < w else out[:, :w - d]
    tmp = out.copy()
    for d in range(1, e + 1):          # columns
        if d < h:
            out[d:, :] = np.maximum(out[d:, :], tmp[:h - d, :])
            out[:h - d, :] = np.maximum(out[:h - d, :], tmp[d:, :])
    return out


def expand_pixels(mask_area: int) -> int:
    """infer_api.py:294-298."""
    mask_radius = np.sqrt(mask_area / np.pi)
    return max(1, int(mask_radius * 0.15))


# ---------------------------------------------------------------------------------------------- one frame
def face_points(lms: np.ndarray, ymin: int, ymax: int, xmin: int, xmax: int, width: int) -> np.ndarray:
    """infer_api.py:281-289 (float64 arithmetic, truncation to int32)."""
    pts = lms[:33].astype(np.float64).copy()
    pts[:, 0] -= xmin
    pts[:, 1] -= ymin
    pts[:, 0] *= width / (xmax - xmin)
    pts[:, 1] *= width / (ymax - ymin)
    return pts.astype(np.int32)


def prepare_frame(img: np.ndarray, lms: np.ndarray):
    """infer_api.py:201-239 -> (box, crop168)."""
    h, w = img.shape[:2]
    box = crop_box(lms, h, w)
    ymin, ymax, xmin, xmax, _ = box
    return box, resize_linear_u8(img[ymin:ymax, xmin:xmax], (168, 168))


def paste_back(img: np.ndarray, lms: np.ndarray, mask: Optional[np.ndarray], box, crop168: np.ndarray,
               pred_u8: np.ndarray) -> np.ndarray:
    """infer_api.py:268-346 for one frame: returns the synthesised full frame (the input image, modified)."""
    img = img.copy()
    ymin, ymax, xmin, xmax, width = box
    crop = crop168.copy()
    crop[4:164, 4:164] = pred_u8
    crop = resize_linear_u8(crop, (width, width))
    face_mask = fill_poly((ymax - ymin, xmax - xmin), face_points(lms, ymin, ymax, xmin, xmax, width))
    e = expand_pixels(int(np.sum(face_mask > 0)))
    final_mask = dilate_square(face_mask, e)
    final_f = final_mask / 255.0                                   # float64 0.0 / 1.0
    final3 = np.repeat(final_f[..., np.newaxis], 3, axis=2)
    target = img[ymin:ymax, xmin:xmax]
    if crop.shape != target.shape:                                 # :320-324: shapes differ -> original frame
        return img
    if mask is not None:
        resized = resize_linear_f32(mask.astype(np.float32), (crop.shape[1], crop.shape[0]))
        resized3 = np.repeat(resized[..., np.newaxis], 3, axis=2)
        inverted = 1.0 - resized3                                  # float32
        combined = final3 * (1.0 - inverted)                       # float64
        result = (crop * combined) + (target * (1.0 - combined))
    else:
        result = (crop * final3) + (target * (1.0 - final3))
    img[ymin:ymax, xmin:xmax] = result                             # float64 -> uint8: truncation
    return img


def process_batch(images: Sequence[np.ndarray], landmarks: Sequence[np.ndarray], masks: Sequence[Optional[np.ndarray]],
                  predict) -> List[np.ndarray]:
    """The whole of process_batch around the model call.  ``predict``: [B,6,160,160] fp32 -> [B,3,160,160] fp32."""
    from . import frame_loop_oracle
    prep = [prepare_frame(img, lms) for img, lms in zip(images, landmarks)]
    x = frame_loop_oracle.crops_to_model_input(np.stack([c for _, c in prep]))
    pred_u8 = frame_loop_oracle.predictions_to_uint8(predict(x))
    return [paste_back(img, lms, m, box, crop, p) for img, lms, m, (box, crop), p in
            zip(images, landmarks, masks, prep, pred_u8)]
