"""Device-side pieces of the reference frame loop, host API.

The reference's ``FrameSynthesizer.process_batch`` (image_infer_v1/tools/frame_synthesizer/
infer_api.py:192-357) does, per frame and on the CPU: crop + ``cv2.resize`` to 168x168, slice /
mask / normalise / transpose / concat into the model input, one ``.cpu()`` per prediction, scale
to uint8, resize back and blend.  The parts that are pure indexing are done here on the GPU in
one launch per batch, bit-exactly; ``cv2.resize`` and the polygon blend keep their host code
(their fixed-point arithmetic cannot be pinned without cv2)."""
from __future__ import annotations

import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def crops_to_model_input(crops168: torch.Tensor) -> torch.Tensor:
    """[B,168,168,3] uint8 BGR (the resized crops, infer_api.py:235) on a ROCm device ->
    [B,6,160,160] fp32: channels 0-2 the inner crop / 255, channels 3-5 the same with the mouth
    rectangle blacked out (infer_api.py:238-245)."""
    if crops168.dtype != torch.uint8 or crops168.dim() != 4 or tuple(crops168.shape[1:]) != (168, 168, 3):
        raise RuntimeError(f"crops must be uint8 [B,168,168,3], got {crops168.dtype} {tuple(crops168.shape)}")
    if crops168.device.type != "cuda":
        raise RuntimeError("crops must be on a ROCm device (no CPU fallback)")
    crops168 = crops168.contiguous()
    b = crops168.shape[0]
    x = torch.empty((b, 6, 160, 160), dtype=torch.float32, device=crops168.device)
    if b:
        with torch.cuda.device(crops168.device):
            _lib.check(_lib.load().casync_op_crop_to_input(crops168.data_ptr(), x.data_ptr(), b,
                                                           _stream(crops168.device)), "crop_to_input")
    return x


def predictions_to_uint8(pred: torch.Tensor) -> torch.Tensor:
    """[B,3,160,160] fp32 in (0,1) -> [B,160,160,3] uint8 BGR = ``np.array(pred.transpose(1,2,0) *
    255, dtype=np.uint8)`` (infer_api.py:265-266), so one D2H copy of 77 KB per frame replaces a
    307 KB float copy per frame."""
    if pred.dtype != torch.float32 or pred.dim() != 4 or tuple(pred.shape[1:]) != (3, 160, 160):
        raise RuntimeError(f"pred must be float32 [B,3,160,160], got {pred.dtype} {tuple(pred.shape)}")
    if pred.device.type != "cuda":
        raise RuntimeError("pred must be on a ROCm device (no CPU fallback)")
    pred = pred.contiguous()
    b = pred.shape[0]
    out = torch.empty((b, 160, 160, 3), dtype=torch.uint8, device=pred.device)
    if b:
        with torch.cuda.device(pred.device):
            _lib.check(_lib.load().casync_op_pred_to_u8(pred.data_ptr(), out.data_ptr(), b, _stream(pred.device)),
                       "pred_to_u8")
    return out
