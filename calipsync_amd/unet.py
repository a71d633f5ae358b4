"""Drop-in host side of the MI355X engine: ``Model(6, "hubert")`` with the reference's
``forward(x, audio_feat)`` signature and checkpoint format.

It replaces, for inference, the reference's ``Model`` (module/unet.py:273-345 ==
image_infer_v1/models/unet.py) as used by the frame loop
(image_infer_v1/tools/frame_synthesizer/infer_api.py:41-43, 259-260):

    net = Model(6, "hubert").to(device)
    net.load_state_dict(torch.load(ckpt))      # same 582 keys, strict
    net.eval()
    with torch.no_grad():
        pred = net(batch_tensor, hubert_tensor)   # [B,3,160,160] in (0,1)

The module tree only *holds* parameters (so ``state_dict`` / ``load_state_dict`` /
``.to()`` / ``.parameters()`` behave like the reference's); all arithmetic runs in
hand-written HIP kernels behind the C ABI of ``libcasync_hip.so``.  PyTorch is used for
device memory and streams only.  There is no CPU or eager fallback: without a ROCm
device or without the built library ``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, arch, pack


class _Holder(nn.Module):
    """A parameter container node (no forward)."""


def _install(root: nn.Module, key: str, tensor: torch.Tensor, is_buffer: bool) -> None:
    parts = key.split(".")
    node = root
    for p in parts[:-1]:
        nxt = node._modules.get(p)
        if nxt is None:
            nxt = _Holder()
            node.add_module(p, nxt)
        node = nxt
    if is_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor))


def _default_init(key: str, shape, role: str) -> torch.Tensor:
    """PyTorch-default-like initial values (the reference relies on nn defaults;
    gamma starts at 0, module/unet.py:205)."""
    if role in ("conv_weight", "linear_weight"):
        fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)           # kaiming_uniform(a=sqrt(5))
        return torch.empty(shape).uniform_(-bound, bound)
    if role == "conv_bias":
        return torch.zeros(shape)
    if role in ("bn_weight", "bn_var"):
        return torch.ones(shape)
    if role in ("bn_bias", "bn_mean", "gamma"):
        return torch.zeros(shape)
    if role == "bn_count":
        return torch.zeros(shape, dtype=torch.long)
    raise ValueError(role)


class Model(nn.Module):
    """CASync lip-sync U-Net on MI355X.  Same constructor and call contract as the
    reference ``Model(n_channels=6, mode='hubert', n_blocks=4)`` (module/unet.py:274)."""

    def __init__(self, n_channels: int = 6, mode: str = "hubert", n_blocks: int = 4, *,
                 precision: str = "fp32"):
        """``precision`` (keyword-only extension of the reference signature): "fp32" is the
        parity path (|delta| < 1e-3 vs the reference, measured ~1e-6); "bf16" stores activations
        and feeds the matrix cores in bf16 with fp32 accumulation (BASELINE configs[2]; ~1e-2)."""
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._dtype = 1 if precision == "bf16" else 0
        if n_channels != 6:
            raise ValueError("the inference contract is 6 input channels (reference crop + masked crop)")
        if mode != "hubert":
            raise NotImplementedError(
                "only mode='hubert' is on the inference path (infer_api.py:41); "
                "AudioConvWenet is out of the hot-path contract")
        if n_blocks != arch.N_ATT_BLOCKS:
            raise NotImplementedError("the engine is built for n_blocks=4 (reference default)")
        self.n_channels = n_channels
        for key, shape, _dtype, role in arch.manifest():
            _install(self, key, _default_init(key, shape, role),
                     is_buffer=role in ("bn_mean", "bn_var", "bn_count"))
        self._engine: Optional[int] = None          # casync_handle
        self._engine_device: Optional[torch.device] = None
        self._packed: Optional[torch.Tensor] = None  # packed weights on the device
        self._workspace: Optional[torch.Tensor] = None   # the largest arena any forward needed so far
        self._options: Dict[str, int] = {}           # per-engine switches (casync_set_option), re-applied on rebuild
        self.eval()

    # ------------------------------------------------------------------ weights
    def _invalidate(self) -> None:
        self._packed = None

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._invalidate()
        return res

    def _apply(self, fn, recurse=True):
        res = super()._apply(fn, recurse)
        self._invalidate()
        self._workspace = None
        return res

    @property
    def precision(self) -> str:
        return "bf16" if self._dtype else "fp32"

    def set_precision(self, precision: str) -> "Model":
        """Switch the engine's activation storage type; the next forward rebuilds the engine."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        new = 1 if precision == "bf16" else 0
        if new != self._dtype:
            self._dtype = new
            if self._engine is not None:
                _lib.load().casync_destroy(self._engine)
            self._engine, self._packed = None, None
            self._workspace = None
        return self

    def set_option(self, name: str, value: int) -> "Model":
        """Engine switch by name for THIS model (``casync_set_option``: "lanes", "trunk_lanes", "overlap",
        "gemm_streamk", "fuse_ir", "fuse_q", ... -- DESIGN.md section 4).  The CASYNC_* environment only sets
        the process defaults, once; this is how tests and tools A/B a switch at run time."""
        self._options[name] = int(value)
        if self._engine is not None:
            _lib.set_option(name, value, self._engine)
        return self

    def get_option(self, name: str) -> int:
        if self._engine is not None:
            return _lib.get_option(name, self._engine)
        return self._options.get(name, _lib.get_option(name))

    def refresh_weights(self) -> None:
        """Re-fold and re-upload after parameters were changed in place."""
        self._invalidate()

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _ensure_engine(self, dev: torch.device) -> None:
        lib = _lib.load()
        if self._engine is None or self._engine_device != dev:
            if self._engine is not None:
                lib.casync_destroy(self._engine)
            h = C.c_void_p()
            _lib.check(lib.casync_create_ex(dev.index or 0, self._dtype, C.byref(h)), "casync_create_ex")
            self._engine, self._engine_device = h, dev
            self._packed = None
            for name, value in self._options.items():
                _lib.set_option(name, value, h)

    def packed_weights_host(self) -> np.ndarray:
        """BN-folded flat fp32 buffer (engine layout) from the current parameters."""
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        return pack.pack(sd)

    def adopt_packed(self, packed_dev: torch.Tensor) -> None:
        """Use an already packed device buffer (e.g. filled by an RCCL broadcast)."""
        dev = packed_dev.device
        if dev.type != "cuda":
            raise RuntimeError("packed weights must live on a ROCm device")
        if packed_dev.dtype != torch.float32 or not packed_dev.is_contiguous():
            raise ValueError("packed weights must be contiguous float32")
        with torch.cuda.device(dev):
            self._ensure_engine(dev)
            _lib.check(_lib.load().casync_load_weights_device(self._engine, packed_dev.data_ptr(),
                                                              packed_dev.numel()), "casync_load_weights_device")
        self._packed = packed_dev

    def _ensure_weights(self, dev: torch.device) -> None:
        self._ensure_engine(dev)
        if self._packed is None:
            host = torch.from_numpy(self.packed_weights_host())
            self.adopt_packed(host.to(dev))

    def _ws(self, batch: int, dev: torch.device) -> torch.Tensor:
        """Workspace arena for `batch` frames: any arena at least as large as the engine asks for works
        (include/casync_hip.h), so the largest one ever needed is kept and reused by smaller batches
        (the last batch of a clip is smaller, infer_api.py:385-386) instead of reallocating."""
        need = _lib.load().casync_workspace_bytes_h(self._engine, batch)
        ws = self._workspace
        if ws is None or ws.device != dev or ws.numel() < need:
            self._workspace = None          # release the old arena before taking the bigger one
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._workspace = ws
        return ws

    # ------------------------------------------------------------------ forward
    def _check_inputs(self, x: torch.Tensor, audio_feat: torch.Tensor) -> torch.device:
        if self.training:
            raise RuntimeError("the MI355X engine implements the eval-mode forward only; call .eval()")
        if x.dim() != 4 or tuple(x.shape[1:]) != (6, arch.FACE_HW, arch.FACE_HW):
            raise RuntimeError(f"x must be [B,6,160,160], got {tuple(x.shape)}")
        if audio_feat.dim() != 4 or tuple(audio_feat.shape[1:]) != (32, 32, 32):
            raise RuntimeError(f"audio_feat must be [B,32,32,32], got {tuple(audio_feat.shape)}")
        if x.shape[0] != audio_feat.shape[0]:
            raise RuntimeError("x and audio_feat must have the same batch size")
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("casync_amd.Model runs on a ROCm device only (no CPU fallback); "
                               "move it with .to('cuda:0')")
        if x.device != dev or audio_feat.device != dev:
            raise RuntimeError(f"inputs must be on {dev}, got {x.device} / {audio_feat.device}")
        if x.dtype != torch.float32 or audio_feat.dtype != torch.float32:
            raise RuntimeError("inputs must be float32 (the reference feeds fp32, infer_api.py:256-257)")
        return dev

    @torch.no_grad()
    def forward(self, x: torch.Tensor, audio_feat: torch.Tensor) -> torch.Tensor:
        dev = self._check_inputs(x, audio_feat)
        batch = x.shape[0]
        if batch == 0:
            return torch.empty((0, 3, arch.FACE_HW, arch.FACE_HW), dtype=torch.float32, device=dev)
        x = x.contiguous()
        audio_feat = audio_feat.contiguous()
        with torch.cuda.device(dev):
            self._ensure_weights(dev)
            ws = self._ws(batch, dev)
            out = torch.empty((batch, 3, arch.FACE_HW, arch.FACE_HW), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().casync_forward(self._engine, x.data_ptr(), audio_feat.data_ptr(),
                                                  out.data_ptr(), batch, ws.data_ptr(), ws.numel(),
                                                  stream), "casync_forward")
        return out

    @torch.no_grad()
    def forward_windows(self, x: torch.Tensor, features: torch.Tensor, frame_indices) -> torch.Tensor:
        """Forward with the HuBERT windows gathered on the device.

        ``features``: the clip's whole HuBERT array ``[T, 2, 1024]`` fp32 on the model's device
        (uploaded once); ``frame_indices``: the video frame index of every batch entry.  Entry b
        sees ``features[i-8:i+8]`` zero-padded past both ends and reshaped to ``(32,32,32)`` --
        the window ``FrameSynthesizer._get_audio_features`` (reference infer_api.py:99-145)
        builds on the host, bit for bit, including its corner cases (a clip shorter than the pad, an
        index past the end or negative: the reference's truncated ``zeros_like`` pads then miss 16 rows
        and it falls back to an all-zero window).  Equivalent to ``forward(x, windows)``."""
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("casync_amd.Model runs on a ROCm device only (no CPU fallback)")
        if features.dim() != 3 or tuple(features.shape[1:]) != (2, 1024) or features.dtype != torch.float32:
            raise RuntimeError(f"features must be float32 [T,2,1024], got {tuple(features.shape)} {features.dtype}")
        if features.device != dev or x.device != dev:
            raise RuntimeError(f"x and features must be on {dev}")
        idx = torch.as_tensor(frame_indices, dtype=torch.int32).to(dev).contiguous()
        batch = x.shape[0]
        if idx.dim() != 1 or idx.numel() != batch:
            raise RuntimeError("frame_indices must have one entry per batch element")
        if x.dim() != 4 or tuple(x.shape[1:]) != (6, arch.FACE_HW, arch.FACE_HW) or x.dtype != torch.float32:
            raise RuntimeError(f"x must be float32 [B,6,160,160], got {tuple(x.shape)}")
        if self.training:
            raise RuntimeError("the MI355X engine implements the eval-mode forward only; call .eval()")
        if batch == 0:
            return torch.empty((0, 3, arch.FACE_HW, arch.FACE_HW), dtype=torch.float32, device=dev)
        x, features = x.contiguous(), features.contiguous()
        with torch.cuda.device(dev):
            self._ensure_weights(dev)
            ws = self._ws(batch, dev)
            out = torch.empty((batch, 3, arch.FACE_HW, arch.FACE_HW), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().casync_forward_windows(
                self._engine, x.data_ptr(), features.data_ptr(), features.shape[0], idx.data_ptr(),
                out.data_ptr(), batch, ws.data_ptr(), ws.numel(), stream), "casync_forward_windows")
        return out

    # ------------------------------------------------------------------ debugging / measurement
    @torch.no_grad()
    def tap(self, name: str, batch: int) -> torch.Tensor:
        """NCHW copy of a named intermediate of the last forward at this batch size."""
        dev = self._device()
        ws = self._workspace
        if ws is None:
            raise RuntimeError("tap: no forward has run yet")
        lib = _lib.load()
        probe = torch.empty(batch * 160 * 160 * 32, dtype=torch.bfloat16 if self._dtype else torch.float32,
                            device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        per = _lib.check(lib.casync_tap(self._engine, name.encode(), batch, ws.data_ptr(),
                                        probe.data_ptr(), probe.numel(), stream), f"casync_tap({name})")
        shapes = {"x1": (160, 32), "x2": (80, 64), "x3": (40, 128), "x4": (20, 256), "x5": (10, 512),
                  "a": (10, 512), "tx": (10, 1024), "kx": (10, 1024), "fuse": (10, 256),
                  "u1": (20, 128), "u2": (40, 64), "u3": (80, 32), "u4": (160, 32),
                  "audio_conv2": (32, 128), "audio_conv3": (16, 256), "audio_conv4": (16, 256),
                  "audio_conv5": (10, 512)}
        for i in range(4):
            shapes[f"att{i}"] = (10, 1024)
        hw, c = shapes[name]
        assert per == hw * hw * c
        return probe[:batch * per].view(batch, hw, hw, c).permute(0, 3, 1, 2).float().contiguous()

    @torch.no_grad()
    def profile(self, x: torch.Tensor, audio_feat: torch.Tensor) -> List[dict]:
        """Per-launch HIP-event timing of one forward (synchronises)."""
        dev = self._check_inputs(x, audio_feat)
        batch = x.shape[0]
        with torch.cuda.device(dev):
            self._ensure_weights(dev)
            ws = self._ws(batch, dev)
            out = torch.empty((batch, 3, arch.FACE_HW, arch.FACE_HW), dtype=torch.float32, device=dev)
            arr = (_lib.KernelTime * 1024)()
            stream = torch.cuda.current_stream(dev).cuda_stream
            n = _lib.check(_lib.load().casync_profile_forward(
                self._engine, x.contiguous().data_ptr(), audio_feat.contiguous().data_ptr(),
                out.data_ptr(), batch, ws.data_ptr(), ws.numel(), stream, arr, 1024),
                "casync_profile_forward")
        return [{"name": arr[i].name.decode(), "kernel": arr[i].kernel.decode(), "ms": arr[i].ms, "ms_raw": arr[i].ms_raw, "flops": arr[i].flops,
                 "bytes": arr[i].bytes} for i in range(n)]

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().casync_destroy(self._engine)
        except Exception:
            pass
