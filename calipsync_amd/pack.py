"""Fold eval-mode BatchNorm into the conv / linear weights and pack them for the engine.

Input: the reference's 582-entry ``state_dict`` (names in ``arch.manifest()``).
Output: one flat fp32 buffer laid out as ``casync_packed_*`` (include/casync_hip.h)
describes -- the engine library owns the layout, this module owns the arithmetic.

Folding is done in float64 and rounded once to fp32:
    BN(y) = s*y + t,  s = gamma / sqrt(var + eps),  t = beta - mean*s
    BN(W x + b) = (s W) x + (s b + t)
(reference: nn.BatchNorm2d/1d in eval(), module/unet.py:18,28,32,163,168,174,228,
230,260,301,310,311).
"""
from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from . import arch


def _np64(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float64)


def _bn_affine(sd: Mapping, p: str):
    s = _np64(sd[f"{p}.weight"]) / np.sqrt(_np64(sd[f"{p}.running_var"]) + arch.BN_EPS)
    t = _np64(sd[f"{p}.bias"]) - _np64(sd[f"{p}.running_mean"]) * s
    return s, t


def _fold_ir(sd: Mapping, b: arch.IRBlock, out: Dict[str, np.ndarray]) -> None:
    """PW1/DW/PW2 of one inverted residual (module/unet.py:16-34), BN folded."""
    c = f"{b.prefix}.conv"
    s1, t1 = _bn_affine(sd, f"{c}.1")
    s2, t2 = _bn_affine(sd, f"{c}.4")
    s3, t3 = _bn_affine(sd, f"{c}.7")
    w1 = _np64(sd[f"{c}.0.weight"]).reshape(b.cexp, b.cin)
    wd = _np64(sd[f"{c}.3.weight"]).reshape(b.cexp, 9)            # [C][ky*3+kx]
    w2 = _np64(sd[f"{c}.6.weight"]).reshape(b.cout, b.cexp)
    out[f"{b.prefix}.pw1.w"] = w1 * s1[:, None]                   # [N][K]
    out[f"{b.prefix}.pw1.b"] = t1
    out[f"{b.prefix}.dw.w"] = (wd * s2[:, None]).T.copy()         # tap-major [9][C]
    out[f"{b.prefix}.dw.b"] = t2
    out[f"{b.prefix}.pw2.w"] = w2 * s3[:, None]
    out[f"{b.prefix}.pw2.b"] = t3


def _fold_dense3x3(sd: Mapping, conv: str, bn: str, out: Dict[str, np.ndarray]) -> None:
    """Dense 3x3 conv + bias + BN (module/unet.py:161-168): rows [N], cols (ky,kx,cin)."""
    s, t = _bn_affine(sd, bn)
    w = _np64(sd[f"{conv}.weight"])                                # [N][Cin][3][3]
    n = w.shape[0]
    out[f"{conv}.w"] = (w.transpose(0, 2, 3, 1).reshape(n, -1)) * s[:, None]
    out[f"{conv}.b"] = _np64(sd[f"{conv}.bias"]) * s + t


def fold(sd: Mapping) -> Dict[str, np.ndarray]:
    """All packed tensors by engine name (float64; rounded when written)."""
    out: Dict[str, np.ndarray] = {}
    blocks = arch.all_ir_blocks()
    for b in blocks:
        if b.prefix == "inc.inconv.0":
            tmp: Dict[str, np.ndarray] = {}
            _fold_ir(sd, b, tmp)
            # both 1x1 weights INPUT-channel major ([cin][cout]): the weights of two adjacent output channels are one
            # aligned pair, the operand of the kernel's packed FMAs (ops.hip, inc_kernel)
            out["inc.inconv.0.fused"] = np.concatenate([
                np.ascontiguousarray(tmp[f"{b.prefix}.pw1.w"].T).reshape(-1), tmp[f"{b.prefix}.pw1.b"],
                tmp[f"{b.prefix}.dw.w"].reshape(-1), tmp[f"{b.prefix}.dw.b"],
                np.ascontiguousarray(tmp[f"{b.prefix}.pw2.w"].T).reshape(-1), tmp[f"{b.prefix}.pw2.b"]])
        else:
            _fold_ir(sd, b, out)
    # Up blocks (module/unet.py:90-96): x = cat([up(lo), skip]).  The bilinear upsample is linear per channel and the
    # 1x1 conv linear per pixel, so they commute: W1 . cat(up(lo), skip) = up(W1a . lo) + W1b . skip.  The engine runs
    # the W1a half at the LOW resolution (a quarter of the pixels) -- the two halves are packed as matrices of their own.
    for st in arch.decoder():
        b = st[0]
        w1 = out[f"{b.prefix}.pw1.w"]
        c_lo = b.cin // 2
        out[f"{b.prefix}.pw1a.w"] = np.ascontiguousarray(w1[:, :c_lo])
        out[f"{b.prefix}.pw1b.w"] = np.ascontiguousarray(w1[:, c_lo:])
    _fold_dense3x3(sd, "audio_model.conv3", "audio_model.bn3", out)
    _fold_dense3x3(sd, "audio_model.conv5", "audio_model.bn5", out)
    s7, t7 = _bn_affine(sd, "audio_model.bn7")
    out["audio_model.bn7.s"], out["audio_model.bn7.t"] = s7, t7

    # MLP fusion + bn_tx (module/unet.py:240-246, 323-326):
    #   tx = bn_tx(cat + bn2(fc2(lrelu(bn1(fc1(cat))))))
    s1, t1 = _bn_affine(sd, "mlp_fusion.bn1")
    s2, t2 = _bn_affine(sd, "mlp_fusion.bn2")
    stx, ttx = _bn_affine(sd, "bn_tx")
    out["mlp_fusion.fc1.w"] = _np64(sd["mlp_fusion.fc1.weight"]) * s1[:, None]
    out["mlp_fusion.fc1.b"] = _np64(sd["mlp_fusion.fc1.bias"]) * s1 + t1
    out["mlp_fusion.fc2.w"] = _np64(sd["mlp_fusion.fc2.weight"]) * (s2 * stx)[:, None]
    out["mlp_fusion.fc2.b"] = (_np64(sd["mlp_fusion.fc2.bias"]) * s2 + t2) * stx + ttx
    out["mlp_fusion.fc2.rs"] = stx

    # attention blocks (module/unet.py:201-217, 256-269)
    kv_w, kv_b = [], []
    for i in range(arch.N_ATT_BLOCKS):
        p = f"attention_blocks.{i}"
        ca = f"{p}.cross_attention"
        kv_w += [_np64(sd[f"{ca}.key_conv.weight"]).reshape(64, 512),
                 _np64(sd[f"{ca}.value_conv.weight"]).reshape(512, 512)]
        kv_b += [_np64(sd[f"{ca}.key_conv.bias"]), _np64(sd[f"{ca}.value_conv.bias"])]
        out[f"{p}.p1.w"] = _np64(sd[f"{p}.attention_adjust_p_1.weight"]).reshape(512, 1024)
        out[f"{p}.p1.b"] = _np64(sd[f"{p}.attention_adjust_p_1.bias"])
        out[f"{p}.q.w"] = _np64(sd[f"{ca}.query_conv.weight"]).reshape(64, 512)
        out[f"{p}.q.b"] = _np64(sd[f"{ca}.query_conv.bias"])
        # q = query_conv(p_1(x)) = (Wq Wp1) x + (Wq bp1 + bq): composed in float64, so the query
        # projection is 64 extra output columns of the p_1 GEMM (module/unet.py:201,209,256,264)
        out[f"{p}.p1q.w"] = np.concatenate([out[f"{p}.p1.w"], out[f"{p}.q.w"] @ out[f"{p}.p1.w"]], 0)
        out[f"{p}.p1q.b"] = np.concatenate([out[f"{p}.p1.b"], out[f"{p}.q.w"] @ out[f"{p}.p1.b"] + out[f"{p}.q.b"]])
        out[f"{p}.gamma"] = _np64(sd[f"{ca}.gamma"]).reshape(1)
        sb, tb = _bn_affine(sd, f"{p}.bn")
        out[f"{p}.b1.w"] = _np64(sd[f"{p}.attention_adjust_b_1.weight"]).reshape(1024, 512) * sb[:, None]
        out[f"{p}.b1.b"] = _np64(sd[f"{p}.attention_adjust_b_1.bias"]) * sb + tb
        out[f"{p}.b1.rs"] = sb
    out["att.kv.w"] = np.concatenate(kv_w, 0)
    out["att.kv.b"] = np.concatenate(kv_b, 0)
    skx, tkx = _bn_affine(sd, "bn_kx")
    out["bn_kx.s"], out["bn_kx.t"] = skx, tkx

    # head: sigmoid(outc_bn(outc(x)))  (module/unet.py:342-344)
    so, to = _bn_affine(sd, "outc_bn")
    out["outc.w"] = _np64(sd["outc.conv.weight"]).reshape(3, 32) * so[:, None]
    out["outc.b"] = _np64(sd["outc.conv.bias"]) * so + to
    return out


def pack(sd: Mapping, layout=None) -> np.ndarray:
    """state_dict -> flat fp32 buffer in the engine's packed layout."""
    if layout is None:
        from . import _lib
        layout = _lib.packed_layout()
    items, total = layout
    folded = fold(sd)
    buf = np.zeros(total, dtype=np.float32)
    seen = set()
    for name, off, size in items:
        if name not in folded:
            raise KeyError(f"engine expects packed tensor {name!r} that the packer does not produce")
        v = np.ascontiguousarray(folded[name]).reshape(-1)
        if v.size != size:
            raise ValueError(f"packed tensor {name}: {v.size} floats, engine expects {size}")
        buf[off:off + size] = v.astype(np.float32)
        seen.add(name)
    extra = set(folded) - seen
    if extra:
        raise KeyError(f"packer produced tensors the engine does not know: {sorted(extra)[:4]}")
    return buf
