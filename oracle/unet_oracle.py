"""CPU oracle for the CASync U-Net forward -- TEST INFRASTRUCTURE ONLY.

This is a CPU restatement, in this repo's own words, of the reference's
per-frame inference forward (reference ``module/unet.py:314-345``).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline --
never as the product path (the product path is the HIP engine in
``calipsync_amd/csrc`` and fails loudly without it).

Parity pin: ``tests/golden/*.npz`` hold outputs and intermediates produced in
the build container by importing the reference module itself
(``tests/golden/make_golden.py``); ``tests/test_oracle.py`` checks this
restatement against them (<= 1e-5).  The reference has no tests or golden
vectors of its own for this path.

It is a *functional* restatement on the raw, unfolded 582-entry state_dict:
eval-mode BatchNorm with running statistics (eps 1e-5), LeakyReLU(0.01), NCHW
fp32, arithmetic by ``torch.nn.functional`` on the CPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

_EPS = 1e-5
_SLOPE = 0.01


def _bn(sd, p: str, x: torch.Tensor) -> torch.Tensor:
    """Eval-mode batch norm from running stats (nn.BatchNorm2d/1d in eval())."""
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"],
                        sd[f"{p}.weight"], sd[f"{p}.bias"], False, 0.0, _EPS)


def _act(x: torch.Tensor) -> torch.Tensor:
    return F.leaky_relu(x, _SLOPE)


def inverted_residual(sd, p: str, x: torch.Tensor, stride: int, res: bool) -> torch.Tensor:
    """PW -> BN -> LReLU -> DW3x3(stride, pad 1) -> BN -> LReLU -> PW -> BN -> LReLU,
    then ``x + .`` if ``res`` (module/unet.py:16-40; residual after the last
    activation)."""
    c = f"{p}.conv"
    h = _act(_bn(sd, f"{c}.1", F.conv2d(x, sd[f"{c}.0.weight"])))
    w_dw = sd[f"{c}.3.weight"]
    h = F.conv2d(h, w_dw, None, stride, 1, 1, w_dw.shape[0])
    h = _act(_bn(sd, f"{c}.4", h))
    h = _act(_bn(sd, f"{c}.7", F.conv2d(h, sd[f"{c}.6.weight"])))
    return x + h if res else h


def double_conv(sd, p: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    """module/unet.py:43-55."""
    x = inverted_residual(sd, f"{p}.double_conv.0", x, stride, False)
    return inverted_residual(sd, f"{p}.double_conv.1", x, 1, True)


def audio_encoder(sd, a: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """AudioConvHubert.forward (module/unet.py:177-194): [B,32,32,32] -> [B,512,10,10]."""
    p = "audio_model"
    a = inverted_residual(sd, f"{p}.conv1", a, 1, False)
    a = inverted_residual(sd, f"{p}.conv2", a, 1, False)
    if taps is not None:
        taps["audio_conv2"] = a
    a = F.conv2d(a, sd[f"{p}.conv3.weight"], sd[f"{p}.conv3.bias"], 2, 1)
    a = _act(_bn(sd, f"{p}.bn3", a))
    if taps is not None:
        taps["audio_conv3"] = a
    a = inverted_residual(sd, f"{p}.conv4", a, 1, True)
    if taps is not None:
        taps["audio_conv4"] = a
    a = F.conv2d(a, sd[f"{p}.conv5.weight"], sd[f"{p}.conv5.bias"], 2, 3)   # pad 3: 16 -> 10
    a = _act(_bn(sd, f"{p}.bn5", a))
    if taps is not None:
        taps["audio_conv5"] = a
    a = inverted_residual(sd, f"{p}.conv6", a, 1, True)
    a = inverted_residual(sd, f"{p}.conv7", a, 1, True)
    return _act(_bn(sd, f"{p}.bn7", a))


def mlp_fusion(sd, x5: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """MLPFusion.forward (module/unet.py:233-249): per-pixel 2-layer MLP on
    cat(x5, a); BN1d over the channel axis; no activation after the 2nd BN."""
    b, c, h, w = x5.shape
    t = torch.cat([x5.reshape(b, c, h * w), a.reshape(b, c, h * w)], 1).transpose(1, 2)  # [B,HW,2C]
    t = F.linear(t, sd["mlp_fusion.fc1.weight"], sd["mlp_fusion.fc1.bias"])
    t = _act(_bn(sd, "mlp_fusion.bn1", t.transpose(1, 2)).transpose(1, 2))
    t = F.linear(t, sd["mlp_fusion.fc2.weight"], sd["mlp_fusion.fc2.bias"])
    t = _bn(sd, "mlp_fusion.bn2", t.transpose(1, 2))                                    # [B,2C,HW]
    return t.reshape(b, 2 * c, h, w)


def cross_attention(sd, p: str, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """CrossAttention.forward (module/unet.py:207-218).  Single head, no
    1/sqrt(d) scale, softmax over the *audio position* axis, out = gamma*O + x."""
    b, c, h, w = x.shape
    n = h * w
    q = F.conv2d(x, sd[f"{p}.query_conv.weight"], sd[f"{p}.query_conv.bias"]).reshape(b, -1, n)
    k = F.conv2d(y, sd[f"{p}.key_conv.weight"], sd[f"{p}.key_conv.bias"]).reshape(b, -1, n)
    v = F.conv2d(y, sd[f"{p}.value_conv.weight"], sd[f"{p}.value_conv.bias"]).reshape(b, -1, n)
    energy = torch.bmm(q.transpose(1, 2), k)              # [B, n_face, n_audio]
    att = torch.softmax(energy, dim=-1)
    o = torch.bmm(v, att.transpose(1, 2)).reshape(b, c, h, w)
    return sd[f"{p}.gamma"] * o + x


def attention_block(sd, p: str, x: torch.Tensor, a: torch.Tensor, tx: torch.Tensor) -> torch.Tensor:
    """AttentionBlock.forward (module/unet.py:263-270)."""
    ox = F.conv2d(x, sd[f"{p}.attention_adjust_p_1.weight"], sd[f"{p}.attention_adjust_p_1.bias"])
    ox = cross_attention(sd, f"{p}.cross_attention", ox, a)
    ox = F.conv2d(ox, sd[f"{p}.attention_adjust_b_1.weight"], sd[f"{p}.attention_adjust_b_1.bias"])
    return _act(_bn(sd, f"{p}.bn", ox + tx))


def up_block(sd, p: str, lo: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """Up.forward (module/unet.py:90-97): bilinear x2 align_corners=True,
    zero-pad to the skip's size (a no-op at these sizes), cat([up, skip])."""
    up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=True)
    dy = skip.shape[2] - up.shape[2]
    dx = skip.shape[3] - up.shape[3]
    if dy or dx:
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(sd, f"{p}.conv", torch.cat([up, skip], 1), 1)


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, audio: torch.Tensor,
            taps: Optional[dict] = None, n_blocks: int = 4) -> torch.Tensor:
    """Model.forward (module/unet.py:314-345).

    x [B,6,160,160] fp32, audio [B,32,32,32] fp32 -> [B,3,160,160] in (0,1).
    ``taps``, if given, receives named intermediates (NCHW tensors)."""
    x1 = inverted_residual(sd, "inc.inconv.0", x, 1, False)
    x2 = double_conv(sd, "down1.maxpool_conv.0", x1, 2)
    x3 = double_conv(sd, "down2.maxpool_conv.0", x2, 2)
    x4 = double_conv(sd, "down3.maxpool_conv.0", x3, 2)
    x5 = double_conv(sd, "down4.maxpool_conv.0", x4, 2)
    a = audio_encoder(sd, audio, taps)
    tx = _bn(sd, "bn_tx", torch.cat([x5, a], 1) + mlp_fusion(sd, x5, a))
    ox = tx
    kx = tx
    att = []
    for i in range(n_blocks):
        ox = attention_block(sd, f"attention_blocks.{i}", ox, a, tx)
        kx = ox + kx
        att.append(ox)
    kx = _act(_bn(sd, "bn_kx", kx))
    f = double_conv(sd, "fuse_conv.0", kx, 1)
    f = double_conv(sd, "fuse_conv.1", f, 1)
    u1 = up_block(sd, "up1", f, x4)
    u2 = up_block(sd, "up2", u1, x3)
    u3 = up_block(sd, "up3", u2, x2)
    u4 = up_block(sd, "up4", u3, x1)
    out = F.conv2d(u4, sd["outc.conv.weight"], sd["outc.conv.bias"])
    out = torch.sigmoid(_bn(sd, "outc_bn", out))
    if taps is not None:
        taps.update(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, a=a, tx=tx, kx=kx, fuse=f,
                    u1=u1, u2=u2, u3=u3, u4=u4, out=out)
        for i, t in enumerate(att):
            taps[f"att{i}"] = t
    return out


def to_torch(sd_np: dict, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """NumPy state_dict (calipsync_amd.recipe) -> torch CPU tensors."""
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(v.copy())
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
