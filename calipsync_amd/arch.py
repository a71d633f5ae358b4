"""Architecture table of the CASync lip-sync U-Net (the hot path).

One compact description from which everything else is derived:

* the 582-entry ``state_dict`` manifest (names, shapes, dtypes) that the
  reference checkpoint format uses (reference ``module/unet.py:273-312`` builds
  the module tree; key names follow from ``nn.Sequential`` indices there),
* the execution plan of the HIP engine (``csrc/engine.hip`` mirrors the same
  order; ``tests/test_abi.py`` checks both sides agree on the packed layout),
* the per-frame work figures (MACs, canonical bytes) used by ``bench.py``.

Nothing here computes: it is pure data.  The stage names are ours; the
``state_dict`` prefixes are the reference's and must not change.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Tuple

# Channel ladder used by every stage (reference module/unet.py:277, :152).
CH = (32, 64, 128, 256, 512)
FACE_HW = 160          # face crop is 160x160 (reference infer_api.py:238)
AUDIO_HW = 32          # HuBERT window reshaped to [32,32,32] (infer_api.py:134)
EXPAND = 2             # expand_ratio used by every block in this model
BN_EPS = 1e-5          # nn.BatchNorm default, eval mode
LRELU_SLOPE = 0.01     # nn.LeakyReLU default slope
N_ATT_BLOCKS = 4


@dataclass(frozen=True)
class IRBlock:
    """One inverted-residual block: PW expand -> DW3x3 -> PW project.

    Reference: ``InvertedResidual`` module/unet.py:8-40 (BN + LeakyReLU after
    each of the three convs; ``x + conv(x)`` when ``res``)."""
    prefix: str      # state_dict prefix of the nn.Sequential named ``conv``
    cin: int
    cout: int
    stride: int
    res: bool
    hw_in: int       # input spatial size (square)

    @property
    def cexp(self) -> int:
        return self.cin * EXPAND

    @property
    def hw_out(self) -> int:
        return self.hw_in if self.stride == 1 else (self.hw_in + 2 - 3) // 2 + 1


def _double(prefix: str, cin: int, cout: int, stride: int, hw: int) -> List[IRBlock]:
    """``DoubleConvDW`` (module/unet.py:43-55): strided IR then residual IR."""
    first = IRBlock(f"{prefix}.double_conv.0", cin, cout, stride, False, hw)
    second = IRBlock(f"{prefix}.double_conv.1", cout, cout, 1, True, first.hw_out)
    return [first, second]


def face_encoder() -> List[List[IRBlock]]:
    """inc + down1..down4 (module/unet.py:290-294)."""
    stages = [[IRBlock("inc.inconv.0", 6, CH[0], 1, False, FACE_HW)]]
    hw = FACE_HW
    for i in range(4):
        st = _double(f"down{i + 1}.maxpool_conv.0", CH[i], CH[i + 1], 2, hw)
        stages.append(st)
        hw = st[-1].hw_out
    return stages


def audio_encoder_blocks() -> dict:
    """``AudioConvHubert`` (module/unet.py:147-194)."""
    return {
        "conv1": IRBlock("audio_model.conv1", 32, CH[1], 1, False, 32),
        "conv2": IRBlock("audio_model.conv2", CH[1], CH[2], 1, False, 32),
        # conv3: dense 3x3 s2 p1 128->256, 32->16 ; bn3 ; LeakyReLU
        "conv4": IRBlock("audio_model.conv4", CH[3], CH[3], 1, True, 16),
        # conv5: dense 3x3 s2 p3 256->512, 16->10 ; bn5 ; LeakyReLU
        "conv6": IRBlock("audio_model.conv6", CH[4], CH[4], 1, True, 10),
        "conv7": IRBlock("audio_model.conv7", CH[4], CH[4], 1, True, 10),
    }


def fuse_blocks() -> List[IRBlock]:
    """``fuse_conv`` (module/unet.py:286-289): 1024->512->256 at 10x10."""
    return (_double("fuse_conv.0", CH[4] * 2, CH[4], 1, 10)
            + _double("fuse_conv.1", CH[4], CH[3], 1, 10))


def decoder() -> List[List[IRBlock]]:
    """up1..up4 (module/unet.py:296-299): bilinear x2, cat skip, DoubleConvDW."""
    spec = [("up1", CH[4], CH[3] // 2, 20), ("up2", CH[3], CH[2] // 2, 40),
            ("up3", CH[2], CH[1] // 2, 80), ("up4", CH[1], CH[0], 160)]
    return [_double(f"{n}.conv", cin, cout, 1, hw) for n, cin, cout, hw in spec]


def all_ir_blocks() -> List[IRBlock]:
    """Every IR block in ``state_dict`` registration order of the reference."""
    a = audio_encoder_blocks()
    out = [a["conv1"], a["conv2"], a["conv4"], a["conv6"], a["conv7"]]
    out += fuse_blocks()
    for st in face_encoder():
        out += st
    for st in decoder():
        out += st
    return out


# --------------------------------------------------------------------------
# state_dict manifest
# --------------------------------------------------------------------------
Entry = Tuple[str, Tuple[int, ...], str, str]   # (key, shape, dtype, role)


def _bn(prefix: str, c: int) -> Iterator[Entry]:
    yield (f"{prefix}.weight", (c,), "float32", "bn_weight")
    yield (f"{prefix}.bias", (c,), "float32", "bn_bias")
    yield (f"{prefix}.running_mean", (c,), "float32", "bn_mean")
    yield (f"{prefix}.running_var", (c,), "float32", "bn_var")
    yield (f"{prefix}.num_batches_tracked", (), "int64", "bn_count")


def _conv(prefix: str, cout: int, cin_per_group: int, k: int, bias: bool) -> Iterator[Entry]:
    yield (f"{prefix}.weight", (cout, cin_per_group, k, k), "float32", "conv_weight")
    if bias:
        yield (f"{prefix}.bias", (cout,), "float32", "conv_bias")


def _ir(b: IRBlock) -> Iterator[Entry]:
    p = f"{b.prefix}.conv"
    yield from _conv(f"{p}.0", b.cexp, b.cin, 1, False)
    yield from _bn(f"{p}.1", b.cexp)
    yield from _conv(f"{p}.3", b.cexp, 1, 3, False)
    yield from _bn(f"{p}.4", b.cexp)
    yield from _conv(f"{p}.6", b.cout, b.cexp, 1, False)
    yield from _bn(f"{p}.7", b.cout)


def manifest() -> List[Entry]:
    """The 582 ``state_dict`` entries, in the reference's registration order
    (module/unet.py:281-312: audio_model, fuse_conv, inc, down1-4, up1-4, outc,
    outc_bn, mlp_fusion, attention_blocks, bn_kx, bn_tx)."""
    out: List[Entry] = []
    a = audio_encoder_blocks()
    out += _ir(a["conv1"])
    out += _ir(a["conv2"])
    out += _conv("audio_model.conv3", CH[3], CH[2], 3, True)
    out += _bn("audio_model.bn3", CH[3])
    out += _ir(a["conv4"])
    out += _conv("audio_model.conv5", CH[4], CH[3], 3, True)
    out += _bn("audio_model.bn5", CH[4])
    out += _ir(a["conv6"])
    out += _ir(a["conv7"])
    out += _bn("audio_model.bn7", CH[4])
    for b in fuse_blocks():
        out += _ir(b)
    for st in face_encoder():
        for b in st:
            out += _ir(b)
    for st in decoder():
        for b in st:
            out += _ir(b)
    out += _conv("outc.conv", 3, CH[0], 1, True)
    out += _bn("outc_bn", 3)
    c2 = CH[4] * 2
    out += [("mlp_fusion.fc1.weight", (c2, c2), "float32", "linear_weight"),
            ("mlp_fusion.fc1.bias", (c2,), "float32", "conv_bias")]
    out += _bn("mlp_fusion.bn1", c2)
    out += [("mlp_fusion.fc2.weight", (c2, c2), "float32", "linear_weight"),
            ("mlp_fusion.fc2.bias", (c2,), "float32", "conv_bias")]
    out += _bn("mlp_fusion.bn2", c2)
    for i in range(N_ATT_BLOCKS):
        p = f"attention_blocks.{i}"
        out.append((f"{p}.cross_attention.gamma", (1,), "float32", "gamma"))
        out += _conv(f"{p}.cross_attention.query_conv", CH[4] // 8, CH[4], 1, True)
        out += _conv(f"{p}.cross_attention.key_conv", CH[4] // 8, CH[4], 1, True)
        out += _conv(f"{p}.cross_attention.value_conv", CH[4], CH[4], 1, True)
        out += _conv(f"{p}.attention_adjust_p_1", CH[4], c2, 1, True)
        out += _conv(f"{p}.attention_adjust_b_1", c2, CH[4], 1, True)
        out += _bn(f"{p}.bn", c2)
    out += _bn("bn_kx", c2)
    out += _bn("bn_tx", c2)
    return out


# --------------------------------------------------------------------------
# Work figures per frame -- used by bench.py's roofline block
# --------------------------------------------------------------------------
def stagewise_bound(peak_flops: float, peak_bytes_per_s: float, elem_bytes: int = 4) -> dict:
    """SURVEY.md 8(d) "stage-wise" roofline: per stage (inc, down1-4, audio encoder, MLP fusion,
    each attention block, fuse_conv, up1-4, outc) max(canonical bytes / HBM peak, flops / matrix
    peak), summed -- the bound the north star's "60 % of the governing roofline" is priced on
    (fp32: 60.7 us/frame = 16.5 k frames/s per GPU)."""
    stages: dict = {}

    def conv(stage, cin, cout, hw_in, hw_out, k, groups=1):
        st = stages.setdefault(stage, [0, 0])
        st[0] += 2 * hw_out * hw_out * cout * (cin // groups) * k * k
        st[1] += (cin * hw_in * hw_in + cout * hw_out * hw_out) * elem_bytes

    for b in all_ir_blocks():
        stage = b.prefix.split(".")[0]
        conv(stage, b.cin, b.cexp, b.hw_in, b.hw_in, 1)
        conv(stage, b.cexp, b.cexp, b.hw_in, b.hw_out, 3, groups=b.cexp)
        conv(stage, b.cexp, b.cout, b.hw_out, b.hw_out, 1)
    conv("audio_model", CH[2], CH[3], 32, 16, 3)
    conv("audio_model", CH[3], CH[4], 16, 10, 3)
    c2 = CH[4] * 2
    conv("mlp", c2, c2, 10, 10, 1)
    conv("mlp", c2, c2, 10, 10, 1)
    for i in range(N_ATT_BLOCKS):
        st = f"attention_blocks.{i}"
        conv(st, c2, CH[4], 10, 10, 1)
        conv(st, CH[4], CH[4] // 8, 10, 10, 1)
        conv(st, CH[4], CH[4] // 8, 10, 10, 1)
        conv(st, CH[4], CH[4], 10, 10, 1)
        conv(st, CH[4], c2, 10, 10, 1)
        stages[st][0] += 2 * (100 * 100 * (CH[4] // 8) + 100 * 100 * CH[4])
    conv("outc", CH[0], 3, 160, 160, 1)
    per = {k: max(v[1] / peak_bytes_per_s, v[0] / peak_flops) for k, v in stages.items()}
    total = sum(per.values())
    return {"seconds_per_frame": total, "frames_per_s": 1.0 / total,
            "hbm_governed": sorted(k for k, v in stages.items() if v[1] / peak_bytes_per_s >= v[0] / peak_flops)}


def work_per_frame() -> dict:
    """MACs and canonical conv-granularity activation elements per frame.

    "Canonical" = for every Conv2d/Linear of the reference, input elements +
    output elements (BN / activation / residual folded); SURVEY.md §8(d)."""
    macs = 0
    elems = 0

    def conv(cin, cout, hw_in, hw_out, k, groups=1):
        nonlocal macs, elems
        macs += hw_out * hw_out * cout * (cin // groups) * k * k
        elems += cin * hw_in * hw_in + cout * hw_out * hw_out

    for b in all_ir_blocks():
        conv(b.cin, b.cexp, b.hw_in, b.hw_in, 1)
        conv(b.cexp, b.cexp, b.hw_in, b.hw_out, 3, groups=b.cexp)
        conv(b.cexp, b.cout, b.hw_out, b.hw_out, 1)
    conv(CH[2], CH[3], 32, 16, 3)          # audio conv3
    conv(CH[3], CH[4], 16, 10, 3)          # audio conv5
    c2 = CH[4] * 2
    conv(c2, c2, 10, 10, 1)                # fc1
    conv(c2, c2, 10, 10, 1)                # fc2
    for _ in range(N_ATT_BLOCKS):
        conv(c2, CH[4], 10, 10, 1)         # p_1
        conv(CH[4], CH[4] // 8, 10, 10, 1)  # query
        conv(CH[4], CH[4] // 8, 10, 10, 1)  # key
        conv(CH[4], CH[4], 10, 10, 1)      # value
        conv(CH[4], c2, 10, 10, 1)         # b_1
    conv(CH[0], 3, 160, 160, 1)            # outc
    att_macs = N_ATT_BLOCKS * (100 * 100 * (CH[4] // 8) + 100 * 100 * CH[4])
    return {
        "conv_macs": macs,
        "attention_macs": att_macs,
        "flops": 2 * (macs + att_macs),
        "canonical_elems": elems,
        "canonical_bytes_f32": elems * 4,
        "io_bytes_f32": (6 * 160 * 160 + 32 * 32 * 32 + 3 * 160 * 160) * 4,
    }
