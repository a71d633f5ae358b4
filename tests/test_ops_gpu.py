"""-m gpu: every HIP operator through the C ABI vs a plain PyTorch fp32 reference of the
same op (CPU).  Tolerance: fp32 rounding only (the whole-net bar is 1e-3; single ops sit
at ~1e-6 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from calipsync_amd import _lib, pack
from gpu_util import dev, nchw, nhwc, ok, options, ptr, stream

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    return float((got - ref).abs().max() / max(1e-6, float(ref.abs().max())))


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (1000, 64, 64), (6400, 1024, 512), (257, 32, 128),
                                   (100, 2304, 512), (300, 256, 1152), (25600, 32, 64), (5, 64, 512)])
def test_pw_gemm_plain(lib, m, n, k):
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ref = F.leaky_relu(a.double() @ w.double().T + b.double(), 0.01).float()
    ad, wd, bd = a.to(dev()), w.to(dev()), b.to(dev())
    c = torch.empty(m, n, device=dev())
    ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, stream()))
    assert rel_err(c.cpu(), ref) < 2e-6


@pytest.mark.parametrize("m,n,k", [(6400, 512, 1024), (3200, 512, 512), (6400, 2304, 512), (3200, 1024, 2048),
                                   (6400, 64, 512), (1111, 512, 256), (100, 512, 2304), (12800, 128, 1024)])
def test_pw_gemm_stream_k_remainder(lib, m, n, k):
    """Shapes whose tile count is not a multiple of 256: the remainder's k-iterations are split
    over workgroups (stream-K) and recombined in a fixed order -> same accuracy, bit-repeatable,
    counters left clean for the next launch (second and third launch identical)."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    pre = torch.randn(m, n, generator=g)
    ref = F.leaky_relu(a.double() @ w.double().T + b.double() + pre.double(), 0.01).float()
    ad, wd, bd, pd = a.to(dev()), w.to(dev()), b.to(dev()), pre.to(dev())
    outs = []
    for _ in range(3):
        c = torch.full((m, n), float("nan"), device=dev())
        ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, ptr(pd), n, 0, 0, 0, 0, 0,
                                 stream()))
        outs.append(c.cpu())
    assert rel_err(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_pw_gemm_stream_k_random_shapes_match_plain_tiles(lib):
    """Stress: 48 random shapes, stream-K on vs off (plain tiles) on the same operands -- equal to fp32
    reassociation, and the stream-K result repeats bit for bit (no arrival-order dependence)."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for _ in range(48):
        m = int(rng.integers(1, 30000))
        n = int(rng.integers(1, 37)) * 64
        k = int(rng.integers(8, 73)) * 32
        g = torch.Generator().manual_seed(m * 31 + n + k)
        ad = torch.randn(m, k, generator=g).to(dev())
        wd = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev())
        bd = torch.randn(n, generator=g).to(dev())
        outs = []
        for sk in (1, 1, 0):
            with options(gemm_streamk=sk):
                c = torch.full((m, n), float("nan"), device=dev())
                ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, stream()))
                outs.append(c)
        assert torch.equal(outs[0], outs[1]), (m, n, k)
        worst = max(worst, rel_err(outs[0], outs[2]))
    assert worst < 5e-6, worst


@pytest.mark.parametrize("m,n,k", [(800, 576, 1024), (800, 1024, 512), (100, 2304, 512), (1, 32, 32), (65, 96, 64),
                                   (3200, 256, 1024), (777, 160, 96), (8, 1024, 1024)])
def test_pw_gemm_small_m_tile_with_k_split_in_the_workgroup(lib, m, n, k):
    """The 64x32 tile of the small-batch plan (gemm_cfg=4): two wave pairs take the two halves of every k-tile and the
    epilogue adds them in LDS.  Forced here over ragged M, N that is only a multiple of 32, an odd number of k-tiles,
    with the whole epilogue (scaled pre-residual, activation, post-residual, affine, accumulator in/out); the result
    repeats bit for bit and, as the cost model's own choice, equals the forced one."""
    g = torch.Generator().manual_seed(m * 13 + n + k)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    bias, ps, s2, t2 = (torch.randn(n, generator=g) for _ in range(4))
    pre, post = torch.randn(m, n, generator=g), torch.randn(m, n, generator=g)
    v = a.double() @ w.double().T + bias.double() + ps.double() * pre.double()
    v = F.leaky_relu(v, 0.01) + post.double()
    ref = F.leaky_relu(v * s2.double() + t2.double(), 0.01).float()
    D = lambda t: t.to(dev())
    ad, wd, bd, psd, s2d, t2d, pred, postd = map(D, (a, w, bias, ps, s2, t2, pre, post))
    outs = []
    for cfg in (4, 4, -1):
        with options(gemm_cfg=cfg):
            c = torch.full((m, n), float("nan"), device=dev())
            ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, ptr(pred), n, ptr(psd), ptr(postd), n,
                                     ptr(s2d), ptr(t2d), stream()))
            outs.append(c.cpu())
    assert rel_err(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[2], ref) < 2e-6


def test_pw_gemm_epilogue_and_strides(lib):
    """lda/ldc slices of wider buffers + pre-residual (scaled) + post-residual + affine."""
    g = torch.Generator().manual_seed(3)
    m, n, k, lda, ldc = 777, 128, 64, 96, 160
    abuf = torch.randn(m, lda, generator=g)
    w = torch.randn(n, k, generator=g) / 8
    bias, ps, s2, t2 = (torch.randn(n, generator=g) for _ in range(4))
    pre, post = torch.randn(m, n, generator=g), torch.randn(m, n + 32, generator=g)
    a = abuf[:, 32:32 + k]
    v = a.double() @ w.double().T + bias.double() + ps.double() * pre.double()
    v = F.leaky_relu(v, 0.01) + post[:, :n].double()
    ref = F.leaky_relu(v * s2.double() + t2.double(), 0.01).float()
    D = lambda t: t.to(dev())
    abuf_d, cbuf = D(abuf), torch.full((m, ldc), -7.0, device=dev())
    wd, bd, psd, s2d, t2d, pred, postd = map(D, (w, bias, ps, s2, t2, pre, post))
    ok(lib.casync_op_pw_gemm(abuf_d.data_ptr() + 32 * 4, lda, ptr(wd), ptr(bd), cbuf.data_ptr() + 16 * 4,
                             ldc, m, n, k, 1, ptr(pred), n, ptr(psd), ptr(postd), n + 32, ptr(s2d),
                             ptr(t2d), stream()))
    out = cbuf.cpu()
    assert rel_err(out[:, 16:16 + n], ref) < 2e-6
    assert (out[:, :16] == -7).all() and (out[:, 16 + n:] == -7).all()     # nothing outside the slice


def test_pw_gemm_operand_beyond_2gib_takes_the_pointer_addressed_kernel(lib):
    """The ring kernels address A and W through buffer descriptors with 32-bit offsets; an operand whose extent
    reaches 2 GiB (here: 4096 rows of a 2.3 GB buffer, lda = 140000 floats) must fall back to the register-staged
    kernel and still be right."""
    m, n, k, lda = 4096, 128, 64, 140000
    assert ((m - 1) * lda + k) * 4 >= 2 ** 31
    g = torch.Generator().manual_seed(5)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g) / 8
    b = torch.randn(n, generator=g)
    abuf = torch.empty(m, lda, device=dev())
    abuf[:, :k] = a.to(dev())
    wd, bd = w.to(dev()), b.to(dev())
    c = torch.empty(m, n, device=dev())
    ok(lib.casync_op_pw_gemm(ptr(abuf), lda, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, stream()))
    ref = F.leaky_relu(a.double() @ w.double().T + b.double(), 0.01).float()
    assert rel_err(c.cpu(), ref) < 2e-6


def test_pw_gemm_rejects_bad_shapes(lib):
    a = torch.zeros(64, 48, device=dev())
    assert lib.casync_op_pw_gemm(ptr(a), 48, ptr(a), 0, ptr(a), 48, 64, 48, 48, 0, 0, 0, 0, 0, 0, 0, 0, stream()) < 0
    assert b"multiple" in lib.casync_last_error()


@pytest.mark.parametrize("b,h,w,c,stride", [(2, 160, 160, 12, 1), (3, 80, 80, 64, 2), (2, 10, 10, 2048, 1),
                                            (1, 21, 13, 8, 2), (2, 32, 32, 128, 1), (1, 7, 9, 4, 1)])
def test_dw3x3(lib, b, h, w, c, stride):
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(c, 1, 3, 3, generator=g) / 3
    bias = torch.randn(c, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, bias, stride, 1, 1, c), 0.01)
    wp = wt.reshape(c, 9).T.contiguous().to(dev())       # tap-major [9][C]
    xd, bd = nhwc(x), bias.to(dev())
    out = torch.empty(b, ref.shape[2], ref.shape[3], c, device=dev())
    ok(lib.casync_op_dw3x3(ptr(xd), ptr(wp), ptr(bd), ptr(out), b, h, w, c, stride, stream()))
    assert rel_err(nchw(out), ref) < 2e-6


@pytest.mark.parametrize("frames", [1, 3, 11])
@pytest.mark.parametrize("hw,c,ldg", [(20, 1024, 1024), (40, 512, 512), (20, 1024, 1280), (40, 512, 516), (10, 64, 64)])
def test_dw3x3_over_the_commuted_upsample(lib, hw, c, ldg, frames):
    """dw3x3_ups_lds_kernel (engine plan skip_early, B < 12): the depthwise conv of up1.0 (20x20x1024) / up2.0 (40x40x512)
    whose input is LReLU(pre + up(g)) -- module/unet.py:90-97 (bilinear x2, align_corners=True, cat) behind the expand
    conv's LeakyReLU (:17-20), then Conv2d(groups=C, k=3) + BN + LeakyReLU (:21-30) -- against PyTorch in float64 in the
    reference's order; ragged `ldg` (g is a column slice of a wider buffer), NaN outside the columns the kernel may read."""
    gen = torch.Generator().manual_seed(hw * 7 + c + frames)
    pre = torch.randn(frames, c, hw, hw, generator=gen)
    g = torch.randn(frames, c, hw // 2, hw // 2, generator=gen)
    wt = torch.randn(c, 1, 3, 3, generator=gen) / 3
    bias = torch.randn(c, generator=gen)
    e = F.leaky_relu(pre.double() + F.interpolate(g.double(), scale_factor=2, mode="bilinear", align_corners=True), 0.01)
    ref = F.leaky_relu(F.conv2d(e, wt.double(), bias.double(), 1, 1, 1, c), 0.01).float()
    gd = torch.full((frames, hw // 2, hw // 2, ldg), float("nan"), device=dev())
    gd[..., :c] = nhwc(g)
    wp = wt.reshape(c, 9).T.contiguous().to(dev())
    out = torch.full((frames, hw, hw, c), float("nan"), device=dev())
    pd, bd = nhwc(pre), bias.to(dev())       # named: a temporary's block may be handed to the next allocation
    ok(lib.casync_op_dw3x3_ups(ptr(pd), ptr(gd), ldg, ptr(wp), ptr(bd), ptr(out), frames, hw, hw, c, stream()))
    assert rel_err(nchw(out), ref) < 3e-6


def test_dw3x3_over_the_commuted_upsample_rejects_bad_shapes(lib):
    z = torch.zeros(64, device=dev())
    for args in ((1, 21, 20, 64), (1, 20, 20, 6), (0, 20, 20, 64), (1, 160, 160, 64)):    # odd height, C % 4, no frames, no slab
        b, h, w, c = args
        assert lib.casync_op_dw3x3_ups(ptr(z), ptr(z), c, ptr(z), ptr(z), ptr(z), b, h, w, c, stream()) != 0
    assert lib.casync_op_dw3x3_ups(ptr(z), ptr(z), 32, ptr(z), ptr(z), ptr(z), 1, 20, 20, 64, stream()) != 0   # ldg < c


@pytest.mark.parametrize("b,h,w,c,stride,padv,cout", [(2, 32, 32, 128, 2, 1, 256), (2, 16, 16, 256, 2, 3, 512),
                                                        (70, 16, 16, 256, 2, 3, 512), (3, 9, 13, 64, 1, 1, 64),
                                                        (1, 7, 5, 32, 1, 0, 128), (5, 11, 11, 32, 2, 2, 64)])
def test_conv3x3_implicit_gemm(lib, b, h, w, c, stride, padv, cout):
    """Dense 3x3 (audio conv3: pad 1, conv5: pad 3, 16 -> 10) with the taps gathered by the GEMM's own
    loads: same result as F.conv2d, including ragged sizes, pad 0 and stream-K-able shapes."""
    g = torch.Generator().manual_seed(h * 31 + c)
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(cout, c, 3, 3, generator=g) / (3 * c ** 0.5)
    bias = torch.randn(cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x.double(), wt.double(), bias.double(), stride, padv), 0.01).float()
    ho, wo = ref.shape[2], ref.shape[3]
    xd, bd = nhwc(x), bias.to(dev())
    wp = wt.permute(0, 2, 3, 1).reshape(cout, 9 * c).contiguous().to(dev())
    out = torch.full((b, ho, wo, cout), float("nan"), device=dev())
    ok(lib.casync_op_conv3x3(ptr(xd), ptr(wp), ptr(bd), ptr(out), b, h, w, c, cout, stride, padv, 1, stream()))
    assert rel_err(nchw(out), ref) < 3e-6


@pytest.mark.parametrize("h,c", [(10, 256), (20, 128), (40, 64), (80, 32), (3, 4)])
def test_upsample2x_align_corners(lib, h, c):
    x = torch.randn(2, c, h, h, generator=torch.Generator().manual_seed(h))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ldc = 2 * c
    out = torch.full((2, 2 * h, 2 * h, ldc), 5.0, device=dev())
    ok(lib.casync_op_upsample2x(ptr(nhwc(x)), ptr(out), ldc, 2, h, h, c, stream()))
    o = out.cpu()
    assert (o[..., c:] == 5).all()                                   # skip half untouched
    assert (o[..., :c].permute(0, 3, 1, 2) - ref).abs().max() < 5e-5   # fma-contraction-level differences


@pytest.mark.parametrize("b", [3, 32, 170])      # channel split 4 / 2 / 1 workgroups per query block
def test_cross_attention(lib, b):
    """module/unet.py:212-217 with gamma != 0: unscaled scores, softmax over audio positions."""
    g = torch.Generator().manual_seed(11)
    q = torch.randn(b, 100, 64, generator=g) * 0.5
    kv = torch.randn(b, 100, 2304, generator=g) * 0.5
    res = torch.randn(b, 100, 512, generator=g)
    gamma = torch.tensor([0.37])
    k, v = kv[:, :, 576:640], kv[:, :, 640:1152]                     # block 1's slice of the KV rows
    att = torch.softmax(q.double() @ k.double().transpose(1, 2), -1)
    ref = (gamma.double() * (att @ v.double()) + res.double()).float()
    qd, kvd, rd, gd = q.to(dev()), kv.to(dev()), res.to(dev()), gamma.to(dev())
    out = torch.empty(b, 100, 512, device=dev())
    ok(lib.casync_op_cross_attention(ptr(qd), 64, kvd.data_ptr() + 576 * 4, 2304, kvd.data_ptr() + 640 * 4, 2304,
                                     ptr(rd), 512, ptr(gd), ptr(out), 512, b, stream()))
    assert rel_err(out.cpu(), ref) < 3e-6


def test_cross_attention_peaked_softmax(lib):
    """Large-magnitude scores (one-hot rows) must not overflow: max-subtracted softmax."""
    g = torch.Generator().manual_seed(5)
    q = torch.randn(1, 100, 64, generator=g) * 6
    k = torch.randn(1, 100, 64, generator=g) * 6
    v = torch.randn(1, 100, 512, generator=g)
    res = torch.zeros(1, 100, 512)
    att = torch.softmax(q.double() @ k.double().transpose(1, 2), -1)
    ref = (att @ v.double()).float()
    out = torch.empty(1, 100, 512, device=dev())
    gd = torch.ones(1, device=dev())
    qd, kd, vd, rd = (t.to(dev()) for t in (q, k, v, res))
    ok(lib.casync_op_cross_attention(ptr(qd), 64, ptr(kd), 64, ptr(vd), 512, ptr(rd), 512, ptr(gd), ptr(out), 512, 1, stream()))
    assert torch.isfinite(out).all() and rel_err(out.cpu(), ref) < 1e-4


def test_nchw_to_nhwc(lib):
    x = torch.randn(3, 32, 32, 32)
    out = torch.empty(3, 1024, 32, device=dev())
    xd = x.to(dev())
    ok(lib.casync_op_nchw_to_nhwc(ptr(xd), ptr(out), 3, 32, 1024, stream()))
    assert torch.equal(out.cpu(), x.reshape(3, 32, 1024).transpose(1, 2))


def test_inc_block(lib, recipe_sd, golden):
    """Whole `inc` inverted residual from the NCHW crop vs the oracle's module."""
    from calipsync_amd import recipe
    from oracle import unet_oracle
    sd = unet_oracle.to_torch(recipe_sd)
    x, _ = recipe.make_inputs(2)
    xt = torch.from_numpy(x)
    ref = unet_oracle.inverted_residual(sd, "inc.inconv.0", xt, 1, False)
    packed = torch.from_numpy(pack.fold(recipe_sd)["inc.inconv.0.fused"].astype(np.float32)).to(dev())
    out = torch.full((2, 160, 160, 64), 9.0, device=dev())
    xd = xt.to(dev())
    ok(lib.casync_op_inc(ptr(xd), ptr(packed), out.data_ptr() + 32 * 4, 64, 2, stream()))
    o = out.cpu()
    assert (o[..., :32] == 9).all()
    assert (o[..., 32:].permute(0, 3, 1, 2) - ref).abs().max() < 1e-5


def test_outc_head(lib):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 32, 160, 160, generator=g)
    w = torch.randn(3, 32, generator=g) / 4
    b = torch.randn(3, generator=g)
    ref = torch.sigmoid(torch.einsum("oc,bchw->bohw", w, x) + b[None, :, None, None])
    out = torch.empty(2, 3, 160, 160, device=dev())
    wd, bd = w.to(dev()), b.to(dev())
    ok(lib.casync_op_outc(ptr(nhwc(x)), 32, ptr(wd), ptr(bd), ptr(out), 2, stream()))
    assert (out.cpu() - ref).abs().max() < 2e-6


IR_CASES = [  # (state_dict prefix, cin, cout, stride, res, h, w)
    ("up4.conv.double_conv.0", 64, 32, 1, False, 160, 160),
    ("up4.conv.double_conv.1", 32, 32, 1, True, 160, 160),
    ("up3.conv.double_conv.0", 128, 32, 1, False, 80, 80),
    ("up3.conv.double_conv.1", 32, 32, 1, True, 24, 40),        # ragged tiles
    ("down1.maxpool_conv.0.double_conv.0", 32, 64, 2, False, 160, 160),
    ("down1.maxpool_conv.0.double_conv.1", 64, 64, 1, True, 80, 80),
    ("down2.maxpool_conv.0.double_conv.0", 64, 128, 2, False, 80, 80),   # 40-wide output: partial tile
    ("down2.maxpool_conv.0.double_conv.0", 64, 128, 2, False, 37, 51),   # odd sizes
    ("audio_model.conv1", 32, 64, 1, False, 32, 32),
    ("audio_model.conv2", 64, 128, 1, False, 32, 32),
]


@pytest.mark.parametrize("prefix,cin,cout,stride,res,h,w", IR_CASES)
def test_ir_fused_block(lib, recipe_sd, prefix, cin, cout, stride, res, h, w):
    """Fused inverted residual vs the oracle's module on the recipe weights, reading from and
    writing into channel slices of wider buffers (as the engine does for the concats)."""
    from oracle import unet_oracle
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    g = torch.Generator().manual_seed(h * 7 + cin)
    b = 2
    x = torch.randn(b, cin, h, w, generator=g)
    ref = unet_oracle.inverted_residual(sd, prefix, x, stride, res)
    ho, wo = ref.shape[2], ref.shape[3]
    ld_in, ld_out = cin + 32, cout + 16
    xin = torch.full((b, h, w, ld_in), 3.0)
    xin[..., 32:] = x.permute(0, 2, 3, 1)
    xin = xin.to(dev())
    out = torch.full((b, ho, wo, ld_out), -5.0, device=dev())
    T = lambda k: torch.from_numpy(f[f"{prefix}.{k}"].astype(np.float32)).contiguous().to(dev())
    w1, b1, wd, bd, w2, b2 = T("pw1.w"), T("pw1.b"), T("dw.w"), T("dw.b"), T("pw2.w"), T("pw2.b")
    ok(lib.casync_op_ir_fused(xin.data_ptr() + 32 * 4, ld_in, ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                              ptr(b2), out.data_ptr() + 16 * 4, ld_out, b, h, w, cin, cout, stride,
                              int(res), stream()))
    o = out.cpu()
    assert (o[..., :16] == -5).all()
    got = o[..., 16:].permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 3e-6, rel_err(got, ref)


def test_ir_fused_rejects_unknown_shape(lib):
    z = torch.zeros(1024, device=dev())
    st = lib.casync_op_ir_fused(ptr(z), 48, ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 48, 1, 4, 4,
                                48, 48, 1, 0, stream())
    assert st < 0 and b"no instance" in lib.casync_last_error()


@pytest.mark.parametrize("prefix,cin,h", [("up4.conv.double_conv.0", 64, 48), ("up3.conv.double_conv.0", 128, 32)])
def test_ir_fused_with_upsample(lib, recipe_sd, prefix, cin, h):
    """Up.forward's interpolate + cat + first inverted residual in one kernel vs the oracle."""
    from oracle import unet_oracle
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    g = torch.Generator().manual_seed(cin)
    b, c_lo = 2, cin // 2
    lo = torch.randn(b, c_lo, h // 2, h // 2, generator=g)
    skip = torch.randn(b, cin - c_lo, h, h, generator=g)
    up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=True)
    ref = unet_oracle.inverted_residual(sd, prefix, torch.cat([up, skip], 1), 1, False)
    cat = torch.full((b, h, h, cin), 77.0)                   # upsampled half is never materialised
    cat[..., c_lo:] = skip.permute(0, 2, 3, 1)
    cat, lod = cat.to(dev()), nhwc(lo)
    out = torch.empty(b, h, h, 32, device=dev())
    T = lambda k: torch.from_numpy(f[f"{prefix}.{k}"].astype(np.float32)).contiguous().to(dev())
    w1, b1, wd, bd, w2, b2 = T("pw1.w"), T("pw1.b"), T("dw.w"), T("dw.b"), T("pw2.w"), T("pw2.b")
    ok(lib.casync_op_ir_fused_up(ptr(lod), c_lo, c_lo, ptr(cat), cin, ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                                 ptr(b2), ptr(out), 32, b, h, h, cin, 32, stream()))
    assert rel_err(nchw(out), ref) < 5e-6


@pytest.mark.parametrize("prefix,cin,h,w,b", [
    ("up4.conv.double_conv.0", 64, 48, 32, 2), ("up4.conv.double_conv.0", 64, 160, 160, 1), ("up4.conv.double_conv.0", 64, 20, 52, 3),
    ("up3.conv.double_conv.0", 128, 32, 48, 2), ("up3.conv.double_conv.0", 128, 80, 80, 1)])
def test_ir_fused_with_commuted_upsample(lib, recipe_sd, prefix, cin, h, w, b):
    """The fused Up block with the upsample commuted behind the expand conv (casync_op_ir_fused_upg): G = W1a . lo by the
    plain GEMM at the low resolution, the fused kernel runs over the skip half and adds up(G) from LDS.  Against the oracle
    (the reference's order: interpolate, cat, conv; module/unet.py:90-97) and against the upsample-on-load kernel; sizes
    with ragged tiles (20 x 52), the product sizes (160 / 80) and strided operands."""
    from oracle import unet_oracle
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    g = torch.Generator().manual_seed(cin + h)
    c_lo, cexp = cin // 2, 2 * cin
    lo = torch.randn(b, c_lo, h // 2, w // 2, generator=g)
    skip = torch.randn(b, cin - c_lo, h, w, generator=g)
    up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=True)
    ref = unet_oracle.inverted_residual(sd, prefix, torch.cat([up, skip], 1), 1, False)
    cat = torch.full((b, h, w, cin), 77.0)                   # upsampled half is never materialised
    cat[..., c_lo:] = skip.permute(0, 2, 3, 1)
    cat, lod = cat.to(dev()), nhwc(lo)
    T = lambda k: torch.from_numpy(f[f"{prefix}.{k}"].astype(np.float32)).contiguous().to(dev())
    w1, b1, wd, bd, w2, b2 = T("pw1.w"), T("pw1.b"), T("dw.w"), T("dw.b"), T("pw2.w"), T("pw2.b")
    w1a, w1b = T("pw1a.w"), T("pw1b.w")
    assert torch.equal(torch.cat([w1a, w1b], 1), w1)
    ld_g = cexp + 16
    G = torch.full((b * (h // 2) * (w // 2), ld_g), 55.0, device=dev())
    ok(lib.casync_op_pw_gemm(ptr(lod), c_lo, ptr(w1a), 0, ptr(G), ld_g, G.shape[0], cexp, c_lo, 0, 0, 0, 0, 0, 0, 0, 0, stream()))
    out = torch.full((b, h, w, 48), -5.0, device=dev())
    ok(lib.casync_op_ir_fused_upg(ptr(G), ld_g, cat.data_ptr() + c_lo * 4, cin, ptr(w1b), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                                  ptr(b2), out.data_ptr() + 16 * 4, 48, b, h, w, cin, 32, stream()))
    o = out.cpu()
    assert (o[..., :16] == -5).all()
    got = o[..., 16:].permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 5e-6, rel_err(got, ref)
    old = torch.empty(b, h, w, 32, device=dev())
    ok(lib.casync_op_ir_fused_up(ptr(lod), c_lo, c_lo, ptr(cat), cin, ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                                 ptr(b2), ptr(old), 32, b, h, w, cin, 32, stream()))
    assert rel_err(got, nchw(old)) < 3e-6


@pytest.mark.parametrize("hw,stride,cin,cexp,frames", [
    (10, 1, 512, 1024, 5), (10, 1, 1024, 2048, 2), (10, 1, 256, 512, 33), (16, 1, 256, 512, 3), (20, 1, 256, 512, 2),
    (20, 2, 256, 512, 3), (20, 1, 128, 256, 2), (10, 1, 64, 128, 1),
    (10, 1, 48, 96, 4), (16, 1, 16, 32, 2),        # cin % 32 != 0 at 2..9 frames: NOT the deep ring's 32-channel k-tiles
    (40, 1, 128, 256, 3), (40, 2, 128, 256, 2), (40, 1, 16, 32, 1), (40, 2, 64, 128, 17)])
def test_pw_dw_fused(lib, hw, stride, cin, cexp, frames):
    """Expand 1x1 + LeakyReLU + depthwise 3x3 + LeakyReLU in one kernel (pw_dw.hip) vs plain PyTorch: odd frame
    counts (a half-empty frame pair), both channel-tile widths, stride 2, strided operands."""
    g = torch.Generator().manual_seed(hw * 100 + cin + frames)
    x = torch.randn(frames, cin, hw, hw, generator=g)
    w1 = torch.randn(cexp, cin, generator=g) / cin ** 0.5
    b1 = torch.randn(cexp, generator=g) * 0.3
    wd = torch.randn(cexp, 1, 3, 3, generator=g) / 3
    bd = torch.randn(cexp, generator=g) * 0.3
    e = F.leaky_relu(F.conv2d(x.double(), w1.double()[:, :, None, None], b1.double()), 0.01)
    ref = F.leaky_relu(F.conv2d(e, wd.double(), bd.double(), stride, 1, 1, cexp), 0.01).float()
    ho = ref.shape[2]
    lda, ldd = cin + 32, cexp + 16
    xin = torch.full((frames, hw, hw, lda), 5.0)
    xin[..., 32:] = x.permute(0, 2, 3, 1)
    xin = xin.to(dev())
    out = torch.full((frames, ho, ho, ldd), -7.0, device=dev())
    wdp = wd.reshape(cexp, 9).T.contiguous().to(dev())          # tap-major [9][C]
    w1d, b1d, bdd = w1.to(dev()), b1.to(dev()), bd.to(dev())
    ok(lib.casync_op_pw_dw(xin.data_ptr() + 32 * 4, lda, ptr(w1d), ptr(b1d), ptr(wdp), ptr(bdd), out.data_ptr() + 16 * 4, ldd,
                           frames, hw, stride, cin, cexp, 0, 0, stream()))
    o = out.cpu()
    assert (o[..., :16] == -7).all()
    assert rel_err(o[..., 16:].permute(0, 3, 1, 2), ref) < 3e-6


@pytest.mark.parametrize("hw,c_lo,cexp,frames,fused", [(20, 256, 1024, 3, True), (20, 256, 1024, 17, True), (40, 128, 512, 2, False),
                                                   (20, 64, 256, 2, False), (16, 32, 128, 1, True), (40, 128, 512, 3, True),
                                                   (40, 32, 64, 1, True)])
def test_up_block_expand_with_commuted_upsample(lib, hw, c_lo, cexp, frames, fused):
    """An Up block's expand conv with the bilinear upsample commuted behind it (module/unet.py:90-96 + 17-20):
    lrelu(W1 . cat(up(lo), skip) + b) == lrelu(up(W1a . lo) + W1b . skip + b).  `G = W1a . lo` runs at the low resolution
    through the plain GEMM; the consumer is the GEMM epilogue (casync_op_pw_gemm_ups) or, with the depthwise conv behind
    it, casync_op_pw_dw.  Reference: plain PyTorch in float64, the reference's own order of operations."""
    g = torch.Generator().manual_seed(hw + c_lo + frames)
    lo = torch.randn(frames, c_lo, hw // 2, hw // 2, generator=g)
    skip = torch.randn(frames, c_lo, hw, hw, generator=g)
    w1 = torch.randn(cexp, 2 * c_lo, generator=g) / (2 * c_lo) ** 0.5
    b1 = torch.randn(cexp, generator=g) * 0.3
    x = torch.cat([F.interpolate(lo.double(), scale_factor=2, mode="bilinear", align_corners=True), skip.double()], 1)
    e = F.leaky_relu(F.conv2d(x, w1.double()[:, :, None, None], b1.double()), 0.01)
    lod, skd = nhwc(lo), nhwc(skip)
    w1a, w1b, b1d = w1[:, :c_lo].contiguous().to(dev()), w1[:, c_lo:].contiguous().to(dev()), b1.to(dev())
    G = torch.empty(frames * (hw // 2) ** 2, cexp, device=dev())
    ok(lib.casync_op_pw_gemm(ptr(lod), c_lo, ptr(w1a), 0, ptr(G), cexp, G.shape[0], cexp, c_lo, 0, 0, 0, 0, 0, 0, 0, 0, stream()))
    if not fused:
        out = torch.empty(frames * hw * hw, cexp, device=dev())
        ok(lib.casync_op_pw_gemm_ups(ptr(skd), c_lo, ptr(w1b), ptr(b1d), ptr(out), cexp, out.shape[0], cexp, c_lo, 1, ptr(G), cexp,
                                     hw, hw, stream()))
        got = out.view(frames, hw, hw, cexp).permute(0, 3, 1, 2).cpu()
        assert rel_err(got, e.float()) < 3e-6
        return
    wd = torch.randn(cexp, 1, 3, 3, generator=g) / 3
    bd = torch.randn(cexp, generator=g) * 0.3
    ref = F.leaky_relu(F.conv2d(e, wd.double(), bd.double(), 1, 1, 1, cexp), 0.01).float()
    out = torch.empty(frames, hw, hw, cexp, device=dev())
    wdp, bdd = wd.reshape(cexp, 9).T.contiguous().to(dev()), bd.to(dev())
    ok(lib.casync_op_pw_dw(ptr(skd), c_lo, ptr(w1b), ptr(b1d), ptr(wdp), ptr(bdd), ptr(out), cexp, frames, hw, 1, c_lo, cexp,
                           ptr(G), cexp, stream()))
    assert rel_err(out.permute(0, 3, 1, 2).cpu(), ref) < 3e-6


@pytest.fixture()
def bf16_ops(lib):
    lib.casync_op_set_dtype(1)
    yield lib
    lib.casync_op_set_dtype(0)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (1000, 64, 128), (6400, 1024, 512), (257, 32, 128), (300, 256, 1152)])
def test_pw_gemm_bf16(bf16_ops, m, n, k):
    """bf16 operands, fp32 accumulate: exact products of the bf16-rounded inputs, so the only
    error vs an fp64 reference on the SAME rounded inputs is the final bf16 rounding (2^-9)."""
    lib = bf16_ops
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).bfloat16()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).bfloat16()
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g).bfloat16()
    ref = F.leaky_relu(a.double() @ w.double().T + b.double(), 0.01) + res.double()
    ad, wd, bd, rd = a.to(dev()), w.to(dev()), b.to(dev()), res.to(dev())
    c = torch.empty(m, n, device=dev(), dtype=torch.bfloat16)
    ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, 0, 0, 0, ptr(rd), n, 0, 0, stream()))
    err = (c.cpu().double() - ref).abs() / (ref.abs() + 1.0)
    assert float(err.max()) < 2 ** -8, float(err.max())


@pytest.mark.parametrize("dwm", [2, 0])
@pytest.mark.parametrize("prefix,cin,cout,stride,res,h,w", IR_CASES)
def test_ir_fused_block_bf16(bf16_ops, recipe_sd, prefix, cin, cout, stride, res, h, w, dwm):
    """bf16 fused inverted residual (bf16 MFMA, bf16 E/D tiles) vs the fp32 oracle module on the
    same bf16-rounded input and weights: the error is the bf16 rounding of E, D and the output --
    and, with the depthwise conv on the matrix pipe (`ir_dw_mfma`, round 6), of the nine taps."""
    from oracle import unet_oracle
    lib = bf16_ops
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    g = torch.Generator().manual_seed(h * 7 + cin)
    b = 2
    x = torch.randn(b, cin, h, w, generator=g).bfloat16().float()
    ref = unet_oracle.inverted_residual(sd, prefix, x, stride, res)
    ho, wo = ref.shape[2], ref.shape[3]
    ld_in, ld_out = cin + 32, cout + 16
    xin = torch.full((b, h, w, ld_in), 3.0)
    xin[..., 32:] = x.permute(0, 2, 3, 1)
    xin = xin.bfloat16().to(dev())
    out = torch.full((b, ho, wo, ld_out), -5.0, device=dev(), dtype=torch.bfloat16)
    F32 = lambda k: torch.from_numpy(f[f"{prefix}.{k}"].astype(np.float32)).contiguous().to(dev())
    B16 = lambda k: F32(k).bfloat16()
    w1, b1, wd, bd, w2, b2 = B16("pw1.w"), F32("pw1.b"), F32("dw.w"), F32("dw.b"), B16("pw2.w"), F32("pw2.b")
    with options(ir_dw_mfma=dwm):
        ok(lib.casync_op_ir_fused(xin.data_ptr() + 32 * 2, ld_in, ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                                  ptr(b2), out.data_ptr() + 16 * 2, ld_out, b, h, w, cin, cout, stride,
                                  int(res), stream()))
    o = out.float().cpu()
    assert (o[..., :16] == -5).all()
    got = o[..., 16:].permute(0, 3, 1, 2)
    err = rel_err(got, ref)
    mean = float((got - ref).abs().mean() / ref.abs().max())
    print(f"ir_fused bf16 {prefix} dwm={dwm}: max rel {err:.3e} mean rel {mean:.3e}")
    assert err < 2e-2, err
    assert mean < 1.5e-3, mean


def test_ir_fused_with_upsample_bf16(bf16_ops, recipe_sd):
    from oracle import unet_oracle
    lib = bf16_ops
    sd = unet_oracle.to_torch(recipe_sd)
    f = pack.fold(recipe_sd)
    prefix, cin, h = "up4.conv.double_conv.0", 64, 48
    g = torch.Generator().manual_seed(9)
    b, c_lo = 2, cin // 2
    lo = torch.randn(b, c_lo, h // 2, h // 2, generator=g).bfloat16().float()
    skip = torch.randn(b, cin - c_lo, h, h, generator=g).bfloat16().float()
    up = F.interpolate(lo, scale_factor=2, mode="bilinear", align_corners=True)
    ref = unet_oracle.inverted_residual(sd, prefix, torch.cat([up, skip], 1), 1, False)
    cat = torch.full((b, h, h, cin), 77.0)
    cat[..., c_lo:] = skip.permute(0, 2, 3, 1)
    cat, lod = cat.bfloat16().to(dev()), nhwc(lo).bfloat16()
    out = torch.empty(b, h, h, 32, device=dev(), dtype=torch.bfloat16)
    F32 = lambda k: torch.from_numpy(f[f"{prefix}.{k}"].astype(np.float32)).contiguous().to(dev())
    w1, b1, wd, bd, w2, b2 = F32("pw1.w").bfloat16(), F32("pw1.b"), F32("dw.w"), F32("dw.b"), F32("pw2.w").bfloat16(), F32("pw2.b")
    for dwm in (2, 0):
        with options(ir_dw_mfma=dwm):
            ok(lib.casync_op_ir_fused_up(ptr(lod), c_lo, c_lo, ptr(cat), cin, ptr(w1), ptr(b1), ptr(wd), ptr(bd), ptr(w2),
                                         ptr(b2), ptr(out), 32, b, h, h, cin, 32, stream()))
        assert rel_err(nchw(out.float()), ref) < 2e-2, dwm


@pytest.mark.parametrize("b,h,w,c,stride", [(2, 40, 40, 512, 1), (3, 20, 20, 256, 2), (2, 10, 10, 2048, 1), (1, 21, 13, 8, 2)])
def test_dw3x3_bf16(bf16_ops, b, h, w, c, stride):
    lib = bf16_ops
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(b, c, h, w, generator=g).bfloat16().float()
    wt = torch.randn(c, 1, 3, 3, generator=g) / 3
    bias = torch.randn(c, generator=g)
    ref = F.leaky_relu(F.conv2d(x, wt, bias, stride, 1, 1, c), 0.01)
    wp = wt.reshape(c, 9).T.contiguous().to(dev())
    xd, bd = nhwc(x).bfloat16(), bias.to(dev())
    out = torch.empty(b, ref.shape[2], ref.shape[3], c, device=dev(), dtype=torch.bfloat16)
    ok(lib.casync_op_dw3x3(ptr(xd), ptr(wp), ptr(bd), ptr(out), b, h, w, c, stride, stream()))
    assert rel_err(nchw(out.float()), ref) < 2 ** -8



@pytest.mark.parametrize("bn", [64, 128])
@pytest.mark.parametrize("hw,stride,cin,cexp,frames", [
    (10, 1, 512, 1024, 5), (10, 1, 1024, 2048, 2), (10, 1, 256, 512, 33), (16, 1, 256, 512, 3), (20, 1, 256, 512, 2),
    (20, 2, 256, 512, 3), (20, 1, 128, 256, 2), (10, 1, 64, 128, 1), (20, 1, 512, 1024, 3),
    (40, 1, 128, 256, 3), (40, 2, 128, 256, 2), (40, 1, 32, 64, 1), (40, 2, 64, 128, 17), (40, 1, 256, 512, 2)])
def test_pw_dw_fused_bf16(bf16_ops, hw, stride, cin, cexp, frames, bn):
    """The bf16 engine's expand 1x1 + LeakyReLU + depthwise 3x3 + LeakyReLU kernel (pw_dw_bf16.hip: 64- / 128-channel tiles,
    bf16 E image in LDS) vs PyTorch in float64 on the SAME bf16-rounded operands, E rounded to bf16 where the kernel rounds
    it: what is left is the last-bit rounding of E (fp32 vs fp64 sums) and of the bf16 output.  Odd frame counts, strips,
    stride 2, strided operands, untouched columns beside D."""
    lib = bf16_ops
    if bn == 128 and hw > 16:
        pytest.skip("128-channel tiles exist for the 10x10 / 16x16 instances only")
    g = torch.Generator().manual_seed(hw * 100 + cin + frames)
    x = torch.randn(frames, cin, hw, hw, generator=g).bfloat16()
    w1 = (torch.randn(cexp, cin, generator=g) / cin ** 0.5).bfloat16()
    b1 = torch.randn(cexp, generator=g) * 0.3
    wd = torch.randn(cexp, 1, 3, 3, generator=g) / 3
    bd = torch.randn(cexp, generator=g) * 0.3
    e = F.leaky_relu(F.conv2d(x.double(), w1.double()[:, :, None, None], b1.double()), 0.01).float().bfloat16()
    ref = F.leaky_relu(F.conv2d(e.double(), wd.double(), bd.double(), stride, 1, 1, cexp), 0.01).float()
    ho = ref.shape[2]
    lda, ldd = cin + 32, cexp + 16
    xin = torch.full((frames, hw, hw, lda), 5.0, dtype=torch.bfloat16)
    xin[..., 32:] = x.permute(0, 2, 3, 1)
    xin = xin.to(dev())
    out = torch.full((frames, ho, ho, ldd), -7.0, device=dev(), dtype=torch.bfloat16)
    wdp = wd.reshape(cexp, 9).T.contiguous().to(dev())          # tap-major [9][C], fp32
    w1d, b1d, bdd = w1.to(dev()), b1.to(dev()), bd.to(dev())
    with options(fuse_dw_bf16_bn=bn):
        ok(lib.casync_op_pw_dw(xin.data_ptr() + 32 * 2, lda, ptr(w1d), ptr(b1d), ptr(wdp), ptr(bdd), out.data_ptr() + 16 * 2, ldd,
                               frames, hw, stride, cin, cexp, 0, 0, stream()))
    o = out.float().cpu()
    assert (o[..., :16] == -7).all()
    got = o[..., 16:].permute(0, 3, 1, 2)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) / scale < 2 ** -7
    assert float((got - ref).abs().mean()) / scale < 2 ** -11


def test_pw_dw_fused_bf16_rejects_bad_shapes(bf16_ops):
    lib = bf16_ops
    z = torch.zeros(4096, device=dev(), dtype=torch.bfloat16)
    zf = torch.zeros(4096, device=dev())
    for hw, stride, cin, cexp in ((12, 1, 64, 128), (10, 2, 64, 128), (20, 1, 48, 128), (20, 1, 64, 96)):
        assert lib.casync_op_pw_dw(ptr(z), cin, ptr(z), ptr(zf), ptr(zf), ptr(zf), ptr(z), cexp, 1, hw, stride, cin, cexp, 0, 0, stream()) != 0
    assert lib.casync_op_pw_dw(ptr(z), 64, ptr(z), ptr(zf), ptr(zf), ptr(zf), ptr(z), 128, 1, 10, 1, 64, 128, ptr(zf), 128, stream()) != 0


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("b,scale", [(1, 0.5), (3, 0.5), (70, 0.5), (2, 4.0)])
def test_cross_attention_bf16(bf16_ops, b, scale, mfma):
    """The bf16 engine's attention core (module/unet.py:209-217): attention_bf16.hip -- both products on bf16 matrix
    instructions, the softmax in the registers of two lanes, V read through the hardware transpose read -- and the
    fp32-MFMA kernel on bf16 storage (att_bf16 = 0), against float64 on the SAME bf16-rounded operands.  Q / K / V are
    column slices of wider buffers (the engine's p1q and KV rows), `scale` 4 gives peaked (near one-hot) softmax rows.
    What is left is the bf16 rounding of P (2^-9 on weights <= 1) and of the output."""
    lib = bf16_ops
    g = torch.Generator().manual_seed(17 + b)
    p1q = (torch.randn(b, 100, 576, generator=g) * scale).bfloat16()          # [ox | q]: res = columns 0..511, q = 512..575
    kv = (torch.randn(b, 100, 2304, generator=g) * scale).bfloat16()
    gamma = torch.tensor([-0.75])
    q, res = p1q[:, :, 512:], p1q[:, :, :512]
    k, v = kv[:, :, 1152:1216], kv[:, :, 1216:1728]                          # block 2's slice of the KV rows
    att = torch.softmax(q.double() @ k.double().transpose(1, 2), -1)
    ref = (gamma.double() * (att @ v.double()) + res.double()).float()
    pd, kvd, gd = p1q.to(dev()), kv.to(dev()), gamma.to(dev())
    out = torch.full((b, 100, 520), 3.0, device=dev(), dtype=torch.bfloat16)
    with options(att_bf16=mfma):
        ok(lib.casync_op_cross_attention(pd.data_ptr() + 512 * 2, 576, kvd.data_ptr() + 1152 * 2, 2304, kvd.data_ptr() + 1216 * 2, 2304,
                                         ptr(pd), 576, ptr(gd), ptr(out), 520, b, stream()))
    o = out.float().cpu()
    assert (o[..., 512:] == 3).all()
    err = (o[..., :512] - ref).abs()
    top = float(ref.abs().max())
    assert torch.isfinite(o).all() and float(err.max()) / top < 2 ** -7 and float(err.mean()) / top < 2 ** -10


@pytest.mark.parametrize("hw,c_lo,cexp,frames", [(20, 256, 1024, 3), (20, 256, 1024, 17), (40, 128, 512, 2), (40, 128, 512, 5), (20, 64, 128, 1)])
def test_up_block_expand_with_commuted_upsample_bf16(bf16_ops, hw, c_lo, cexp, frames):
    """The first inverted residual of an Up stage in the bf16 engine (module/unet.py:90-97 then :17-30): the upsample commuted
    behind the expand conv -- G = W1a . lo by a bf16 GEMM at the low resolution, pw_dw_bf16 runs W1b . skip and adds the
    bilinear x2 upsample of G from an LDS tile before the LeakyReLU, then the depthwise conv -- against PyTorch in float64 in
    the REFERENCE's order (upsample, cat, conv) on the same bf16-rounded operands.  Differences: the bf16 rounding of G (a
    second rounding the un-commuted chain does not have), of E and of the output."""
    lib = bf16_ops
    g = torch.Generator().manual_seed(hw + c_lo + frames)
    lo = torch.randn(frames, c_lo, hw // 2, hw // 2, generator=g).bfloat16()
    skip = torch.randn(frames, c_lo, hw, hw, generator=g).bfloat16()
    w1 = (torch.randn(cexp, 2 * c_lo, generator=g) / (2 * c_lo) ** 0.5).bfloat16()
    b1 = torch.randn(cexp, generator=g) * 0.3
    wd = torch.randn(cexp, 1, 3, 3, generator=g) / 3
    bd = torch.randn(cexp, generator=g) * 0.3
    x = torch.cat([F.interpolate(lo.double(), scale_factor=2, mode="bilinear", align_corners=True), skip.double()], 1)
    e = F.leaky_relu(F.conv2d(x, w1.double()[:, :, None, None], b1.double()), 0.01)
    ref = F.leaky_relu(F.conv2d(e, wd.double(), bd.double(), 1, 1, 1, cexp), 0.01).float()
    w1a, w1b = w1[:, :c_lo].contiguous().to(dev()), w1[:, c_lo:].contiguous().to(dev())
    lod, skd = nhwc(lo), nhwc(skip)
    m_lo = frames * (hw // 2) ** 2
    G = torch.empty(m_lo, cexp, device=dev(), dtype=torch.bfloat16)
    ok(lib.casync_op_pw_gemm(ptr(lod), c_lo, ptr(w1a), 0, ptr(G), cexp, m_lo, cexp, c_lo, 0, 0, 0, 0, 0, 0, 0, 0, stream()))
    out = torch.empty(frames, hw, hw, cexp, device=dev(), dtype=torch.bfloat16)
    wdp, b1d, bdd = wd.reshape(cexp, 9).T.contiguous().to(dev()), b1.to(dev()), bd.to(dev())
    ok(lib.casync_op_pw_dw(ptr(skd), c_lo, ptr(w1b), ptr(b1d), ptr(wdp), ptr(bdd), ptr(out), cexp, frames, hw, 1, c_lo, cexp,
                           ptr(G), cexp, stream()))
    got = out.float().permute(0, 3, 1, 2).cpu()
    top = float(ref.abs().max())
    err = (got - ref).abs()
    assert float(err.max()) / top < 2 ** -6 and float(err.mean()) / top < 2 ** -10


@pytest.mark.parametrize("m,n,k", [(25600, 512, 256), (33001, 256, 512), (40000, 128, 1024)])
def test_pw_gemm_bf16_two_stage_ring_128(bf16_ops, m, n, k):
    """Round-6 experiment instance (`gemm_ring128`): the 128x128 bf16 tile on a two-stage LDS-DMA ring, launches of more than
    256 tiles (ragged M included), against the register-staged kernel on the same operands: bit-identical (same products, same
    k order) and within the bf16 rounding of an fp64 reference."""
    lib = bf16_ops
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).bfloat16()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).bfloat16()
    b = torch.randn(n, generator=g)
    ad, wd, bd = a.to(dev()), w.to(dev()), b.to(dev())
    outs = []
    for ring in (0, 1):
        c = torch.full((m, n), 9.0, device=dev(), dtype=torch.bfloat16)
        with options(gemm_ring128=ring, gemm_cfg=0):
            ok(lib.casync_op_pw_gemm(ptr(ad), k, ptr(wd), ptr(bd), ptr(c), n, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, stream()))
        outs.append(c.cpu())
    ref = F.leaky_relu(a[:4096].double() @ w.double().T + b.double(), 0.01)
    err = (outs[1][:4096].double() - ref).abs() / (ref.abs() + 1.0)
    assert float(err.max()) < 2 ** -8
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("mfma", [1, 0])
def test_inc_block_bf16(bf16_ops, recipe_sd, mfma):
    """`inc` of the bf16 engine from the NCHW fp32 crop vs the oracle's module: the projection on the matrix pipe
    (`inc_mfma`, round 6: depthwise output and W2 rounded to bf16 before the product) and the round-5 VALU kernel; a padded
    leading dimension, nothing written outside the slice."""
    from calipsync_amd import recipe
    from oracle import unet_oracle
    lib = bf16_ops
    sd = unet_oracle.to_torch(recipe_sd)
    x, _ = recipe.make_inputs(3)
    xt = torch.from_numpy(x)
    ref = unet_oracle.inverted_residual(sd, "inc.inconv.0", xt, 1, False)
    packed = torch.from_numpy(pack.fold(recipe_sd)["inc.inconv.0.fused"].astype(np.float32)).to(dev())
    out = torch.full((3, 160, 160, 64), 9.0, device=dev(), dtype=torch.bfloat16)
    xd = xt.to(dev())
    with options(inc_mfma=mfma):
        ok(lib.casync_op_inc(ptr(xd), ptr(packed), out.data_ptr() + 32 * 2, 64, 3, stream()))
    o = out.float().cpu()
    assert (o[..., :32] == 9).all()
    got = o[..., 32:].permute(0, 3, 1, 2)
    err, mean = rel_err(got, ref), float((got - ref).abs().mean() / ref.abs().max())
    print(f"inc bf16 mfma={mfma}: max rel {err:.3e} mean rel {mean:.3e}")
    assert err < (6e-3 if mfma else 4e-3) and mean < 6e-4
