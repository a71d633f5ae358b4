#!/usr/bin/env python3
"""Round 6: which single bf16 operator, looping on a second host thread, disturbs the fp32 model's forward?
    python tools/experiments/op_beside_model.py      (CASYNC_NO_FWD_GATE has no bearing: operators do not take the gate)"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib, pack, recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
sd_np = recipe.make_state_dict()
sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
folded = pack.fold(sd_np)
x, a = recipe.make_inputs_range(0, int(os.environ.get("FRAMES", "96")))   # FRAMES=8: the small-batch plan (pw_dw deep ring, single lane)
xt, at = torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev)
net = Model(6, "hubert").to(dev)
net.load_state_dict(sd)
ref = net(xt, at).clone()
torch.cuda.synchronize()
ITERS = int(os.environ.get("ITERS", "30"))


def F32(k):
    return torch.from_numpy(folded[k].astype(np.float32)).contiguous().to(dev)


def op_ir(prefix, cin, cout, stride, res, hw, frames=64, dtype=1):
    tdt = torch.bfloat16 if dtype else torch.float32
    xin = torch.randn(frames, hw, hw, cin, device=dev).to(tdt)
    ho = (hw + 2 - 3) // stride + 1
    out = torch.empty(frames, ho, ho, cout, device=dev, dtype=tdt)
    w1, b1, wd, bd, w2, b2 = F32(prefix + ".pw1.w").to(tdt), F32(prefix + ".pw1.b"), F32(prefix + ".dw.w"), F32(prefix + ".dw.b"), F32(prefix + ".pw2.w").to(tdt), F32(prefix + ".pw2.b")
    keep = (xin, out, w1, b1, wd, bd, w2, b2)

    def run(s):
        lib.casync_op_set_dtype(dtype)
        st = lib.casync_op_ir_fused(xin.data_ptr(), cin, w1.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                    out.data_ptr(), cout, frames, hw, hw, cin, cout, stride, int(res), s)
        assert st == 0, lib.casync_last_error()
    run.keep = keep
    return run


def op_gemm(m, n, k, dtype=1):
    tdt = torch.bfloat16 if dtype else torch.float32
    A = torch.randn(m, k, device=dev).to(tdt)
    W = (torch.randn(n, k, device=dev) / k ** 0.5).to(tdt)
    b = torch.randn(n, device=dev)
    C = torch.empty(m, n, device=dev, dtype=tdt)

    def run(s):
        lib.casync_op_set_dtype(dtype)
        st = lib.casync_op_pw_gemm(A.data_ptr(), k, W.data_ptr(), b.data_ptr(), C.data_ptr(), n, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, s)
        assert st == 0, lib.casync_last_error()
    run.keep = (A, W, b, C)
    return run


def op_inc(frames=64, dtype=1):
    tdt = torch.bfloat16 if dtype else torch.float32
    xin = torch.rand(frames, 6, 160, 160, device=dev)
    packed = F32("inc.inconv.0.fused")
    out = torch.empty(frames, 160, 160, 32, device=dev, dtype=tdt)

    def run(s):
        lib.casync_op_set_dtype(dtype)
        assert lib.casync_op_inc(xin.data_ptr(), packed.data_ptr(), out.data_ptr(), 32, frames, s) == 0
    run.keep = (xin, packed, out)
    return run


def beside(name, op):
    stop, bad = [False], []

    def load():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop[0]:
                for _ in range(8):
                    op(s.cuda_stream)
                s.synchronize()

    def work():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(ITERS):
                out = net(xt, at)
                s.synchronize()
                if not torch.equal(out, ref):
                    bad.append(i)
    tl, tw = threading.Thread(target=load), threading.Thread(target=work)
    tl.start(); tw.start(); tw.join(); stop[0] = True; tl.join()
    print(f"{name:64s} mismatching forwards: {len(bad)} of {ITERS}", flush=True)


def with_opts(op, **kv):
    def run(s):
        old = {k: _lib.get_option(k) for k in kv}
        for k, v in kv.items():
            _lib.set_option(k, v)
        try:
            op(s)
        finally:
            for k, v in old.items():
                _lib.set_option(k, v)
    return run


if os.environ.get("TWO_KERNELS"):
    # the smallest reproducer: ONE fp32 fused Up block (upsample commuted, casync_op_ir_fused_upg on up4.0's shape) looping on this
    # thread and compared with its own first result, ONE GEMM looping on the other thread
    frames, h, cin, cexp, c_lo = 32, 160, 64, 128, 32
    prefix = "up4.conv.double_conv.0"
    G_ = torch.randn(frames * 80 * 80, cexp, device=dev)
    skip = torch.randn(frames, h, h, cin - c_lo, device=dev)
    w1b, b1, wd, bd, w2, b2 = (F32(prefix + k) for k in (".pw1b.w", ".pw1.b", ".dw.w", ".dw.b", ".pw2.w", ".pw2.b"))
    out = torch.empty(frames, h, h, 32, device=dev)

    def upg(s):
        lib.casync_op_set_dtype(0)
        st = lib.casync_op_ir_fused_upg(G_.data_ptr(), cexp, skip.data_ptr(), cin - c_lo, w1b.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                        w2.data_ptr(), b2.data_ptr(), out.data_ptr(), 32, frames, h, h, cin, 32, s)
        assert st == 0, lib.casync_last_error()

    upg(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref_out = out.clone()

    def beside_op(name, load_op):
        stop, bad = [False], []

        def load():
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                while not stop[0]:
                    for _ in range(8):
                        load_op(s.cuda_stream)
                    s.synchronize()

        def work():
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for i in range(200):
                    upg(s.cuda_stream)
                    s.synchronize()
                    if not torch.equal(out, ref_out):
                        d = (out - ref_out).abs()
                        n_bad = int((d.amax(-1) > 0).sum())
                        import time
                        time.sleep(0.02)                              # ... is it a LATE write?  look again without relaunching
                        torch.cuda.synchronize()
                        later = int(((out - ref_out).abs().amax(-1) > 0).sum())
                        bad.append((i, n_bad, later))
        tl, tw = threading.Thread(target=load), threading.Thread(target=work)
        tl.start(); tw.start(); tw.join(); stop[0] = True; tl.join()
        print(f"fused fp32 Up block beside {name:58s} wrong results: {len(bad)} of 200   (pixels wrong at sync, 20 ms later: {[b[1:] for b in bad[:6]]})", flush=True)

    g16 = op_gemm(25600, 1024, 512)
    g32 = op_gemm(25600, 1024, 512, dtype=0)
    if os.environ.get("VICTIM"):
        # other fp32 victims in the same harness: the plain fused block (up4.1), the upsample-on-load block (UPS = 1), inc, a GEMM
        which = os.environ["VICTIM"]
        if which == "plain":
            victim = op_ir("up4.conv.double_conv.1", 32, 32, 1, True, 160, frames=32, dtype=0)
            vout = victim.keep[1]
        elif which == "down1":
            victim = op_ir("down1.maxpool_conv.0.double_conv.1", 64, 64, 1, True, 80, frames=64, dtype=0)
            vout = victim.keep[1]
        elif which == "ups1":
            lo1 = torch.randn(frames, 80, 80, 32, device=dev)
            cat1 = torch.randn(frames, h, h, 64, device=dev)
            w1 = F32(prefix + ".pw1.w")
            vout = torch.empty(frames, h, h, 32, device=dev)

            def victim(s):
                lib.casync_op_set_dtype(0)
                st = lib.casync_op_ir_fused_up(lo1.data_ptr(), 32, 32, cat1.data_ptr(), 64, w1.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(), w2.data_ptr(),
                                               b2.data_ptr(), vout.data_ptr(), 32, frames, h, h, 64, 32, s)
                assert st == 0, lib.casync_last_error()
        elif which == "gemm":
            victim = op_gemm(25600, 1024, 512, dtype=0)
            vout = victim.keep[3]
        elif which == "gemm_k32":
            victim = op_gemm(204800, 128, 32, dtype=0)
            vout = victim.keep[3]
        upg = victim
        out = vout
        upg(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref_out = out.clone()
        print("victim:", which)
    if os.environ.get("BURN"):
        # register-only matrix-instruction loops as the co-runner (tools/experiments/ubench/mfma_burn.hip, built to $BURN)
        import ctypes
        burn = ctypes.CDLL(os.environ["BURN"]).burn
        burn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(256, device=dev)
        for kind, name in ((0, "v_mfma_f32_32x32x16_bf16"), (1, "v_mfma_f32_16x16x32_bf16"), (2, "v_mfma_f32_32x32x2_f32"), (3, "v_mfma_f32_16x16x4_f32"),
                           (4, "v_pk_fma_f32 only")):
            beside_op("a register-only loop of " + name, lambda s, k=kind: burn(k, 20000, 1024, s, sink.data_ptr()))
        sys.exit(0)
    beside_op("nothing", lambda s: None)
    beside_op("bf16 GEMM 25600x1024x512, default tile", g16)
    beside_op("fp32 GEMM 25600x1024x512, default tile", g32)
    beside_op("bf16 ir_fused up4.1", op_ir("up4.conv.double_conv.1", 32, 32, 1, True, 160))
    sys.exit(0)

if os.environ.get("GEMM_MATRIX"):
    g16, g32 = op_gemm(25600, 1024, 512), op_gemm(25600, 1024, 512, dtype=0)
    for name, op in (("bf16 GEMM default (cost model)", g16),
                     ("bf16 GEMM gemm_cfg=0 (128x128 register-staged)", with_opts(g16, gemm_cfg=0)),
                     ("bf16 GEMM gemm_cfg=0 gemm_ring128=1 (128x128 two-stage ring)", with_opts(g16, gemm_cfg=0, gemm_ring128=1)),
                     ("bf16 GEMM gemm_cfg=1 (128x64 ring)", with_opts(g16, gemm_cfg=1)),
                     ("bf16 GEMM gemm_cfg=2 (64x64 ring)", with_opts(g16, gemm_cfg=2)),
                     ("bf16 GEMM gemm_cfg=1 gemm_glds=0 (128x64 register-staged)", with_opts(g16, gemm_cfg=1, gemm_glds=0)),
                     ("bf16 GEMM gemm_cfg=0 gemm_persist=0", with_opts(g16, gemm_cfg=0, gemm_persist=0)),
                     ("fp32 GEMM gemm_cfg=0 gemm_glds=0 (128x128 register-staged)", with_opts(g32, gemm_cfg=0, gemm_glds=0)),
                     ("fp32 GEMM gemm_cfg=1 (128x64 ring)", with_opts(g32, gemm_cfg=1)),
                     ("fp32 GEMM default", g32)):
        beside(name, op)
    sys.exit(0)
beside("bf16 ir_fused up4.1 (32->64->32, 160x160, residual)", op_ir("up4.conv.double_conv.1", 32, 32, 1, True, 160))
beside("fp32 ir_fused up4.1", op_ir("up4.conv.double_conv.1", 32, 32, 1, True, 160, dtype=0))
beside("bf16 ir_fused down1.0 (32->64->64, stride 2)", op_ir("down1.maxpool_conv.0.double_conv.0", 32, 64, 2, False, 160))
beside("bf16 ir_fused down1.1 (64->128->64, 80x80)", op_ir("down1.maxpool_conv.0.double_conv.1", 64, 64, 1, True, 80, frames=128))
beside("bf16 pw_gemm 25600 x 1024 x 512", op_gemm(25600, 1024, 512))
beside("bf16 pw_gemm 102400 x 128 x 512 (ring 128x64)", op_gemm(102400, 128, 512))
beside("fp32 pw_gemm 25600 x 1024 x 512", op_gemm(25600, 1024, 512, dtype=0))
beside("bf16 inc", op_inc())
beside("fp32 inc", op_inc(dtype=0))
