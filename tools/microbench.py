#!/usr/bin/env python3
"""Micro-benchmarks of single engine operators through the C ABI (GPU box only).

    python tools/microbench.py ir   [--batch 64] [--iters 20]     # fused inverted-residual shapes
    python tools/microbench.py gemm [--batch 64]                  # the plan's GEMM shapes
Times with HIP events on torch's current stream; prints ms, TFLOP/s and GB/s per shape."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from calipsync_amd import _lib  # noqa: E402

DEV = torch.device("cuda:0")


def time_ms(fn, iters):
    # warm the clock as well as the caches: after ~1 s idle the chip restarts at ~2.1 GHz and takes tens of
    # ms of load to reach its 2.39 GHz (profiles/r2_mfma_clock.txt); ~0.2 s of the op itself first
    import time
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


IR_SHAPES = [("up4.ir0", 64, 32, 1, 0, 160), ("up4.ir1", 32, 32, 1, 1, 160), ("up3.ir0", 128, 32, 1, 0, 80),
             ("up3.ir1", 32, 32, 1, 1, 80), ("down1.ir0", 32, 64, 2, 0, 160), ("down1.ir1", 64, 64, 1, 1, 80),
             ("down2.ir0", 64, 128, 2, 0, 80), ("audio.conv1", 32, 64, 1, 0, 32), ("audio.conv2", 64, 128, 1, 0, 32)]


def bench_ir(args):
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    bf = args.dtype == "bf16"
    lib.casync_op_set_dtype(1 if bf else 0)
    tdt = torch.bfloat16 if bf else torch.float32
    for name, cin, cout, stride, res, hw in IR_SHAPES:
        if args.only and args.only not in name:
            continue
        B, ce = args.batch, 2 * cin
        ho = hw // stride
        x = torch.randn(B, hw, hw, cin, device=DEV).to(tdt)
        w1, b1 = (torch.randn(ce, cin, device=DEV) / cin ** 0.5).to(tdt), torch.randn(ce, device=DEV)
        wd, bd = torch.randn(9, ce, device=DEV) / 3, torch.randn(ce, device=DEV)
        w2, b2 = (torch.randn(cout, ce, device=DEV) / ce ** 0.5).to(tdt), torch.randn(cout, device=DEV)
        out = torch.empty(B, ho, ho, cout, device=DEV, dtype=tdt)

        def fn():
            st = lib.casync_op_ir_fused(x.data_ptr(), cin, w1.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                        w2.data_ptr(), b2.data_ptr(), out.data_ptr(), cout, B, hw, hw, cin, cout,
                                        stride, res, s)
            assert st == 0, lib.casync_last_error()
        ms = time_ms(fn, args.iters)
        m_in, m_out = B * hw * hw, B * ho * ho
        flops = 2.0 * (m_in * cin * ce + 9 * m_out * ce + m_out * ce * cout)
        byts = x.element_size() * (m_in * cin + m_out * cout)
        print(f"{name:12s} cin={cin:3d} cout={cout:3d} s={stride} hw={hw:3d}  {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TF  "
              f"{byts / ms / 1e6:7.1f} GB/s", flush=True)


GEMM_SHAPES = [  # (name, rows per frame, N, K)
    ("sq4096", 64, 4096, 4096), ("big", 400, 1024, 1024),
    ("p1", 100, 512, 1024), ("b1", 100, 1024, 512), ("fc", 100, 1024, 1024), ("kv", 100, 2304, 512),
    ("fuse.pw1", 100, 2048, 1024), ("fuse.pw2", 100, 512, 2048), ("q", 100, 64, 512),
    ("conv5", 100, 512, 2304), ("conv3", 256, 256, 1152), ("d4.pw1", 400, 512, 256), ("u1.pw1", 400, 1024, 512),
    ("u1.pw2", 400, 128, 1024), ("u2.pw1", 1600, 512, 256), ("u2.pw2", 1600, 64, 512), ("d3.pw1", 1600, 256, 128),
    ("d2.ir1.pw1", 1600, 256, 128), ("d2.ir1.pw2", 1600, 128, 256)]


def bench_gemm(args):
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lib.casync_op_set_dtype(1 if args.dtype == "bf16" else 0)
    shapes = GEMM_SHAPES
    if args.shape:   # --shape M,N,K[;M,N,K...]: explicit sizes (M taken as is)
        shapes = [("custom", int(v.split(",")[0]) / args.batch, int(v.split(",")[1]), int(v.split(",")[2]))
                  for v in args.shape.split(";")]
    for name, rows, n, k in shapes:
        if args.only and args.only not in name:
            continue
        m = int(round(rows * args.batch))
        if k % (64 if args.dtype == "bf16" else 32):
            continue
        # --rotate R: R operand sets used round-robin, so that with R x (A + W + C) > 256 MB every launch finds
        # its operands in HBM (as inside the network), not in the L2 / Infinity Cache a repeated launch leaves warm
        sets = [(torch.randn(m, k, device=DEV).to(tdt), (torch.randn(n, k, device=DEV) / k ** 0.5).to(tdt),
                 torch.empty(m, n, device=DEV, dtype=tdt)) for _ in range(max(1, args.rotate))]
        a = sets[0][0]
        bias = torch.randn(n, device=DEV)
        turn = [0]

        def fn():
            a_, w_, c_ = sets[turn[0] % len(sets)]
            turn[0] += 1
            st = lib.casync_op_pw_gemm(a_.data_ptr(), k, w_.data_ptr(), bias.data_ptr(), c_.data_ptr(), n, m, n, k, 1,
                                       0, 0, 0, 0, 0, 0, 0, s)
            assert st == 0, lib.casync_last_error()
        ms = time_ms(fn, args.iters)
        gb = (m * k + m * n + n * k) * a.element_size() / ms / 1e6
        print(f"{name:12s} M={m:7d} N={n:5d} K={k:5d}  {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:6.1f} TF  {gb:7.1f} GB/s", flush=True)


DW_SHAPES = [(10, 1024), (10, 2048), (16, 512), (20, 512), (20, 1024), (40, 256), (40, 512), (80, 128)]


def bench_dw(args):
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    bf = args.dtype == "bf16"
    lib.casync_op_set_dtype(1 if bf else 0)
    tdt = torch.bfloat16 if bf else torch.float32
    for hw, c in DW_SHAPES:
        B = args.batch
        x = torch.randn(B, hw, hw, c, device=DEV).to(tdt)
        w, b = torch.randn(9, c, device=DEV) / 3, torch.randn(c, device=DEV)
        out = torch.empty_like(x)

        def fn():
            st = lib.casync_op_dw3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, hw, hw, c, 1, s)
            assert st == 0, lib.casync_last_error()
        ms = time_ms(fn, args.iters)
        gb = 2 * x.numel() * x.element_size() / ms / 1e6
        print(f"dw {hw:3d}x{hw:<3d} C={c:5d}  {ms * 1e3:8.1f} us  {gb:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["ir", "gemm", "dw"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--shape", default="")
    ap.add_argument("--rotate", type=int, default=1, help="gemm: operand sets used round-robin (cold operands)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    a = ap.parse_args()
    {"ir": bench_ir, "gemm": bench_gemm, "dw": bench_dw}[a.what](a)
