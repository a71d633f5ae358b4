#!/usr/bin/env python3
"""Per-phase shader cycles of the fused inverted-residual kernel (wave 0 of each workgroup), on the plan's
shapes: python tools/experiments/ir_timeline.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
RES = {"down1.ir1": 1}
s = torch.cuda.current_stream().cuda_stream
SHAPES = [("up4.ir0", 64, 32, 1, 160), ("up4.ir1", 32, 32, 1, 160), ("up3.ir0", 128, 32, 1, 80), ("down1.ir0", 32, 64, 2, 160),
          ("down1.ir1", 64, 64, 1, 80), ("down2.ir0", 64, 128, 2, 80)]
stamps = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
for name, cin, cout, stride, hw in SHAPES:
    ce, ho = 2 * cin, hw // stride
    x = torch.randn(B, hw, hw, cin, device=dev)
    w1, b1 = torch.randn(ce, cin, device=dev) / cin ** 0.5, torch.randn(ce, device=dev)
    wd, bd = torch.randn(9, ce, device=dev) / 3, torch.randn(ce, device=dev)
    w2, b2 = torch.randn(cout, ce, device=dev) / ce ** 0.5, torch.randn(cout, device=dev)
    out = torch.empty(B, ho, ho, cout, device=dev)

    def run():
        st = lib.casync_op_ir_fused(x.data_ptr(), cin, w1.data_ptr(), b1.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                    w2.data_ptr(), b2.data_ptr(), out.data_ptr(), cout, B, hw, hw, cin, cout, stride, 0, s)
        assert st == 0, lib.casync_last_error()
    for _ in range(100):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    stamps.zero_()
    lib.casync_debug_ir_stamps(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.casync_debug_ir_stamps(0)
    raw = stamps.cpu().numpy().reshape(-1, 8)
    raw = raw[raw[:, :5].sum(1) > 0]
    st = raw[:, :5].astype(np.float64)
    extra = ""
    if raw[:, 5].any():              # round 4: cycles wave 0 waited at the chunk barriers (not part of any phase)
        extra = f"  [+ barrier wait {np.median(raw[:, 5]) / (ce // 16):5.0f}/chunk, LDS-DMA issue {np.median(raw[:, 6]) / (ce // 16):5.0f}/chunk (inside P2 before round-4b)]"
    med = np.median(st, axis=0)
    tot = med.sum()
    nch = ce // 16
    flops = 2.0 * B * (hw * hw * cin * ce + 9 * ho * ho * ce + ho * ho * ce * cout)
    print(f"{name:10s} cin={cin:3d} cout={cout:3d} s={stride} hw={hw:3d}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF | wave-0 cycles per "
          f"workgroup {tot:8.0f}: prologue {med[0]:6.0f} ({100 * med[0] / tot:4.1f}%)  P1 {med[1]:6.0f} ({100 * med[1] / tot:4.1f}%, "
          f"{med[1] / nch:5.0f}/chunk)  P2 {med[2]:6.0f} ({100 * med[2] / tot:4.1f}%, {med[2] / nch:5.0f}/chunk)  P3 {med[3]:6.0f} "
          f"({100 * med[3] / tot:4.1f}%, {med[3] / nch:5.0f}/chunk)  epilogue {med[4]:6.0f} ({100 * med[4] / tot:4.1f}%)" + extra)
