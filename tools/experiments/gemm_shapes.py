#!/usr/bin/env python3
"""Per-launch time of casync_op_pw_gemm on a list of shapes under sets of engine options (HIP events, back to back).

    python tools/experiments/gemm_shapes.py bf16 "25600,1024,512;25600,512,1024" "gemm_pf=1" "gemm_pf=2" "gemm_cfg=1"
Prints one line per shape: us per launch and TFLOP/s for every option set."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
bf16 = sys.argv[1] == "bf16"
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[2].split(";")]
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split()) for a in sys.argv[3:]] or [{}]
lib.casync_op_set_dtype(1 if bf16 else 0)
tdt = torch.bfloat16 if bf16 else torch.float32
s = torch.cuda.current_stream().cuda_stream
for m, n, k in shapes:
    pad = int(os.environ.get("LDA_PAD", "0"))      # extra elements per A row (leading dimension k + pad)
    cpad = int(os.environ.get("LDC_PAD", "0"))
    abuf = torch.randn(m, k + pad, device=dev).to(tdt)
    a = abuf
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(tdt)
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n + cpad, device=dev, dtype=tdt)

    def run():
        st = lib.casync_op_pw_gemm(a.data_ptr(), k + pad, w.data_ptr(), bias.data_ptr(), c.data_ptr(), n + cpad, m, n, k, 1, 0, 0, 0, 0, 0, 0, 0, s)
        assert st == 0, lib.casync_last_error()
    line = f"M={m:7d} N={n:5d} K={k:5d} |"
    for opts in sets:
        old = {kk: _lib.get_option(kk) for kk in opts}
        for kk, v in opts.items():
            _lib.set_option(kk, v)
        for _ in range(200):
            run()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(100):
            run()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 10
        line += f" {us:7.1f} us {2.0 * m * n * k / us / 1e6:6.0f} TF |"
        for kk, v in old.items():
            _lib.set_option(kk, v)
    print(line, flush=True)
