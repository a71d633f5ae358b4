#!/usr/bin/env python3
"""Where does a GEMM launch spend its time?  In-kernel 100 MHz stamps (casync_debug_gemm_stamps) per
workgroup: entry, end of the first tile's k loop, end of its epilogue, (second tile), exit.

    python tools/experiments/gemm_timeline.py "6400,512,1024;3200,512,1024" [cfg] [streamk] [bf16]
Prints, per shape: launch duration from HIP events (back-to-back average, clock-warmed) and from the
stamps (last exit - first entry), spread of workgroup entry times, median k-loop / epilogue durations of
the first tile, tiles per workgroup, and the idle tail (median exit vs last exit)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1].split(";")]
if len(sys.argv) > 2:
    _lib.set_option("gemm_cfg", int(sys.argv[2]))
if len(sys.argv) > 3:
    _lib.set_option("gemm_streamk", int(sys.argv[3]))
bf16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"
lib.casync_op_set_dtype(1 if bf16 else 0)
tdt = torch.bfloat16 if bf16 else torch.float32
s = torch.cuda.current_stream().cuda_stream
stamps = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
for m, n, k in shapes:
    a = torch.randn(m, k, device=dev).to(tdt)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).to(tdt)
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev, dtype=tdt)

    def run():
        st = lib.casync_op_pw_gemm(a.data_ptr(), k, w.data_ptr(), bias.data_ptr(), c.data_ptr(), n, m, n, k, 1, 0, 0, 0, 0, 0,
                                   0, 0, s)
        assert st == 0, lib.casync_last_error()
    t_end = torch.cuda.Event(enable_timing=True)
    t0 = torch.cuda.Event(enable_timing=True)
    for _ in range(int(0.3 / max(2.0 * m * n * k / 100e12, 1e-5))):   # ~0.3 s of this GEMM: clock at its loaded level
        run()
    t0.record()
    for _ in range(50):
        run()
    t_end.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t_end) / 50
    stamps.zero_()
    lib.casync_debug_gemm_stamps(stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.casync_debug_gemm_stamps(0)
    st = stamps.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    t_first = st[:, 0].min()
    us = lambda v: (v - t_first) / 100.0
    entry, exit_ = us(st[:, 0]), us(st[:, 5])
    kloop = (st[:, 1] - st[:, 0]) / 100.0
    epi = (st[:, 2] - st[:, 1]) / 100.0
    kloop2 = (st[:, 3] - st[:, 2]) / 100.0
    epi2 = (st[:, 4] - st[:, 3]) / 100.0
    clk = np.median(st[:, 7] / np.maximum(st[:, 5] - st[:, 0], 1) * 0.1)      # GHz: shader cycles per 10 ns tick
    print(f"clock {clk:.3f} GHz | ", end="")
    print(f"M={m} N={n} K={k}: events {ms * 1e3:7.1f} us/launch ({2.0 * m * n * k / ms / 1e9:6.1f} TF) | stamps: {len(st)} WGs, "
          f"kernel {exit_.max():6.1f} us, entry spread {entry.max():5.1f} us (median {np.median(entry):4.1f}), "
          f"tile0 k-loop median {np.median(kloop):6.1f} us (max {kloop.max():6.1f}), epilogue median {np.median(epi):5.1f} us "
          f"(max {epi.max():5.1f}), tile1 k-loop median {np.median(kloop2[st[:, 3] > 0]) if (st[:, 3] > 0).any() else 0:6.1f} us, epilogue {np.median(epi2[st[:, 4] > 0]) if (st[:, 4] > 0).any() else 0:5.1f} us, tiles/WG {st[:, 6].min()}-{st[:, 6].max()}, exit median {np.median(exit_):6.1f} / last {exit_.max():6.1f} us")
