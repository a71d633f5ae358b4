#!/usr/bin/env python3
"""Shader clock INSIDE the timed two-lane forward: every GEMM workgroup stamps s_memtime / s_memrealtime at
entry and exit (casync_debug_gemm_stamps); after ~2 s of back-to-back forwards the buffer holds the last GEMM
launch's stamps.  python tools/experiments/clock_in_forward.py [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import _lib, recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
net = Model(6, "hubert").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
x, a = recipe.make_inputs(B)
x, a = torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev)
stamps = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
lib = _lib.load()
for _ in range(10):
    net(x, a)
torch.cuda.synchronize()
lib.casync_debug_gemm_stamps(stamps.data_ptr())
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 2.0:
    for _ in range(10):
        net(x, a)
    n += 10
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.casync_debug_gemm_stamps(0)
st = stamps.cpu().numpy().reshape(-1, 8)
st = st[(st[:, 0] != 0) & (st[:, 5] > st[:, 0])]
clk = st[:, 7] / (st[:, 5] - st[:, 0]) * 0.1
print(f"B={B}: {n * B / dt:.0f} frames/s; shader clock in the GEMM workgroups (n={len(st)}): median {np.median(clk):.3f} GHz, "
      f"p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}")
