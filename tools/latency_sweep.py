#!/usr/bin/env python3
"""Forward latency / throughput by batch size and lane count: python tools/latency_sweep.py [--dtype f32|bf16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--batches", default="1,2,4,8,16,32,64,128")
ap.add_argument("--lanes", default="1,2")
args = ap.parse_args()
dev = torch.device("cuda", 0)
net = Model(6, "hubert", precision="bf16" if args.dtype == "bf16" else "fp32").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
for b in [int(v) for v in args.batches.split(",")]:
    x_np, a_np = recipe.make_inputs(b)
    x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
    for lanes in args.lanes.split(","):
        net.set_option("lanes", int(lanes))
        for _ in range(5):
            net(x, a)
        torch.cuda.synchronize()
        n = max(10, min(200, 2000 // b))
        t0 = time.perf_counter()
        for _ in range(n):
            net(x, a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={b:4d} lanes={lanes}  {dt * 1e3:8.3f} ms/forward  {b / dt:9.1f} frames/s", flush=True)
