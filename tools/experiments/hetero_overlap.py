#!/usr/bin/env python3
"""free-running, de-phased forwards instead of two lock-stepped lanes inside one forward.

Inside Model.forward(B=64) the two 32-frame lanes run the SAME kernel at the same time, so a stall-heavy fused block is
only ever paired with itself.  Here two engines (same packed weights, own workspaces and streams) each forward 32 frames
back to back with no join between them, the second one started half a forward late, so one stream's trunk GEMMs run beside
the other's fused blocks.  Three alternating rounds against the headline forward on ONE box:

    python tools/experiments/hetero_overlap.py [steps=60]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
base = Model(6, "hubert").to(dev)
base.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
base.eval()
x_np, a_np = recipe.make_inputs(64)
x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
ref = base(x, a)
packed = base._packed


def engine(lanes):
    m = Model(6, "hubert").to(dev)
    m.set_option("lanes", lanes)
    m.adopt_packed(packed)
    return m.eval()


def headline():
    for _ in range(10):
        base(x, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        base(x, a)
    torch.cuda.synchronize()
    return 64 * steps / (time.perf_counter() - t0)


def free_running(n_streams, per, lanes, skew_us):
    """n_streams engines of `per` frames each (n_streams * per frames per round), started skew_us apart."""
    nets = [engine(lanes) for _ in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    xs = [x[(i * per) % 64:(i * per) % 64 + per].contiguous() for i in range(n_streams)]
    as_ = [a[(i * per) % 64:(i * per) % 64 + per].contiguous() for i in range(n_streams)]
    outs = [None] * n_streams
    for i, s in enumerate(streams):                 # warm-up (also checks the result)
        with torch.cuda.stream(s):
            for _ in range(5):
                outs[i] = nets[i](xs[i], as_[i])
    torch.cuda.synchronize()
    for i in range(n_streams):
        assert (outs[i] - ref[(i * per) % 64:(i * per) % 64 + per]).abs().max() < 1e-5
    t0 = time.perf_counter()
    for k in range(steps):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                if k == 0 and i and skew_us:
                    torch.cuda._sleep(int(skew_us * 2400 * i))      # ~cycles: de-phase the streams once, at the start
                outs[i] = nets[i](xs[i], as_[i])
    torch.cuda.synchronize()
    return n_streams * per * steps / (time.perf_counter() - t0)


configs = [("Model.forward(B=64): two lock-stepped lanes (headline)", headline),
           ("2 streams x 32 frames, one lane each, in phase", lambda: free_running(2, 32, 1, 0)),
           ("2 streams x 32 frames, one lane each, second 2.5 ms late", lambda: free_running(2, 32, 1, 2500)),
           ("2 streams x 64 frames, two lanes each, second 2.5 ms late", lambda: free_running(2, 64, 2, 2500)),
           ("3 streams x 32 frames, one lane each, 1.7 ms apart", lambda: free_running(3, 32, 1, 1700))]
for rnd in range(3):
    for name, fn in configs:
        print(f"round {rnd + 1}  {fn():9.1f} frames/s  {name}", flush=True)
