#!/usr/bin/env python3
"""Round 6: why `secondary.bf16_b512` (the bf16 B=512 leg inside the default fp32 bench run) reads lower than
`bench.py --dtype bf16` on its own.  Times the bf16 forward (5 warm-up + 10 timed) in one process: fresh, right after fp32
B=512 work, after a pause, and with the fp32 model and its arena released."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda", 0)


def timed(net, x, a, warm=5, steps=10):
    for _ in range(warm):
        net(x, a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net(x, a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


sd = {k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()}
x_np, a_np = recipe.make_inputs_range(0, 512)
x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
n16 = Model(6, "hubert", precision="bf16").to(dev)
n16.load_state_dict(sd)
print(f"bf16 fresh:                 {512 / timed(n16, x, a):9.1f} frames/s")
n32 = Model(6, "hubert").to(dev)
n32.load_state_dict(sd)
for _ in range(3):
    f = 512 / timed(n32, x, a, 2, 8)
print(f"fp32 B=512:                 {f:9.1f} frames/s")
print(f"bf16 right after fp32:      {512 / timed(n16, x, a):9.1f} frames/s")
time.sleep(5)
print(f"bf16 after a 5 s pause:     {512 / timed(n16, x, a):9.1f} frames/s")
del n32
torch.cuda.empty_cache()
print(f"bf16, fp32 arena released:  {512 / timed(n16, x, a):9.1f} frames/s")
n16b = Model(6, "hubert", precision="bf16").to(dev)
n16b.load_state_dict(sd)
print(f"a NEW bf16 model:           {512 / timed(n16b, x, a):9.1f} frames/s")
