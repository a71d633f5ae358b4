#!/usr/bin/env python3
"""Soak: many forwards with changing batch sizes, both precisions and both input forms; every repeat of a
configuration must reproduce its first output bit for bit (races in the ring / stream-K / lane hand-offs would show
as flips).  python tools/soak.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
sd = {k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()}
nets = {}
for prec in ("fp32", "bf16"):
    nets[prec] = Model(6, "hubert", precision=prec).to(dev)
    nets[prec].load_state_dict(sd)
xs, aud = recipe.make_inputs(96)
xs, aud = torch.from_numpy(xs).to(dev), torch.from_numpy(aud).to(dev)
xs, aud = xs.repeat(3, 1, 1, 1)[:264].contiguous(), aud.repeat(3, 1, 1, 1)[:264].contiguous()   # 264 frames: the bf16 engine's three-lane plan (round 6)
feats = torch.randn(600, 2, 1024, device=dev)
first, n, flips = {}, 0, 0
rng = np.random.default_rng(0)
t_end = time.time() + budget
while time.time() < t_end:
    prec = "fp32" if rng.random() < 0.7 else "bf16"
    b = int(rng.choice([1, 2, 5, 8, 11, 12, 31, 32, 33, 63, 64, 65, 96] + ([264] if rng.random() < 0.15 else [])))
    form = int(rng.integers(0, 2))
    net = nets[prec]
    if form == 0:
        out = net(xs[:b], aud[:b])
    else:
        out = net.forward_windows(xs[:b], feats, list(range(5, 5 + b)))
    key = (prec, b, form)
    if key not in first:
        first[key] = out.clone()
    elif not torch.equal(out, first[key]):
        flips += 1
        print("MISMATCH", key, float((out - first[key]).abs().max()))
    n += 1
torch.cuda.synchronize()
print(f"soak: {n} forwards over {len(first)} configurations in {budget:.0f} s, {flips} mismatches")
sys.exit(1 if flips else 0)
