#!/usr/bin/env python3
"""N forwards at a small batch (default B=8: FrameSynthesizer's default batch and the reference's own benchmark shape,
image_infer_v1/models/unet.py:342-347), for `rocprofv3 --kernel-trace -- python3 tools/experiments/small_forward.py 8 30`;
prints the wall time per forward.  tools/gap_table.py turns the trace into a per-launch gap table."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
net = Model(6, "hubert").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
net.eval()
x_np, a_np = recipe.make_inputs(B)
x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
for _ in range(10):
    net(x, a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    net(x, a)
torch.cuda.synchronize()
print(f"B={B}: {1e3 * (time.perf_counter() - t0) / N:.4f} ms per forward over {N}")
