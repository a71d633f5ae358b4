"""Per-launch table of one lane of the default two-lane schedule (isolated launches, B=64 -> 32 frames per lane)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe
from calipsync_amd.unet import Model
net = Model(6, "hubert").to("cuda:0")
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
x, a = recipe.make_inputs(64)
x, a = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
net.set_option("lanes", 2)
net(x, a); rows = net.profile(x, a); rows = net.profile(x, a)
half = len(rows) // 2
tot = 0
for r in rows[:half]:
    tot += r["ms"]
    if "gemm" in r["kernel"]:
        print(f"{r['name']:46s} {r['kernel']:46s} {r['ms']*1e3:7.1f} us {r['flops']/max(r['ms'],1e-9)/1e9:6.1f} TF")
print("lane total ms", tot, "gemm ms", sum(r["ms"] for r in rows[:half] if "gemm" in r["kernel"]))
