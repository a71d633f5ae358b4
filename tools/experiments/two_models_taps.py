#!/usr/bin/env python3
"""Which stage of the fp32 forward goes wrong while a bf16 model forwards at the same time?  (round 6)"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()}
B = int(os.environ.get("B32", "96"))
x, a = recipe.make_inputs_range(0, 96)
xt, at = torch.from_numpy(x).to(dev)[:B].contiguous(), torch.from_numpy(a).to(dev)[:B].contiguous()
xb, ab = torch.from_numpy(x).to(dev).repeat(3, 1, 1, 1)[:264].contiguous(), torch.from_numpy(a).to(dev).repeat(3, 1, 1, 1)[:264].contiguous()
TAPS = ["x1", "x2", "x3", "x4", "x5", "audio_conv2", "audio_conv3", "audio_conv4", "audio_conv5", "a", "tx", "att0", "att1", "att2", "att3", "kx",
        "fuse", "u1", "u2", "u3", "u4"]
n32 = Model(6, "hubert").to(dev)
n32.load_state_dict(sd)
for kv in os.environ.get("OPTS32", "").split():
    n32.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n16 = Model(6, "hubert", precision="bf16").to(dev)
n16.load_state_dict(sd)
for kv in os.environ.get("OPTS16", "bf16_plan=0 ir_dw_mfma=0").split():
    n16.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ref = n32(xt, at).clone()
ref_taps = {t: n32.tap(t, B).clone() for t in TAPS}
n16(xb, ab)
torch.cuda.synchronize()
stop = False
report = []


def load():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        while not stop:
            n16(xb, ab)
            s.synchronize()


def work():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(int(os.environ.get("ITERS", "20"))):
            out = n32(xt, at)
            s.synchronize()
            if not torch.equal(out, ref):
                wrong = []
                for t in TAPS:
                    got = n32.tap(t, B)
                    s.synchronize()
                    if not torch.equal(got, ref_taps[t]):
                        d = (got - ref_taps[t]).abs()
                        frames = (d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
                        f0 = frames[0]
                        df = d[f0]                                   # [C, H, W]
                        ys = (df.amax(0).amax(1) > 0).nonzero().flatten().tolist()
                        xs_ = (df.amax(0).amax(0) > 0).nonzero().flatten().tolist()
                        cs = (df.amax(1).amax(1) > 0).nonzero().flatten().tolist()
                        npx = int((df.amax(0) > 0).sum())
                        wrong.append((t, round(float(d.max()), 4), frames[:5], len(frames), f"frame {f0}: rows {ys[0]}..{ys[-1]} ({len(ys)}), cols {xs_[0]}..{xs_[-1]} ({len(xs_)}), "
                                      f"{npx} pixels, channels {cs[0]}..{cs[-1]} ({len(cs)})"))
                report.append((i, wrong))


tl = threading.Thread(target=load)
tw = threading.Thread(target=work)
tl.start()
tw.start()
tw.join()
stop = True
tl.join()
print(f"{len(report)} mismatching forwards")
for i, wrong in report[:6]:
    print(" forward", i, [w[0] for w in wrong])
    for w in wrong[:3]:
        print("    ", w)
