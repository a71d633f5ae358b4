#!/usr/bin/env python3
"""Round 6: do two models forwarding at the same time from two host threads disturb each other's results?
Each thread runs its model on a torch stream of its own; every output is compared with the model's single-threaded result.
    python tools/experiments/two_models.py            (CASYNC_LIB=... for another build of the library)"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda", 0)
sd = {k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()}
x, a = recipe.make_inputs_range(0, 96)
xt, at = torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev)
xb, ab = xt.repeat(3, 1, 1, 1)[:264].contiguous(), at.repeat(3, 1, 1, 1)[:264].contiguous()


def make(prec, **opts):
    m = Model(6, "hubert", precision=prec).to(dev)
    m.load_state_dict(sd)
    for k, v in opts.items():
        m.set_option(k, v)
    return m


def run(pairs, iters=int(os.environ.get('ITERS', '30'))):
    refs = [m(xi, ai).clone() for m, xi, ai in pairs]
    torch.cuda.synchronize()
    bad = [[] for _ in pairs]

    def work(k):
        m, xi, ai = pairs[k]
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(iters):
                out = m(xi, ai)
                s.synchronize()
                if not torch.equal(out, refs[k]):
                    d = (out - refs[k]).abs()
                    frames = (d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
                    rows = (d[frames[0]].amax(0).amax(1) > 0).nonzero().flatten().tolist()
                    bad[k].append((i, round(float(d.max()), 4), "frames", frames[:6], len(frames), "rows of the first", rows[:3], len(rows)))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(pairs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return bad


cases = {
    "fp32 (96 frames) alone, on a thread":               lambda: [(make("fp32"), xt, at)],
    "fp32 + bf16 (264 frames, three-lane plan)":        lambda: [(make("fp32"), xt, at), (make("bf16"), xb, ab)],
    "fp32 + fp32 (two models, same inputs)":            lambda: [(make("fp32"), xt, at), (make("fp32"), xt, at)],
    "fp32 overlap=0 + bf16":                            lambda: [(make("fp32", overlap=0), xt, at), (make("bf16"), xb, ab)],
    "fp32 lanes=1 + bf16":                              lambda: [(make("fp32", lanes=1), xt, at), (make("bf16"), xb, ab)],
    "fp32 lanes=1 overlap=0 + bf16":                    lambda: [(make("fp32", lanes=1, overlap=0), xt, at), (make("bf16"), xb, ab)],
    "fp32 (8 frames, single lane) + bf16":              lambda: [(make("fp32"), xt[:8].contiguous(), at[:8].contiguous()), (make("bf16"), xb, ab)],
}
class Load:
    """Not a model: an unrelated GPU load (torch matmuls on a stream of its own)."""
    def __init__(self):
        self.a = torch.randn(4096, 4096, device=dev)

    def __call__(self, x, a):
        for _ in range(6):
            self.a = (self.a @ self.a).clamp_(-1, 1)
        return x[:1, :3].clone() * 0


class Load16(Load):
    def __init__(self):
        self.a = torch.randn(4096, 4096, device=dev).bfloat16()


class Copy:
    """HBM pressure: big device-to-device copies."""
    def __init__(self, dtype=torch.float32):
        self.a = torch.randn(256 << 20, device=dev).to(dtype)
        self.b = torch.empty_like(self.a)

    def __call__(self, x, a):
        for _ in range(4):
            self.b.copy_(self.a)
            self.a.copy_(self.b)
        return x[:1, :3].clone() * 0


only = os.environ.get("ONLY")
if only:
    cases.clear()
cases["fp32 + fp32 (264 frames, default options)"] = lambda: [(make("fp32"), xt, at), (make("fp32"), xb, ab)]
cases["fp32 + fp32 (264 frames, lanes=1 overlap=0)"] = lambda: [(make("fp32"), xt, at), (make("fp32", lanes=1, overlap=0), xb, ab)]
cases["fp32 + fp32 (264 frames, lanes=3 overlap=0)"] = lambda: [(make("fp32"), xt, at), (make("fp32", lanes=3, overlap=0), xb, ab)]
cases["fp32 + bf16 (96 frames, round-5 kernels)"] = lambda: [(make("fp32"), xt, at), (make("bf16", bf16_plan=0, ir_dw_mfma=0), xt, at)]
cases["fp32 + a bf16 matmul load"] = lambda: [(make("fp32"), xt, at), (Load16(), xt, at)]
cases["fp32 + 1 GiB fp32 copies"] = lambda: [(make("fp32"), xt, at), (Copy(), xt, at)]
cases["fp32 + 0.5 GiB bf16 copies"] = lambda: [(make("fp32"), xt, at), (Copy(torch.bfloat16), xt, at)]
cases["fp32 lanes=1 + 1 GiB fp32 copies"] = lambda: [(make("fp32", lanes=1), xt, at), (Copy(), xt, at)]
cases["fp32 + bf16 lanes=1 overlap=0 (round-5 kernels)"] = lambda: [(make("fp32"), xt, at), (make("bf16", bf16_plan=0, ir_dw_mfma=0, lanes=1, overlap=0), xb, ab)]
cases["fp32 + an unrelated matmul load"] = lambda: [(make("fp32"), xt, at), (Load(), xt, at)]
cases["fp32 (8 frames) + an unrelated matmul load"] = lambda: [(make("fp32"), xt[:8].contiguous(), at[:8].contiguous()), (Load(), xt, at)]
cases["bf16 + bf16"] = lambda: [(make("bf16"), xb, ab), (make("bf16"), xb, ab)]
cases["fp32 + bf16 with bf16_plan=0 ir_dw_mfma=0"] = lambda: [(make("fp32"), xt, at), (make("bf16", bf16_plan=0, ir_dw_mfma=0), xb, ab)]
for name, mk in cases.items():
    bad = run(mk())
    print(f"{name:52s} mismatches per model: {[len(b) for b in bad]}  {[b[:2] for b in bad if b]}", flush=True)
