"""Host-side cost of enqueueing one forward (no sync inside the loop) vs the GPU time per forward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import recipe
from calipsync_amd.unet import Model
net = Model(6, "hubert").to("cuda:0")
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
for B in (64,):
    x, a = recipe.make_inputs(B)
    x, a = torch.from_numpy(x).cuda(), torch.from_numpy(a).cuda()
    for lanes in ("1", "2"):
        net.set_option("lanes", int(lanes))
        for _ in range(5): net(x, a)
        torch.cuda.synchronize()
        # enqueue time with an idle GPU queue: sync before each call
        enq = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); net(x, a); enq.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): net(x, a)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 20
        # single forward latency (enqueue + execute, queue empty)
        lat = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); net(x, a); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        print(f"B={B} lanes={lanes}: host enqueue {1e3*sorted(enq)[10]:.3f} ms, back-to-back {1e3*tot:.3f} ms/forward, isolated latency {1e3*sorted(lat)[10]:.3f} ms")
