import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from calipsync_amd import recipe
from calipsync_amd.unet import Model
dev = torch.device("cuda", 0)
net = Model(6, "hubert").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
net.set_option("lanes", 1)
for b in (1, 8):
    x_np, a_np = recipe.make_inputs(b)
    x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
    for _ in range(3): net(x, a)
    rows = net.profile(x, a); rows = net.profile(x, a)
    print(f"B={b}: {len(rows)} launches, sum of kernel ms = {sum(r['ms'] for r in rows):.3f}")
    top = sorted(rows, key=lambda r: -r["ms"])[:8]
    for r in top: print(f"   {r['name']:50s} {r['kernel']:48s} {r['ms']*1e3:7.1f} us")
    torch.cuda.synchronize()
    for ov in ("1", "0"):
        net.set_option("overlap", int(ov))
        for _ in range(3): net(x, a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): net(x, a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"   overlap={ov}: host enqueue {1e3*(t1-t0)/100:.3f} ms/forward, total {1e3*(t2-t0)/100:.3f} ms/forward")
    net.set_option("overlap", 1)
