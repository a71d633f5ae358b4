"""Achievable HBM read / write / copy bandwidth with plain torch kernels (1 GiB buffers)."""
import torch, time
n = 1 << 28
a = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
b = torch.empty_like(a)
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
ms = t(lambda: a.fill_(1.0)); print(f"write (fill)  {n*4/ms/1e6:8.1f} GB/s")
ms = t(lambda: a.sum());      print(f"read (sum)    {n*4/ms/1e6:8.1f} GB/s")
ms = t(lambda: b.copy_(a));   print(f"copy (r+w)    {2*n*4/ms/1e6:8.1f} GB/s total")
ms = t(lambda: torch.add(a, 1.0, out=b)); print(f"add (r+w)     {2*n*4/ms/1e6:8.1f} GB/s total")
for mb in (32, 128, 512):
    m = mb << 18
    x = a[:m]; y = b[:m]
    ms = t(lambda: y.copy_(x), 50); print(f"copy {mb:4d} MB  {2*m*4/ms/1e6:8.1f} GB/s total")
    ms = t(lambda: x.fill_(2.0), 50); print(f"fill {mb:4d} MB  {m*4/ms/1e6:8.1f} GB/s")
