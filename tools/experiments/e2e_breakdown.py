#!/usr/bin/env python3
"""Where the time of process_batch_device goes (B = 64 synthetic 1080p frames): host slicing, upload, device work,
download, host paste with and without the reference's per-frame img.copy()."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from calipsync_amd import frame_bench, frame_loop, recipe  # noqa: E402
from calipsync_amd.unet import Model  # noqa: E402

dev = torch.device("cuda:0")
net = Model(6, "hubert").to(dev)
net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
B = 64
imgs, lms = frame_bench.synthetic_frames(B)
feats = torch.randn(B * 4 + 16, 2, 1024, device=dev)
idx = list(range(B))
for _ in range(3):
    frame_loop.process_batch_device(net, imgs, lms, [None] * B, features=feats, frame_indices=idx)
torch.cuda.synchronize()


def t(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


ms, _ = t(lambda: frame_loop.process_batch_device(net, imgs, lms, [None] * B, features=feats, frame_indices=idx))
print(f"process_batch_device           {ms:7.2f} ms / batch of {B}")
boxes = [frame_loop.crop_box(l, 1080, 1920) for l in lms]
ms, regs = t(lambda: np.concatenate([np.ascontiguousarray(im[b[0]:b[1], b[2]:b[3]]).reshape(-1) for im, b in zip(imgs, boxes)]))
print(f"  host crop slices + concat    {ms:7.2f} ms   ({regs.nbytes / 1e6:.1f} MB)")
ms, rd = t(lambda: torch.from_numpy(regs).to(dev))
print(f"  upload (pageable)            {ms:7.2f} ms")
pin = torch.from_numpy(regs).pin_memory()
ms, _ = t(lambda: pin.to(dev, non_blocking=True))
print(f"  upload (pinned)              {ms:7.2f} ms")
x = torch.rand(B, 6, 160, 160, device=dev)
ms, _ = t(lambda: net.forward_windows(x, feats, idx))
print(f"  forward_windows              {ms:7.2f} ms")
ms, host = t(lambda: rd.cpu().numpy())
print(f"  download (pageable)          {ms:7.2f} ms")
hp = torch.empty(rd.shape, dtype=torch.uint8).pin_memory()
ms, _ = t(lambda: (hp.copy_(rd, non_blocking=True), torch.cuda.synchronize()))
print(f"  download (pinned)            {ms:7.2f} ms")
ms, _ = t(lambda: [im.copy() for im in imgs])
print(f"  64 x img.copy() (1080p)      {ms:7.2f} ms")


def paste(copy):
    off = 0
    out = []
    for im, b in zip(imgs, boxes):
        o = im.copy() if copy else im
        h, w = b[1] - b[0], b[3] - b[2]
        o[b[0]:b[1], b[2]:b[3]] = host[off:off + h * w * 3].reshape(h, w, 3)
        off += h * w * 3
        out.append(o)
    return out


ms, _ = t(lambda: paste(True))
print(f"  host paste with copies       {ms:7.2f} ms")
ms, _ = t(lambda: paste(False))
print(f"  host paste in place          {ms:7.2f} ms")
