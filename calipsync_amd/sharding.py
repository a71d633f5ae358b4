"""Multi-GPU plumbing for the frame loop: one process per GPU, frames sharded, weights
broadcast once.

The reference is single-device (``cuda:0`` everywhere: inference.py:18,
frame_synthesizer/infer_api.py:14) and has no distributed code.  Frames are independent in
eval mode, so the only exchange this path needs is ONE broadcast of the
packed, BN-folded weight buffer (~79 MB fp32) from rank 0 at start-up -- over RCCL/xGMI on
the GPU box (backend "nccl"), over gloo in the CPU tests.  No steady-state collective.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [start, start+count) of `total` frames owned by `rank`.

    The first ``total % world`` ranks take one extra frame; empty shards are legal."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def shard_frames(x: torch.Tensor, audio: torch.Tensor, rank: int, world: int):
    """This rank's slice of a global batch (views, no copy)."""
    s, n = shard_range(x.shape[0], rank, world)
    return x[s:s + n], audio[s:s + n]


def forward_chunked(net, x: torch.Tensor, audio: torch.Tensor, chunk: int = 512, keep: bool = True):
    """Walk a rank's shard through ONE model in chunks of at most `chunk` frames (BASELINE configs[3]: 4096 frames over
    8 GPUs = one 512-frame chunk per rank; fewer GPUs walk more chunks).  Every chunk reuses the model's arena (sized by
    the first, largest chunk), launches are enqueued back to back on the current stream with no host sync in between.
    Returns the outputs concatenated in frame order (``keep=False``: nothing -- a throughput walk that leaves no
    4096-frame tensor behind).  Frames are independent, so this equals one forward over the whole shard
    to fp32 rounding."""
    if chunk <= 0:
        raise ValueError("chunk must be positive")
    if x.shape[0] != audio.shape[0]:
        raise ValueError("x and audio must have the same number of frames")
    outs = []
    for s in range(0, x.shape[0], chunk):
        o = net(x[s:s + chunk], audio[s:s + chunk])
        if keep:
            outs.append(o)
    if not keep:
        return None
    if not outs:
        return net(x, audio)          # empty shard: the model's own empty result
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def packed_total() -> int:
    from . import _lib
    return int(_lib.load().casync_packed_total())


def broadcast_packed_weights(model_on_src, device: torch.device, src: int = 0,
                             group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Rank `src` folds + packs its model's weights; everyone receives the flat buffer.

    ``model_on_src`` is the calipsync_amd.unet.Model holding the checkpoint on rank `src`
    (ignored -- may be None -- elsewhere).  Returns a contiguous fp32 tensor on `device`
    that ``Model.adopt_packed`` can take without another copy."""
    n = packed_total()
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else src
    if rank == src:
        if model_on_src is None:
            raise ValueError("the source rank must pass its model")
        host = model_on_src.packed_weights_host()
        assert host.shape == (n,) and host.dtype == np.float32
        buf = torch.from_numpy(host).to(device)
    else:
        buf = torch.empty(n, dtype=torch.float32, device=device)
    if distributed:
        if buf.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal on a shared GPU: stage through host memory (gloo + device tensors is slow / partial)
            host = buf.cpu()
            dist.broadcast(host, src=src, group=group)
            buf.copy_(host)
        else:
            dist.broadcast(buf, src=src, group=group)
    return buf
