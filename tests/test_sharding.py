"""Frame sharding + the one-time weight broadcast, world_size 2 and 8 over gloo on the CPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from calipsync_amd.sharding import shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            pos = 0
            for s, n in spans:
                assert s == pos and n >= 0
                pos += n
            assert pos == total
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, q, n_frames=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from calipsync_amd import recipe
        from calipsync_amd.sharding import broadcast_packed_weights, shard_frames
        from calipsync_amd.unet import Model
        net = None
        if rank == 0:
            net = Model(6, "hubert")
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
        buf = broadcast_packed_weights(net, torch.device("cpu"))
        # every rank must now hold rank 0's folded weights bit for bit
        digest = torch.tensor([float(buf.double().sum()), float(buf.double().abs().sum()), float(buf.numel())],
                              dtype=torch.float64)
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        x = torch.arange(n_frames * 6, dtype=torch.float32).reshape(n_frames, 6)
        xs, _ = shard_frames(x, x, rank, world)
        q.put((rank, [g.tolist() for g in gathered], xs[:, 0].tolist(), bool((buf != 0).any())))
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_frame_shards_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, g0, f0, nz0), (r1, g1, f1, nz1) = res
    assert g0 == g1 and g0[0] == g0[1]          # identical buffers on both ranks
    assert nz0 and nz1
    assert f0 == [0.0, 6.0, 12.0] and f1 == [18.0, 24.0]   # 5 frames -> 3 + 2, contiguous


def test_weight_broadcast_and_frame_shards_world8():
    """The world size of BASELINE configs[3] (one process per GPU of an 8-GPU node), rehearsed over gloo on the CPU: ONE
    broadcast leaves rank 0's folded weights on all eight ranks, and the 4096-frame job is cut into eight contiguous
    512-frame shards (here: 11 frames -> 2 2 2 1 1 1 1 1, contiguous, in rank order)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q, 11)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=480) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    digests = [g for _, g, _, _ in res]
    assert all(d == digests[0] for d in digests) and all(row == digests[0][0] for row in digests[0])   # eight identical buffers
    assert all(nz for _, _, _, nz in res)
    frames = [f for _, _, f, _ in res]
    assert [len(f) for f in frames] == [2, 2, 2, 1, 1, 1, 1, 1]
    assert sum(frames, []) == [6.0 * i for i in range(11)]
    assert [shard_range(4096, r, 8) for r in range(8)] == [(512 * r, 512) for r in range(8)]


def test_forward_chunked_walks_a_shard_in_order():
    """configs[3] on fewer than 8 GPUs: a rank walks its shard in chunks of <= 512 frames through one model
    (sharding.forward_chunked).  Host logic only: a stand-in model records what it was handed."""
    from calipsync_amd.sharding import forward_chunked
    calls = []

    def fake(x, a):
        assert x.shape[0] == a.shape[0]
        calls.append(x.shape[0])
        return x[:, :1] + a[:, :1]

    x = torch.arange(11, dtype=torch.float32).reshape(11, 1).repeat(1, 3)
    a = 100 * x
    out = forward_chunked(fake, x, a, 4)
    assert calls == [4, 4, 3]
    assert torch.equal(out, x[:, :1] + a[:, :1])
    calls.clear()
    assert forward_chunked(fake, x, a, 4, keep=False) is None and calls == [4, 4, 3]
    calls.clear()
    assert torch.equal(forward_chunked(fake, x, a, 512), x[:, :1] + a[:, :1]) and calls == [11]
    assert forward_chunked(fake, x[:0], a[:0], 4).shape[0] == 0                 # an empty shard is legal
    with pytest.raises(ValueError):
        forward_chunked(fake, x, a, 0)
    with pytest.raises(ValueError):
        forward_chunked(fake, x, a[:3], 4)
    # shards of the 4096-frame job: every rank's walk covers its range exactly once
    from calipsync_amd.sharding import shard_range
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            s, n = shard_range(4096, r, world)
            seen += [(s + c, min(512, n - c)) for c in range(0, n, 512)]
        assert sum(n for _, n in seen) == 4096 and [s for s, _ in seen] == sorted(s for s, _ in seen)
        assert all(seen[i][0] + seen[i][1] == seen[i + 1][0] for i in range(len(seen) - 1))
