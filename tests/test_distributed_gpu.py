"""-m gpu: the RCCL code path on the one GPU the box has -- init_process_group("nccl", world_size=1),
the packed-weight broadcast, adopt_packed without a copy, one forward -- and bench.py's own launcher."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

from calipsync_amd import _lib, recipe
from calipsync_amd.sharding import broadcast_packed_weights, shard_range
from calipsync_amd.unet import Model

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_world1_broadcast_adopt_forward(recipe_sd):
    """backend "nccl" IS RCCL on ROCm: the same calls bench.py makes per rank at N > 1, in a world of one."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        src = Model(6, "hubert").to(dev)
        src.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
        packed = broadcast_packed_weights(src, dev)
        # a real collective on the buffer (world 1 short-circuits inside broadcast_packed_weights):
        dist.broadcast(packed, src=0)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        assert float(t.sum()) == 4.0
        net = Model(6, "hubert").to(dev)          # default-init parameters: everything comes from `packed`
        net.adopt_packed(packed)
        assert net._packed.data_ptr() == packed.data_ptr()          # adopted, not copied
        start, count = shard_range(5, 0, 1)
        x, a = recipe.make_inputs_range(start, count)
        out = net(torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev))
        ref = src(torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev))
        assert torch.equal(out, ref)
    finally:
        dist.destroy_process_group()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two child ranks (sharing the one GPU: gloo rehearsal),
    one JSON line from rank 0, exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "16", "--strong-frames", "64", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["world_size"] == 2 and r["config"]["global_batch"] == 32
    assert r["value"] > 0 and r["config"]["strong_scaling"]["global_batch"] == 64
    # the line proves what ran: ranks, backend, distinct devices, the one broadcast
    c = r["config"]["rccl"]
    assert c["world"] == 2 and c["backend"] == "gloo" and c["unique_devices"] == 1 and c["hosts"] == 1   # the 1-GPU rehearsal
    assert c["broadcast_bytes"] == 4 * _lib.load().casync_packed_total() and c["broadcast_ms"] > 0 and c["collectives_per_step"] == 0


def test_bench_rehearses_a_many_rank_launch():
    """The launcher, the port choice, `shard_range`, the strong leg and the `config.rccl` block at the largest world size the
    one-GPU box allows: FIVE ranks sharing the GPU over gloo (its process guard kills a run with more than six GPU processes,
    and this pytest process is one; the driver's N = 8 run is the same code with RANK 0..7 -- `tests/test_sharding.py` covers
    the 4096-over-8 split and `tests/test_bench_launcher.py` the N = 8 watchdog on the CPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "5", "--steps", "2", "--warmup", "1",
                          "--batch", "8", "--strong-frames", "320", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 5 and r["config"]["world_size"] == 5 and r["config"]["global_batch"] == 40 and r["scaling"] == "weak"
    c = r["config"]["rccl"]
    assert c["world"] == 5 and c["backend"] == "gloo" and c["unique_devices"] == 1 and c["collectives_per_step"] == 0
    s = r["config"]["strong_scaling"]
    assert s["global_batch"] == 320 and s["frames_per_gpu"] == 64 and s["scaling"] == "strong" and s["n_gpus"] == 5 and s["value"] > 0


_SHARD_WORKER = r"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from calipsync_amd import _lib, recipe
from calipsync_amd.sharding import broadcast_packed_weights, shard_range
from calipsync_amd.unet import Model
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world)      # two ranks share the one GPU: RCCL refuses that
src = None
if rank == 0:
    src = Model(6, "hubert").to(dev)
    src.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe.make_state_dict().items()})
packed = broadcast_packed_weights(src, dev)
net = Model(6, "hubert").to(dev)
net.adopt_packed(packed)
net.set_option("gemm_streamk", 0)                                  # batch-invariant bits
start, count = shard_range(6, rank, world)
x, a = recipe.make_inputs_range(start, count)
out = net(torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev))
torch.save({"start": start, "count": count, "out": out.cpu()}, sys.argv[2] + f".{rank}")
dist.barrier()
dist.destroy_process_group()
"""


def test_world2_shards_equal_the_single_process_forward(recipe_sd, tmp_path):
    """Two ranks (sharing the GPU, weights broadcast over gloo) each run their contiguous shard of 6 frames
    (shard_range(6, r, 2)); the concatenated shard outputs equal the single-process forward bit for bit."""
    worker = tmp_path / "worker.py"
    worker.write_text(_SHARD_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(worker), REPO, str(tmp_path / "out")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    parts = sorted((torch.load(str(tmp_path / f"out.{r}")) for r in range(2)), key=lambda d: d["start"])
    assert [(d["start"], d["count"]) for d in parts] == [(0, 3), (3, 3)]
    dev = torch.device("cuda", 0)
    net = Model(6, "hubert").to(dev)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in recipe_sd.items()})
    net.set_option("gemm_streamk", 0)
    x, a = recipe.make_inputs_range(0, 6)
    ref = net(torch.from_numpy(x).to(dev), torch.from_numpy(a).to(dev)).cpu()
    assert torch.equal(torch.cat([d["out"] for d in parts]), ref)
