#!/usr/bin/env python3
"""Headline benchmark: 160x160 lip-sync frames/sec through the MI355X engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is one pass of the hot path (``Model.forward``) over one batch of synthetic
frames per GPU.  Workload = BASELINE.json configs[1]: batch 64 per GPU, 160x160 fp32,
synthetic crops U(0,1) + random HuBERT windows N(0,1) (the reference's own self-benchmark
shapes, image_infer_v1/models/unet.py:342-347), golden-recipe weights.  Inputs are
resident in HBM before the timed region; outputs stay on the device.

N > 1: one process per GPU (torch.distributed, backend "nccl" == RCCL).  Either the driver starts
the ranks (``python -m torch.distributed.run ... bench.py --gpus N``: RANK/WORLD_SIZE in the
environment) or ``python bench.py --gpus N`` starts them itself, as fresh child processes, BEFORE the
parent touches the GPU.  Frames are independent, so the batch is sharded with no data-path
collective ("weak" scaling: 64 frames per GPU); the only collective is ONE broadcast of the packed,
BN-folded weight buffer from rank 0 at start-up.  At N > 1 the line also carries the
BASELINE configs[3] split (4096 frames over the ranks) beside the weak point.

Prints one JSON line (rank 0) with the whole-job frames/sec, the roofline of the dominant
kernel (live HIP-event timing) and, at N=1, the CPU oracle timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: f32-input MFMA == f32 vector peak (155.4 measured here,
                            # profiles/r2_mfma_clock.txt)
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
PROFILE_TAG = "r6"          # profiles/<tag>_pmc_traffic.json, <tag>_mfma_busy.json, <tag>_*_kernel_stats_replay.csv feed
                            # roofline.traffic / mfma_busy / avg_kernel_us_rocprof -- only when their source_hash matches


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per GPU per step (default: 64 for f32 = configs[1], 512 for bf16 = configs[2])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="engine activation storage: f32 = parity path (default), bf16 = BASELINE configs[2]")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling as the headline: a FIXED number of frames per step split over the ranks "
                         "(SURVEY 8d: 4096); default 0 = weak scaling, --batch frames per GPU")
    ap.add_argument("--strong-frames", type=int, default=4096,
                    help="N > 1 only: frames of the configs[3] split timed beside the weak point (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU-oracle leg")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--replay-only", action="store_true",
                    help="skip the timed region; run --steps isolated replays (the mode rocprofv3 / PMC passes "
                         "are collected in, so their per-kernel figures match the roofline block)")
    ap.add_argument("--e2e", action="store_true",
                    help="also time the device frame loop (process_batch: crop+resize -> forward -> paste-back "
                         "blend on synthetic 1080p frames, one D2H per batch) and report it in config.e2e")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (N=1, default f32 run only: bf16 B=512 = configs[2], the device "
                         "frame loop, the reference's own B=8 benchmark shape; ~20 s)")
    return ap.parse_args()


def host_cores() -> int:
    """Threads for the CPU leg: affinity mask, capped by the cgroup CPU quota and by the GPU
    box's per-GPU CPU share (16); override with CASYNC_CPU_THREADS."""
    if os.environ.get("CASYNC_CPU_THREADS"):
        return max(1, int(os.environ["CASYNC_CPU_THREADS"]))
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return min(n, 16)


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh child processes (one rank per GPU)
    with the torch.distributed environment, relay rank 0's JSON line, fail if any rank fails.  Nothing in
    this parent process has touched the GPU (no torch import even), so nothing is re-executed from a
    process that initialised HIP.

    Watchdog: every child is polled; the first rank that exits non-zero ends the job -- the others are terminated
    (fresh children only, never a re-exec), the parent prints that rank's stderr tail and returns its exit code, so a
    rank that dies at import or at RCCL init costs seconds, not the collective timeout rank 0 would otherwise sit in.
    A rendezvous port that turns out to be taken is retried on a new one."""
    import tempfile
    from calipsync_amd import build as _build   # plain Python, no torch / HIP
    try:
        _build.build()        # once, here: N ranks must not all find the library stale and rebuild it side by side
    except Exception as exc:  # no compiler on this box: the ranks refuse a stale library themselves
        print(f"bench.py: library not rebuilt in the launcher ({exc})", file=sys.stderr)
    limit = float(os.environ.get("CASYNC_BENCH_TIMEOUT", "1500"))
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs, outs, errs = [], [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            outs.append(tempfile.TemporaryFile(mode="w+"))
            errs.append(tempfile.TemporaryFile(mode="w+"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=outs[r], stderr=errs[r], text=True))
        t0, failed = time.monotonic(), None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = (r, p.returncode)
                    break
            if failed is None and time.monotonic() - t0 > limit:
                failed = (-1, 124)
            if failed is None:
                time.sleep(0.1)
        if failed is None:
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
        for p in procs:                      # stop whatever is still running (only after a failure / the time limit)
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

        def tail(f, lines=15):
            f.seek(0)
            return "".join(f.readlines()[-lines:])
        if failed is None:
            sys.stdout.write(tail(outs[0], 5))
            sys.stdout.flush()
            sys.stderr.write(tail(errs[0], 40))
            return 0
        r, rc = failed
        err_text = tail(errs[r]) if r >= 0 else ""
        if "address already in use" in err_text.lower() and attempt < 2:
            print(f"bench.py: rendezvous port {port} was taken, retrying on another one", file=sys.stderr)
            continue
        if r < 0:
            print(f"bench.py: ranks still running after {limit:.0f} s (CASYNC_BENCH_TIMEOUT), terminated", file=sys.stderr)
        else:
            print(f"bench.py: rank {r} exited with code {rc}; the other ranks were terminated.  Its stderr tail:\n{err_text}",
                  file=sys.stderr)
        return rc if rc else 1
    return 1


def cpu_baseline(sd_np, seconds: float):
    """The CPU oracle (a port of the reference's forward, torch-CPU fp32) on the host cores, at
    B in {1, 8, 64}: median of the timed forwards per batch size, the B=64
    figure (the GPU line's own batch) is `value`.  Bounded: ~`seconds` of CPU work in all."""
    import torch
    from calipsync_amd import recipe
    from oracle import unet_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = unet_oracle.to_torch(sd_np)
    by_batch, spent = {}, 0.0
    for b, reps in ((1, 10), (8, 5), (64, 3)):
        x, a = recipe.make_inputs(b)
        xt, at = torch.from_numpy(x), torch.from_numpy(a)
        t0 = time.perf_counter()
        unet_oracle.forward(sd, xt, at)                      # warm-up
        warm = time.perf_counter() - t0
        if b == 64 and spent + warm * (1 + reps) > 1.6 * seconds:
            reps = 1
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            unet_oracle.forward(sd, xt, at)
            times.append(time.perf_counter() - t0)
        spent += warm + sum(times)
        med = sorted(times)[len(times) // 2]
        by_batch[str(b)] = {"frames_per_s": round(b / med, 2), "ms_per_forward": round(1e3 * med, 1), "timed_forwards": reps}
    return {"value": by_batch["64"]["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
            "by_batch": by_batch,
            "sample": f"torch-CPU fp32 oracle, same weights/input recipe, median forward at B=1/8/64 "
                      f"({by_batch['1']['timed_forwards']}/{by_batch['8']['timed_forwards']}/{by_batch['64']['timed_forwards']} "
                      f"timed after one warm-up each), {spent:.1f} s of CPU work; value = the B=64 figure"}


def load_profile_json(name, want_hash):
    """profiles/<tag>_<name>.json if it was collected from THIS version of the kernels (its `source_hash` ==
    sha256 of calipsync_amd/csrc, calipsync_amd.build.source_hash()), else (None, reason)."""
    path = os.path.join(REPO, "profiles", f"{PROFILE_TAG}_{name}.json")
    if not os.path.exists(path):
        return None, f"profiles/{PROFILE_TAG}_{name}.json not collected yet"
    d = json.load(open(path))
    if d.get("source_hash") != want_hash:
        return None, (f"profiles/{PROFILE_TAG}_{name}.json was collected from other kernel sources "
                      f"(source_hash {str(d.get('source_hash'))[:12]} != {want_hash[:12]})")
    return d, None


def rocprof_avg_us(tag, kernel, want_hash):
    """AverageNs of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of `bench.py --replay-only`
    (profiles/<tag>_kernel_stats_replay.csv + .meta.json carrying the source hash), or (None, reason)."""
    import csv
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_names import short
    path = os.path.join(REPO, "profiles", f"{tag}_kernel_stats_replay.csv")
    meta = path.replace(".csv", ".meta.json")
    if not (os.path.exists(path) and os.path.exists(meta)):
        return None, f"profiles/{tag}_kernel_stats_replay.csv not collected yet"
    if json.load(open(meta)).get("source_hash") != want_hash:
        return None, f"profiles/{tag}_kernel_stats_replay.csv was collected from other kernel sources"
    for r in csv.DictReader(open(path)):
        if short(r["Name"]) == kernel:
            return round(float(r["AverageNs"]) / 1e3, 2), None
    return None, f"{kernel} not in profiles/{tag}_kernel_stats_replay.csv"


def hbm_moved_per_frame(pmc, launches: dict, frames: int):
    """HBM bytes the engine really moves per frame: sum over the step's kernels of the PMC pass's HBM-side bytes per
    launch (TCC fetch + write, profiles/<tag>_pmc_traffic*.json) x that kernel's launches per step, / frames.  The
    granularity the kernels implement, as opposed to the canonical un-fused conv granularity.  Returns
    (bytes_per_frame, kernels_without_counters)."""
    total, missing = 0.0, []
    for k, n in launches.items():
        if k in pmc["kernels"]:
            total += pmc["kernels"][k]["hbm_bytes_per_launch"] * n
        else:
            missing.append(k)
    return total / frames, missing


def time_forward(net, x, a, warmup, steps, dev):
    import torch
    for _ in range(warmup):
        out = net(x, a)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = net(x, a)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return dt / steps


def strong_leg(net, x, a, total_frames, rank, world, steps, sync, max_over_ranks, dtype="f32"):
    """BASELINE configs[3]: a FIXED `total_frames`-frame job (4096) split over the ranks, each rank walking its contiguous
    shard in chunks of <= 512 frames (the per-GPU size configs[3] names) through ONE Model / one arena
    (calipsync_amd.sharding.forward_chunked).  Same timing contract as the headline: barrier + sync on both sides, max
    over ranks.  The chunk's frames are the resident frames tiled up to the chunk size (timing is data-blind)."""
    from calipsync_amd.sharding import forward_chunked, shard_range
    per_rank = shard_range(total_frames, rank, world)[1]
    chunk = min(512, max(1, per_rank))
    reps = (per_rank + x.shape[0] - 1) // x.shape[0]
    xs, as_ = (x, a) if reps == 1 else (x.repeat(reps, 1, 1, 1), a.repeat(reps, 1, 1, 1))
    xs, as_ = xs[:per_rank].contiguous(), as_[:per_rank].contiguous()
    forward_chunked(net, xs, as_, chunk, keep=False)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        forward_chunked(net, xs, as_, chunk, keep=False)
    sync()
    sdt = max_over_ranks(time.perf_counter() - t0)
    return {"global_batch": total_frames, "frames_per_gpu": per_rank, "chunk": chunk, "chunks_per_step": -(-per_rank // chunk),
            "steps": steps, "value": round(total_frames * steps / sdt, 1), "unit": "frames/s",
            "ms_per_step": round(1e3 * sdt / steps, 3), "scaling": "strong", "n_gpus": world, "dtype": dtype}


def secondary_block(net, packed, x, a, dev, src_hash):
    """Figures next to the headline that the driver's plain `bench.py` run would otherwise never produce
    (N = 1 only; ~20 s): the reference's own benchmark shape (B=8 fp32), BASELINE configs[2] (bf16, B=512, with its
    own roofline) and the device frame loop.  None of them replaces `value`."""
    import torch
    from calipsync_amd import arch, frame_bench
    from calipsync_amd.unet import Model
    sec = {}
    work = arch.work_per_frame()
    # the reference's own self-benchmark shape: image_infer_v1/models/unet.py:342-347 (B=8 fp32)
    s8 = time_forward(net, x[:8], a[:8], 10, 50, dev)
    sec["b8_fp32"] = {"value": round(8 / s8, 1), "unit": "frames/s", "ms_per_step": round(1e3 * s8, 3), "batch": 8,
                      "dtype": "f32", "steps": 50,
                      "mfma_frac": round(8 / s8 * work["flops"] / (MFMA_F32_PEAK_TF * 1e12), 4),
                      "shape": "reference self-benchmark, image_infer_v1/models/unet.py:342-347"}
    # device frame loop, median of three runs per mode
    sec["e2e"] = frame_bench.run(net, dev, batch=x.shape[0])
    # 512 distinct synthetic frames of the same recipe: the per-GPU shard of BASELINE configs[2] / configs[3]
    from calipsync_amd import recipe
    from calipsync_amd.sharding import forward_chunked
    x_np, a_np = recipe.make_inputs_range(0, 512)
    x16, a16 = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)
    # BASELINE configs[3], single-GPU anchor: one 512-frame fp32 forward (the shard every rank of the
    # 8-GPU job runs) with its own whole-net fractions, and the WHOLE 4096-frame job walked through this one GPU in
    # 8 chunks of 512 through one arena -- the N = 1 point `config.strong_scaling` of an N > 1 line is divided by
    s512 = time_forward(net, x16, a16, 2, 8, dev)
    ex512 = sum(r["flops"] for r in net.profile(x16, a16)) / 512
    stage32 = arch.stagewise_bound(MFMA_F32_PEAK_TF * 1e12, HBM_PEAK_GBS * 1e9, 4)
    sec["fp32_b512"] = {"value": round(512 / s512, 1), "unit": "frames/s", "ms_per_step": round(1e3 * s512, 3), "batch": 512,
                        "dtype": "f32", "steps": 8, "lanes": net.get_option("lanes"),
                        "whole_net": {"mfma_frac": round(512 / s512 * work["flops"] / (MFMA_F32_PEAK_TF * 1e12), 4),
                                      "executed_gflop_per_frame": round(ex512 / 1e9, 3),
                                      "mfma_frac_executed": round(512 / s512 * ex512 / (MFMA_F32_PEAK_TF * 1e12), 4),
                                      "frac_of_stagewise_bound": round(512 / s512 / stage32["frames_per_s"], 4)}}
    sec["strong_n1"] = strong_leg(net, x16, a16, 4096, 0, 1, 4, lambda: torch.cuda.synchronize(dev), lambda t: t)
    net16 = Model(6, "hubert", precision="bf16").to(dev)
    net16.adopt_packed(packed)
    s16 = time_forward(net16, x16, a16, 5, 10, dev)     # (five warm-up forwards: the first ones create the third lane's streams and set kernel attributes)
    per = {}
    for row in net16.profile(x16, a16):
        c = per.setdefault(row["kernel"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
        c["ms"] += row["ms"]; c["flops"] += row["flops"]; c["bytes"] += row["bytes"]; c["n"] += 1
    dom_name, dom = max(per.items(), key=lambda kv: kv[1]["ms"])
    gbs, tf = dom["bytes"] / dom["ms"] / 1e6, dom["flops"] / dom["ms"] / 1e9
    stage = arch.stagewise_bound(MFMA_BF16_PEAK_TF * 1e12, HBM_PEAK_GBS * 1e9, 2)
    fps16 = 512 / s16
    canon16 = work["canonical_bytes_f32"] // 2
    ex16 = sum(c["flops"] for c in per.values()) / 512
    sec["bf16_b512"] = {
        "value": round(fps16, 1), "unit": "frames/s", "ms_per_step": round(1e3 * s16, 3), "batch": 512, "steps": 10,
        "dtype": "bf16 (fp32 accumulate; NOT the parity path)",
        "roofline": {"kernel": dom_name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_tflops": round(tf, 1),
                     "avg_launch_ms": round(dom["ms"] / dom["n"], 4),
                     "share_of_replay": round(dom["ms"] / sum(c["ms"] for c in per.values()), 3),
                     "measured": "HIP events, one serialised replay of the timed run's launches"},
        # whole net, at the granularity the kernels IMPLEMENT: the bytes the fused engine
        # really moves (PMC) and the flops it executes -- this is the figure to judge the bf16 engine by
        "whole_net": {"mfma_frac": round(fps16 * work["flops"] / (MFMA_BF16_PEAK_TF * 1e12), 4),
                      "mfma_frac_executed": round(fps16 * ex16 / (MFMA_BF16_PEAK_TF * 1e12), 4),
                      "hbm_bytes_moved_per_frame": None, "hbm_frac_moved": None,
                      # context only: bytes of the UN-FUSED conv-granularity network (SURVEY 8d "canonical"), which a
                      # fused engine does not move
                      "canonical_mb_per_frame": round(canon16 / 1e6, 2),
                      "hbm_frac_canonical": round(fps16 * canon16 / (HBM_PEAK_GBS * 1e9), 4),
                      "frac_of_stagewise_bound": round(fps16 / stage["frames_per_s"], 4),
                      "canonical_note": "hbm_frac_canonical / frac_of_stagewise_bound are quoted on un-fused "
                                        "conv-granularity bytes: context only, not what the kernels move"},
        "frac_of_stagewise_bound": round(fps16 / stage["frames_per_s"], 4)}
    pmc, why = load_profile_json("pmc_traffic_bf16_b512", src_hash)
    wn = sec["bf16_b512"]["whole_net"]
    if pmc:
        moved, missing = hbm_moved_per_frame(pmc, {k: c["n"] for k, c in per.items()}, 512)
        wn["hbm_bytes_moved_per_frame"] = round(moved)
        wn["hbm_frac_moved"] = round(fps16 * moved / (HBM_PEAK_GBS * 1e9), 4)
        wn["hbm_moved_source"] = pmc["source"] + (f"; no counters for {missing}" if missing else "")
    else:
        wn["hbm_moved_source"] = why
    del net16, x16, a16
    torch.cuda.empty_cache()
    return sec


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # before torch / HIP are even imported

    # test hook (tests/test_bench_launcher.py): this rank dies before it imports anything
    if os.environ.get("CASYNC_BENCH_FAIL_RANK") == os.environ.get("RANK", ""):
        sys.exit(3)

    import numpy as np  # noqa: F401
    import torch
    if args.batch <= 0:
        args.batch = 64 if args.dtype == "f32" else 512
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    torch.set_num_threads(max(1, host_cores() // max(1, world)))   # the cgroup's share, split over the ranks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path to benchmark")
    # one rank per GPU.  Rehearsal hook: on a box with fewer GPUs than ranks (the 1-GPU dev box)
    # ranks wrap around the visible devices and the weight broadcast goes over gloo, because RCCL
    # refuses two ranks on one device; the code path is otherwise identical.
    n_dev = torch.cuda.device_count()
    shared = world > n_dev
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared else "nccl"
        import datetime
        init_to = datetime.timedelta(seconds=120)   # a peer that never arrives fails this rank in two minutes
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=init_to)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=init_to)

    from calipsync_amd import arch, recipe
    from calipsync_amd.unet import Model
    from calipsync_amd.sharding import broadcast_packed_weights, shard_range

    # ---- weights: rank 0 folds + packs, ONE RCCL broadcast, every rank adopts the buffer
    sd_np = recipe.make_state_dict() if rank == 0 else None
    net = Model(6, "hubert", precision="bf16" if args.dtype == "bf16" else "fp32").to(dev)
    if rank == 0:
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t_bc = time.perf_counter()
    packed = broadcast_packed_weights(net if rank == 0 else None, dev)
    torch.cuda.synchronize(dev)
    broadcast_ms = 1e3 * (time.perf_counter() - t_bc)   # rank 0: fold + pack + upload + the collective; others: the collective
    net.adopt_packed(packed)

    # ---- N > 1: what actually ran, so the line proves itself: one identity per rank -- the device's UUID
    #      (or PCI bus id) and the host -- all-gathered; over RCCL every rank must own a device of its own
    comm = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        ident = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or dev_index)
        idents = [None] * world
        dist.all_gather_object(idents, (socket.gethostname(), ident))
        comm = {"world": dist.get_world_size(), "backend": dist.get_backend(), "unique_devices": len(set(idents)),
                "hosts": len({h for h, _ in idents}), "broadcast_ms": round(broadcast_ms, 2),
                "broadcast_bytes": int(packed.numel() * packed.element_size()),
                "collectives_per_step": 0}
        if comm["backend"] == "nccl" and comm["unique_devices"] != world:
            if rank == 0:
                print(f"bench.py: {world} RCCL ranks but only {comm['unique_devices']} distinct devices {sorted(set(idents))}: "
                      "refusing to report a scaling point", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(4)

    # ---- inputs: this rank's contiguous shard of the global synthetic batch, resident in HBM
    B = args.batch
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} must be a multiple of the {world} ranks")
        B = args.global_batch // world
    start, count = shard_range(B * world, rank, world)
    x_np, a_np = recipe.make_inputs_range(start, count)
    x, a = torch.from_numpy(x_np).to(dev), torch.from_numpy(a_np).to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(seconds: float) -> float:
        if world == 1:
            return seconds
        t = torch.tensor([seconds], device="cpu" if shared else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.replay_only:
        # the mode rocprofv3 / the PMC passes are collected in: ONLY serialised replays (warm-up included), so the
        # per-kernel statistics contain nothing but the launches the roofline block is quoted on.  `value` is then
        # the serialised rate (no lane overlap) and the line says so.
        for _ in range(max(1, args.warmup)):
            net.profile(x, a)
        sync()
        t0 = time.perf_counter()
        net.profile(x, a)
        sync()
        dt = max_over_ranks((time.perf_counter() - t0) * args.steps)
    else:
        for _ in range(args.warmup):
            out = net(x, a)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = net(x, a)
        sync()
        dt = max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(out).all()

    # ---- N > 1: BASELINE configs[3] beside the weak point -- a fixed 4096-frame job split over the
    #      ranks, each rank walking its contiguous shard in chunks of <= 512 frames (the per-GPU size
    #      configs[3] names; one arena serves every chunk).  Same timing contract: barrier + sync on
    #      both sides, max over ranks.  Inputs are the weak leg's frames tiled (timing is data-blind).
    strong = None
    if world > 1 and args.strong_frames and not args.global_batch and not args.replay_only:
        strong = strong_leg(net, x, a, args.strong_frames, rank, world, max(2, args.steps // 4), sync, max_over_ranks, args.dtype)

    # ---- per-kernel timing with HIP events on the launch stream (rank 0)
    result = None
    lanes = net.get_option("lanes")
    plan16 = net.get_option("bf16_plan")
    if args.dtype == "bf16" and plan16 > 0 and B >= plan16 and lanes == 2:
        lanes = 3                   # the bf16 engine's large-batch plan (engine.hip run_forward): three lanes, gemm_ring128, overlap 0
    trunk_lanes = net.get_option("trunk_lanes")
    if rank == 0:
        per = {}
        reps = args.steps if args.replay_only else 3
        # The timed region overlaps the sub-batch lanes (and the audio streams) on the GPU, so
        # per-kernel durations there are not separable.  The roofline pass replays the SAME launches
        # (same lanes, hence the same kernels, tile choices and grids) serialised on one stream with
        # an event pair around each (casync_profile_forward).
        # every launch is timed `reps` times; its figure is the MEDIAN of those (one slow launch -- a cold instruction
        # cache, a clock dip -- otherwise moves a whole kernel family's average), counted `reps` times so that the
        # sums below keep their meaning
        runs = [net.profile(x, a) for _ in range(reps)]
        for i, row in enumerate(runs[0]):
            ms = sorted(r[i]["ms"] for r in runs)[reps // 2]
            c = per.setdefault(row["kernel"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
            c["ms"] += ms * reps; c["flops"] += row["flops"] * reps; c["bytes"] += row["bytes"] * reps; c["n"] += reps
        if args.kernel_table:
            tot = sum(c["ms"] for c in per.values())
            for k, c in sorted(per.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{k:40s} {c['n'] // reps:4d} launches {c['ms'] / reps:9.3f} ms {100 * c['ms'] / tot:5.1f}%  "
                      f"{c['flops'] / c['ms'] / 1e9:8.1f} TFLOP/s {c['bytes'] / c['ms'] / 1e6:8.1f} GB/s", file=sys.stderr)
            for r in runs[-1]:
                print(f"  {r['name']:48s} {r['kernel']:38s} {r['ms']:8.3f} ms {r['flops'] / max(r['ms'], 1e-6) / 1e9:8.1f} TF "
                      f"{r['bytes'] / max(r['ms'], 1e-6) / 1e6:8.1f} GB/s", file=sys.stderr)
        dom_name, dom = max(per.items(), key=lambda kv: kv[1]["ms"])
        dom_raw = sum(sorted(r[i]["ms_raw"] for r in runs)[reps // 2] * reps for i, row in enumerate(runs[0]) if row["kernel"] == dom_name)
        executed_flops = sum(c["flops"] for c in per.values()) / reps      # of one B-frame forward, as launched
        tf = dom["flops"] / dom["ms"] / 1e9
        gbs = dom["bytes"] / dom["ms"] / 1e6
        mfma_peak = MFMA_F32_PEAK_TF if args.dtype == "f32" else MFMA_BF16_PEAK_TF
        mfma_bound = dom_name.startswith(("pw_gemm", "ir_fused")) and tf / mfma_peak >= gbs / HBM_PEAK_GBS
        tag = f"{PROFILE_TAG}_{'f32_b64' if args.dtype == 'f32' else 'bf16_b512'}"
        roofline = {
            "kernel": dom_name,
            "bound": "mfma" if mfma_bound else "hbm",
            "achieved": round(tf if mfma_bound else gbs, 2),
            "peak": mfma_peak if mfma_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if mfma_bound else "GB/s",
            "frac": round((tf / mfma_peak) if mfma_bound else (gbs / HBM_PEAK_GBS), 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
            "algorithmic_gflop_per_launch": round(dom["flops"] / dom["n"] / 1e9, 3),
            "launches_per_step": dom["n"] // reps,
            "avg_launch_ms": round(dom["ms"] / dom["n"], 4),
            # share of the SERIALISED replay's kernel time (the timed two-lane step overlaps kernels, so it is shorter
            # than the replay and a kernel runs slower inside it)
            "share_of_replay": round(dom["ms"] / sum(c["ms"] for c in per.values()), 3),
            "avg_launch_ms_raw": round(dom_raw / dom["n"], 4),     # the event pairs as measured (both are printed)
            "measured": "HIP events around every launch (minus the calibrated cost of an empty event pair); the timed "
                        "run's own launches (same lanes, tiles, grids) serialised on one stream; "
                        f"profiles/{tag}_kernel_stats_replay.csv is rocprofv3 --kernel-trace --stats of `bench.py --replay-only`",
        }
        # HBM-side bytes and MFMA-busy of that kernel from the committed PMC passes (tools/collect_profiles.sh
        # with --replay-only, i.e. these same launches; the counters are per launch, so they do not depend on
        # the step count).  Only for the two configurations the passes were collected on.
        # They are quoted only when they were collected from THIS version of the kernels (source_hash), else null + why.
        from calipsync_amd import build as _build
        src_hash = _build.source_hash()
        roofline["source_hash"] = src_hash
        default_cfg = (args.dtype == "f32" and B == 64) or (args.dtype == "bf16" and B == 512)
        moved_per_frame = moved_missing = None
        if default_cfg:
            pmc, why = load_profile_json("pmc_traffic" if args.dtype == "f32" else "pmc_traffic_bf16_b512", src_hash)
            if pmc and dom_name in pmc["kernels"]:
                roofline["traffic"] = pmc["kernels"][dom_name]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = pmc["source"]
            else:
                roofline["traffic_source"] = why or f"{dom_name} not in the PMC summary"
            if pmc:
                moved_per_frame, moved_missing = hbm_moved_per_frame(pmc, {k: c["n"] // reps for k, c in per.items()}, B)
            busy, why = load_profile_json("mfma_busy" if args.dtype == "f32" else "mfma_busy_bf16_b512", src_hash)
            if busy and dom_name in busy["kernels"]:
                # the matrix pipes' busy cycles over the kernel's duration in the TRACE pass at the clock the peak is quoted at
                # (>= achieved / peak by construction); the PMC pass itself runs the kernel slower: both are printed
                bk = busy["kernels"][dom_name]
                roofline["mfma_busy"] = bk.get("mfma_busy_of_trace_time", bk["mfma_busy_of_kernel_time"])
                roofline["mfma_busy_pmc_pass"] = bk["mfma_busy_of_kernel_time"]
                roofline["pmc_pass_slowdown"] = bk.get("pmc_pass_slowdown")
                roofline["mfma_busy_source"] = busy["source"]
            else:
                roofline["mfma_busy"] = None
                roofline["mfma_busy_source"] = why or f"{dom_name} not in the PMC summary"
            avg_us, why = rocprof_avg_us(tag, dom_name, src_hash)
            roofline["avg_kernel_us_rocprof"] = avg_us
            if avg_us is None:
                roofline["avg_kernel_us_rocprof_source"] = why
            else:
                roofline["frac_rocprof"] = round(dom["flops"] / dom["n"] / (avg_us * 1e-6) / 1e12 / mfma_peak, 4) if mfma_bound \
                    else round(dom["bytes"] / dom["n"] / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        work = arch.work_per_frame()
        canon_bytes = work["canonical_bytes_f32"] // (1 if args.dtype == "f32" else 2)
        fps = world * B * args.steps / dt
        per_gpu = fps / world
        stage = arch.stagewise_bound(mfma_peak * 1e12, HBM_PEAK_GBS * 1e9, 4 if args.dtype == "f32" else 2)
        two_lane = B >= 16 * lanes and lanes > 1
        result = {
            "metric": "160x160 lip-sync frames/sec (whole node)",
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 (fp32 accumulate; NOT the parity path)",
            "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU 160x160 fp32 crops + HuBERT windows through Model.forward "
                                   f"({'BASELINE configs[1]' if args.dtype == 'f32' else 'bf16 engine, BASELINE configs[2] family'}); "
                                   "frames sharded, weights broadcast once",
                       "global_batch": B * world, "parallelism": f"frames-dp{world}",
                       "lanes_per_gpu": lanes if two_lane else 1,   # the engine runs small batches as one lane
                       "trunk_lanes": (trunk_lanes or lanes) if two_lane else 1,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": {"nccl": "nccl (RCCL)", "gloo": "gloo"}.get(backend, "none (single process)"),
                       **({"rccl": comm} if comm else {}),
                       **({"rehearsal": f"{world} ranks share {n_dev} GPU(s), gloo"} if shared else {}),
                       **({"strong_scaling": strong} if strong else {})},
            **({"replay_only": "serialised replays only: value is NOT the two-lane rate"} if args.replay_only else {}),
            "roofline": roofline,
            "whole_net": {"mfma_frac": round(per_gpu * work["flops"] / (mfma_peak * 1e12), 4),
                          "hbm_frac_canonical": round(per_gpu * canon_bytes / (HBM_PEAK_GBS * 1e9), 4),
                          "gflop_per_frame": round(work["flops"] / 1e9, 3),
                          # mfma_frac is quoted on the REFERENCE's flops.  The engine executes fewer: with
                          # ups_commute the upsampled half of every Up block's expand conv runs at a quarter of the pixels
                          # (up(W1a.lo) = W1a.up(lo)).  What the matrix pipes actually did:
                          "executed_gflop_per_frame": round(executed_flops / B / 1e9, 3),
                          "mfma_frac_executed": round(per_gpu * executed_flops / B / (mfma_peak * 1e12), 4),
                          # the granularity the kernels implement: HBM bytes the fused engine really moves per
                          # frame (PMC, hash-gated like roofline.traffic) and the share of the HBM roof that is; the
                          # canonical figures around it are quoted on UN-FUSED conv-granularity bytes (context only)
                          "hbm_bytes_moved_per_frame": None if moved_per_frame is None else round(moved_per_frame),
                          "hbm_frac_moved": None if moved_per_frame is None else
                          round(per_gpu * moved_per_frame / (HBM_PEAK_GBS * 1e9), 4),
                          **({"hbm_moved_missing_kernels": moved_missing} if moved_missing else {}),
                          "canonical_mb_per_frame": round(canon_bytes / 1e6, 2),
                          # per stage max(canonical bytes / 8 TB/s, flops / matrix peak), summed
                          "stagewise_bound_fps_per_gpu": round(stage["frames_per_s"], 1),
                          "frac_of_stagewise_bound": round(per_gpu / stage["frames_per_s"], 4)},
        }
        if world == 1 and args.dtype == "f32" and not args.no_secondary and not args.replay_only and not args.global_batch:
            from calipsync_amd import build as _build
            result["secondary"] = secondary_block(net, packed, x, a, dev, _build.source_hash())
        if args.e2e and world == 1:    # measured once per run: the secondary block's figure when it exists
            from calipsync_amd import frame_bench
            result["config"]["e2e"] = result["secondary"]["e2e"] if "secondary" in result else frame_bench.run(net, dev, batch=B)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd_np, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
