#!/usr/bin/env python3
"""Headline benchmark: 160x160 lip-sync frames/sec through the MI355X engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is one pass of the hot path (``Model.forward``) over one batch of synthetic
frames per GPU.  Workload = BASELINE.json configs[1]: batch 64 per GPU, 160x160 fp32,
synthetic crops U(0,1) + random HuBERT windows N(0,1) (the reference's own self-benchmark
shapes, image_infer_v1/models/unet.py:342-347), golden-recipe weights.  Inputs are
resident in HBM before the timed region; outputs stay on the device.

N > 1: one process per GPU (torch.distributed, backend "nccl" == RCCL).  Frames are
independent, so the batch is sharded with no data-path collective ("weak" scaling: 64
frames per GPU); the only collective is ONE broadcast of the packed, BN-folded weight
buffer from rank 0 at start-up.

Prints one JSON line (rank 0) with the whole-job frames/sec, the roofline of the dominant
kernel (live HIP-event timing) and, at N=1, the CPU oracle timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: f32-input MFMA == f32 vector peak
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


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
                    help="strong scaling: a FIXED number of frames per step split over the ranks "
                         "(SURVEY 8d: 4096); default 0 = weak scaling, --batch frames per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--kernel-table", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--replay-only", action="store_true",
                    help="skip the timed region; run --steps isolated replays (the mode rocprofv3 / PMC passes "
                         "are collected in, so their per-kernel figures match the roofline block)")
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


def cpu_baseline(sd_np, seconds: float):
    """The CPU oracle (a port of the reference's forward, torch-CPU fp32) on the host cores."""
    from calipsync_amd import recipe
    from oracle import unet_oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = unet_oracle.to_torch(sd_np)
    b = 8
    x, a = recipe.make_inputs(b)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    unet_oracle.forward(sd, xt, at)                      # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        unet_oracle.forward(sd, xt, at)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 50:
            break
    return {"value": round(n * b / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} forwards of batch {b} (same weights/input recipe), torch-CPU fp32, {dt:.1f} s"}


def main():
    args = parse()
    if args.batch <= 0:
        args.batch = 64 if args.dtype == "f32" else 512
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from calipsync_amd import arch, recipe, _lib
    from calipsync_amd.unet import Model
    from calipsync_amd.sharding import broadcast_packed_weights, shard_range

    # ---- weights: rank 0 folds + packs, ONE RCCL broadcast, every rank adopts the buffer
    sd_np = recipe.make_state_dict() if rank == 0 else None
    net = Model(6, "hubert", precision="bf16" if args.dtype == "bf16" else "fp32").to(dev)
    if rank == 0:
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    packed = broadcast_packed_weights(net if rank == 0 else None, dev)
    net.adopt_packed(packed)

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

    for _ in range(args.warmup):
        out = net(x, a)
    sync()
    t0 = time.perf_counter()
    for _ in range(1 if args.replay_only else args.steps):
        out = net(x, a)
    sync()
    dt = (time.perf_counter() - t0) * (args.steps if args.replay_only else 1)
    if world > 1:
        t = torch.tensor([dt], device="cpu" if shared else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out).all()

    # ---- per-kernel timing with HIP events on the launch stream (rank 0)
    result = None
    lanes = int(os.environ.get("CASYNC_LANES", "2"))
    if rank == 0:
        per = {}
        reps = args.steps if args.replay_only else 3
        # The timed region overlaps `lanes` sub-batch lanes (and the audio stream) on the GPU, so
        # per-kernel durations there are not separable.  The roofline pass replays the SAME launches
        # (same lanes, hence the same kernels, tile choices and grids) serialised on one stream with
        # an event pair around each (casync_profile_forward).
        for _ in range(reps):
            for row in net.profile(x, a):
                c = per.setdefault(row["kernel"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
                c["ms"] += row["ms"]; c["flops"] += row["flops"]; c["bytes"] += row["bytes"]; c["n"] += 1
        if args.kernel_table:
            tot = sum(c["ms"] for c in per.values())
            for k, c in sorted(per.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{k:40s} {c['n'] // reps:4d} launches {c['ms'] / reps:9.3f} ms {100 * c['ms'] / tot:5.1f}%  "
                      f"{c['flops'] / c['ms'] / 1e9:8.1f} TFLOP/s {c['bytes'] / c['ms'] / 1e6:8.1f} GB/s", file=sys.stderr)
            rows = net.profile(x, a)
            for r in rows:
                print(f"  {r['name']:48s} {r['kernel']:38s} {r['ms']:8.3f} ms {r['flops'] / max(r['ms'], 1e-6) / 1e9:8.1f} TF "
                      f"{r['bytes'] / max(r['ms'], 1e-6) / 1e6:8.1f} GB/s", file=sys.stderr)
        dom_name, dom = max(per.items(), key=lambda kv: kv[1]["ms"])
        tf = dom["flops"] / dom["ms"] / 1e9
        gbs = dom["bytes"] / dom["ms"] / 1e6
        mfma_peak = MFMA_F32_PEAK_TF if args.dtype == "f32" else MFMA_BF16_PEAK_TF
        mfma_bound = dom_name.startswith(("pw_gemm", "ir_fused")) and tf / mfma_peak >= gbs / HBM_PEAK_GBS
        roofline = {
            "kernel": dom_name,
            "bound": "mfma" if mfma_bound else "hbm",
            "achieved": round(tf if mfma_bound else gbs, 2),
            "peak": mfma_peak if mfma_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if mfma_bound else "GB/s",
            "frac": round((tf / mfma_peak) if mfma_bound else (gbs / HBM_PEAK_GBS), 4),
            "traffic": None,
            "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
            "launches_per_step": dom["n"] // reps,
            "avg_launch_ms": round(dom["ms"] / dom["n"], 4),
            "share_of_step": round(dom["ms"] / sum(c["ms"] for c in per.values()), 3),
            "measured": "HIP events around every launch; the timed run's own launches (same lanes, tiles, grids) "
                        "serialised on one stream; profiles/r1_final_kernel_stats_replay.csv is rocprofv3 "
                        "--kernel-trace --stats of `bench.py --replay-only`",
        }
        # HBM-side bytes per launch of that kernel from the committed PMC passes (collected with
        # tools/collect_profiles.sh with --replay-only, i.e. these same launches; B=64 fp32 only -- the counters are per
        # launch, so they do not depend on the step count)
        pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")
        if args.dtype == "f32" and B == 64 and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            hit = pmc["kernels"].get(dom_name)
            if hit:
                roofline["traffic"] = hit["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/r1_pmc_traffic.json: " + pmc["source"]
        work = arch.work_per_frame()
        canon_bytes = work["canonical_bytes_f32"] // (1 if args.dtype == "f32" else 2)
        fps = world * B * args.steps / dt
        per_gpu = fps / world
        stage = arch.stagewise_bound(mfma_peak * 1e12, HBM_PEAK_GBS * 1e9, 4 if args.dtype == "f32" else 2)
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
                       "lanes_per_gpu": lanes if B >= 16 * lanes else 1,   # the engine runs small batches as one lane
                       **({"rehearsal": f"{world} ranks share {n_dev} GPU(s), gloo"} if shared else {})},
            **({"replay_only": True} if args.replay_only else {}),
            "roofline": roofline,
            "whole_net": {"mfma_frac": round(per_gpu * work["flops"] / (mfma_peak * 1e12), 4),
                          "hbm_frac_canonical": round(per_gpu * canon_bytes / (HBM_PEAK_GBS * 1e9), 4),
                          "gflop_per_frame": round(work["flops"] / 1e9, 3),
                          "canonical_mb_per_frame": round(canon_bytes / 1e6, 2),
                          # per stage max(canonical bytes / 8 TB/s, flops / matrix peak), summed
                          "stagewise_bound_fps_per_gpu": round(stage["frames_per_s"], 1),
                          "frac_of_stagewise_bound": round(per_gpu / stage["frames_per_s"], 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd_np, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
