#!/usr/bin/env python3
"""Concurrency of the timed two-lane schedule from a rocprofv3 --kernel-trace CSV:
    python tools/experiments/trace_overlap.py <kernel_trace.csv> [steps_to_skip]
For the last forward in the trace: wall time, sum of kernel durations, time with 0 / 1 / 2 / 3+ kernels in
flight, and the per-kernel-family share of the time in which it ran ALONE (nothing else in flight)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|^void |\(.*", "", r["Kernel_Name"]).strip()
    if name.startswith(("at::", "__amd")) or "elementwise" in name:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
ev.sort()
# one forward = from an inc_kernel pair / nchw kernels to the last outc_kernel: split on outc launches
outc_ends = [e for s, e, n in ev if n.startswith("outc_kernel")]
n_lanes = 2
fw_end = outc_ends[-1]
fw_prev_end = outc_ends[-1 - n_lanes]
cur = [(s, e, n) for s, e, n in ev if s >= fw_prev_end - 0 and e <= fw_end and s > outc_ends[-1 - n_lanes] - 1]
cur = [x for x in cur if x[0] >= min(s for s, e, n in ev if s > fw_prev_end - 2_000_000 and e > fw_prev_end)]
t0, t1 = min(s for s, e, n in cur), max(e for s, e, n in cur)
pts = []
for s, e, n in cur:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
active = collections.Counter()
by_level = collections.Counter()
alone = collections.Counter()
last = t0
for t, d, n in pts:
    lvl = sum(active.values())
    dt = t - last
    by_level[min(lvl, 3)] += dt
    if lvl == 1:
        alone[next(k for k, v in active.items() if v > 0)] += dt
    last = t
    active[n] += d
wall = t1 - t0
print(f"forward wall {wall / 1e3:.1f} us, {len(cur)} launches, sum of kernel durations {sum(e - s for s, e, n in cur) / 1e3:.1f} us")
for lvl in range(4):
    print(f"  {lvl}{'+' if lvl == 3 else ' '} kernels in flight: {by_level[lvl] / 1e3:8.1f} us  {100 * by_level[lvl] / wall:5.1f} %")
print("  time a kernel family ran ALONE (top):")
for k, v in alone.most_common(8):
    print(f"    {k:60s} {v / 1e3:8.1f} us")
dur = collections.defaultdict(list)
for s, e, n in cur:
    dur[n].append(e - s)
print("  per family in this forward (concurrent durations):")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"    {k:60s} n={len(v):3d} total {sum(v) / 1e3:8.1f} us  avg {sum(v) / len(v) / 1e3:7.1f} us")
