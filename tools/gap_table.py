#!/usr/bin/env python3
"""Per-launch gap table of ONE forward from a rocprofv3 --kernel-trace CSV (small batches: the launch chain is the bound).

    python tools/gap_table.py <kernel_trace.csv> [forwards_from_the_end=2]

For the chosen forward (delimited by outc_kernel launches) it lists every launch in start order with its duration,
the idle time on the device before it (start minus the latest end of any earlier kernel: negative = overlapped with
another stream's kernel), and at the end: wall time of the forward, sum of durations, time with nothing running."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|^void |\(.*", "", r["Kernel_Name"]).strip()
    if name.startswith(("at::", "__amd")) or "elementwise" in name:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r.get("Grid_Size", 0) or 0),
               int(r.get("Workgroup_Size", 0) or 0)))
ev.sort()
ends = [i for i, e in enumerate(ev) if e[2].startswith("outc_kernel")]
lo, hi = ends[-1 - back] + 1, ends[-back] + 1 if back else len(ev)
cur = ev[lo:hi]
t0 = cur[0][0]
busy_until, idle, tot = cur[0][0], 0, 0
print(f"{'#':>3s} {'start us':>9s} {'dur us':>8s} {'gap us':>8s} {'wgs':>6s}  kernel")
for i, (s, e, n, grid, wg) in enumerate(cur):
    gap = s - busy_until
    if gap > 0:
        idle += gap
    print(f"{i:3d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f} {grid // max(wg, 1):6d}  {n[:70]}")
    busy_until = max(busy_until, e)
    tot += e - s
wall = max(e for s, e, *_ in cur) - t0
print(f"forward: {len(cur)} launches, wall {wall / 1e3:.1f} us, sum of durations {tot / 1e3:.1f} us, device idle {idle / 1e3:.1f} us "
      f"({100 * idle / wall:.1f} %)")
