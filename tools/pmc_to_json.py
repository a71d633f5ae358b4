#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> per-kernel HBM-side bytes per launch.

    python tools/pmc_to_json.py <fetch_dir> <write_dir> <out.json>

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB; on gfx950
FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) streaming reads at 64 B -> doubled;
WRITE_SIZE is exact for 16-B-per-lane stores.  All of this engine's bulk loads are 16 B/lane
(global_load_dwordx4 / global_load_lds_dwordx4)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_names import short  # noqa: E402
from calipsync_amd.build import source_hash  # noqa: E402  (sha256 of the kernel sources the counters were collected from)


def per_kernel(d, counter):
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = short(r["Kernel_Name"])
        agg[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or "elementwise" in k:
        continue
    fk, n = fetch.get(k, (0.0, 0))
    wk, _ = write.get(k, (0.0, 0))
    out[k] = {"launches_sampled": n, "fetch_bytes_per_launch": round(2 * fk * 1024), "write_bytes_per_launch": round(wk * 1024),
              "hbm_bytes_per_launch": round((2 * fk + wk) * 1024)}
json.dump({"source_hash": source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of bench.py " + (sys.argv[4] if len(sys.argv) > 4 else "B=64 fp32") + ", "
                     "--replay-only (the timed run's launches, serialised); FETCH_SIZE x2 (gfx950), KB -> bytes",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k:55s} n={v['launches_sampled']:4d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")
