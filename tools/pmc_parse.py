#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_parse.py <dir> [substr]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
files = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=lambda f: -__import__("os").path.getmtime(f))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(files[0])):
    name = r["Kernel_Name"]
    if sub and sub not in name:
        continue
    short = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", name)
    agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:55s} {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
