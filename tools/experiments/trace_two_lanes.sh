#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the timed two-lane run -> concurrency profile of one forward (trace_overlap.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 3 > $O/log 2>&1
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 $R/tools/experiments/trace_overlap.py $f > $O/overlap.txt 2>&1
cat $O/overlap.txt
python3 - <<PY
import csv, re, collections
rows = list(csv.DictReader(open("$f")))
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|^void |\(.*", "", r["Kernel_Name"]).strip()
    if name.startswith(("at::", "__amd")) or "elementwise" in name: continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
ev.sort()
outc = [e for e in ev if e[2].startswith("outc_kernel")]
t_end = outc[-1][1]; t_begin = outc[-3][1]
cur = [e for e in ev if e[0] >= t_begin and e[1] <= t_end]
# timeline in 100 us buckets: number of kernels in flight and which
t0 = min(e[0] for e in cur)
print("queues", collections.Counter(e[3] for e in cur))
for b in range(0, int((t_end - t0) / 1e5) + 1):
    lo, hi = t0 + b * 1e5, t0 + (b + 1) * 1e5
    act = collections.Counter()
    for s, e, n, q, st in cur:
        ov = min(e, hi) - max(s, lo)
        if ov > 0: act[n[:34]] += ov / 1e5
    print("%5.1f ms  conc %.2f  " % (b / 10, sum(act.values())) + "  ".join("%s %.2f" % kv for kv in act.most_common(4)))
PY
find $O -name "*.csv" -size +20M -delete
