#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the engine library on the fused inverted-residual shapes.
#   bash tools/experiments/r3_ab_ir.sh <tag>   (baseline = calipsync_amd/lib/ab/libcasync_base.so, new = the in-tree library)
set -e
R=$GRAFT_REPO_ROOT
tag=${1:-ab1}
O=$R/gpurun_out/r3_$tag
mkdir -p $O
cd $R
export TMPDIR=/tmp
BASE=$R/calipsync_amd/lib/ab/libcasync_base.so
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
CASYNC_LIB=$BASE timeout -k 10 200 python tools/microbench.py ir --batch 32 > $O/ir_base.log 2>&1
timeout -k 10 200 python tools/microbench.py ir --batch 32 > $O/ir_new.log 2>&1
paste -d'\n' $O/ir_base.log $O/ir_new.log
CASYNC_LIB=$BASE timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
timeout -k 10 300 python bench.py --no-cpu-baseline --kernel-table > $O/bench_new.json 2> $O/bench_new.err
python - <<PY
import json
for t in ("base", "new"):
    d = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["whole_net"]["mfma_frac"])
PY
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -d $O/pmc_lds --output-format csv -- python3 $R/tools/microbench.py ir --batch 32 --iters 3 > $O/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ir_fused" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, "conflict %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]), "valu %.0f lds %.0f" % (m["SQ_INSTS_VALU"], m["SQ_INSTS_LDS"]))
PY
find $O -name "*.csv" -size +5M -delete
