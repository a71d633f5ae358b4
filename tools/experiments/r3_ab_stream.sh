#!/bin/bash
# Run ON THE GPU BOX: row-streaming fused block (ir_stream.hip) against the tile kernel, same library, option ir_stream.
set -e
R=$GRAFT_REPO_ROOT
tag=${1:-stream1}
O=$R/gpurun_out/r3_$tag
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for mode in 0 1; do
  CASYNC_IR_STREAM=$mode timeout -k 10 200 python tools/microbench.py ir --batch 32 > $O/ir_stream$mode.log 2>&1
done
paste -d'\n' $O/ir_stream0.log $O/ir_stream1.log
for m in 1 3 4 8; do
  echo "ir_stream_min=$m"; CASYNC_IR_STREAM_MIN=$m timeout -k 10 200 python tools/microbench.py ir --batch 32 --only up > $O/ir_min$m.log 2>&1; cat $O/ir_min$m.log | grep -v amdgpu.ids
done
CASYNC_IR_STREAM=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_tile.json 2> $O/bench_tile.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --kernel-table > $O/bench_stream.json 2> $O/bench_stream.err
python - <<PY
import json
for t in ("tile", "stream"):
    d = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["whole_net"]["mfma_frac"])
PY
grep -E "ir_stream|ir_fused" $O/bench_stream.err | head -12
cd /tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $O/pmc$i --output-format csv -- python3 $R/tools/microbench.py ir --batch 32 --iters 3 > $O/pmc$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ir_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[1][:60] if "(anonymous" in r["Kernel_Name"] else r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "conflict %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)), "valu/wave %.0f lds/wave %.0f" % (m["SQ_INSTS_VALU"] / max(m.get("SQ_WAVES", 1), 1), m["SQ_INSTS_LDS"] / max(m.get("SQ_WAVES", 1), 1)),
          "waves %.0f cyc %.0f mfma_busy %.3f cu_busy %.3f" % (m.get("SQ_WAVES", 0), cyc, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1024 * cyc, 1), m.get("SQ_BUSY_CU_CYCLES", 0) / max(256 * cyc, 1)))
PY
find $O -name "*.csv" -size +5M -delete
