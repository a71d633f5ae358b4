#!/bin/bash
# Run ON THE GPU BOX: GPU test suite, then the fused-block microbench with the streaming kernel off / on, then bench.py.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_${1:-quick}
mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
for mode in 0 1; do
  CASYNC_IR_STREAM=$mode timeout -k 10 200 python tools/microbench.py ir --batch 32 2>&1 | grep -v amdgpu.ids > $O/ir_stream$mode.log
done
paste -d'\n' $O/ir_stream0.log $O/ir_stream1.log
if [ "$2" != "nobench" ]; then
CASYNC_IR_STREAM=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_tile.json 2> $O/bench_tile.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --kernel-table > $O/bench_stream.json 2> $O/bench_stream.err
python - <<PY
import json
for t in ("tile", "stream"):
    try:
        d = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["whole_net"]["mfma_frac"])
    except Exception as e:
        print(t, "failed", e)
PY
grep -E "ir_stream|ir_fused" $O/bench_stream.err | head -10
fi
exit $rc
