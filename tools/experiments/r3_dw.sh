#!/bin/bash
# Run ON THE GPU BOX: the fused expand + depthwise kernel (pw_dw.hip): op tests, then bench.py with option fuse_dw off / on.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_${1:-dw}
mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head -20
for m in 0 1; do
CASYNC_FUSE_DW=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --kernel-table > $O/bench_dw$m.json 2> $O/bench_dw$m.err
done
python - <<PY
import json
for t in ("0", "1"):
    try:
        d = json.loads(open("$O/bench_dw%s.json" % t).read().strip().splitlines()[-1])
        print("fuse_dw", t, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["whole_net"]["mfma_frac"])
    except Exception as e:
        print(t, "failed", e)
PY
grep -E "pw_dw|dw3x3|upsample" $O/bench_dw1.err | head -30
exit $rc
