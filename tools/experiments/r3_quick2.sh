#!/bin/bash
# Run ON THE GPU BOX: stream-kernel tests, microbench off / on, phase timeline.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_${1:-quick2}
mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "ir_ or upsample" > $O/pytest_ir.log 2>&1; rc=$?
tail -3 $O/pytest_ir.log
[ $rc -ne 0 ] && grep -E "^(FAILED|ERROR)" $O/pytest_ir.log | head -20
for mode in 0 1; do
  CASYNC_IR_STREAM=$mode timeout -k 10 200 python tools/microbench.py ir --batch 32 2>&1 | grep -v amdgpu.ids > $O/ir_stream$mode.log
done
paste -d'\n' $O/ir_stream0.log $O/ir_stream1.log | head -12
timeout -k 10 200 python tools/experiments/ir_timeline.py 32 2>&1 | grep -v amdgpu | head -3
exit $rc
