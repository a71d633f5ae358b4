#!/bin/bash
# Run ON THE GPU BOX: start-delay sweep of the row-streaming fused block (options ir_stream_stagger / _skew / _wgs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_${1:-sweep}
mkdir -p $O; cd $R; export TMPDIR=/tmp
run() { env "$@" timeout -k 10 120 python tools/microbench.py ir --batch 32 --only "${ONLY:-up}" 2>&1 | grep -v amdgpu.ids | awk -v tag="$*" '{print tag, $0}'; }
run CASYNC_IR_STREAM=0
for a in 0 25 50 100 200; do for b in 0 4; do run CASYNC_IR_STREAM_STAGGER=$a CASYNC_IR_STREAM_SKEW=$b; done; done
for w in 4 6 9; do run CASYNC_IR_STREAM_WGS=$w CASYNC_IR_STREAM_MIN=1; done
ONLY=down1.ir1 run CASYNC_IR_STREAM=0
for a in 0 50 100 200; do ONLY=down1.ir1 run CASYNC_IR_STREAM_STAGGER=$a; done
