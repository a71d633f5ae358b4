#!/bin/bash
# Run ON THE GPU BOX: end-to-end A/B of engine options or libraries (CASYNC_LIB=gpurun_out/lib/libcasync_base.so), alternating, three rounds of `bench.py --steps $AB_STEPS` each.
#   bash tools/experiments/ab_bench.sh <tag> "CASYNC_X=0 CASYNC_Y=1" "CASYNC_X=1" ...
#   AB_ARGS="--batch 512" AB_STEPS=10 bash tools/experiments/ab_bench.sh b512 ...     (other bench.py arguments, e.g. the 512-frame shard or --dtype bf16)
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/ab_$tag
mkdir -p $O; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps ${AB_STEPS:-40} --warmup ${AB_WARMUP:-10} $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round cfg $i [$cfg]', d['value'], d['ms_per_step'])"
  done
done
