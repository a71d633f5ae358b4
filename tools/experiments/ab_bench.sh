#!/bin/bash
# Run ON THE GPU BOX: end-to-end A/B of engine options or libraries (CASYNC_LIB=calipsync_amd/lib/libcasync_base.so), alternating, three rounds of `bench.py --steps 40` each.
#   bash tools/experiments/ab_bench.sh <tag> "CASYNC_X=0 CASYNC_Y=1" "CASYNC_X=1" ...
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/ab_$tag
mkdir -p $O; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round cfg $i [$cfg]', d['value'], d['ms_per_step'])"
  done
done
