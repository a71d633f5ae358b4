#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library, alternating, three rounds:
# bench.py B=64 fp32 (frames/s) and tools/experiments/small_forward.py at B=8 and B=1 (ms per forward).
#   bash tools/experiments/ab_lib.sh base hip        # tools/experiments/ablib/libcasync_<name>.so
# The copies live in tools/experiments/ablib/ (git-ignored; gpurun_out/ does not travel to the box): make one with
#   mkdir -p tools/experiments/ablib && cp calipsync_amd/lib/libcasync_hip.so tools/experiments/ablib/libcasync_base.so
# and DELETE the directory when the comparison is recorded -- every file in the tree ships with every GPU lease.
#   AB_ARGS / AB_STEPS as in ab_bench.sh; AB_SMALL=0 skips the B=8 / B=1 legs.
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  for lib in "$@"; do
    export CASYNC_LIB=tools/experiments/ablib/libcasync_$lib.so
    v=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps ${AB_STEPS:-40} --warmup ${AB_WARMUP:-10} $AB_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    if [ "${AB_SMALL:-1}" = 1 ]; then
      b8=$(timeout -k 10 100 python tools/experiments/small_forward.py 8 200 2>/dev/null | tail -1)
      b1=$(timeout -k 10 100 python tools/experiments/small_forward.py 1 200 2>/dev/null | tail -1)
    fi
    echo "round $round $lib | bench $AB_ARGS: $v | $b8 | $b1"
  done
done
