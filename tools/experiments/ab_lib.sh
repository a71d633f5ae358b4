#!/bin/bash
# Run ON THE GPU BOX: A/B of two builds of the library (calipsync_amd/lib/libcasync_<name>.so), alternating, three rounds:
# bench.py B=64 fp32 (frames/s) and tools/experiments/small_forward.py at B=8 and B=1 (ms per forward).
#   bash tools/experiments/ab_lib.sh v1 hip
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  for lib in "$@"; do
    export CASYNC_LIB=calipsync_amd/lib/libcasync_$lib.so
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    b8=$(timeout -k 10 100 python tools/experiments/small_forward.py 8 200 2>/dev/null | tail -1)
    b1=$(timeout -k 10 100 python tools/experiments/small_forward.py 1 200 2>/dev/null | tail -1)
    echo "round $round $lib | B=64: $v | $b8 | $b1"
  done
done
