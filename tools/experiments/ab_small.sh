#!/bin/bash
# Run ON THE GPU BOX: A/B of library builds at small batches (ms per forward), alternating, three rounds; then bf16 B=512.
#   bash tools/experiments/ab_small.sh base new      (copies under tools/experiments/ablib/, see ab_lib.sh)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  for lib in "$@"; do
    export CASYNC_LIB=tools/experiments/ablib/libcasync_$lib.so
    line="round $round $lib |"
    for B in 1 8 16 31; do line="$line $(timeout -k 10 100 python tools/experiments/small_forward.py $B 200 2>/dev/null | tail -1 | sed 's/ ms per forward over 200//')"; done
    bf=$(timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")
    echo "$line | bf16 B=512: $bf"
  done
done
