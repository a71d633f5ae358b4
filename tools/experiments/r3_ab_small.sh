#!/bin/bash
# Run ON THE GPU BOX: small-batch A/B of engine options (single-lane regime): bench.py --batch B for B in 1 8 16 32.
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for b in 1 8 16 32; do
  for cfg in "$@"; do
    env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --batch $b --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b [$cfg]', d['value'], d['ms_per_step'])"
  done
done
