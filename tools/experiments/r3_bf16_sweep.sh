#!/bin/bash
# Run ON THE GPU BOX: option sweep of the bf16 engine at B = 512 (two alternating rounds of bench.py --dtype bf16 --steps 10).
R=$GRAFT_REPO_ROOT; cd $R
for r in 1 2; do for c in "CASYNC_X=1" "CASYNC_GEMM_CONC_TILES=512" "CASYNC_GEMM_CONC=0" "CASYNC_GEMM_CONC=2" "CASYNC_LANES=4" "CASYNC_LANES=1"; do env $c timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --dtype bf16 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 [$c]', d['value'], d['ms_per_step'])"; done; done
