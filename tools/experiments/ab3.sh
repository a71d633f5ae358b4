R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  for lib in "$@"; do
    export CASYNC_LIB=calipsync_amd/lib/libcasync_$lib.so
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    b31=$(timeout -k 10 100 python tools/experiments/small_forward.py 31 100 2>/dev/null | tail -1)
    echo "round $round $lib | B=64: $v | $b31"
  done
done
