R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
 for lib in v1 hip; do
  CASYNC_LIB=calipsync_amd/lib/libcasync_$lib.so timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round $lib', d['value'], d['ms_per_step'])"
 done
done
