#!/bin/bash
# Run ON THE GPU BOX: ms per forward at single-lane batch sizes under option sets, alternating, two rounds.
#   bash tools/experiments/b_sweep.sh "CASYNC_X=0" "CASYNC_X=1"
cd $GRAFT_REPO_ROOT
for r in 1 2; do for e in "$@"; do
  line="[$e]"
  for B in 1 4 8 12 16 24 31; do line="$line $(env $e timeout -k 10 100 python tools/experiments/small_forward.py $B 100 2>/dev/null | tail -1 | sed 's/ ms per forward over 100//')"; done
  echo "$line"
done; done
