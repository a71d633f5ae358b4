#!/bin/bash
# Run ON THE GPU BOX: ms per forward around the two-lane threshold (32 frames), one lane against two.
cd $GRAFT_REPO_ROOT
for r in 1 2; do for e in "CASYNC_LANES=1" "CASYNC_LANES=2"; do
  line="[$e]"
  for B in 32 36 40 48 64; do line="$line $(env $e timeout -k 10 100 python tools/experiments/small_forward.py $B 60 2>/dev/null | tail -1 | sed 's/ ms per forward over 60//')"; done
  echo "$line"
done; done
