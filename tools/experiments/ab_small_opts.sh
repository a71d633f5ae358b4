#!/bin/bash
# Run ON THE GPU BOX: A/B of engine options at small batches (ms per forward, tools/experiments/small_forward.py), alternating, three rounds.
#   AB_BATCHES="1 4 8 11" bash tools/experiments/ab_small_opts.sh "CASYNC_X=0" "CASYNC_X=1" ...
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
for round in 1 2 3; do
  for cfg in "$@"; do
    line="round $round [$cfg] |"
    for B in ${AB_BATCHES:-1 8}; do line="$line $(env $cfg timeout -k 10 100 python tools/experiments/small_forward.py $B 200 2>/dev/null | tail -1 | sed 's/ ms per forward over 200//')"; done
    echo "$line"
  done
done
