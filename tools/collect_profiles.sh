#!/bin/bash
# Run ON THE GPU BOX (gpurun): collects the rocprofv3 evidence that profiles/ keeps.
#   bash tools/collect_profiles.sh <tag>
# 1. kernel-trace + stats of `bench.py --replay-only` (the timed run's launches, serialised: the mode
#    bench.py's roofline block is measured in) and of the timed two-lane run itself;
# 2. PMC passes (FETCH_SIZE, WRITE_SIZE separately, no trace domains mixed in) of the replay
set -e
tag=${1:-r2}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/replay --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 10 --warmup 3 > $out/replay.log 2>&1
echo "stats replay done"
rocprofv3 --kernel-trace --stats -d $out/timed --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/timed.log 2>&1
echo "stats timed done"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 2 --warmup 1 > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 2 --warmup 1 > $out/pmc_write.log 2>&1
echo "pmc write done"
find $out -name "*.csv" | head -20
