#!/bin/bash
# Run ON THE GPU BOX (gpurun): collects the rocprofv3 evidence that profiles/ keeps.
#   bash tools/collect_profiles.sh <tag> [f32|bf16]
# 1. kernel-trace + stats of `bench.py --replay-only` (the timed run's launches, serialised: the mode
#    bench.py's roofline block is measured in) and of the timed two-lane run itself;
# 2. PMC passes (FETCH_SIZE, WRITE_SIZE separately, no trace domains mixed in) of the replay;
# 3. MFMA-busy / wait / LDS counter groups of the replay (tools/collect_pmc_busy.sh).
# Fold with tools/pmc_to_json.py and tools/pmc_busy_to_json.py, copy the stats CSVs into profiles/.
set -e
tag=${1:-r2}
dt=${2:-f32}
extra=""
[ "$dt" = bf16 ] && extra="--dtype bf16"
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$dt
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out/replay --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 10 --warmup 3 $extra > $out/replay.log 2>&1
echo "stats replay done"
rocprofv3 --kernel-trace --stats -d $out/timed --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 $extra > $out/timed.log 2>&1
echo "stats timed done"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 2 --warmup 1 $extra > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 2 --warmup 1 $extra > $out/pmc_write.log 2>&1
echo "pmc write done"
bash $R/tools/collect_pmc_busy.sh ${tag}_$dt $extra
# the trace CSVs are large and not needed once the stats exist
find $out -name "*kernel_trace.csv" -size +20M -delete
find $out $GRAFT_REPO_ROOT/gpurun_out/pmc_busy_${tag}_$dt -name "*.csv" | head -30
