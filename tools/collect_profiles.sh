#!/bin/bash
# Run ON THE GPU BOX (gpurun): collects the rocprofv3 evidence that profiles/ keeps.
#   bash tools/collect_profiles.sh <tag>
# 1. kernel-trace + stats of bench.py in the roofline-replay mode (one lane, plain tiles) and in the
#    default two-lane mode;  2. PMC passes (FETCH_SIZE, WRITE_SIZE separately, no trace domains mixed in)
set -e
tag=${1:-r1}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASYNC_LANES=1 CASYNC_GEMM_STREAMK=0 rocprofv3 --kernel-trace --stats -d $out/lanes1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/lanes1.log 2>&1
echo "stats lanes1 done"
rocprofv3 --kernel-trace --stats -d $out/lanes2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $out/lanes2.log 2>&1
echo "stats lanes2 done"
CASYNC_LANES=1 CASYNC_GEMM_STREAMK=0 CASYNC_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
CASYNC_LANES=1 CASYNC_GEMM_STREAMK=0 CASYNC_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $out/pmc_write.log 2>&1
echo "pmc write done"
find $out -name "*.csv" | head -20
