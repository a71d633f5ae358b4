#!/bin/bash
# Run ON THE GPU BOX (gpurun): collects AND folds the rocprofv3 evidence that profiles/ keeps.
#   bash tools/collect_profiles.sh <tag> [f32|bf16]
# 1. bench.py's own line (+ --kernel-table) -> <tag>_bench*.json, <tag>_launch_table*.txt
# 2. kernel-trace + stats of `bench.py --replay-only` (ONLY serialised replays, warm-up included: the mode bench.py's
#    roofline block is measured in) and of the timed two-lane run -> <tag>_*_kernel_stats_{replay,timed}.csv + .meta.json
# 3. PMC passes of the replay (FETCH_SIZE, WRITE_SIZE separately, no trace domains mixed in) -> <tag>_pmc_traffic*.json
# 4. MFMA-busy / wait / LDS counter groups of the replay (tools/collect_pmc_busy.sh) -> <tag>_mfma_busy*.json
# Every JSON carries source_hash = sha256 of calipsync_amd/csrc (calipsync_amd.build.source_hash): bench.py quotes
# counters only when the hash matches the sources of the library it runs.  Results land in gpurun_out/profiles_<tag>/
# (copy them into profiles/ and commit).
set -e
tag=${1:-r5}
dt=${2:-f32}
R=$GRAFT_REPO_ROOT
extra=""; sfx=""; cfg="f32_b64"; what="B=64 fp32"
if [ "$dt" = bf16 ]; then extra="--dtype bf16"; sfx="_bf16_b512"; cfg="bf16_b512"; what="B=512 bf16"; fi
out=$R/gpurun_out/prof_${tag}_$dt
dst=$R/gpurun_out/profiles_$tag
mkdir -p $out $dst
cd /tmp && export TMPDIR=/tmp
hash=$(cd $R && python3 -c "from calipsync_amd.build import source_hash; print(source_hash())")
rocprofv3 --kernel-trace --stats -d $out/replay --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --replay-only --steps 10 --warmup 3 $extra > $out/replay.log 2>&1
echo "stats replay done"
rocprofv3 --kernel-trace --stats -d $out/timed --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 3 $extra > $out/timed.log 2>&1
echo "stats timed done"
for m in replay timed; do
  f=$(find $out/$m -name "*kernel_stats.csv" | head -1)
  cp $f $dst/${tag}_${cfg}_kernel_stats_$m.csv
  echo "{\"source_hash\": \"$hash\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary $([ $m = replay ] && echo --replay-only) --steps 10 --warmup 3 $extra\"}" > $dst/${tag}_${cfg}_kernel_stats_$m.meta.json
done
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --replay-only --steps 2 --warmup 1 $extra > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-secondary --replay-only --steps 2 --warmup 1 $extra > $out/pmc_write.log 2>&1
echo "pmc write done"
python3 $R/tools/pmc_to_json.py $out/pmc_fetch $out/pmc_write $dst/${tag}_pmc_traffic$sfx.json "$what" > $out/pmc_traffic.txt
bash $R/tools/collect_pmc_busy.sh ${tag}_$dt --no-secondary $extra
python3 $R/tools/pmc_busy_to_json.py $R/gpurun_out/pmc_busy_${tag}_$dt $dst/${tag}_mfma_busy$sfx.json $dst/${tag}_${cfg}_kernel_stats_replay.csv > $out/mfma_busy.txt
# the trace CSVs are large and not needed once the stats exist
find $out $R/gpurun_out/pmc_busy_${tag}_$dt -name "*.csv" -size +5M -delete
# bench line + launch table LAST, so that they can quote the counters collected above (copy the JSONs into profiles/ first)
mkdir -p $R/profiles && cp $dst/${tag}_* $R/profiles/
cd $R
if [ "$dt" = f32 ]; then
  python3 bench.py --e2e --kernel-table > $dst/${tag}_bench.json 2> $dst/${tag}_launch_table.txt
else
  python3 bench.py --dtype bf16 --kernel-table > $dst/${tag}_bench_bf16_b512.json 2> $dst/${tag}_launch_table_bf16_b512.txt
fi
ls -la $dst
