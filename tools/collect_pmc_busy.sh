#!/bin/bash
# Run ON THE GPU BOX (gpurun): MFMA-busy / LDS-stall counters of `bench.py --replay-only` (the timed
# run's own launches, serialised), one counter group per pass, no trace domains mixed in.
#   bash tools/collect_pmc_busy.sh <tag> [extra bench args]
# -> gpurun_out/pmc_busy_<tag>/g{1,2,3}/...counter_collection.csv ; tools/pmc_busy_to_json.py folds them.
tag=${1:-r2}
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_busy_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp -d $out/g$i --output-format csv -- python3 $R/bench.py --no-cpu-baseline --replay-only --steps 2 --warmup 1 "$@" > $out/g$i.log 2>&1 || echo "group $i failed: $grp"
  echo "pmc group $i done"
done
