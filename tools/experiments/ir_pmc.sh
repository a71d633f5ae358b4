#!/bin/bash
# PMC counters of the fused inverted-residual kernels (microbench shapes), one counter group per pass
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ir
mkdir -p $out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $out/g$i --output-format csv -- python3 $R/tools/microbench.py ir --iters 3 > $out/g$i.log 2>&1 || echo "group $i failed: $grp"
done
echo done
