#!/bin/bash
# MFMA-busy and stall counters of the fp32 GEMM kernels on the plan's shapes (B=64)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $out/g$i --output-format csv -- python3 $R/tools/microbench.py gemm --iters 3 > $out/g$i.log 2>&1 || echo "group $i failed: $grp"
done
echo done
