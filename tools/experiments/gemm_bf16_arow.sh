#!/bin/bash
# bf16 short-K GEMMs of the B=512 plan: the A-stationary kernel (cfg 7) against the 128x128 register-staged one (cfg 0).
S="25600,1024,512;25600,2304,512;25600,512,512;102400,512,256;102400,256,512;409600,128,256"
for cfg in 0 7; do
  echo "== cfg $cfg"
  CASYNC_GEMM_CFG=$cfg timeout -k 10 100 python tools/microbench.py gemm --dtype bf16 --rotate 4 --shape "$S" 2>&1 | grep -v amdgpu
done
