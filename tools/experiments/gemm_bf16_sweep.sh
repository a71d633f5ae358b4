#!/bin/bash
# bf16 GEMMs of the B=512 plan (M = 256 frames x pixels per lane): tile config x ring depth (GPU box).
S="25600,1024,512;25600,512,1024;25600,1024,1024;25600,576,1024;25600,2304,512;25600,2048,1024;102400,512,256;102400,256,512;409600,256,128;409600,128,256"
for pipe in 0 3 4; do for cfg in 0 1 2; do
  echo "== pipe $pipe cfg $cfg"
  CASYNC_GEMM_CFG=$cfg timeout -k 10 100 python tools/microbench.py gemm --dtype bf16 --rotate 4 --shape "$S" 2>&1 | grep -v amdgpu
done; done
