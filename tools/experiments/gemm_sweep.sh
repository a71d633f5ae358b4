#!/bin/bash
# GEMM tile config x stream-K x pipelined ring on the plan's shapes (GPU box): bash tools/experiments/gemm_sweep.sh
S="6400,512,1024;6400,1024,512;6400,1024,1024;6400,2304,512;6400,2048,1024;6400,512,2048;6400,576,1024;3200,512,1024;3200,1024,512;3200,2048,1024;12800,1024,512;12800,128,1024;12800,256,128;51200,512,256;51200,256,128;51200,64,512;25600,512,256;4096,4096,4096"
for pipe in 0 1; do for cfg in 0 1 2; do for sk in 0 1; do
  echo "== pipe $pipe cfg $cfg sk $sk"
  CASYNC_GEMM_CFG=$cfg CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --shape "$S" 2>&1 | grep -v amdgpu
done; done; done
