#!/bin/bash
# forced tile config x stream-K on/off over the plan's awkward GEMM shapes (run on the GPU box)
S="6400,512,1024;6400,1024,512;6400,1024,1024;6400,2304,512;6400,512,256;3200,512,1024;3200,1024,512;25600,512,256;12800,512,256;25600,256,512"
for cfg in 0 1 2; do for sk in 0 1; do echo "== cfg $cfg sk $sk"; CASYNC_GEMM_CFG=$cfg CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --shape "$S" 2>&1 | grep -v amdgpu; done; done
