#!/bin/bash
# Run ON THE GPU BOX: the GEMM shapes of a B=8 forward (M = 800 rows at 10x10, 3200 at 20x20) under each tile config, with and
# without stream-K:  bash tools/experiments/small_m_gemm.sh [cfgs...]
S="800,1024,1024;800,576,1024;800,1024,512;800,2304,512;800,2048,1024;800,512,2048;800,512,1024;800,256,1024;800,512,256;800,256,512;3200,1024,512;3200,256,1024;100,1024,1024;100,576,1024"
cd $GRAFT_REPO_ROOT
for cfg in ${@:-auto 2}; do for sk in 1 0; do
  echo "== cfg $cfg stream-K $sk"
  if [ $cfg = auto ]; then CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --batch 1 --iters 200 --shape "$S" 2>&1 | grep -v amdgpu
  else CASYNC_GEMM_CFG=$cfg CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --batch 1 --iters 200 --shape "$S" 2>&1 | grep -v amdgpu; fi
done; done
