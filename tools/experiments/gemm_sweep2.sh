#!/bin/bash
# Tile config x stream-K of the lean two-stage loop on the trunk shapes at one- and two-lane M (GPU box).
S="3200,1024,512;3200,512,1024;3200,1024,1024;3200,576,1024;3200,2048,1024;3200,2304,512;6400,1024,512;6400,512,1024;6400,1024,1024;6400,576,1024;6400,2048,1024;6400,2304,512"
for cfg in -1 0 1 2; do for sk in 0 1; do
  echo "== cfg $cfg sk $sk"
  CASYNC_GEMM_CFG=$cfg CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --shape "$S" 2>&1 | grep -v amdgpu
done; done
