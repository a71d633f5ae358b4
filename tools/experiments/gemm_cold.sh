#!/bin/bash
# Warm (one operand set, L2 / Infinity Cache hits) against cold (24 sets > 256 MB, HBM) launches of the trunk GEMMs.
S="3200,1024,512;3200,512,1024;3200,1024,1024;6400,1024,512;6400,512,1024"
for rot in 1 24; do for cfg in 2 1; do for sk in 0 1; do
  echo "== rotate $rot cfg $cfg sk $sk"
  CASYNC_GEMM_CFG=$cfg CASYNC_GEMM_STREAMK=$sk timeout -k 10 100 python tools/microbench.py gemm --rotate $rot --shape "$S" 2>&1 | grep -v amdgpu
done; done; done
