#!/bin/bash
# tile policy of the GEMM picker in the two-lane (concurrent) schedule, by batch size
for b in 32 64 128 256; do
  for mode in "0 0" "1 0" "2 0" "3 512" "3 1024" "3 2048" "3 4096"; do
    set -- $mode
    echo -n "B=$b mode=$1 tiles=$2 "
    CASYNC_GEMM_CONC=$1 CASYNC_GEMM_CONC_TILES=$2 python bench.py --no-cpu-baseline --steps 30 --batch $b 2>/dev/null | cut -c60-90
  done
done
