#!/bin/bash
# tile policy of the GEMM picker in the two-lane (concurrent) schedule, by batch size:
# mode 0 = cost model as for a launch that owns the chip, 1 = always 64x64, 2 = no 128x128,
# 3 = 64x64 up to CASYNC_GEMM_CONC_TILES tiles (default policy: 3 / 2048)
for b in ${BATCHES:-32 64 128 256}; do
  for mode in "0 0" "1 0" "3 1024" "3 2048" "3 4096" "3 8192"; do
    set -- $mode
    echo -n "B=$b mode=$1 tiles=$2 "
    CASYNC_GEMM_CONC=$1 CASYNC_GEMM_CONC_TILES=$2 python bench.py --no-cpu-baseline --steps 30 --batch $b 2>/dev/null | cut -c60-90
  done
done
