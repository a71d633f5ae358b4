#!/bin/bash
# Run ON THE GPU BOX: the two-kernel reproducer (op_beside_model.py TWO_KERNELS=1: the fp32 fused Up block beside a looping bf16 GEMM) once per
# library variant in tools/experiments/ablib/ -- binary-patched builds of the failing kernel (profiles/r6_two_models.txt section 7).
#   bash tools/experiments/ablib_run.sh out.txt variant [variant ...]
out=$1; shift
for v in "$@"; do
  echo "== $v" >> $out
  CASYNC_LIB=tools/experiments/ablib/libcasync_$v.so TWO_KERNELS=1 timeout -k 10 200 python tools/experiments/op_beside_model.py 2>&1 | grep "fused fp32" | grep "bf16 GEMM" | sed 's/fused fp32 Up block beside bf16 GEMM 25600x1024x512, default tile *//' >> $out || exit 1
done
cat $out
