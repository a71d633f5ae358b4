cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for e in "X=0" "CASYNC_GEMM_CFG=2" "CASYNC_GEMM_STREAMK=0" "CASYNC_GEMM_CFG=2 CASYNC_GEMM_STREAMK=0"; do
  echo "[$e] $(env $e timeout -k 10 100 python tools/experiments/small_forward.py 8 200 2>/dev/null | tail -1) | $(env $e timeout -k 10 100 python tools/experiments/small_forward.py 16 100 2>/dev/null | tail -1)"
done; done
