set -e
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bf16
mkdir -p $out; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
S="51200,512,1024;51200,1024,512;204800,512,256;204800,256,512;819200,256,128;819200,128,256"
python3 $R/tools/microbench.py gemm --dtype bf16 --shape "$S" --iters 10 2>&1 | grep -v amdgpu
rocprofv3 --pmc FETCH_SIZE -d $out/f --output-format csv -- python3 $R/tools/microbench.py gemm --dtype bf16 --shape "$S" --iters 3 > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/w --output-format csv -- python3 $R/tools/microbench.py gemm --dtype bf16 --shape "$S" --iters 3 > $out/w.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $out/s --output-format csv -- python3 $R/tools/microbench.py gemm --dtype bf16 --shape "$S" --iters 3 > $out/s.log 2>&1 || true
echo done
