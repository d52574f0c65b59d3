set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "gemm or host or padded or transform or linear or stack or sharded or bf16out" > gpurun_out/r3_gemm_tests.log 2>&1 || { tail -40 gpurun_out/r3_gemm_tests.log; exit 1; }
tail -3 gpurun_out/r3_gemm_tests.log
FS=128,256 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py > gpurun_out/r3_exp_gemm.log 2>&1 || { tail -20 gpurun_out/r3_exp_gemm.log; exit 1; }
cat gpurun_out/r3_exp_gemm.log
N=1000000 FS=128 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py 2>&1 | grep "F="
