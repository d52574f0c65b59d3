set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or transform or linear or stack or layer" > gpurun_out/r3_gemm_tests.log 2>&1 || { tail -40 gpurun_out/r3_gemm_tests.log; exit 1; }
tail -3 gpurun_out/r3_gemm_tests.log
FS=128,256 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py 2>&1 | grep "F="
