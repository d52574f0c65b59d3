set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3_gpu_tests.log
