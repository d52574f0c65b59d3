set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_host_api.py -x -q -m gpu > gpurun_out/r3_host_tests.log 2>&1 || { tail -40 gpurun_out/r3_host_tests.log; exit 1; }
tail -3 gpurun_out/r3_host_tests.log
./tests/cpp/bench_host_api 10000000 100000000 256 5 0 0 2 1 2>&1 | tail -3
