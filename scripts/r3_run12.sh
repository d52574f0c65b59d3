set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "batchnorm_sums or host_api or sharded or hub_rows or fused_prologue or through_layer or drop_in or driver" > gpurun_out/r3_sums_tests.log 2>&1 || { tail -40 gpurun_out/r3_sums_tests.log; exit 1; }
tail -3 gpurun_out/r3_sums_tests.log
./tests/cpp/bench_host_api 10000000 100000000 256 5 0 2 2 1 2>&1 | tail -4
