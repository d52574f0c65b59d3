set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_loopback.py -x -q -m gpu > gpurun_out/r3_shard_tests.log 2>&1 || { tail -40 gpurun_out/r3_shard_tests.log; exit 1; }
tail -3 gpurun_out/r3_shard_tests.log
