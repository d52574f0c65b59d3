set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hub_rows or split_rows or fused_prologue" > gpurun_out/r3_hub_tests.log 2>&1 || { tail -30 gpurun_out/r3_hub_tests.log; exit 1; }
tail -3 gpurun_out/r3_hub_tests.log
L=gpurun_out/r3_exp_hub2.log
: > $L
CHUNKS=256,512,1024,4096 timeout -k 10 300 python scripts/exp_hub.py >> $L 2>&1
cat $L
cd /tmp && export TMPDIR=/tmp
CHUNKS=1024 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub2 -- python3 $GRAFT_REPO_ROOT/scripts/exp_hub.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub2.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub2 -name "*kernel_stats.csv" | head -1 | xargs head -5 | cut -c1-200
