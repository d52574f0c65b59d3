set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_exp_hub.log
: > $L
CHUNKS=0,512,1024,2048,4096,8192 timeout -k 10 300 python scripts/exp_hub.py >> $L 2>&1
GNNX_HIP_LIB=exp GNNX_SPMM_HUB=chunk CHUNKS=1024,4096 timeout -k 10 300 python scripts/exp_hub.py >> $L 2>&1
GNNX_HIP_LIB=exp GNNX_SPMM_HUB_LAS=4 CHUNKS=1024,4096 timeout -k 10 300 python scripts/exp_hub.py >> $L 2>&1
cat $L
cd /tmp && export TMPDIR=/tmp
CHUNKS=4096 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub -- python3 $GRAFT_REPO_ROOT/scripts/exp_hub.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_r3_hub -name "*kernel_stats.csv" | head -1 | xargs head -8
