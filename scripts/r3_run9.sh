set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CHUNKS=16,32,64,128,256,1024 timeout -k 10 300 python scripts/exp_hub.py > gpurun_out/r3_exp_hub3.log 2>&1 || { tail gpurun_out/r3_exp_hub3.log; exit 1; }
cat gpurun_out/r3_exp_hub3.log
