set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scripts/exp_ceilings.py > gpurun_out/r3_ceilings.log 2>&1 || { tail -20 gpurun_out/r3_ceilings.log; exit 1; }
cat gpurun_out/r3_ceilings.log
