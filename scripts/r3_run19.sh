set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_exp_side.log
: > $L
for rep in 1 2; do
for side in 1 0; do
  GNNX_HIP_LIB=exp GNNX_SPMM_SIDE=$side CHUNKS=1024 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
  GNNX_HIP_LIB=exp GNNX_SPMM_SIDE=$side CHUNKS=1024,256 N=1000000 E=10000000 F=128 SEED=1 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
done; done
cat $L
