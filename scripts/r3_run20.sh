set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hub_rows or split_rows or batchnorm_sums or fused_prologue or bf16_feature or baseline_sizes or sym_mode" > gpurun_out/r3_side_tests.log 2>&1 || { tail -40 gpurun_out/r3_side_tests.log; exit 1; }
tail -3 gpurun_out/r3_side_tests.log
L=gpurun_out/r3_exp_side.log
: > $L
for side in 1 0; do
  GNNX_HIP_LIB=exp GNNX_SPMM_SIDE=$side CHUNKS=1024 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
  GNNX_HIP_LIB=exp GNNX_SPMM_SIDE=$side CHUNKS=1024,256 N=1000000 E=10000000 F=128 SEED=1 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
done
CHUNKS=1024 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
CHUNKS=1024 N=1000000 E=10000000 F=128 SEED=1 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
CHUNKS=1024 N=2400000 E=62000000 F=100 SEED=3 timeout -k 10 300 python scripts/exp_hub.py 2>&1 | grep "split rows" >> $L
cat $L
