set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_exp_gemm_ablate.log
: > $L
for ab in 0 1 2 3; do
  echo "ablate $ab" >> $L
  GNNX_HIP_LIB=exp GNNX_GEMM_ABLATE=$ab FS=128,256 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py 2>&1 | grep "F=" >> $L
done
cat $L
