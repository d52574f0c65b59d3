set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_exp_gemm128.log
: > $L
GNNX_HIP_LIB=exp FS=128 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py >> $L 2>&1
GNNX_HIP_LIB=exp GNNX_GEMM_GEO128=22 FS=128 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py >> $L 2>&1
GNNX_HIP_LIB=exp GNNX_GEMM_GEO128=22 N=1000000 FS=128 TALL_ONLY=1 timeout -k 10 300 python scripts/exp_gemm.py >> $L 2>&1
grep "F=" $L
