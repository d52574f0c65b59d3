set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3_gpu_tests.log
for wl in rmat10m_100m_f256 rmat1m_10m_f128 products_2p4m_62m_f100; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-order-control --workload $wl > gpurun_out/r3_bench_$wl.json 2> gpurun_out/r3_bench_$wl.err || { tail -20 gpurun_out/r3_bench_$wl.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/r3_bench_$wl.json')); print('$wl', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()})"
done
timeout -k 10 300 python scripts/exp_ceilings.py > gpurun_out/r3_ceilings.log 2>&1 || { tail -20 gpurun_out/r3_ceilings.log; exit 1; }
cat gpurun_out/r3_ceilings.log
