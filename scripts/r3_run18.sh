set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hub_rows or split_rows or batchnorm_sums or fused_prologue or bf16_feature or baseline_sizes" > gpurun_out/r3_side_tests.log 2>&1 || { tail -40 gpurun_out/r3_side_tests.log; exit 1; }
tail -3 gpurun_out/r3_side_tests.log
for wl in rmat10m_100m_f256 rmat1m_10m_f128 products_2p4m_62m_f100; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cpp-api --no-order-control --workload $wl > gpurun_out/r3_bench_$wl.json 2> gpurun_out/r3_bench_$wl.err || { tail -20 gpurun_out/r3_bench_$wl.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/r3_bench_$wl.json')); r=d['roofline']; print('$wl', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()}, 'frac', round(r['frac'],3))"
done
timeout -k 10 300 python bench.py --workload cora_2708_10556 --hip-graph --steps 200 --warmup 20 --no-cpu-baseline --no-cpp-api --no-ceilings | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cora hipgraph', d['ms_per_step'])"
timeout -k 10 300 python bench.py --workload rmat1m_10m_f128 --hip-graph --steps 50 --warmup 5 --no-cpu-baseline --no-cpp-api --no-ceilings | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rmat1m hipgraph', d['ms_per_step'])"
