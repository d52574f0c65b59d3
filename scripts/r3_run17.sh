set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r3_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()"
for wl in rmat1m_10m_f128 products_2p4m_62m_f100; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $wl > gpurun_out/r3_bench_$wl.json 2> gpurun_out/r3_bench_$wl.err || { tail -20 gpurun_out/r3_bench_$wl.err; exit 1; }
  python -c "
import json,sys; d=json.load(open('gpurun_out/r3_bench_$wl.json')); r=d['roofline']; print('$wl', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()}, 'frac', round(r['frac'],3), 'traffic', r['traffic'], (d['cpp_api'] or {}).get('hot_path_ms'), (d['cpp_api'] or {}).get('full_layer_ms'))"
done
timeout -k 10 300 python bench.py --workload cora_2708_10556 --hip-graph --steps 200 --warmup 20 --no-cpu-baseline --no-cpp-api --no-ceilings | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cora hipgraph', d['ms_per_step'])"
timeout -k 10 600 python bench.py --train-layers 2 --steps 5 --warmup 2 --no-cpu-baseline --no-ceilings | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train2', d['ms_per_step'])"
