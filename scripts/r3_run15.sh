set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
GIT_HEAD=e45684d bash scripts/profile_all.sh r03 2>&1 | tail -12
cd $GRAFT_REPO_ROOT
mkdir -p profiles && cp gpurun_out/profiles_r03/* profiles/
( time timeout -k 10 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_default.json')); print(d['ms_per_step'], d['value']); r=d['roofline']; print({k:r[k] for k in ('achieved','peak','frac','traffic','avg_launch_ms','effective_GBps')}); print(r['traffic_source']); print(r['ceilings']); print(d['cpp_api']); print(d['vertex_order_control']); print(d['kernels_ms'])"
