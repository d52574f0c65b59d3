set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_default.json')); print(d['ms_per_step'], d['value']); print(json.dumps(d['roofline'], indent=1)[:3000]); print(d['cpp_api']); print(d['vertex_order_control'])"
GIT_HEAD=c36991e bash scripts/profile_all.sh r03 2>&1 | tail -8
