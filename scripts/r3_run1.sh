set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hub_rows or split_rows or fused_prologue or relabelling" > gpurun_out/r3_hub_tests.log 2>&1 || { tail -30 gpurun_out/r3_hub_tests.log; exit 1; }
tail -3 gpurun_out/r3_hub_tests.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_hub.json 2> gpurun_out/r3_bench_hub.err || { tail -20 gpurun_out/r3_bench_hub.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_hub.json')); print(d['ms_per_step'], d['kernels_ms'], d['vertex_order_control'])"
