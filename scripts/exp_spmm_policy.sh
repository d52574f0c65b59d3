#!/usr/bin/env bash
# Cache policy of the streaming aggregation kernel's one-touch traffic (EXPERIMENTS build, GNNX_SPMM_POLICY bit mask:
# 1 colidx / vals loads nontemporal, 2 Y stores nontemporal, 4 Y stores sc1) on the headline graph: kernel times per variant.
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
mkdir -p gpurun_out
for pol in 0 1 2 4 3 5; do
  GNNX_HIP_LIB=exp GNNX_SPMM_POLICY=$pol python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings \
    > gpurun_out/exp_policy_$pol.json 2> gpurun_out/exp_policy_$pol.err
  python3 - "$pol" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/exp_policy_{sys.argv[1]}.json"))
print("policy", sys.argv[1], "ms_per_step %.2f" % d["ms_per_step"], {k: round(v, 3) for k, v in d["kernels_ms"].items()})
PY
done
