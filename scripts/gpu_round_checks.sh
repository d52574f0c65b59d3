#!/usr/bin/env bash
# The builder's end-of-round check on one MI355X box (through gpurun, from the repo root):
#   GIT_HEAD=$(git rev-parse --short HEAD); gpurun --timeout 1200 -- "GIT_HEAD=$GIT_HEAD bash scripts/gpu_round_checks.sh r03"
# gpu-marked tests, smoke(), the round's profile set (scripts/profile_all.sh) and the default bench line.  Everything it writes lands
# under gpurun_out/ (profiles: gpurun_out/profiles_<tag>/ -- copy them into profiles/ afterwards).
set -euo pipefail
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1 || { tail -40 gpurun_out/${TAG}_gpu_tests.log; exit 1; }
tail -3 gpurun_out/${TAG}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()"
bash scripts/profile_all.sh "$TAG" 2>&1 | tail -12
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_default.json"))
r = d["roofline"]
print(d["ms_per_step"], d["value"], {k: r[k] for k in ("achieved", "peak", "frac", "traffic", "avg_launch_ms")}, d["cpp_api"], d["kernels_ms"])
PY
