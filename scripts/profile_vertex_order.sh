#!/usr/bin/env bash
# L2 / fabric counters of the forward aggregation in the two vertex orders (run ON THE GPU BOX through gpurun, from the repo root):
#   bash scripts/profile_vertex_order.sh
# --pmc only, never with a trace domain.  Raw output under gpurun_out/prof_vo_*; scripts/summarize_vertex_order.py reads it.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for order in scrambled as-generated; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum -f csv -d "$OUT/prof_vo_${order}_a" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-order-control --vertex-order $order > "$OUT/prof_vo_${order}_a.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum -f csv -d "$OUT/prof_vo_${order}_b" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-order-control --vertex-order $order > "$OUT/prof_vo_${order}_b.log" 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_TAG_STALL_sum -f csv -d "$OUT/prof_vo_${order}_c" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-order-control --vertex-order $order > "$OUT/prof_vo_${order}_c.log" 2>&1
  echo "profiled $order"
done
