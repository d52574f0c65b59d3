#!/usr/bin/env bash
# Round profile set (run ON THE GPU BOX through gpurun, from the repo root):
#   GIT_HEAD=$(git rev-parse --short HEAD) gpurun -- 'GIT_HEAD='$GIT_HEAD' bash scripts/profile_all.sh r03'
# For every workload: one `rocprofv3 --kernel-trace --stats` pass and three separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE, TCC hit / miss; never
# combined with a trace domain) of the SAME bench command; for the GEMMs one SQ-counter pass of scripts/exp_gemm.py.  Raw output
# lands under gpurun_out/prof_<tag>_*; scripts/summarize_profile.py turns it into the small files committed under profiles/.
set -euo pipefail
TAG=${1:-r05}
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run_set() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof_${TAG}_${name}_stats" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings "$@" > "$OUT/prof_${TAG}_${name}_stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/prof_${TAG}_${name}_fetch" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings "$@" > "$OUT/prof_${TAG}_${name}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/prof_${TAG}_${name}_write" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings "$@" > "$OUT/prof_${TAG}_${name}_write.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d "$OUT/prof_${TAG}_${name}_tcc" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings "$@" > "$OUT/prof_${TAG}_${name}_tcc.log" 2>&1
  local nnz
  nnz=$(python3 -c "import json,sys; print([json.loads(l)['config']['nnz'] for l in open(sys.argv[1]) if l.startswith('{')][-1])" "$OUT/prof_${TAG}_${name}_stats.log")
  (cd "$ROOT" && WORKLOAD="${WL:-rmat10m_100m_f256}" PROFILED_NNZ="$nnz" PROFILE_CMD="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-order-control --no-cpp-api --no-ceilings $*" \
     python3 scripts/summarize_profile.py "${TAG}_bench_${name}" "$OUT/prof_${TAG}_${name}_stats" "$OUT/prof_${TAG}_${name}_fetch" "$OUT/prof_${TAG}_${name}_write" "$OUT/prof_${TAG}_${name}_tcc")
  echo "profiled $name"
}
# one rank of eight of the sharded step (exchange stubbed): per-rank compute of ranks 0 and 7 in both schedules -- the
# first run of the script only warms the box up -- and the kernel timeline of rank 0.  FIRST: after the counter passes below a box runs
# every kernel launch some 50 us slower for a while (the profiling power state lingers), which a step of ~25 launches shows
for i in 1 2; do python3 "$ROOT/scripts/exp_shard_compute.py" 8 --ranks 0,7 > "$OUT/prof_${TAG}_shard_warmup.log" 2>&1 || true; done   # (a fresh box launches slowly at first: the step's ~25 launches then show gaps of 30-40 us each -- `wait_*` entries of 0.03 instead of 0.005 ms say so)
python3 "$ROOT/scripts/exp_shard_compute.py" 8 --ranks 0,7 > "$OUT/prof_${TAG}_shard_overlap.log" 2>&1
python3 "$ROOT/scripts/exp_shard_compute.py" 8 --ranks 0,7 --schedule training > "$OUT/prof_${TAG}_shard_training.log" 2>&1
cat "$OUT/prof_${TAG}_shard_overlap.log" "$OUT/prof_${TAG}_shard_training.log" | grep '^{' > "$ROOT/profiles/${TAG}_shard_compute_per_rank.jsonl"
rocprofv3 --kernel-trace -f csv -d "$OUT/prof_${TAG}_rank0" -- python3 "$ROOT/scripts/exp_shard_compute.py" 8 --ranks 0 > "$OUT/prof_${TAG}_rank0.log" 2>&1
python3 "$ROOT/scripts/trace_to_text.py" "$OUT/prof_${TAG}_rank0" --last 19 \
  --header "rocprofv3 --kernel-trace -- python3 scripts/exp_shard_compute.py 8 --ranks 0   (rank 0 of 8 of RMAT 10M/100M F=256, overlap schedule, exchange stubbed; git ${GIT_HEAD:-?})" \
  > "$ROOT/profiles/${TAG}_shard_rank0_trace.txt"
echo "profiled one rank of eight"
if [ -z "${ONLY_SHARD:-}" ]; then   # (ONLY_SHARD=1: just the sharded rank above)
WL=rmat10m_100m_f256 run_set rmat10m
WL=rmat1m_10m_f128 run_set rmat1m_10m_f128 --workload rmat1m_10m_f128
WL=products_2p4m_62m_f100 run_set products_2p4m_62m_f100 --workload products_2p4m_62m_f100
# whole 2-layer training step on the headline graph: kernel stats only
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof_${TAG}_train2_stats" -- python3 "$ROOT/bench.py" --train-layers 2 --steps 3 --warmup 1 --no-cpu-baseline --no-ceilings > "$OUT/prof_${TAG}_train2_stats.log" 2>&1
(cd "$ROOT" && python3 scripts/summarize_profile.py "${TAG}_bench_rmat10m_train2" "$OUT/prof_${TAG}_train2_stats")
echo "profiled train2"
# GEMM SQ counters (MFMA busy, waits, LDS conflicts) on the three 10M x 256 x 256 products
TALL_ONLY=1 FS=256 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv \
  -d "$OUT/prof_${TAG}_gemm_sq" -- python3 "$ROOT/scripts/exp_gemm.py" > "$OUT/prof_${TAG}_gemm_sq.log" 2>&1
TALL_ONLY=1 FS=256 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof_${TAG}_gemm_stats" -- python3 "$ROOT/scripts/exp_gemm.py" > "$OUT/prof_${TAG}_gemm_stats.log" 2>&1
(cd "$ROOT" && python3 scripts/summarize_profile.py --sq "${TAG}_gemm_sq_counters" "$OUT/prof_${TAG}_gemm_sq" "$OUT/prof_${TAG}_gemm_stats")
echo "profiled gemm sq"
# the same for the 128-wide products (10 M x 128 x 128: the 256 x 128 / 128 x 128 LDS-DMA geometry and its k-major operand image)
TALL_ONLY=1 FS=128 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv \
  -d "$OUT/prof_${TAG}_gemm128_sq" -- python3 "$ROOT/scripts/exp_gemm.py" > "$OUT/prof_${TAG}_gemm128_sq.log" 2>&1
TALL_ONLY=1 FS=128 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof_${TAG}_gemm128_stats" -- python3 "$ROOT/scripts/exp_gemm.py" > "$OUT/prof_${TAG}_gemm128_stats.log" 2>&1
(cd "$ROOT" && python3 scripts/summarize_profile.py --sq "${TAG}_gemm128_sq_counters" "$OUT/prof_${TAG}_gemm128_sq" "$OUT/prof_${TAG}_gemm128_stats")
echo "profiled gemm128 sq"
fi
# nothing but gpurun_out/ travels back from the GPU box: leave a copy of the summaries there (the builder moves them into profiles/)
mkdir -p "$OUT/profiles_${TAG}" && cp "$ROOT"/profiles/${TAG}_* "$OUT/profiles_${TAG}/"
