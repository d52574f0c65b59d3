set -e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3_full -- $GRAFT_REPO_ROOT/tests/cpp/bench_host_api 10000000 100000000 256 5 0 1 2 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_r3_full.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r3_full -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n)[:90]
    print(f'{n:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms total {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
