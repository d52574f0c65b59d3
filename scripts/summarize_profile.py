#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python scripts/summarize_profile.py <tag> <stats_dir> [<fetch_dir> <write_dir> [<tcc_dir>]]

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary, names shortened) and, when the two
--pmc passes are given, profiles/<tag>_hbm_traffic.json with per-kernel per-launch HBM bytes:
    traffic = 2 * FETCH_SIZE + WRITE_SIZE   (KiB counters -> bytes)
The factor 2 is the gfx950 correction of MI355X_MICROARCH.md section HBM (FETCH_SIZE tallies 128-B requests at
64 B); it is calibrated in the same run on colsum_stage1, which reads its [N,F] input exactly once
(expected 4*N*F bytes).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def file_sha256(path):
    import hashlib
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:110]


def one(pattern):
    # rocprofv3 nests its output one directory deep (<dir>/<host>/<pid>_*.csv) unless -o names the files
    f = glob.glob(pattern) or glob.glob(pattern.replace(os.sep + "*" + os.sep, os.sep, 1))
    if not f:
        sys.exit(f"no file matches {pattern}")
    return f[0]


def counters(d, counter, scale=1024.0):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(os.path.join(d, "*", "*_counter_collection.csv")))):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * scale for k, v in acc.items()}  # (KiB -> bytes,) mean per launch


def sq_summary(tag, sq_dir, stats_dir):
    """profiles/<tag>.json: mean SQ counters per GEMM kernel + mean duration from the stats pass of the same command."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(one(os.path.join(sq_dir, "*", "*_counter_collection.csv")))):
        k = short(r["Kernel_Name"])
        if re.search(r"gemm|splitk|mfma", k):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {short(r["Name"]): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(one(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))))}
    res = {}
    for k, cs in acc.items():
        row = {c: sum(v) / len(v) for c, v in cs.items()}
        row["mean_duration_ms"] = dur.get(k)
        if row.get("mean_duration_ms") and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            # 1024 SIMDs; clock taken as 2.4 GHz (the GEMMs hold 2.33-2.41 GHz: DESIGN.md section 4.2)
            row["mfma_pipe_busy_frac_at_2p4GHz"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["mean_duration_ms"] * 1e-3 * 2.4e9 * 1024)
        res[k] = row
    out = os.path.join(ROOT, "profiles", f"{tag}.json")
    json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                          "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -- python3 scripts/exp_gemm.py (FS=256: 10M x 256 x 256 products + one 4096^3); "
                          "durations from a separate --kernel-trace --stats pass of the same command",
               "note": "means over the dispatches of each kernel; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, "
                       "SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)", "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out)


def main():
    if sys.argv[1] == "--sq":
        return sq_summary(sys.argv[2], sys.argv[3], sys.argv[4])
    tag, stats_dir = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    rows = list(csv.DictReader(open(one(os.path.join(stats_dir, "*", "*_kernel_stats.csv")))))
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "pct", "min_ms", "max_ms"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e6:.4f}",
                        r["Percentage"], f"{float(r['MinNs']) / 1e6:.4f}", f"{float(r['MaxNs']) / 1e6:.4f}"])
    print("wrote", out)
    if len(sys.argv) >= 5:
        fetch, write = counters(sys.argv[3], "FETCH_SIZE"), counters(sys.argv[4], "WRITE_SIZE")
        tcc_hit = counters(sys.argv[5], "TCC_HIT_sum", 1.0) if len(sys.argv) >= 6 else {}
        tcc_miss = counters(sys.argv[5], "TCC_MISS_sum", 1.0) if len(sys.argv) >= 6 else {}
        res = {}
        for k in sorted(set(fetch) | set(write)):
            if not re.search(r"spmm|gemm_kernel|gemm_stream_kernel|gemm_dma|colsum_stage1|splitk|rows_kernel", k):
                continue
            fb, wb = fetch.get(k, 0.0), write.get(k, 0.0)
            res[k] = {"fetch_size_bytes_raw": fb, "write_size_bytes": wb, "hbm_bytes_corrected": 2 * fb + wb}
            if k in tcc_hit or k in tcc_miss:   # L2 (TCC) hit rate of the kernel: requests served by the XCD L2s / all requests
                h, m = tcc_hit.get(k, 0.0), tcc_miss.get(k, 0.0)
                res[k].update({"tcc_hit": h, "tcc_miss": m, "l2_hit_rate": h / (h + m) if h + m > 0 else None})
        out = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")
        json.dump({"workload": os.environ.get("WORKLOAD", "rmat10m_100m_f256"),
                   # provenance bench.py prints beside `traffic` so that a stale file shows: the commit the profiled build was made
                   # from (the GPU box has no .git: the caller passes it) and the graph's non-zeros as the profiled bench reported them
                   "git_head": os.environ.get("GIT_HEAD"), "spmm_source_sha256": file_sha256(os.path.join(ROOT, "gnn.cpp_amd", "csrc", "gnnx_spmm.hip")),
                   "nnz": int(os.environ["PROFILED_NNZ"]) if os.environ.get("PROFILED_NNZ") else None,
                   "command": os.environ.get("PROFILE_CMD", "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"),
                   "note": "per launch; hbm_bytes_corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction, "
                           "MI355X_MICROARCH.md section HBM; calibrate on colsum_stage1 = 4*N*F bytes read once); "
                           "l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) from a third pass (MI355X_MICROARCH.md section L2)",
                   "kernels": res}, open(out, "w"), indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
