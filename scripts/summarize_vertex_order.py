#!/usr/bin/env python3
"""profiles/r02_vertex_order_counters.json from the raw --pmc output of scripts/profile_vertex_order.sh: per launch of the forward
aggregation (spmm_stream_kernel<64, 4, 8, 0, ...>), in the two vertex orders."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for order in ("scrambled", "as-generated"):
    acc = defaultdict(list)
    for part in "abc":
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_vo_{order}_{part}", "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "spmm_stream_kernel<64, 4, 8, 0" in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    d = {k: sum(v) / len(v) for k, v in acc.items()}
    if "TCC_EA0_RDREQ_sum" in d and "TCC_EA0_RDREQ_LEVEL_sum" in d:
        d["avg_ea_read_latency_cycles"] = d["TCC_EA0_RDREQ_LEVEL_sum"] / d["TCC_EA0_RDREQ_sum"]
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    out[order] = d
json.dump({"kernel": "spmm_stream_kernel<64, 4, 8, 0, 64, float> (forward aggregation), RMAT 10M/100M F=256, per launch",
           "command": "rocprofv3 --pmc <3 counters per pass> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-order-control --vertex-order <order>",
           "orders": out}, open(os.path.join(ROOT, "profiles", "r02_vertex_order_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
