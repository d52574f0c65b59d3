#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> a small text timeline (start / end / duration / kernel / grid), the last `--last` ms of the run.
usage: trace_to_text.py DIR_OR_CSV [--last MS] [--header TEXT]"""
import argparse
import csv
import glob
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("path")
ap.add_argument("--last", type=float, default=20.0)
ap.add_argument("--header", default="")
args = ap.parse_args()
f = args.path if args.path.endswith(".csv") else sorted(glob.glob(os.path.join(args.path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
keep = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - args.last * 1e6]
t0 = int(keep[0]["Start_Timestamp"])
if args.header:
    print(args.header)
print("start_ms end_ms dur_ms kernel grid")
for r in keep:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"\(.*$", "", name)[:70]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:8.3f} {e:8.3f} {e - s:8.3f}  {name}  grid={r['Grid_Size_X']}")
