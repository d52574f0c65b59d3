#!/usr/bin/env python3
"""GPU box: the in-run ceilings of bench.py (bench_kernels/ceilings.hip) on their own, for several table sizes / degrees."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
from __graft_entry__ import load_package  # noqa: E402
load_package()
import importlib  # noqa: E402
capi = importlib.import_module("gnncpp_amd.capi")
for v in range(6):
    print("copy variant", v, round(bench.copy_ceiling(capi, dev, variant=v), 1), flush=True)
print("torch copy_", end=" ")
src = torch.full((2 ** 30,), 1.0, device=dev)
dst = torch.empty_like(src)
print(round(2 * src.numel() * 4 / (bench._timed_ms(capi, lambda st: dst.copy_(src) is None and 0) * 1e-3) / 1e9, 1), flush=True)
del src, dst
for mb in (38, 160, 8192):
    for v in range(5):
        print("gather table_mb", mb, "variant", v, round(bench.gather_ceiling(capi, dev, table_mb=mb, variant=v), 1), flush=True)
