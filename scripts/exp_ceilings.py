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
print(bench.measure_ceilings(capi, dev, verbose=True))
for mb in (38, 160, 640, 8192):
    for deg in (8, 16):
        print(mb, deg, bench.gather_ceiling(capi, dev, table_mb=mb, deg=deg))
