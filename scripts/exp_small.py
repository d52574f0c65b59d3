#!/usr/bin/env python3
"""Tuning experiment (GPU box): colsum / halo pack timings."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        fn()
    a, b = capi.Event(), capi.Event()
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st)
    b.sync()
    return a.elapsed_ms(b) / reps


for n, F in ((10_000_000, 256), (1_000_000, 128), (2_400_000, 100), (100_000, 16)):
    G = ops.uniform_pm1(1, (n, F), device=dev)
    out = torch.empty(F, dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.colsum(G, out=out))
    print(f"colsum N={n} F={F}: {ms:.3f} ms  {4.0 * n * F / ms / 1e6:.0f} GB/s", flush=True)
    idx = torch.randperm(n, device=dev)[: n // 2].int()
    o = torch.empty((n // 2, F), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.gather_rows(G, idx, out=o))
    print(f"gather_rows N/2 rows F={F}: {ms:.3f} ms  {8.0 * (n // 2) * F / ms / 1e6:.0f} GB/s (read+write)", flush=True)
    del G, o
