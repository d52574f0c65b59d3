#!/usr/bin/env python3
"""Experiment (GPU box): does tiling the feature dimension (smaller gathered rows -> more of them per L2) pay on the 10M/100M graph?"""
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


def timeit(fn, reps=5):
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


def main():
    n, e, F = 10_000_000, 100_000_000, 256
    src, dst = ops.rmat_edges(2, n, e, 0.57, 0.19, 0.19, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    del src, dst
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
    g.make_plans(4096, F)
    H = ops.uniform_pm1(1, (n, F), device=dev)
    out = torch.empty_like(H)
    bias = torch.zeros(F, dtype=torch.float32, device=dev)
    print(f"one pass F=256: {timeit(lambda: ops.aggregate_fwd(g, H, bias, out=out)):.3f} ms", flush=True)
    for tile in (128, 64):
        def run():
            for f0 in range(0, F, tile):
                ops.aggregate_fwd(g, H[:, f0:f0 + tile], bias[f0:f0 + tile], out=out[:, f0:f0 + tile])
        print(f"{F // tile} passes of {tile} features: {timeit(run):.3f} ms", flush=True)


main()
