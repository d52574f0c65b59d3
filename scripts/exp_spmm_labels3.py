#!/usr/bin/env python3
"""Degree-sorted vertex orders against the multiplicative relabelling (hub rows as one dense block)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
dev = torch.device("cuda:0")
n, e, F = 10_000_000, 100_000_000, 256
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H = ops.uniform_pm1(1, (n, F), device=dev)
out = torch.empty((n, F), dtype=torch.float32, device=dev)
src, dst = ops.rmat_edges(2, n, e, 0.57, 0.19, 0.19, device=dev)


def med(fn, reps=7):
    ts = []
    for _ in range(reps + 1):
        a, b = capi.Event(), capi.Event()
        a.record(st)
        fn()
        b.record(st)
        b.sync()
        ts.append(a.elapsed_ms(b))
    return float(np.median(ts[1:]))


ar = torch.arange(n, dtype=torch.int64, device=dev)
indeg = torch.bincount(dst.long(), minlength=n)
outdeg = torch.bincount(src.long(), minlength=n)


def rank_of(key):   # new id = position in the descending order of key
    order = torch.sort(key, descending=True, stable=True).indices
    nid = torch.empty(n, dtype=torch.int64, device=dev)
    nid[order] = ar
    return nid


mul = (ar * 2654435761) % n
hot = indeg >= 128
nhot = int(hot.sum())
# hubs first (dense block, in scrambled relative order), everything else scrambled behind them
key = torch.where(hot, 2 * n - mul, n - mul)
cands = {"mul 2654435761": mul, "in-degree descending": rank_of(indeg), "in+out degree descending": rank_of(indeg + outdeg),
         f"hubs (in-degree>=128: {nhot}) first, rest scrambled": rank_of(key)}
for name, nid in cands.items():
    g = ops.CsrGraph.from_coo(src, dst, n, relabel=nid.to(torch.int32))
    g.make_plans(4096, F)
    f = med(lambda: ops.aggregate_fwd(g, H, None, out=out))
    b = med(lambda: ops.aggregate_bwd(g, H, out=out))
    print(f"{name:58s} fwd {f:.2f}  bwd {b:.2f} ms", flush=True)
    del g
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
