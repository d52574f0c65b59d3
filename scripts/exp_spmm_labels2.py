#!/usr/bin/env python3
"""Multiplicative relabelling vs random permutations vs other multipliers: is there anything left between them?"""
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
cands = {"mul 2654435761": (ar * 2654435761) % n, "mul 40503 (16-bit golden)": (ar * 40503) % n,
         "mul 7919": (ar * 7919) % n, "mul 1000003": (ar * 1000003) % n}
for sd in (1, 2, 3):
    g_ = torch.Generator(device=dev)
    g_.manual_seed(sd)
    cands[f"random perm {sd}"] = torch.randperm(n, device=dev, generator=g_)
for name, nid in cands.items():
    assert int(torch.unique(nid).numel()) == n, name
    g = ops.CsrGraph.from_coo(src, dst, n, relabel=nid.to(torch.int32))
    g.make_plans(4096, F)
    f = med(lambda: ops.aggregate_fwd(g, H, None, out=out))
    b = med(lambda: ops.aggregate_bwd(g, H, out=out))
    print(f"{name:28s} fwd {f:.2f}  bwd {b:.2f} ms", flush=True)
    del g
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
