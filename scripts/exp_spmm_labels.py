#!/usr/bin/env python3
"""R-MAT hubs are the vertex ids with few one-bits, i.e. feature-row addresses with few one-bits.  Is the aggregation's time
sensitive to that?  Same graph with its vertex labels permuted at random (an isomorphic graph: same degrees, same nnz)."""
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


def time_spmm(g, fn, reps=5):
    ts = []
    for _ in range(reps + 1):
        a, b = capi.Event(), capi.Event()
        a.record(st)
        fn(g)
        b.record(st)
        b.sync()
        ts.append(a.elapsed_ms(b))
    return np.median(ts[1:])


for mode in ("as generated", "random labels", "random labels 2", "bit-reversed labels"):
    src, dst = ops.rmat_edges(0, n, e, 0.57, 0.19, 0.19, device=dev)
    if mode.startswith("random"):
        gen = torch.Generator(device=dev)
        gen.manual_seed(len(mode))
        perm = torch.randperm(n, device=dev, generator=gen).to(torch.int32)
        src, dst = perm[src.long()], perm[dst.long()]
    elif mode.startswith("bit"):
        def mix(v):
            v = v.long()
            return ((v * 2654435761) % n).to(torch.int32)     # a multiplicative scramble (bijective: gcd(2654435761, 10^7) = 1)
        src, dst = mix(src), mix(dst)
    g = ops.CsrGraph.from_coo(src, dst, n)
    del src, dst
    g.make_plans(4096, F)
    f = time_spmm(g, lambda g_: ops.aggregate_fwd(g_, H, None, out=out))
    b = time_spmm(g, lambda g_: ops.aggregate_bwd(g_, H, out=out))
    print(f"{mode:22s} nnz {g.nnz}  fwd {f:.2f} ms  bwd {b:.2f} ms", flush=True)
    del g
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
