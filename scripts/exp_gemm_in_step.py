#!/usr/bin/env python3
"""Why are the dense products 4-7 % slower inside the bench step than in a loop of their own?  Times X.W^T (HIP events around the
one launch) after different predecessors on the stream."""
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
src, dst = ops.rmat_edges(0, n, e, 0.57, 0.19, 0.19, device=dev)
g = ops.CsrGraph.from_coo(src, dst, n)
del src, dst
g.make_plans(4096, F)
X = ops.uniform_pm1(1, (n, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
G = ops.uniform_pm1(3, (n, F), device=dev)
H = torch.empty((n, F), dtype=torch.float32, device=dev)
out = torch.empty((n, F), dtype=torch.float32, device=dev)
db = torch.empty(F, dtype=torch.float32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed_gemm(pre, reps=6, gap=False):
    ts = []
    for _ in range(reps):
        pre()
        if gap:
            torch.cuda.synchronize()
        a, b = capi.Event(), capi.Event()
        a.record(st)
        ops.linear_fwd(X, W, out=H)
        b.record(st)
        b.sync()
        ts.append(a.elapsed_ms(b))
    return np.median(ts), min(ts), max(ts)


print("after nothing (sync)        med %.3f min %.3f max %.3f" % timed_gemm(lambda: None, gap=True))
print("after X.W^T                 med %.3f min %.3f max %.3f" % timed_gemm(lambda: ops.linear_fwd(X, W, out=H)))
print("after colsum(G)             med %.3f min %.3f max %.3f" % timed_gemm(lambda: ops.colsum(G, out=db)))
print("after aggregate fwd         med %.3f min %.3f max %.3f" % timed_gemm(lambda: ops.aggregate_fwd(g, G, None, out=out)))
print("after aggregate fwd + sync  med %.3f min %.3f max %.3f" % timed_gemm(lambda: ops.aggregate_fwd(g, G, None, out=out), gap=True))
print("after dH^T.X                med %.3f min %.3f max %.3f" % timed_gemm(lambda: ops.gemm(G, X, transA=True)))
print("after 50 ms idle            med %.3f min %.3f max %.3f" % timed_gemm(lambda: (torch.cuda.synchronize(), __import__("time").sleep(0.05)), gap=True))
