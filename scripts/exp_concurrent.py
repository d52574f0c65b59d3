#!/usr/bin/env python3
"""Do an HBM-bound aggregation and an MFMA-bound dense product overlap when they are launched on two streams?
Bench graph (10M / 100M, F = 256): X.W^T on stream A beside the backward aggregation on stream B, against the two run in sequence.
Also the same with the dense product's persistent grid restricted to fewer CUs (GNNX_GEMM_DMA_GRID, experiment builds only)."""
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
n, e, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("E", 100_000_000)), int(os.environ.get("F", 256))
src, dst = ops.rmat_edges(0, n, e, 0.57, 0.19, 0.19, device=dev)
g = ops.CsrGraph.from_coo(src, dst, n)
del src, dst
g.make_plans(int(os.environ.get("CHUNK", 4096)), F)
X = ops.uniform_pm1(1, (n, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
G = ops.uniform_pm1(3, (n, F), device=dev)
H = torch.empty((n, F), dtype=torch.float32, device=dev)
dH = torch.empty((n, F), dtype=torch.float32, device=dev)
dW = torch.empty((F, F), dtype=torch.float32, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def gemm():
    ops.linear_fwd(X, W, out=H)


def gemm_tn():
    ops.gemm(G, X, transA=True, out=dW)


def spmm():
    ops.aggregate_bwd(g, G, out=dH)


def both(first, second):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        first()
    with torch.cuda.stream(sb):
        second()
    cur.wait_stream(sa)
    cur.wait_stream(sb)


print("X.W^T alone        %.2f ms" % wall(gemm))
print("dH^T.X alone       %.2f ms" % wall(gemm_tn))
print("aggregate^T alone  %.2f ms" % wall(spmm))
print("X.W^T then agg^T, one stream   %.2f ms" % wall(lambda: (gemm(), spmm())))
print("X.W^T || agg^T (gemm first)    %.2f ms" % wall(lambda: both(gemm, spmm)))
print("agg^T || X.W^T (spmm first)    %.2f ms" % wall(lambda: both(spmm, gemm)))
print("dH^T.X || agg^T (gemm first)   %.2f ms" % wall(lambda: both(gemm_tn, spmm)))
print("agg^T || dH^T.X (spmm first)   %.2f ms" % wall(lambda: both(spmm, gemm_tn)))
