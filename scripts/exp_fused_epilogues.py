#!/usr/bin/env python3
"""Timing of the GEMM epilogue fusions at the bench shape: fused call vs the separate passes it replaces."""
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


n, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("F", 256))
X = ops.uniform_pm1(1, (n, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
Wn = W.t().contiguous()
Y = torch.relu(ops.uniform_pm1(3, (n, F), device=dev))
H = torch.empty((n, F), dtype=torch.float32, device=dev)
db = torch.empty(F, dtype=torch.float32, device=dev)
print(f"N={n} F={F}")
print("X.W^T alone                  %.3f ms" % timeit(lambda: ops.linear_fwd(X, W, out=H)))
print("bn_stats(H) (exact two-pass) %.3f ms" % timeit(lambda: ops.bn_stats(H)))
print("X.W^T + stats fused (opt-in) %.3f ms" % timeit(lambda: ops.linear_fwd_bn_stats(X, W, out=H)))
print("dH.W alone                   %.3f ms" % timeit(lambda: ops.gemm(X, Wn, out=H)))
print("relu mask pass               %.3f ms" % timeit(lambda: ops.bn_relu_bwd(Y, Y, H, relu=True)))
print("colsum pass                  %.3f ms" % timeit(lambda: ops.colsum(H, out=db)))
print("dH.W + mask + colsum fused   %.3f ms" % timeit(lambda: ops.gemm_relu_colsum(X, Wn, Y, out=H, colsum_out=db)))
