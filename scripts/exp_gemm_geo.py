#!/usr/bin/env python3
"""Tuning experiment (GPU box, GNNX_HIP_LIB=exp): the row-parallel products under another tile geometry (GNNX_GEMM_GEO256 / GNNX_GEMM_GEO128,
read once per process) -- time and a checksum of the output's BITS, to be compared between processes."""
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


def bits(t):
    return int(t.view(torch.int32).to(torch.int64).sum().item())


def main():
    n = int(os.environ.get("N", 10_000_000))
    tag = f"geo256={os.environ.get('GNNX_GEMM_GEO256', '-')} geo128={os.environ.get('GNNX_GEMM_GEO128', '-')}"
    for F in [int(f) for f in os.environ.get("FS", "256,128").split(",")]:
        X = ops.uniform_pm1(1, (n, F), device=dev)
        W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
        out = torch.empty((n, F), dtype=torch.float32, device=dev)
        fl = 2.0 * n * F * F
        for name, fn in (("X.W^T (NT)", lambda: ops.gemm(X, W, transB=True, out=out)), ("dH.W (NN)", lambda: ops.gemm(X, W, out=out))):
            ms = timeit(fn)
            print(f"{tag} F={F:4d} {name:12s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 1.573:.1f}% of 157.3)  bits {bits(out)}",
                  flush=True)
        del X, out


main()
