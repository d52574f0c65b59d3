#!/usr/bin/env python3
"""Experiment (GPU box): the opt-in split-bf16 GEMM vs the f32 MFMA GEMM -- accuracy against float64 and time."""
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


def check(M, N, K):
    X = ops.uniform_pm1(1, (M, K), device=dev)
    W = ops.uniform_pm1(2, (N, K), scale=K ** -0.5, device=dev)
    ref = X.double() @ W.double().t()
    for name, got in (("f32 mfma", ops.gemm(X, W, transB=True)), ("split NT", ops.gemm_split(X, W, transB=True)),
                      ("split NN", ops.gemm_split(X, W.t().contiguous(), transB=False))):
        err = (got.double() - ref).abs()
        print(f"  {M}x{N}x{K} {name:9s} max err {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e} "
              f"(max |ref| {ref.abs().max().item():.2f})", flush=True)


def main():
    for shp in ((300, 128, 64), (1000, 256, 256), (4097, 128, 128), (5000, 256, 1024)):
        check(*shp)
    n = int(os.environ.get("N", 10_000_000))
    for F in (256, 128):
        X = ops.uniform_pm1(1, (n, F), device=dev)
        W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
        out = torch.empty((n, F), dtype=torch.float32, device=dev)
        fl = 2.0 * n * F * F
        for name, fn in (("f32 mfma X.W^T", lambda: ops.gemm(X, W, transB=True, out=out)),
                         ("split    X.W^T", lambda: ops.gemm_split(X, W, transB=True, out=out)),
                         ("split    dH.W ", lambda: ops.gemm_split(X, W, transB=False, out=out))):
            ms = timeit(fn)
            print(f"F={F:4d} {name} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s f32-equivalent", flush=True)
        del X, out


main()
