#!/usr/bin/env python3
"""Tuning experiment (GPU box): the three GEMMs of the path at the bench shape + a square reference shape."""
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
    n = int(os.environ.get("N", 10_000_000))
    for F in [int(f) for f in os.environ.get("FS", "256,128").split(",")]:
        X = ops.uniform_pm1(1, (n, F), device=dev)
        W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
        out = torch.empty((n, F), dtype=torch.float32, device=dev)
        dW = torch.empty((F, F), dtype=torch.float32, device=dev)
        fl = 2.0 * n * F * F
        for name, fn in (("X.W^T (NT)", lambda: ops.gemm(X, W, transB=True, out=out)),
                         ("dH.W (NN)", lambda: ops.gemm(X, W, out=out)),
                         ("dH^T.X (TN)", lambda: ops.gemm(X, out, transA=True, out=dW))):
            ms = timeit(fn)
            print(f"F={F:4d} {name:12s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / ms / 1e9 / 1.573:.1f}% of 157.3)", flush=True)
        del X, out
    if os.environ.get("TALL_ONLY"):   # profile passes: only the bench's shapes, so that per-kernel means are not a mix
        return
    s = 4096
    A = ops.uniform_pm1(3, (s, s), device=dev)
    B = ops.uniform_pm1(4, (s, s), device=dev)
    Cc = torch.empty((s, s), dtype=torch.float32, device=dev)
    ms = timeit(lambda: ops.gemm(A, B, transB=True, out=Cc))
    print(f"4096^3 NT      {ms:8.3f} ms  {2.0 * s ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    ref = (A[:64].double() @ B.double().T)
    err = float((Cc[:64].double() - ref).abs().max() / ref.abs().max())
    print("rel err vs f64:", err)


main()


def mfma_peak():
    """GNNX_HIP_LIB=exp only: gnnx_mfma_peak_f32 is a measurement entry of the EXPERIMENTS build, not of include/gnnx.h."""
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    fl = C.c_double(0)
    fn = capi.lib().gnnx_mfma_peak_f32
    fn.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
    for wgs in (256, 512, 1024, 2048):
        fn(20000, wgs, C.c_void_p(sink.data_ptr()), C.byref(fl), None)
        ms = timeit(lambda: fn(20000, wgs, C.c_void_p(sink.data_ptr()), C.byref(fl), None), reps=3)
        print(f"mfma peak loop, {wgs} workgroups x 4 waves: {fl.value / ms / 1e9:.1f} TFLOP/s", flush=True)


if os.environ.get("PEAK"):
    mfma_peak()
