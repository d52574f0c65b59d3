#!/usr/bin/env python3
"""Does the row-parallel dense product care about the leading dimension of A / C (power-of-two row stride vs padded)?"""
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
n, F = 10_000_000, 256
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
Wn = W.t().contiguous()


def t(fn, reps=6):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        a, b = capi.Event(), capi.Event()
        a.record(st)
        fn()
        b.record(st)
        b.sync()
        ts.append(a.elapsed_ms(b))
    return float(np.median(ts))


for lda, ldc in ((256, 256), (288, 256), (256, 288), (288, 288), (272, 272), (320, 320), (264, 264)):
    A = torch.zeros((n, lda), dtype=torch.float32, device=dev)
    A[:, :F] = ops.uniform_pm1(1, (n, F), device=dev)
    Cb = torch.zeros((n, ldc), dtype=torch.float32, device=dev)
    Av, Cv = A[:, :F], Cb[:, :F]
    nt = t(lambda: ops.gemm(Av, W, transB=True, out=Cv))
    nn = t(lambda: ops.gemm(Av, Wn, out=Cv))
    dW = torch.empty((F, F), dtype=torch.float32, device=dev)
    tn = t(lambda: ops.gemm(Av, Cv, transA=True, out=dW))
    print(f"lda {lda} ldc {ldc}:  X.W^T {nt:.3f}  dH.W {nn:.3f}  dH^T.X {tn:.3f} ms", flush=True)
    del A, Cb, Av, Cv
    torch.cuda.empty_cache()
