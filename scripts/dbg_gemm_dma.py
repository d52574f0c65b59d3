#!/usr/bin/env python3
"""Debug harness for the LDS-DMA GEMM: one shape per process, operands inside padded allocations with sentinels."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
dev = torch.device("cuda:0")
M, N, K, nn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
PAD = 1 << 22  # floats of padding either side
torch.manual_seed(0)


def padded(rows, cols):
    buf = torch.full((PAD * 2 + rows * cols,), 7777.0, dtype=torch.float32, device=dev)
    return buf, buf[PAD: PAD + rows * cols].view(rows, cols)


xb, X = padded(M, K)
X.copy_(torch.rand((M, K), device=dev) * 2 - 1)
wb, W = padded(N, K) if not nn else padded(K, N)
W.copy_(torch.rand(W.shape, device=dev) * 2 - 1)
cb, Cm = padded(M, N)
torch.cuda.synchronize()
ops.gemm(X, W, transB=not nn, out=Cm)
torch.cuda.synchronize()
print("launched ok", flush=True)
assert bool((cb[:PAD] == 7777.0).all()) and bool((cb[PAD + M * N:] == 7777.0).all()), "C padding overwritten"
ref = X.double() @ (W.double() if nn else W.double().t())
err = (Cm.double() - ref).abs().max().item()
bad = ((Cm.double() - ref).abs() > 1e-4).nonzero()
print(f"M={M} N={N} K={K} nn={nn}: max err {err:.3e}; bad elements {bad.shape[0]}", flush=True)
if bad.shape[0]:
    print("first bad (row, col):", bad[:10].tolist(), "rows mod 256:", sorted(set((bad[:, 0] % 256).tolist()))[:40],
          "cols:", sorted(set(bad[:, 1].tolist()))[:40])
    sys.exit(1)
