#!/usr/bin/env python3
"""Tuning experiment (GPU box): the streaming BatchNorm passes of the full layer at 10 M x 256 (statistics, forward, backward sums + apply)."""
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
tag = f"reduce_blocks={os.environ.get('GNNX_BN_REDUCE_BLOCKS', '-')} apply_blocks={os.environ.get('GNNX_BN_APPLY_BLOCKS', '-')}"
X = ops.uniform_pm1(1, (n, F), device=dev)
dY = ops.uniform_pm1(2, (n, F), device=dev)
gamma, beta = ops.uniform_pm1(3, (F,), device=dev) + 1.5, ops.uniform_pm1(4, (F,), device=dev)
GB = n * F * 4 / 1e9
ms = timeit(lambda: ops.bn_stats(X))
mean, var = ops.bn_stats(X)
print(f"{tag}  stats (2 passes)  {ms:7.3f} ms  {2 * GB / ms:6.2f} TB/s")
out = torch.empty_like(X)
ms = timeit(lambda: ops.bn_relu_fwd(X, mean, var, gamma, beta, out=out))
print(f"{tag}  forward           {ms:7.3f} ms  {2 * GB / ms:6.2f} TB/s")
ms = timeit(lambda: ops.bn_relu_bwd(X, None, dY, mean, var, gamma, beta=beta))
print(f"{tag}  backward (sums + apply) {ms:7.3f} ms  {5 * GB / ms:6.2f} TB/s")
