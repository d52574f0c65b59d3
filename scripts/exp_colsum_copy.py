#!/usr/bin/env python3
"""Tuning experiment (GPU box): column sums with and without the copy onto the gather pitch (gnnx_colsum_copy_f32), 10 M x 256."""
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
G = ops.uniform_pm1(1, (n, F), device=dev)
out = torch.empty(F, dtype=torch.float32, device=dev)
ms = timeit(lambda: ops.colsum(G, out=out))
print(f"colsum                {ms:7.3f} ms  {n * F * 4 / ms / 1e6:7.1f} GB/s read")
ld = ops.gather_row_stride(n, F)
spacer = torch.empty(int(os.environ.get("SPACER_MB", 0)) << 20, dtype=torch.uint8, device=dev) if os.environ.get("SPACER_MB") else None   # another placement of the copy
Gp = ops.empty_gathered(n, F, device=dev)
print(f"G at {G.data_ptr():#x}, copy at {Gp.data_ptr():#x}")
ms = timeit(lambda: ops.colsum_copy(G, Gp, out=out))
print(f"colsum + copy (ld {ld}) {ms:7.3f} ms  {2 * n * F * 4 / ms / 1e6:7.1f} GB/s read + written   NT={os.environ.get('GNNX_COLSUM_NT', '0')}")
assert torch.equal(Gp, G)
