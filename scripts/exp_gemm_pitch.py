#!/usr/bin/env python3
"""The three products of the headline step in the bench's own layout (10 M x 256 x 256; H and dH on the gather pitch), one JSON line.
A/B of two library builds on ONE box: run once per build (GNNX_HIP_LIB=exp picks libgnnx_hip_exp.so).
usage (GPU box): python scripts/exp_gemm_pitch.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
dev = torch.device("cuda:0")


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


M, F = int(os.environ.get("M", 10_000_000)), int(os.environ.get("F", 256))
X = ops.uniform_pm1(1, (M, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
H = ops.empty_gathered(M, F, device=dev)
dH = ops.empty_gathered(M, F, device=dev)
dH.copy_(ops.uniform_pm1(3, (M, F), device=dev))
dX = torch.empty((M, F), dtype=torch.float32, device=dev)
dW = torch.empty((F, F), dtype=torch.float32, device=dev)
rec = {"lib": os.environ.get("GNNX_HIP_LIB", "product"), "M": M, "F": F, "ld_H": H.stride(0)}
timed(lambda: ops.linear_fwd(X, W, out=H), reps=5)   # clocks up
for rnd in range(2):
    rec[f"xwT_{rnd}"] = timed(lambda: ops.linear_fwd(X, W, out=H))
    rec[f"dX_{rnd}"] = timed(lambda: ops.gemm(dH, W, out=dX))
    rec[f"dW_{rnd}"] = timed(lambda: ops.gemm(dH, X, transA=True, out=dW))
print(json.dumps(rec), flush=True)
