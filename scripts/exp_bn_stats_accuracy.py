#!/usr/bin/env python3
"""GPU box: accuracy of the BatchNorm batch statistics at the bench size (10 M x 256) -- the exact two-pass kernels (gnnx_bn_stats_f32)
and the statistics from the transform's epilogue (gnnx_gemm_bn_stats_f32) against float64."""
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
n, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("F", 256))
for offset in (0.0, 3.0):
    X = ops.uniform_pm1(1, (n, F), device=dev) + offset
    W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
    H, m1, v1 = ops.linear_fwd_bn_stats(X, W)
    m2, v2 = ops.bn_stats(H)
    m64 = torch.zeros(F, dtype=torch.float64, device=dev)
    q64 = torch.zeros(F, dtype=torch.float64, device=dev)
    step = 1_000_000
    for r in range(0, n, step):
        m64 += H[r:r + step].double().sum(0)
    m64 /= n
    for r in range(0, n, step):
        q64 += ((H[r:r + step].double() - m64) ** 2).sum(0)
    v64 = q64 / n
    sd = v64.sqrt()
    for name, m, v in (("epilogue (one pass)", m1, v1), ("two-pass kernels", m2, v2)):
        em = ((m.double() - m64).abs() / sd).max().item()          # error of the mean in units of the column's standard deviation
        ev = ((v.double() - v64).abs() / v64).max().item()
        print(f"input offset {offset}: {name:20s} max |mean - mean64| / sd = {em:.3e}   max rel err of var = {ev:.3e}   (f32 eps = 5.96e-08)")
    del X, H
