#!/usr/bin/env python3
"""Achieved HBM rate of every streaming (non-GEMM, non-SpMM) entry point at the bench width: which ones are off their roofline?"""
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
n, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("F", 256))
X = ops.uniform_pm1(1, (n, F), device=dev)
Y = ops.uniform_pm1(2, (n, F), device=dev)
O = torch.empty((n, F), dtype=torch.float32, device=dev)
v = ops.uniform_pm1(3, (n,), device=dev)
vcol = v.reshape(n, 1).contiguous()
b = ops.uniform_pm1(4, (F,), device=dev)
idx = torch.randperm(n, device=dev)[: n // 5].to(torch.int32)
packed = torch.empty((n // 5, F), dtype=torch.float32, device=dev)
mean, var = ops.bn_stats(X)
tgt = (torch.arange(n, device=dev, dtype=torch.int64) * 7 + 3).remainder(F).to(torch.int32)
Xb = torch.empty((n, F), dtype=torch.bfloat16, device=dev)
GB = n * F * 4 / 1e9
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(name, fn, gbytes, reps=4):
    fn()
    a, e = capi.Event(), capi.Event()
    a.record(st)
    for _ in range(reps):
        fn()
    e.record(st)
    e.sync()
    ms = a.elapsed_ms(e) / reps
    print(f"{name:34s} {ms:8.3f} ms  {gbytes / ms:7.2f} TB/s  ({gbytes:.1f} GB)", flush=True)


t("colsum", lambda: ops.colsum(X), GB)
t("rowsum", lambda: ops.rowsum(X), GB)
t("rowscale", lambda: ops.rowscale(X, v, out=O), 2 * GB)
t("bias_add", lambda: ops.bias_add(X, b, out=O), 2 * GB)
t("binary add [N,F]+[N,F]", lambda: ops.binary("add", X, Y, out=O), 3 * GB)
t("binary mul [N,F]*[N,1]", lambda: ops.binary("mul", X, vcol, out=O), 2 * GB)
t("binary add [N,F]+[F]", lambda: ops.binary("add", X, b, out=O), 2 * GB)
t("binary div [N,F]/[F]", lambda: ops.binary("div", X, b, out=O), 2 * GB)
t("axpy", lambda: ops.axpy(0.5, X, O), 3 * GB)
t("gather_rows (N/5 rows)", lambda: ops.gather_rows(X, idx, out=packed), 2 * GB / 5)
t("bn_stats (two passes)", lambda: ops.bn_stats(X), 2 * GB)
t("bn_relu_fwd", lambda: ops.bn_relu_fwd(X, mean, var, b, b, relu=True, out=O), 2 * GB)
t("bn_relu_bwd (sums + apply)", lambda: ops.bn_relu_bwd(X, None, Y, mean, var, b, relu=True, beta=b), 5 * GB)
t("relu mask (bn_relu_bwd no stats)", lambda: ops.bn_relu_bwd(X, X, Y, relu=True), 4 * GB)
t("to_bf16", lambda: ops.to_bf16(X, out=Xb), 1.5 * GB)
t("softmax_ce + colsum", lambda: ops.softmax_ce(X, tgt, colsum_out=b.clone()), 2 * GB)
t("sgd_step (N*F)", lambda: ops.sgd_step(O, X, 1e-3), 3 * GB)
for name in ("transpose", "pow", "fill"):
    if hasattr(ops, name):
        print("has", name)
