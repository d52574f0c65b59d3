#!/usr/bin/env python3
"""128-wide products: gemm_w128_kernel (W resident in LDS, a 32-row strip per wavefront, no barrier in the loop) against
gemm_dma_kernel<4, 2> (EXPERIMENTS build, GNNX_GEMM_W128=1; default: the product kernel), same bits.
usage (GPU box): GNNX_HIP_LIB=exp [GNNX_GEMM_W128=1] python scripts/exp_gemm_w128.py"""
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


def timed(fn, reps=20, warm=5):
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


F = 128
for M in (1_000_000, 10_000_000, 1_000_003):
    X = ops.uniform_pm1(1, (M, F), device=dev)
    W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
    H = torch.empty((M, F), dtype=torch.float32, device=dev)
    dX = torch.empty((M, F), dtype=torch.float32, device=dev)
    timed(lambda: ops.linear_fwd(X, W, out=H), reps=10)   # clocks
    rec = {"w128_env": os.environ.get("GNNX_GEMM_W128", ""), "M": M}
    rec["xwT_ms"] = timed(lambda: ops.linear_fwd(X, W, out=H))
    rec["dX_ms"] = timed(lambda: ops.gemm(X, W, out=dX))
    flop = 2.0 * M * F * F
    rec["xwT_frac"] = round(flop / (rec["xwT_ms"] * 1e-3) / 157.3e12, 3)
    rec["dX_frac"] = round(flop / (rec["dX_ms"] * 1e-3) / 157.3e12, 3)
    idx = torch.cat([torch.arange(0, 3000), torch.arange(M - 3000, M), torch.randint(0, M, (5000,))]).to(dev)
    ref = X[idx].double() @ W.double().t()
    rec["max_err_vs_f64"] = float((H[idx].double() - ref).abs().max())
    rec["checksum"] = float(H.double().sum())
    rec["short_equal"] = bool(torch.equal(ops.gemm(X[M - 1000:].contiguous(), W, transB=True), H[M - 1000:]))
    print(json.dumps(rec), flush=True)
    del X, W, H, dX
