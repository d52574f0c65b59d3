#!/usr/bin/env python3
"""Why does X . W^T (1.25 M x 256 x 256) run 9 % slower than dH . W through the very same kernel?  Operand placement A/B.
usage (GPU box): python scripts/exp_gemm_operands.py"""
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


def timed(fn, reps=20, warm=3):
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


M, F = int(os.environ.get("M", 1_250_000)), 256
X = ops.uniform_pm1(1, (M, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
Wn = W.t().contiguous()
H = torch.empty((M, F), dtype=torch.float32, device=dev)
G = ops.uniform_pm1(3, (M, F), device=dev)
dX = torch.empty((M, F), dtype=torch.float32, device=dev)
rec = {"M": M, "ptr_mod_1MiB": {k: hex(v.data_ptr() % (1 << 20)) for k, v in dict(X=X, H=H, G=G, dX=dX, W=W, Wn=Wn).items()},
       "ptr": {k: hex(v.data_ptr()) for k, v in dict(X=X, H=H, G=G, dX=dX).items()}}
rec["X.W^T -> H"] = timed(lambda: ops.linear_fwd(X, W, out=H))
rec["X.Wn  -> H"] = timed(lambda: ops.gemm(X, Wn, out=H))
rec["G.W   -> dX"] = timed(lambda: ops.gemm(G, W, out=dX))
rec["X.Wn  -> dX"] = timed(lambda: ops.gemm(X, Wn, out=dX))
rec["G.Wn  -> H"] = timed(lambda: ops.gemm(G, Wn, out=H))
rec["G.W^T -> dX"] = timed(lambda: ops.linear_fwd(G, W, out=dX))
rec["H.Wn  -> X (in-place roles swapped)"] = timed(lambda: ops.gemm(H, Wn, out=X))
print(json.dumps(rec), flush=True)
