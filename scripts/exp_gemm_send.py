#!/usr/bin/env python3
"""The sharded step's transform at one rank's size (1.25 M rows x 256 x 256, 1.79 M send rows to 7 peers):
product + pack pass against the product with the pack in its epilogue (gnnx_gemm_nt_rows_to_slots_f32), and the products' last
partial round on smaller tiles against the plain launch (EXPERIMENTS build: GNNX_GEMM_TAIL=0 in a second run).
usage (GPU box): python scripts/exp_gemm_send.py            one JSON line per shape
"""
import json
import os
import sys

import importlib

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
    return a.elapsed_time(b) / reps


SHAPES = [(1_250_000, 256, 7, 0.204), (1_000_000, 128, 7, 0.2), (10_000_000, 256, 0, 0.0), (10_000_000, 128, 0, 0.0)]
if os.environ.get("ONLY_SHAPE"):   # (for a profiler run)
    SHAPES = [SHAPES[int(os.environ["ONLY_SHAPE"])]]
for M, F, peers, frac in SHAPES:
    X = ops.uniform_pm1(1, (M, F), device=dev)
    W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
    H = torch.empty((M, F), dtype=torch.float32, device=dev)
    G = ops.uniform_pm1(3, (M, F), device=dev)
    dX = torch.empty((M, F), dtype=torch.float32, device=dev)
    dW = torch.empty((F, F), dtype=torch.float32, device=dev)
    rec = {"M": M, "F": F, "gemm_tail_env": os.environ.get("GNNX_GEMM_TAIL", ""), "lib": os.environ.get("GNNX_HIP_LIB", "")}
    rec["xwT_ms"] = round(timed(lambda: ops.linear_fwd(X, W, out=H)), 4)
    rec["dX_ms"] = round(timed(lambda: ops.gemm(G, W, out=dX)), 4)
    rec["dW_ms"] = round(timed(lambda: ops.gemm(G, X, transA=True, out=dW)), 4)
    flop = 2.0 * M * F * F
    rec["xwT_frac_of_157.3TF"] = round(flop / (rec["xwT_ms"] * 1e-3) / 157.3e12, 3)
    rec["dX_frac_of_157.3TF"] = round(flop / (rec["dX_ms"] * 1e-3) / 157.3e12, 3)
    rec["dW_frac_of_157.3TF"] = round(flop / (rec["dW_ms"] * 1e-3) / 157.3e12, 3)
    if peers:
        gen = torch.Generator(device="cpu").manual_seed(5)
        parts = [torch.sort(torch.randperm(M, generator=gen)[: int(M * frac)]).values for _ in range(peers)]
        send_idx = torch.cat(parts).to(torch.int32).to(dev)
        table = ops.slot_table(send_idx, M)
        send = torch.empty((send_idx.numel(), F), dtype=torch.float32, device=dev)
        rec["send_rows"] = int(send_idx.numel())
        rec["pack_pass_ms"] = round(timed(lambda: ops.rows_to_slots(H, table, send)), 4)
        rec["xwT_with_pack_in_epilogue_ms"] = round(timed(lambda: ops.linear_fwd_rows_to_slots(X, W, H, table, send)), 4)
        want = ops.gather_rows(ops.linear_fwd(X, W), send_idx)
        rec["same_bits"] = bool(torch.equal(send, want))
        del want, send, table, send_idx
    print(json.dumps(rec), flush=True)
    del X, W, H, G, dX, dW
    torch.cuda.empty_cache()
