#!/usr/bin/env python3
"""Tuning experiment (GPU box): the plan's hub path on the bench graph (RMAT 10M / 100M, F = 256, scrambled vertex order).
    CHUNKS=1024,4096 python scripts/exp_hub.py            # product library: sequential hub kernel
    GNNX_HIP_LIB=exp GNNX_SPMM_HUB=chunk python scripts/exp_hub.py   # measurement build: chunk + combine path (round 2)
    GNNX_HIP_LIB=exp GNNX_SPMM_HUB_LAS=4 python scripts/exp_hub.py   # half-size LDS ring
Prints forward / backward aggregation times per plan chunk (HIP events, mean of 10)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fn()
    a, b = capi.Event(), capi.Event()
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st)
    b.sync()
    return a.elapsed_ms(b) / reps


def main():
    n, e, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("E", 100_000_000)), int(os.environ.get("F", "256"))
    seed = int(os.environ.get("SEED", "2"))
    relabel = None if os.environ.get("ORDER") == "as-generated" else "scramble"
    src, dst = ops.rmat_edges(seed, n, e, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n, relabel=relabel)
    del src, dst
    H = ops.uniform_pm1(1, (n, F), device=dev)
    bias = torch.zeros(F, dtype=torch.float32, device=dev)
    out = torch.empty_like(H)
    tag = f"lib={os.environ.get('GNNX_HIP_LIB', 'product')} n={n} F={F} side={os.environ.get('GNNX_SPMM_SIDE', '1')}"
    for chunk in [int(c) for c in os.environ.get("CHUNKS", "0,1024,2048,4096,8192").split(",")]:
        if chunk:
            g.make_plans(chunk, F)
        else:
            g.plan = g.plan_t = None
        f = timed(lambda: ops.aggregate_fwd(g, H, bias, out=out))
        b = timed(lambda: ops.aggregate_bwd(g, H, out=out))
        ns = g.plan.n_split_rows if g.plan else 0
        print(f"{tag} chunk {chunk:6d} split rows {ns:6d}: fwd {f:7.3f} ms  bwd {b:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
