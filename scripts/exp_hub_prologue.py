#!/usr/bin/env python3
"""Tuning experiment (GPU box): the forward aggregation with the BatchNorm + ReLU prologue on the bench graph, hub rows on
spmm_hub_kernel (default split) or ALL on the producer / consumer kernel (16- or, EXPERIMENTS build, 64-feature slabs).
    python scripts/exp_hub_prologue.py           GNNX_HIP_LIB=exp GNNX_PC_SLAB=64 python scripts/exp_hub_prologue.py"""
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


def timed(fn, reps=8):
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
    n, e, F, seed = 10_000_000, 100_000_000, 256, 2
    src, dst = ops.rmat_edges(seed, n, e, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n, relabel="scramble")
    del src, dst
    H = ops.uniform_pm1(1, (n, F), device=dev)
    bias = torch.zeros(F, dtype=torch.float32, device=dev)
    out = torch.empty_like(H)
    mean, var = ops.bn_stats(H)
    gamma = torch.ones(F, dtype=torch.float32, device=dev)
    beta = torch.zeros(F, dtype=torch.float32, device=dev)
    bn = (mean, var, gamma, beta, 1e-5)
    ref = None
    chunks = [int(c) for c in os.environ.get("CHUNKS", "1024").split(",")]   # hub threshold (rows above it leave the streaming kernel)
    for name, thr, chunk in [("default split", None, c) for c in chunks] + [("all hub rows on the pc kernel", 0, chunks[0])]:
        name = f"chunk {chunk} {name}"
        g.make_plans(chunk, F, big_rows=thr)
        plain = timed(lambda: ops.aggregate_fwd(g, H, bias, out=out))
        pro = timed(lambda: ops.aggregate_fwd(g, H, bias, out=out, bn=bn, relu_in=True))
        o = ops.aggregate_fwd(g, H, bias, bn=bn, relu_in=True)
        same = True if ref is None else torch.equal(o, ref)
        ref = o if ref is None else ref
        print(f"slab {os.environ.get('GNNX_PC_SLAB', '16')} {name}: plain {plain:.3f} ms  BatchNorm+ReLU prologue {pro:.3f} ms  same bits {same}", flush=True)


if __name__ == "__main__":
    main()
