#!/usr/bin/env python3
"""Tuning experiment (GPU box): does a non-power-of-two row stride of the GATHERED matrix remove the channel pile-up of the
as-generated vertex order (RMAT 10M / 100M, F = 256)?  Forward / backward aggregation with H / G stored on strides 256 .. 320,
as-generated labels, against the scrambled labels.      python scripts/exp_spmm_stride.py"""
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
    for label, relabel in (("as-generated", None), ("scrambled", "scramble")):
        g = ops.CsrGraph.from_coo(src, dst, n, relabel=relabel)
        g.make_plans(1024, F)
        out = torch.empty((n, F), dtype=torch.float32, device=dev)
        bias = torch.zeros(F, dtype=torch.float32, device=dev)
        for ld in (tuple(int(v) for v in os.environ.get("STRIDES", "256,264,272,288,320").split(",")) if relabel is None else (256,)):
            Hp = torch.empty((n, ld), dtype=torch.float32, device=dev)
            Hp[:, :F] = ops.uniform_pm1(1, (n, F), device=dev)
            H = Hp[:, :F]
            f = timed(lambda: ops.aggregate_fwd(g, H, bias, out=out))
            b = timed(lambda: ops.aggregate_bwd(g, H, out=out))
            print(f"{label:13s} gathered-row stride {ld:4d} floats ({ld * 4} B): fwd {f:7.3f} ms  bwd {b:7.3f} ms", flush=True)
            del Hp, H
        del g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
