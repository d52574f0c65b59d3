#!/usr/bin/env python3
"""Tuning experiment (GPU box): ONE hub row of known length -- ns per neighbour of the two hub kernels, alone and beside a
streaming load (a second graph's aggregation on another stream).
    python scripts/exp_hub_row.py [F ...]"""
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
    Fs = [int(a) for a in sys.argv[1:]] or [256, 128]
    n = int(os.environ.get("N", 4_000_000))
    for F in Fs:
        X = ops.uniform_pm1(1, (n, F), device=dev)
        for L in (20_000, 250_000) if n > 250_000 else (n // 50, n // 2):
            # row 0 has L random neighbours, every other row is empty
            cols = torch.randperm(n - 1, device=dev)[:L].to(torch.int32) + 1
            src = torch.zeros(L, dtype=torch.int32, device=dev)
            g = ops.CsrGraph.from_coo(src, cols, n, transpose=False, norm=False)
            out = torch.zeros((n, F), dtype=torch.float32, device=dev)
            ref = ops.spmm(g.rowptr, g.colidx, X, out=out.clone())
            res = {}
            for name, thr in (("hub", 2_000_000_000), ("hubpc", 0)):
                g.make_plans(1024, F, big_rows=thr)
                o = ops.spmm(g.rowptr, g.colidx, X, out=out.clone(), plan=g.plan)
                assert os.environ.get("GNNX_PC_EXP") or torch.equal(o[0], ref[0]), name
                ms = timed(lambda: ops.spmm(g.rowptr, g.colidx, X, out=out, plan=g.plan))
                res[name] = ms
            print(f"n={n} F={F} L={L}: hub {res['hub']:.3f} ms  hubpc {res['hubpc']:.3f} ms (both include the streaming kernel over {n} empty rows)", flush=True)


if __name__ == "__main__":
    main()
