#!/usr/bin/env python3
"""Tuning experiment (GPU box): forward-SpMM time on uniform-degree vs RMAT graphs, plan chunk sweep."""
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


def time_spmm(g, H, out, plan, reps=5):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        ops.spmm(g.rowptr, g.colidx, H, out=out, rowscale=g.norm, plan=plan)
    a, b = capi.Event(), capi.Event()
    a.record(st)
    for _ in range(reps):
        ops.spmm(g.rowptr, g.colidx, H, out=out, rowscale=g.norm, plan=plan)
    b.record(st)
    b.sync()
    return a.elapsed_ms(b) / reps


def report(name, g, F, ms):
    B = 4 * (g.n + 1) + 4 * g.nnz + 4 * F * g.nnz + 4 * g.n + 4 * F * g.n
    print(f"{name:40s} nnz={g.nnz:>10d} {ms:8.3f} ms  {B / ms / 1e6:8.1f} GB/s  ({B / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)


def main():
    n, e, F = 10_000_000, 100_000_000, int(os.environ.get("F", "256"))
    H = ops.uniform_pm1(1, (n, F), device=dev)
    out = torch.empty_like(H)
    which = sys.argv[1:] or ["uniform", "rmat"]
    if "uniform" in which:
        src = torch.arange(n, dtype=torch.int32, device=dev).repeat_interleave(10)
        dst = torch.randint(0, n, (e,), dtype=torch.int32, device=dev)
        g = ops.CsrGraph.from_coo(src, dst, n, transpose=False)
        del src, dst
        report("uniform deg 10, random cols", g, F, time_spmm(g, H, out, None))
        del g
    if "rmat" in which:
        src, dst = ops.rmat_edges(2, n, e, device=dev)
        g = ops.CsrGraph.from_coo(src, dst, n, transpose=False)
        del src, dst
        deg = (g.rowptr[1:] - g.rowptr[:-1]).cpu()
        print("rmat degree: zero rows %.1f%%, deg<=2 %.1f%%, max %d, rows>1024: %d (nnz share %.1f%%), rows>256: %d (%.1f%%)" % (
            100 * (deg == 0).float().mean(), 100 * (deg <= 2).float().mean(), int(deg.max()),
            int((deg > 1024).sum()), 100 * deg[deg > 1024].sum() / deg.sum(), int((deg > 256).sum()),
            100 * deg[deg > 256].sum() / deg.sum()))
        report("rmat, no plan", g, F, time_spmm(g, H, out, None))
        for chunk in [int(c) for c in os.environ.get("CHUNKS", "32,64,128,256,512").split(",")]:
            plan = ops.SpmmPlan(g.rowptr, chunk, F)
            report(f"rmat, plan chunk {chunk} ({plan.n_split_rows} hub rows / {plan.n_hub_nnz} nnz)", g, F,
                   time_spmm(g, H, out, plan))
            del plan
        if os.environ.get("HUBSPLIT"):
            thr = int(os.environ["HUBSPLIT"])
            rp = g.rowptr.long()
            degl = rp[1:] - rp[:-1]
            rows = torch.repeat_interleave(torch.arange(n, device=dev), degl)
            keep_h = degl[rows] > thr
            for nm, keep in (("hub rows only", keep_h), ("non-hub rows only", ~keep_h)):
                r2 = rows[keep].int()
                c2 = g.colidx[keep]
                g2 = ops.CsrGraph.from_coo(r2, c2, n, transpose=False)
                for chunk in (0, 256, 1024):
                    plan = ops.SpmmPlan(g2.rowptr, chunk, F) if chunk else None
                    ms = time_spmm(g2, H, out, plan)
                    print(f"  {nm:20s} thr {thr} chunk {chunk:5d}: nnz {g2.nnz:>10d} {ms:8.3f} ms  {(4*F*g2.nnz)/ms/1e6:8.1f} GB/s gathered", flush=True)
                del g2


main()
