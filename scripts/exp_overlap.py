#!/usr/bin/env python3
"""Experiment (GPU box): does an HBM-bound SpMM co-run with an MFMA-bound GEMM on two HIP streams?"""
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


def main():
    n = int(os.environ.get("N", 10_000_000))
    e = n * 10
    F = int(os.environ.get("F", 256))
    src, dst = ops.rmat_edges(2, n, e, 0.57, 0.19, 0.19, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    del src, dst
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
    g.make_plans(4096, F)
    X = ops.uniform_pm1(1, (n, F), device=dev)
    W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
    G = ops.uniform_pm1(3, (n, F), device=dev)
    dH = torch.empty((n, F), dtype=torch.float32, device=dev)
    dH2 = torch.empty((n, F), dtype=torch.float32, device=dev)
    dX = torch.empty((n, F), dtype=torch.float32, device=dev)
    dW = torch.empty((F, F), dtype=torch.float32, device=dev)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def spmm():
        ops.aggregate_bwd(g, G, out=dH2)

    def gemms():
        ops.gemm(dH, W, out=dX)
        ops.gemm(dH, X, transA=True, out=dW)

    def wall(fn, reps=4):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def serial():
        spmm()
        gemms()

    def concurrent():
        cur = torch.cuda.current_stream()
        sA.wait_stream(cur)
        sB.wait_stream(cur)
        with torch.cuda.stream(sA):
            spmm()
        with torch.cuda.stream(sB):
            gemms()
        cur.wait_stream(sA)
        cur.wait_stream(sB)

    spmm(); gemms()
    t_s = wall(spmm)
    t_g = wall(gemms)
    t_ser = wall(serial)
    t_con = wall(concurrent)
    print(f"tile={os.environ.get('GNNX_GEMM_TILE', 'default')}: spmm {t_s:.2f} ms, gemms {t_g:.2f} ms, serial {t_ser:.2f} ms, "
          f"two streams {t_con:.2f} ms (ideal max {max(t_s, t_g):.2f})", flush=True)


main()
