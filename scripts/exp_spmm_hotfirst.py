#!/usr/bin/env python3
"""Keep hub rows in L2?  Vertex labels hot-first (the K vertices gathered most often get ids [0, K), scrambled inside the two classes)
and, in the EXPERIMENTS build, the streaming hint on every gather of a cold row (GNNX_SPMM_POLICY=8, GNNX_SPMM_HOTK=K: read from the
environment by the library at first use, so ONE setting per process -- run this script once per setting).
Prints the forward / backward aggregation time on the headline graph for: scramble (bench default), hot-first labels alone, and
whatever policy the environment selects on the hot-first labels.   usage: K=65536 [GNNX_SPMM_POLICY=8 GNNX_SPMM_HOTK=65536] exp_spmm_hotfirst.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
dev = torch.device("cuda:0")
n, e, F = int(os.environ.get("N", 10_000_000)), int(os.environ.get("E", 100_000_000)), int(os.environ.get("F", 256))
K = int(os.environ.get("K", 65536))
src, dst = ops.rmat_edges(2, n, e, 0.57, 0.19, 0.19, device=dev)


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def run(label, relabel):
    g = ops.CsrGraph.from_coo(src, dst, n, relabel=relabel)
    g.make_plans(1024, F)
    H = ops.uniform_pm1(5, (n, F), device=dev)
    out = torch.empty_like(H)
    bias = torch.zeros(F, device=dev)
    f = timed(lambda: ops.aggregate_fwd(g, H, bias, out=out))
    b = timed(lambda: ops.aggregate_bwd(g, H, out=out))
    print(f"{label}: fwd {f:.2f} ms  bwd {b:.2f} ms   (policy {os.environ.get('GNNX_SPMM_POLICY', '0')}, hot_k {os.environ.get('GNNX_SPMM_HOTK', '0')})", flush=True)
    del g, H, out
    torch.cuda.empty_cache()


run("scramble", "scramble")
# hot-first: rank vertices by in-degree + out-degree (both aggregations gather them), top K first
deg = torch.bincount(dst.long(), minlength=n) + torch.bincount(src.long(), minlength=n)
order = torch.argsort(deg, descending=True, stable=True)            # order[k] = vertex of rank k
mul = ops.CsrGraph.SCRAMBLE_MUL
rank = torch.arange(n, device=dev, dtype=torch.int64)
new_of_rank = torch.where(rank < K, (rank * mul) % K, K + ((rank - K) * mul) % max(n - K, 1))
nid = torch.empty(n, dtype=torch.int64, device=dev)
nid[order] = new_of_rank
share = float(deg[order[:K]].sum()) / float(deg.sum())
print(f"K = {K}: {K * F * 4 / 2**20:.0f} MiB of hot rows carry {share:.3f} of the edge endpoints", flush=True)
run(f"hot-first K={K}", nid.to(torch.int32))
