#!/usr/bin/env python3
"""Does the forward aggregation's time depend on WHERE its feature matrix lives?  Same graph, same values, the gathered matrix in
buffers allocated at different moments (before / after the graph build, fresh hipMalloc segments vs re-used cache blocks)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
dev = torch.device("cuda:0")
n, e, F = 10_000_000, 100_000_000, 256
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
early = [torch.empty((n, F), dtype=torch.float32, device=dev) for _ in range(2)]   # allocated before anything else
src, dst = ops.rmat_edges(0, n, e, 0.57, 0.19, 0.19, device=dev)
g = ops.CsrGraph.from_coo(src, dst, n)
del src, dst
g.make_plans(4096, F)
late = [torch.empty((n, F), dtype=torch.float32, device=dev) for _ in range(2)]     # after the build (cache blocks of the build re-used)
ops._ws_cache.clear()
torch.cuda.empty_cache()
fresh = [torch.empty((n, F), dtype=torch.float32, device=dev) for _ in range(2)]    # after empty_cache: new segments
out = torch.empty((n, F), dtype=torch.float32, device=dev)
src_vals = ops.uniform_pm1(1, (n, F), device=dev)


def time_spmm(H, reps=5):
    H.copy_(src_vals)
    ts = []
    for _ in range(reps):
        a, b = capi.Event(), capi.Event()
        a.record(st)
        ops.aggregate_fwd(g, H, None, out=out)
        b.record(st)
        b.sync()
        ts.append(a.elapsed_ms(b))
    return np.median(ts), min(ts), max(ts)


for rnd in range(2):
    for name, bufs in (("early", early), ("late", late), ("fresh", fresh)):
        for i, H in enumerate(bufs):
            med, lo, hi = time_spmm(H)
            print(f"round {rnd} {name}[{i}] ptr mod 2MB = {H.data_ptr() % (2 << 20):8d}  med {med:.2f}  min {lo:.2f}  max {hi:.2f}", flush=True)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
