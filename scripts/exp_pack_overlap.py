#!/usr/bin/env python3
"""Does the halo pack (a row gather, bandwidth-bound) hide under an MFMA-bound product when the two run on different streams?
One rank of eight of the headline graph: product 1.25 M x 256 x 256, pack of 1.79 M rows of 1 KiB.  Prints sequential and concurrent
times, with the pack launched before / after the product, whole and in 4 chunks (chunk k's pack beside chunk k+1's product)."""
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
M, F, NS = 1_250_000, 256, 1_790_000
X = ops.uniform_pm1(1, (M, F), device=dev)
W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=dev)
H = torch.empty((M, F), dtype=torch.float32, device=dev)
G = ops.uniform_pm1(3, (M, F), device=dev)
idx = torch.sort(torch.randint(0, M, (NS,), device=dev, dtype=torch.int32)).values
send = torch.empty((NS, F), dtype=torch.float32, device=dev)
side = torch.cuda.Stream()


def wall(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def gemm(r0=0, r1=M):
    ops.linear_fwd(X[r0:r1], W, out=H[r0:r1])


def pack(src=G, i0=0, i1=NS):
    ops.gather_rows(src, idx[i0:i1], out=send[i0:i1])


def beside(first_main, then_side):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        then_side()
    first_main()
    cur.wait_stream(side)


print("product alone            %.3f ms" % wall(gemm))
print("pack alone               %.3f ms" % wall(pack))
print("product then pack        %.3f ms" % wall(lambda: (gemm(), pack())))
print("pack (side) || product   %.3f ms" % wall(lambda: beside(gemm, pack)))
# chunked: the pack of H's chunk k (rows it covers: idx is sorted, so a contiguous slice) beside the product of chunk k+1
K = 4
rb = [M * k // K for k in range(K + 1)]
ib = [int(torch.searchsorted(idx, torch.tensor(r, device=dev, dtype=torch.int32))) for r in rb]


def chunked():
    cur = torch.cuda.current_stream()
    for k in range(K):
        gemm(rb[k], rb[k + 1])
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            pack(H, ib[k], ib[k + 1])
    cur.wait_stream(side)


def chunked_seq():
    for k in range(K):
        gemm(rb[k], rb[k + 1])
        pack(H, ib[k], ib[k + 1])


print("4 chunks, product k then pack k on ONE stream   %.3f ms" % wall(chunked_seq))
print("4 chunks, pack k (side) || product k+1          %.3f ms" % wall(chunked))
