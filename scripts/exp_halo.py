#!/usr/bin/env python3
"""Halo volume of the 1-D vertex partition on the bench graph (single process, plans built rank by rank), for the
round-1 contiguous partition and the degree-sorted snake deal: rows / non-zeros per rank, halo rows per rank and --
what an xGMI all-to-all-v is bound by -- rows per (sender, receiver) link, max / mean."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
shard = importlib.import_module("gnncpp_amd.shard")
dev = torch.device("cuda:0")
n, e, F = 10_000_000, 100_000_000, 256
if len(sys.argv) > 1:
    n, e = int(sys.argv[1]), int(sys.argv[2])
src, dst = ops.rmat_edges(2, n, e, device=dev)
w = torch.bincount(src.long(), minlength=n) + torch.bincount(dst.long(), minlength=n) + max(1, round(0.078 * F))


def builder(s_, d_, n_rows, n_cols):
    rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
    return rp[: n_rows + 1].contiguous(), ci


for partition in ("contiguous", "deal"):
    for world in (2, 4, 8):
        part = (shard.deal_partition if partition == "deal" else shard.contiguous_partition)(w, world)
        tot_f = tot_b = 0
        worst = 0
        links = []
        rows, nnzs = [], []
        for rank in range(world):
            p = shard.ShardPlan(src, dst, n, rank, world, None, builder, partition=part)
            tot_f += p.fwd.n_halo
            tot_b += p.bwd.n_halo
            worst = max(worst, p.fwd.n_halo, p.bwd.n_halo)
            links += [c for q, c in enumerate(p.fwd.recv_counts) if q != rank]
            rows.append(p.n_local)
            nnzs.append(p.nnz_local)
            print(f"{partition} world {world} rank {rank}: rows {p.n_local:>9d} nnz {p.nnz_local:>10d} nnz_t {int(p.bwd.colidx.numel()):>10d} "
                  f"halo_fwd {p.fwd.n_halo:>9d} ({p.fwd.n_halo * F * 4 / 1e9:.2f} GB) halo_bwd {p.bwd.n_halo:>9d} "
                  f"recv_counts_fwd {p.fwd.recv_counts}", flush=True)
            del p
            ops._ws_cache.clear()
            torch.cuda.empty_cache()
        mean_link = sum(links) / len(links)
        print(f"{partition} world {world}: total halo rows fwd {tot_f} bwd {tot_b}; worst rank receives {worst * F * 4 / 1e9:.2f} GB per "
              f"exchange; rows per link max {max(links)} mean {mean_link:.0f} max/mean {max(links) / mean_link:.3f} "
              f"(max link {max(links) * F * 4 / 1e9:.3f} GB); rows per rank max/mean {max(rows) / (sum(rows) / world):.3f}; "
              f"nnz per rank max/mean {max(nnzs) / (sum(nnzs) / world):.3f}", flush=True)
