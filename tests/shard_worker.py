"""Worker of tests/test_shard_gloo.py: one rank of a world_size-N gloo job on CPU.

Drives the PRODUCT partition / halo-plan / exchange code (gnn.cpp_amd/shard.py) with CPU tensors; the
arithmetic that the HIP kernels do on a GPU is done here by the oracle (test-side injection -- shard.py
itself never imports it).  Asserts that the sharded forward and backward aggregation of this rank's rows
are BIT-IDENTICAL to the unsharded oracle result, and that dW / dbias all-reduce to the global sums.
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests.helpers import synth  # noqa: E402

shard = importlib.import_module("gnncpp_amd.shard")


def np_csr(src, dst, n_rows, n_cols):
    """Adjacency semantics on (local row, global col) pairs: dedupe + (row, col) order; self loops were
    already dropped on global ids by ShardPlan."""
    key = np.unique(src.numpy().astype(np.int64) * n_cols + dst.numpy().astype(np.int64))
    rows, cols = key // n_cols, key % n_cols
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n_rows), out=rowptr[1:])
    return torch.from_numpy(rowptr.astype(np.int32)), torch.from_numpy(cols.astype(np.int32))


def pack(rows, idx, out):
    out.copy_(rows.index_select(0, idx.long()))


def degree_norm(rowptr, colidx, n_rows, s_out, s_cols, norm_out):
    rp = rowptr.numpy().astype(np.int64)
    if s_out is not None:
        deg = np.diff(rp)
        s_out.reshape(-1).copy_(torch.from_numpy(oracle.powf_table(int(deg.max()) + 2)[deg + 1]))
    if norm_out is not None:
        sc = s_cols.numpy().reshape(-1, 1)
        acc = oracle.aggregate_fwd(rp, colidx.numpy(), np.ascontiguousarray(sc), None, None, n_rows=n_rows)
        norm_out.copy_(torch.from_numpy(acc[:, 0] * sc[:n_rows, 0]))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, e, F = int(os.environ.get("N", 3000)), int(os.environ.get("E", 40000)), int(os.environ.get("F", 24))
    src, dst = synth.rmat_edges(7, n, e)
    H = synth.uniform_pm1(8, (n, F))
    G = synth.uniform_pm1(9, (n, F))
    bias = synth.uniform_pm1(10, (F,))
    # unsharded oracle
    rp, ci = oracle.coo_to_csr(src, dst, n)
    rT, cT = oracle.csr_transpose(rp, ci, n)
    s, norm = oracle.degree_norm(rp, ci, n)
    out_ref = oracle.aggregate_fwd(rp, ci, H, norm, bias)
    dH_ref = oracle.aggregate_bwd(rT, cT, G, norm)

    K = int(os.environ.get("CHUNKS", 1))   # row chunks of the pipelined exchange (chunk-major halo tail / send buffer)
    plan = shard.ShardPlan(torch.from_numpy(src), torch.from_numpy(dst), n, rank, world, dist, np_csr,
                           partition=os.environ.get("PARTITION", "deal"), n_chunks=K)
    assert plan.n_chunks == K and plan.row_chunks[0] == 0 and plan.row_chunks[-1] == plan.n_local
    nl = plan.n_local
    mine = plan.verts.numpy()  # original ids of this rank's rows, in local order
    assert plan.cuts[0] == 0 and plan.cuts[-1] == n and all(a <= b for a, b in zip(plan.cuts, plan.cuts[1:]))
    assert len(mine) == nl and len(np.unique(mine)) == nl
    if os.environ.get("PARTITION", "deal") != "deal":
        assert np.all(np.diff(mine) > 0)   # un-scrambled partitions keep a rank's rows in ascending original id
    assert np.array_equal(plan.nid.numpy()[mine], np.arange(plan.lo, plan.hi)), "new ids are not rank-contiguous"
    counts = torch.tensor([nl], dtype=torch.int64)
    dist.all_reduce(counts)
    assert int(counts.item()) == n, "the partition does not cover every vertex exactly once"
    halo_f, halo_b = plan.orig_ids(plan.fwd.halo).numpy(), plan.orig_ids(plan.bwd.halo).numpy()
    plan.compute_norm(dist, degree_norm, pack)
    assert np.array_equal(plan.norm.numpy(), norm[mine]), "sharded norm differs"
    assert np.array_equal(plan.s_ext[:nl, 0].numpy(), s[mine])
    assert np.array_equal(plan.s_ext[nl:, 0].numpy(), s[halo_f]), "halo s exchange wrong"

    # forward: [local | halo] buffer, one exchange, local aggregation
    f = plan.fwd
    Hext = torch.zeros((nl + f.n_halo, F), dtype=torch.float32)
    if K == 1:
        Hext[:nl] = torch.from_numpy(H[mine])
        _, h = shard.exchange_rows(dist, f, Hext, F, pack, async_op=True)
        h.wait()
    else:
        # the pipeline of the training schedule: chunk k of the local rows is WRITTEN just before chunk k is exchanged (the later
        # chunks are still zero), so a chunk's exchange that read a row of another chunk would deliver zeros
        Hl = torch.from_numpy(H[mine])
        hs = []
        for k in range(K):
            r0, r1 = plan.row_chunks[k], plan.row_chunks[k + 1]
            Hext[r0:r1] = Hl[r0:r1]
            hs.append(shard.exchange_rows(dist, f, Hext, F, pack, async_op=True, chunk=k)[1])
        for h in hs:
            h.wait()
        assert sum(sum(r) for r in f.recv_counts_k) == f.n_halo and sum(sum(r) for r in f.send_counts_k) == int(f.send_idx.numel())
        assert [sum(f.recv_counts_k[k][q] for k in range(K)) for q in range(world)] == f.recv_counts
    assert np.array_equal(Hext[nl:].numpy(), H[halo_f]), "forward halo rows wrong"
    out = oracle.aggregate_fwd(f.rowptr.numpy().astype(np.int64), f.colidx.numpy(), Hext.numpy(), plan.norm.numpy(), bias,
                               n_rows=nl)
    assert np.array_equal(out, out_ref[mine]), "sharded forward aggregation is not bit-identical"

    # backward: pull rows of G for in-neighbours through the transposed shard
    b = plan.bwd
    Gext = torch.zeros((nl + b.n_halo, F), dtype=torch.float32)
    Gext[:nl] = torch.from_numpy(G[mine])
    shard.exchange_rows(dist, b, Gext, F, pack)
    assert np.array_equal(plan.norm_ext_bwd.numpy()[nl:], norm[halo_b]), "backward halo norm wrong"
    dH = oracle.aggregate_bwd(b.rowptr.numpy().astype(np.int64), b.colidx.numpy(), Gext.numpy(), plan.norm_ext_bwd.numpy(),
                              n_rows=nl)
    assert np.array_equal(dH, dH_ref[mine]), "sharded backward aggregation is not bit-identical"

    # nnz bookkeeping and parameter-gradient reduction
    t = torch.tensor([plan.nnz_local], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t.item()) == len(ci)
    dbias = torch.from_numpy(G[mine].astype(np.float64).sum(0))
    dist.all_reduce(dbias)
    assert np.allclose(dbias.numpy(), G.astype(np.float64).sum(0))
    # send lists are consistent: what I send to q is what q receives from me
    sent = torch.tensor(f.send_counts, dtype=torch.int64)
    got = torch.empty(world, dtype=torch.int64)
    dist.all_to_all_single(got, sent)
    assert got.tolist() == f.recv_counts
    dist.barrier()
    if rank == 0:
        print(f"SHARD_OK world={world} cuts={plan.cuts} halo_fwd={f.n_halo} halo_bwd={b.n_halo}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
