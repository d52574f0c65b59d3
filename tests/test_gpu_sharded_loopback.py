"""The sharded step (gnn.cpp_amd/shard.py ShardedBench: plan, norm exchange, halo all-to-all-v, SpMM over [local|halo],
GEMMs, parameter all-reduce) on ONE GPU: P ranks run as threads of this process and talk through an in-process stand-in
for torch.distributed (same call signatures, data moved with device copies).  What it proves that the gloo/CPU tests and
the single-rank rehearsal cannot: the real HIP kernels on the real [local | halo] buffers of every rank reproduce the
single-GPU result -- forward and backward aggregation bit for bit, dW / dbias after the all-reduce within tolerance."""
import importlib
import threading

import numpy as np
import pytest

from tests.helpers import pkg

pytestmark = pytest.mark.gpu


class LoopbackWorld:
    """Minimal torch.distributed look-alike for P threads sharing one device."""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def rank_view(self, rank):
        return _RankDist(self, rank)


class _RankDist:
    def __init__(self, w, rank):
        self.w, self.rank = w, rank

    class _Work:
        def wait(self):
            return True

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        """async_op: the stand-in completes the exchange before returning (every rank shares one device and one stream,
        so there is nothing to overlap); the caller's issue order and buffer lifetimes are what is being exercised."""
        w, P = self.w, self.w.world
        if input_split_sizes is None:
            n = inp.shape[0] // P
            input_split_sizes = [n] * P
            output_split_sizes = [n] * P
        w.slots[self.rank] = (inp, list(input_split_sizes))
        w.bar.wait()
        off = 0
        for q in range(P):
            src, splits = w.slots[q]
            start = sum(splits[: self.rank])
            cnt = splits[self.rank]
            assert cnt == output_split_sizes[q], (self.rank, q, cnt, output_split_sizes[q])
            if cnt:
                out[off: off + cnt].copy_(src[start: start + cnt])
            off += cnt
        import torch
        torch.cuda.synchronize()
        w.bar.wait()
        return self._Work() if async_op else None

    def all_reduce(self, t, op=None):
        import torch
        w = self.w
        w.slots[self.rank] = t.clone()
        w.bar.wait()
        total = w.slots[0].clone()
        for q in range(1, w.world):
            total += w.slots[q]
        torch.cuda.synchronize()
        w.bar.wait()
        t.copy_(total)

    def barrier(self):
        self.w.bar.wait()


@pytest.mark.parametrize("world,schedule,partition,chunk", [(2, "overlap", "deal", 0), (4, "overlap", "deal", 64), (8, "overlap", "deal", 0),
                                                            (4, "sequential", "deal", 0), (8, "sequential", "contiguous", 64),
                                                            (2, "training", "deal", 64), (8, "training", "deal", 0), (3, "training", "deal", 256),
                                                            (4, "replicate-input-halo", "deal", 0),
                                                            (8, "replicate-input-halo", "deal", 256)])
def test_sharded_step_equals_single_gpu(world, schedule, partition, chunk):
    """chunk > 0: every shard runs with a load-balancing plan (hub rows of the shard's [local | halo] CSR on the sequential hub
    kernel), as bench.py does -- same bits.  The single-GPU reference the shards are compared with is itself compared with the CPU
    oracle here, bit for bit, so "equals one GPU" is "equals the oracle" at this size."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("needs a HIP device")
    ops = importlib.import_module("gnncpp_amd.ops")
    capi = importlib.import_module("gnncpp_amd.capi")
    shard = importlib.import_module("gnncpp_amd.shard")
    dev = torch.device("cuda:0")
    n, e, F, abc, seed = 200_000, 2_000_000, 64, (0.57, 0.19, 0.19), 5
    # ---- single-GPU reference (exact mode: no plan)
    src, dst = ops.rmat_edges(seed, n, e, *abc, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    X = ops.uniform_pm1(seed + 10, (n, F), device=dev)
    W = ops.uniform_pm1(seed + 11, (F, F), scale=F ** -0.5, device=dev)
    G = ops.uniform_pm1(seed + 12, (n, F), device=dev)
    bias = torch.zeros(F, dtype=torch.float32, device=dev)
    H = ops.linear_fwd(X, W)
    out_ref = ops.aggregate_fwd(g, H, bias, use_plan=False)
    dH_ref = ops.aggregate_bwd(g, G, use_plan=False)
    dX_ref, dW_ref = ops.linear_bwd(dH_ref, X, W)
    dbias_ref = ops.colsum(G)
    torch.cuda.synchronize()
    import oracle   # checker only
    rp, ci = oracle.coo_to_csr(src.cpu().numpy(), dst.cpu().numpy(), n)
    rT, cT = oracle.csr_transpose(rp, ci, n)
    norm_h = g.norm.cpu().numpy()
    assert np.array_equal(out_ref.cpu().numpy(), oracle.aggregate_fwd(rp, ci, H.cpu().numpy(), norm_h, bias.cpu().numpy()))
    assert np.array_equal(dH_ref.cpu().numpy(), oracle.aggregate_bwd(rT, cT, G.cpu().numpy(), norm_h))

    lw = LoopbackWorld(world)
    runners, errors = [None] * world, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            rep = schedule == "replicate-input-halo"   # opt-in first-layer form: X halo fetched once, one exchange per step
            r = shard.ShardedBench(ops, capi, pkg, lw.rank_view(rank), dev, rank, world, n, e, F, abc, seed, chunk, global_inputs=True,
                                   schedule="overlap" if rep else schedule, partition=partition, replicate_input_halo=rep)
            r.step()
            r.step(timed=True)  # a second step reuses every buffer (send buffers, halo tails) while nothing is in flight
            torch.cuda.synchronize()
            runners[rank] = r
        except Exception as ex:  # noqa: BLE001
            import traceback
            errors.append((rank, traceback.format_exc()))
            lw.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[0][1]
    nnz = sum(r.plan.nnz_local for r in runners)
    assert nnz == g.nnz
    covered = torch.zeros(n, dtype=torch.int32, device=dev)
    for r in runners:
        v = r.plan.verts
        covered[v] += 1
        assert v.numel() == r.plan.n_local   # (covered == 1 below: every vertex exactly once)
        assert torch.equal(r.plan.norm, g.norm[v]), "sharded norm differs"
        assert torch.equal(r.out, out_ref[v]), "sharded forward aggregation not bit-identical to single GPU"
        assert torch.equal(r.dH, dH_ref[v]), "sharded backward aggregation not bit-identical to single GPU"
        assert float((r.dX - dX_ref[v]).abs().max()) <= 1e-5 * max(1.0, float(dX_ref.abs().max()))
        # parameters were all-reduced: every rank holds the global sums
        scale = float((dH_ref.abs().t().double() @ X.abs().double()).max())
        assert float((r.dW - dW_ref).abs().max()) <= 1e-5 * max(1.0, scale)
        assert float((r.dbias - dbias_ref).abs().max()) <= 1e-5 * max(1.0, float(G.abs().double().sum(0).max()))
    assert bool((covered == 1).all()), "the partition does not cover every vertex exactly once"
    cuts = runners[0].plan.cuts
    assert cuts[0] == 0 and cuts[-1] == n and all(r.plan.cuts == cuts for r in runners)
    if partition == "deal":  # every rank gets the same number of rows (+-1) and about the same non-zeros
        rows = [r.plan.n_local for r in runners]
        nz = [r.plan.nnz_local for r in runners]
        assert max(rows) - min(rows) <= 1
        assert max(nz) <= 1.1 * (sum(nz) / world)
        sends = [c for r in runners for q, c in enumerate(r.plan.fwd.send_counts) if q != r.rank]
        assert max(sends) <= 1.35 * (sum(sends) / len(sends)), "per-link send volume is unbalanced"


def test_cabi_one_rank_plan_is_the_single_gpu_relabelled_csr():
    """The single-GPU vertex relabelling behind the C-ABI: gnnx_partition_deal + gnnx_partition_scramble + gnnx_shard_select_edges +
    gnnx_csr_from_coo + gnnx_halo_plan_create with ONE rank produce exactly the CSR pair that ops.CsrGraph.from_coo(relabel=
    "scramble") -- what bench.py runs -- builds with torch index ops: same nid, same rowptr / colidx for A and A^T."""
    import torch
    ops = importlib.import_module("gnncpp_amd.ops")
    sn = importlib.import_module("gnncpp_amd.shard_native")
    dev = torch.device("cuda:0")
    n, e = 50_000, 600_000
    src, dst = ops.rmat_edges(21, n, e, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n, relabel="scramble")
    pn = sn.NativeShardPlan(src, dst, n, 0, 1, None)
    assert pn.cuts == [0, n] and torch.equal(pn.nid, g.nid)
    assert pn.fwd.n_halo == 0 and pn.bwd.n_halo == 0
    assert torch.equal(pn.fwd.rowptr, g.rowptr) and torch.equal(pn.fwd.colidx, g.colidx)
    assert torch.equal(pn.bwd.rowptr, g.rowptr_t) and torch.equal(pn.bwd.colidx, g.colidx_t)


@pytest.mark.parametrize("world", [2, 8])
def test_cabi_plan_equals_shard_py_plan_and_drives_a_step(world):
    """The partition + halo plan behind the C-ABI (gnnx_vertex_weights / gnnx_partition_deal / gnnx_shard_select_edges /
    gnnx_halo_plan_*; what the C++ host builds) equals shard.py's torch plan ARRAY FOR ARRAY, its send lists travel through
    the in-process communicator (gnnx_comm_init_local + gnnx_halo_plan_exchange_requests), and one forward + backward
    aggregation exchanged with gnnx_halo_exchange_rows_f32 reproduces the single-GPU rows bit for bit."""
    import torch
    ops = importlib.import_module("gnncpp_amd.ops")
    shard = importlib.import_module("gnncpp_amd.shard")
    sn = importlib.import_module("gnncpp_amd.shard_native")
    dev = torch.device("cuda:0")
    n, e, F, seed = 60_000, 700_000, 32, 9
    src, dst = ops.rmat_edges(seed, n, e, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    H = ops.uniform_pm1(seed + 1, (n, F), device=dev)
    G = ops.uniform_pm1(seed + 2, (n, F), device=dev)
    out_ref = ops.aggregate_fwd(g, H, None, use_plan=False)
    dH_ref = ops.aggregate_bwd(g, G, use_plan=False)
    torch.cuda.synchronize()
    lw = LoopbackWorld(world)
    comms = sn.local_comms(world)
    errors, done = [], [None] * world

    def builder(s_, d_, n_rows, n_cols):
        rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
        return rp[: n_rows + 1].contiguous(), ci

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            pt = shard.ShardPlan(src, dst, n, rank, world, lw.rank_view(rank), builder, row_weight=3)
            pn = sn.NativeShardPlan(src, dst, n, rank, world, comms[rank], row_weight=3)
            assert pn.cuts == pt.cuts
            assert torch.equal(pn.owner, pt.owner) and torch.equal(pn.nid, pt.nid)
            for a, b in ((pn.fwd, pt.fwd), (pn.bwd, pt.bwd)):
                assert a.n_local == b.n_local and a.n_halo == b.n_halo
                assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.colidx, b.colidx)
                assert torch.equal(a.halo(dev), b.halo.to(torch.int32))
                assert a.recv_counts == b.recv_counts and a.send_counts == b.send_counts
                assert torch.equal(a.send_idx(dev), b.send_idx)
            nl = pn.n_local
            v = pt.verts
            norm = g.norm[v].contiguous()
            # forward: [local | halo] buffer filled by the native exchange
            Hext = torch.zeros((nl + pn.fwd.n_halo, F), dtype=torch.float32, device=dev)
            Hext[:nl] = H[v]
            sbuf = torch.empty((max(pn.fwd.n_send, pn.bwd.n_send, 1), F), dtype=torch.float32, device=dev)
            pn.fwd.exchange_rows(comms[rank], Hext, sbuf)
            assert torch.equal(Hext[nl:], H[pt.orig_ids(pt.fwd.halo)])
            out = ops.spmm(pn.fwd.rowptr, pn.fwd.colidx, Hext, rowscale=norm, n_rows=nl)
            assert torch.equal(out, out_ref[v]), "forward aggregation over the C-ABI plan is not bit-identical"
            # the plan's own slot table is the inverse of its send list, and a transform that packs in its epilogue + the exchange of
            # the packed buffer (gnnx_halo_plan_slot_table / gnnx_gemm_nt_rows_to_slots_f32 / gnnx_halo_exchange_packed_f32: what
            # GCNConv::forward_sharded runs) fills the same [local | halo] buffer as product -> exchange with its own pack
            table = pn.fwd.slot_table(dev)
            Xl = ops.uniform_pm1(seed + 5, (n, F), device=dev)[v].contiguous()
            Wt = ops.uniform_pm1(seed + 6, (F, F), scale=F ** -0.5, device=dev)
            want = torch.zeros_like(Hext)
            ops.linear_fwd(Xl, Wt, out=want[:nl])
            pn.fwd.exchange_rows(comms[rank], want, sbuf)          # (collective calls: unconditional on every rank)
            got = torch.zeros_like(Hext)
            if pn.fwd.n_send:
                assert table is not None and torch.equal(table, ops.slot_table(pt.fwd.send_idx, nl))
                ops.linear_fwd_rows_to_slots(Xl, Wt, got[:nl], table, sbuf[: pn.fwd.n_send])
            else:
                ops.linear_fwd(Xl, Wt, out=got[:nl])
            pn.fwd.exchange_packed(comms[rank], got, sbuf)
            assert torch.equal(got, want), "the packed exchange differs from the exchange with its own pack"
            # backward: rows of G and the per-column norm for the transposed shard
            Gext = torch.zeros((nl + pn.bwd.n_halo, F), dtype=torch.float32, device=dev)
            Gext[:nl] = G[v]
            pn.bwd.exchange_rows(comms[rank], Gext, sbuf)
            nb = torch.zeros((nl + pn.bwd.n_halo, 1), dtype=torch.float32, device=dev)
            nb[:nl, 0] = norm
            sb1 = torch.empty((max(pn.bwd.n_send, 1), 1), dtype=torch.float32, device=dev)
            pn.bwd.exchange_rows(comms[rank], nb, sb1)
            dH = ops.spmm(pn.bwd.rowptr, pn.bwd.colidx, Gext, colscale=nb.reshape(-1).contiguous(), n_rows=nl)
            assert torch.equal(dH, dH_ref[v]), "backward aggregation over the C-ABI plan is not bit-identical"
            # the small all-reduce of the local transport: every rank ends with the same rank-ordered sum
            t = torch.full((1000,), float(rank + 1), dtype=torch.float32, device=dev)
            capi = importlib.import_module("gnncpp_amd.capi")
            capi.call("gnnx_allreduce_sum_f32", comms[rank], ops._ptr(t), t.numel(), ops._stream())
            assert bool((t == world * (world + 1) / 2).all())
            torch.cuda.synchronize()
            done[rank] = True
        except Exception:  # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
            lw.bar.abort()
            sn.comm_destroy(comms[rank])  # peers blocked in a collective get an error instead of hanging
            comms[rank] = None

    threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    for c in comms:
        if c is not None:
            sn.comm_destroy(c)
    assert not errors, errors[0]
    assert all(done)


def test_cross_shard_batchnorm_statistics_and_fused_layer():
    """BatchNorm between transform and aggregation on a sharded graph: every rank all-reduces two [F] vectors and gets the batch
    statistics of the WHOLE graph (shard.sharded_bn_stats), then runs the aggregation with the BatchNorm + ReLU prologue on raw
    exchanged rows of H.  Against one GPU: statistics within 1e-6 relative, layer output rows within 1e-5."""
    import torch
    ops = importlib.import_module("gnncpp_amd.ops")
    shard = importlib.import_module("gnncpp_amd.shard")
    dev = torch.device("cuda:0")
    world, n, e, F, seed = 4, 50_000, 600_000, 48, 21
    src, dst = ops.rmat_edges(seed, n, e, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    H = ops.uniform_pm1(seed + 1, (n, F), device=dev) * 1.5 + 0.2
    gamma = ops.uniform_pm1(seed + 2, (F,), device=dev) + 1.5
    beta = ops.uniform_pm1(seed + 3, (F,), scale=0.3, device=dev)
    bias = ops.uniform_pm1(seed + 4, (F,), scale=0.5, device=dev)
    mean1, var1 = ops.bn_stats(H)
    out1 = ops.aggregate_fwd(g, H, bias, use_plan=False, bn=(mean1, var1, gamma, beta, 1e-5), relu_in=True)
    torch.cuda.synchronize()
    lw = LoopbackWorld(world)
    errors = []

    def builder(s_, d_, n_rows, n_cols):
        rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
        return rp[: n_rows + 1].contiguous(), ci

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            dist = lw.rank_view(rank)
            p = shard.ShardPlan(src, dst, n, rank, world, dist, builder)
            v, nl = p.verts, p.n_local
            Hl = H[v].contiguous()
            mean, var = shard.sharded_bn_stats(dist, ops, Hl, n)
            assert float(((mean - mean1).abs() / mean1.abs().clamp_min(1.0)).max()) <= 1e-6
            assert float(((var - var1).abs() / var1.abs().clamp_min(1.0)).max()) <= 1e-6
            Hext = torch.zeros((nl + p.fwd.n_halo, F), dtype=torch.float32, device=dev)
            Hext[:nl] = Hl
            shard.exchange_rows(dist, p.fwd, Hext, F, lambda rows, idx, out: ops.gather_rows(rows, idx, out=out))
            out = ops.spmm(p.fwd.rowptr, p.fwd.colidx, Hext, rowscale=g.norm[v].contiguous(), bias=bias, n_rows=nl,
                           bn=(mean, var, gamma, beta, 1e-5), relu_in=True)
            ref = out1[v]
            assert float(((out - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 1e-5
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
            lw.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[0]


@pytest.mark.parametrize("world,dims", [(2, [64, 64, 32]), (4, [100, 72, 47]), (3, [32, 32, 32, 16])])
def test_sharded_multi_layer_training_step_equals_single_gpu(world, dims):
    """shard.ShardedGcnStack -- the multi-layer training step on a 1-D vertex partition (BASELINE configs[4] "3-layer GCN fwd+bwd, 1 and 8
    GPUs") -- against ops.GcnStack on one GPU: logits of every vertex bit for bit, the global loss, every parameter gradient
    (summed over the ranks: rounding-level) and the parameters after one SGD step."""
    import torch
    ops = importlib.import_module("gnncpp_amd.ops")
    shard = importlib.import_module("gnncpp_amd.shard")
    capi = importlib.import_module("gnncpp_amd.capi")
    dev = torch.device("cuda:0")
    n, e, seed = 60_000, 500_000, 13
    s_np, d_np = pkg.synth.uniform_edges(seed, n, e)   # no hubs: logits stay O(1), the reference's max-free softmax does not overflow
    src, dst = torch.from_numpy(s_np).to(dev), torch.from_numpy(d_np).to(dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    X = ops.uniform_pm1(seed + 1, (n, dims[0]), device=dev)
    target = (torch.arange(n, device=dev, dtype=torch.int64) * 7 + 3).remainder(dims[-1]).to(torch.int32)
    net = ops.GcnStack(g, dims, seed=77, device=dev, pad_streamed=False)
    for l in range(len(dims) - 1):
        net.b[l].copy_(ops.uniform_pm1(200 + l, (dims[l + 1],), scale=0.2, device=dev))
    b0 = [b.clone() for b in net.b]
    logits = net.forward(X)
    loss_ref, dlog = ops.softmax_ce(logits, target, colsum_out=net.db[-1])
    net.backward(dlog, input_grad=False, have_last_bias_grad=True)
    dW_ref, db_ref = [w.clone() for w in net.dW], [b.clone() for b in net.db]
    logits_ref = logits.clone()
    net.step(lr=0.05)
    W_after = [w.clone() for w in net.W]
    torch.cuda.synchronize()

    lw = LoopbackWorld(world)
    errors, res = [], [None] * world

    def builder(s_, d_, n_rows, n_cols):
        rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
        return rp[: n_rows + 1].contiguous(), ci

    def degree_norm(rowptr, colidx, n_rows, s_out, s_cols, norm_out):
        capi.call("gnnx_degree_norm_f32", ops._ptr(rowptr), ops._ptr(colidx), n_rows, ops._ptr(s_out), ops._ptr(s_cols), ops._ptr(norm_out),
                  ops._stream())

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            dist = lw.rank_view(rank)
            p = shard.ShardPlan(src, dst, n, rank, world, dist, builder)
            p.compute_norm(dist, degree_norm, lambda rows, idx, out: ops.gather_rows(rows, idx, out=out))
            st = shard.ShardedGcnStack(ops, dist, p, dims, seed=77, chunk=0)   # exact mode on both sides (no hub-row chunks)
            for l in range(len(dims) - 1):
                st.b[l].copy_(b0[l])
            v = p.verts
            lg = st.forward(X[v].contiguous())
            assert torch.equal(lg, logits_ref[v]), "logits of a shard differ from the single-GPU stack"
            loss = st.loss_and_backward(lg, target[v].contiguous(), n)
            st.step(lr=0.05)
            torch.cuda.synchronize()
            res[rank] = (float(loss.item()), [w.clone() for w in st.dW], [b.clone() for b in st.db], [w.clone() for w in st.W])
        except Exception:  # noqa: BLE001
            import traceback
            errors.append((rank, traceback.format_exc()))
            lw.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[0][1]
    for rank in range(world):
        loss, dW, db, W = res[rank]
        assert abs(loss - float(loss_ref.item())) <= 1e-5 * max(1.0, abs(float(loss_ref.item())))
        for a, b in zip(dW, dW_ref):
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6)
        for a, b in zip(db, db_ref):
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6)
        for a, b in zip(W, W_after):
            assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("workload", ["rmat10m_100m_f256", "products_2p4m_62m_f100"])
def test_one_rank_of_eight_at_baseline_sizes(workload):
    """BASELINE configs[3] / [4] in their x8 form: ranks 0 and 7 of the P = 8 partition of the FULL-SIZE bench graph, built by the
    product planner both ways (shard.ShardPlan -- with the chunk-major tail of the training schedule -- and the C-ABI plan behind
    gnnx_partition_deal / gnnx_shard_select_edges / gnnx_halo_plan_create, compared array for array), the halo tail filled from the
    single-GPU matrices (no second GPU on this box: the exchange itself is covered by the loopback and gloo tests), then the PLANNED
    forward / backward aggregation and the three dense products on the shard:
      * norm of the shard (s of the halo columns handed over) == the single-GPU norm of those vertices;
      * X.W^T, norm (.) (A.H) + bias, A^T.(norm (.) G) and dH.W of the shard's rows are torch.equal to the single-GPU rows -- which
        test_headline_config_whole_graph_vs_oracle / test_multi_layer_gcn_at_baseline_sizes_vs_oracle pin to the oracle bit for bit;
      * dH^T.X over the shard's rows against float64 (a rank's partial sum has no single-GPU counterpart).
    The [local | halo] renumbering, the hub plan ON A SHARD (hub rows keep all their non-zeros: 1/8 of the rows, the same longest
    row) and the 128-float padded layout of the products-shaped width meet the full-size graphs here."""
    import torch
    ops = importlib.import_module("gnncpp_amd.ops")
    shard = importlib.import_module("gnncpp_amd.shard")
    sn = importlib.import_module("gnncpp_amd.shard_native")
    bench = importlib.import_module("bench")
    dev = torch.device("cuda:0")
    n, e, F, abc, seed = bench.WORKLOADS[workload]
    world, chunk = 8, 1024
    src, dst = ops.rmat_edges(seed, n, e, *abc, device=dev)
    g = ops.CsrGraph.from_coo(src, dst, n)
    g.make_plans(chunk, F)
    Fp = -(-F // 128) * 128   # streamed rows (X, dH, dX, W) on the 128-float stride, as bench.py / GcnStack lay them out
    def padded(t):
        if Fp == F:
            return t
        p_ = torch.zeros((t.shape[0], Fp), dtype=torch.float32, device=dev)
        p_[:, :F] = t
        return p_
    Xp = padded(ops.uniform_pm1(seed + 10, (n, F), device=dev))
    Wp = torch.zeros((Fp, Fp), dtype=torch.float32, device=dev)
    Wp[:F, :F] = ops.uniform_pm1(seed + 11, (F, F), scale=F ** -0.5, device=dev)
    bias = ops.uniform_pm1(seed + 13, (F,), scale=0.1, device=dev)
    G = ops.uniform_pm1(seed + 12, (n, F), device=dev)
    H = ops.linear_fwd(Xp, Wp[:F])                       # [n, F]: gathered rows keep their own width
    out = ops.aggregate_fwd(g, H, bias)
    dHp = torch.zeros((n, Fp), dtype=torch.float32, device=dev)
    ops.aggregate_bwd(g, G, out=dHp[:, :F])
    dXp = ops.gemm(dHp, Wp)
    torch.cuda.synchronize()

    def builder(s_, d_, n_rows, n_cols):
        rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
        return rp[: n_rows + 1].contiguous(), ci

    def degree_norm(rowptr, colidx, n_rows, s_out, s_cols, norm_out):
        capi = importlib.import_module("gnncpp_amd.capi")
        capi.call("gnnx_degree_norm_f32", ops._ptr(rowptr), ops._ptr(colidx), n_rows, ops._ptr(s_out), ops._ptr(s_cols), ops._ptr(norm_out),
                  ops._stream())

    rw = max(1, round(0.078 * F))
    for rank in (0, world - 1):
        pn = sn.NativeShardPlan(src, dst, n, rank, world, None, row_weight=rw)
        pt1 = shard.ShardPlan(src, dst, n, rank, world, None, builder, row_weight=rw)                 # owner-major tail: the C-ABI layout
        assert pn.cuts == pt1.cuts and torch.equal(pn.owner, pt1.owner) and torch.equal(pn.nid, pt1.nid)
        for a, b in ((pn.fwd, pt1.fwd), (pn.bwd, pt1.bwd)):
            assert a.n_local == b.n_local and a.n_halo == b.n_halo and a.recv_counts == b.recv_counts
            assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.colidx, b.colidx) and torch.equal(a.halo(dev), b.halo.to(torch.int32))
        part = (pt1.owner, pt1.nid, pt1.cuts)
        del pn, pt1
        pt = shard.ShardPlan(src, dst, n, rank, world, None, builder, partition=part, n_chunks=4)      # chunk-major tail (training schedule)
        nl, v = pt.n_local, pt.verts
        assert abs(nl - n / world) <= 1
        # ---- norm: s of the halo columns from the single-GPU vector
        s_ext = torch.zeros((nl + pt.fwd.n_halo, 1), dtype=torch.float32, device=dev)
        degree_norm(pt.fwd.rowptr, pt.fwd.colidx, nl, s_ext[:nl], None, None)
        assert torch.equal(s_ext[:nl, 0], g.s[v])
        s_ext[nl:, 0] = g.s[pt.orig_ids(pt.fwd.halo)]
        norm = torch.zeros(nl, dtype=torch.float32, device=dev)
        degree_norm(pt.fwd.rowptr, pt.fwd.colidx, nl, None, s_ext, norm)
        assert torch.equal(norm, g.norm[v]), "shard norm"
        plan_f, plan_b = ops.SpmmPlan(pt.fwd.rowptr, chunk, F), ops.SpmmPlan(pt.bwd.rowptr, chunk, F)
        assert plan_f.n_split_rows > 0 and plan_b.n_split_rows > 0, "a shard of the bench graph has hub rows"
        # ---- forward: transform of the local rows, halo rows of H from the single-GPU matrix, planned aggregation
        Xl = Xp[v].contiguous()
        Hext = torch.empty((nl + pt.fwd.n_halo, F), dtype=torch.float32, device=dev)
        for k in range(pt.n_chunks):   # row chunks, as the training schedule multiplies them
            r0, r1 = pt.row_chunks[k], pt.row_chunks[k + 1]
            ops.linear_fwd(Xl[r0:r1], Wp[:F], out=Hext[r0:r1])
        assert torch.equal(Hext[:nl], H[v]), "X.W^T of the shard's rows (in row chunks)"
        Hext[nl:] = H[pt.orig_ids(pt.fwd.halo)]
        o = ops.spmm(pt.fwd.rowptr, pt.fwd.colidx, Hext, rowscale=norm, bias=bias, plan=plan_f, n_rows=nl)
        assert torch.equal(o, out[v]), f"rank {rank}: planned forward aggregation of the shard != single-GPU rows"
        del Hext, o
        # ---- backward
        Gext = torch.empty((nl + pt.bwd.n_halo, F), dtype=torch.float32, device=dev)
        Gext[:nl] = G[v]
        hb = pt.orig_ids(pt.bwd.halo)
        Gext[nl:] = G[hb]
        norm_ext = torch.cat([norm, g.norm[hb]])
        vals = ops.gather_rows(norm_ext.reshape(-1, 1), pt.bwd.colidx).reshape(-1)
        dHl = torch.zeros((nl, Fp), dtype=torch.float32, device=dev)
        ops.spmm(pt.bwd.rowptr, pt.bwd.colidx, Gext, out=dHl[:, :F], vals=vals, plan=plan_b, n_rows=nl)
        assert torch.equal(dHl, dHp[v]), f"rank {rank}: planned backward aggregation of the shard != single-GPU rows"
        assert torch.equal(ops.gemm(dHl, Wp), dXp[v]), "dH.W of the shard's rows"
        dW = ops.gemm(dHl, Xl, transA=True)
        ref = dHl.double().t() @ Xl.double()
        bound = 1e-5 * torch.maximum(torch.maximum(ref.abs(), dHl.abs().double().t() @ Xl.abs().double()), torch.ones_like(ref))
        assert bool(((dW.double() - ref).abs() <= bound).all()), "dH^T.X of the shard's rows vs float64"
        del Gext, dHl, dW, ref, bound, pt, plan_f, plan_b, Xl, vals
        ops._ws_cache.clear()
        torch.cuda.empty_cache()
