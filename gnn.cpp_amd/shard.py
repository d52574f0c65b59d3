"""1-D vertex (row) partition of the GCN aggregation over the GPUs of one node + halo exchange.

What shards: rows of A are independent; the only coupling is reading neighbour
rows of H owned elsewhere.

Partition (`deal_partition`).  Vertices are sorted by weight (out-degree + in-degree + a per-row term for
the dense work) and DEALT to the ranks in snake order (0..P-1, P-1..0, ...): every rank gets n/P +- 1
rows, the same degree mix, hence the same GEMM rows, the same non-zeros and -- what the xGMI links care about
-- statistically the same send volume to every peer.  (Contiguous ranges on an R-MAT graph gave one rank 3.6 M
tail rows and another 75 k hub rows, 52 k .. 451 k rows per link; see scripts/exp_halo.py.)  Inside a rank the
rows keep their ascending original order.  The partition is expressed as a relabelling `nid[v]`
(new id: rank p owns the contiguous new-id range [cuts[p], cuts[p+1])); world = 1 is the identity.

Rank p holds
  * CSR of its rows of A and CSR of its rows of A^T (built from the edges whose src / dst it owns).  The entries of a
    row are sorted by ORIGINAL column id -- the kernel adds neighbours in storage order, so the order stays the
    reference's descending-original-column order and the sharded result is bit-identical to the single-GPU one --
    and only then renumbered, value by value, to [local rows | halo rows];
  * halo lists: the sorted new ids of remote columns it reads (forward: out-neighbours, backward:
    in-neighbours), grouped by owner because new-id ranges are contiguous;
  * send lists: for every peer, which of its own rows that peer reads (exchanged once, at plan time).
Per aggregation there is ONE exchange step: pack rows by send list (gnnx_gather_rows_f32) -> all-to-all-v
(torch.distributed all_to_all_single with split sizes = grouped ncclSend/ncclRecv on RCCL; every pair of
GPUs has its own xGMI link) straight into the [halo] tail of the feature buffer -> local SpMM over
[local | halo].  Backward pulls too (rows of G for in-neighbours through the transposed CSR): no
scatter-add, no atomics, deterministic.  dW / dbias are summed with a small all-reduce.

Schedule (`ShardedBench.step`).  X and the upstream gradient G are both inputs of the layer step, and the forward
chain (X.W^T -> exchange H -> aggregate) and the backward chain (exchange G -> aggregate^T -> dH.W, dH^T.X) share
nothing but the links.  "overlap" issues both exchanges asynchronously on the communication stream and runs each
chain's compute under the other chain's exchange; every SpMM still runs over the complete [local | halo] buffer, so
the results are the same bits as in "sequential" (GEMM -> exchange -> SpMM -> ... on one stream).

This module is host-side index logic (torch tensor ops, device-agnostic so that the gloo/CPU tests drive
exactly this code) plus the exchange; all arithmetic goes through the C-ABI (ops.py).  The same plan exists
behind the C-ABI (gnnx_partition_deal / gnnx_halo_plan_*, include/gnnx.h) for the C++ host; the two are compared
array for array in tests/test_gpu_sharded_loopback.py.
"""
import torch


def balanced_cuts(weight, world):
    """Cut points [world+1] of contiguous ranges with ~equal total weight (weight: int64 per vertex)."""
    n = int(weight.numel())
    csum = torch.cumsum(weight.to(torch.int64), 0)
    total = int(csum[-1].item()) if n else 0
    targets = torch.tensor([total * (p + 1) // world for p in range(world - 1)], dtype=torch.int64, device=weight.device)
    inner = torch.searchsorted(csum, targets, right=False) + 1 if n else targets
    cuts = [0] + [min(n, int(c)) for c in inner.tolist()] + [n]
    for i in range(1, len(cuts)):  # monotone
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


SCRAMBLE_MUL = 2654435761   # prime, larger than any per-rank row count: k -> (k * SCRAMBLE_MUL) mod n_p is a bijection of [0, n_p)


def scramble_nid(owner, nid, cuts):
    """Second relabelling inside every rank's new-id range (mirrors gnnx_partition_scramble): position k of rank p's n_p rows
    moves to (k * SCRAMBLE_MUL) mod n_p.  R-MAT (and most synthetic power-law generators) put the hubs on the vertex ids with few
    one-bits -- feature rows whose ADDRESSES have few one-bits, so the bits that select the L2 channel / memory channel / Infinity-
    Cache slice are mostly zero and the hottest rows pile onto a few channels (DRAM-credit stalls of the L2 read interface 234 M vs
    129 M cycles per launch, profiles/r02_vertex_order_counters.json); spreading them is worth 28 % of the aggregation at 10 M / 100 M (19.0 -> 13.7 ms).  Row contents and
    the order of a row's entries are untouched: every vertex's result has the same bits, stored at row nid[v]."""
    cuts_t = torch.tensor(cuts, dtype=torch.int64, device=nid.device)
    o = owner.to(torch.int64)
    lo = cuts_t[o]
    cnt = cuts_t[o + 1] - lo
    k = nid.to(torch.int64) - lo
    # k < 2^31 and SCRAMBLE_MUL < 2^32: the product fits in int64
    return (lo + (k * SCRAMBLE_MUL) % cnt).to(torch.int32)


def deal_partition(weight, world, scramble=True):
    """Snake-deal the vertices, heaviest first, to `world` ranks.

    weight: int64 [n] (>= 0).  Returns (owner int32 [n], nid int32 [n], cuts list[world+1]):
    position k of the stable descending-weight order (ties: ascending id) goes to rank j = k % world in even rounds
    k // world and to world-1-j in odd rounds; nid numbers rank p's vertices cuts[p].. in ascending original id and, with
    `scramble`, then spreads them inside the range (scramble_nid).  Mirrors gnnx_partition_deal (+ gnnx_partition_scramble)."""
    n = int(weight.numel())
    dev = weight.device
    if world == 1 or n == 0:
        owner = torch.zeros(n, dtype=torch.int32, device=dev)
        nid = torch.arange(n, dtype=torch.int32, device=dev)
        cuts = [0] + [n] * world
        return owner, (scramble_nid(owner, nid, cuts) if scramble and n else nid), cuts
    w = weight.to(torch.int64)
    key = int(w.max().item()) - w                      # ascending stable sort == descending weight, ties by id
    order = torch.sort(key, stable=True).indices
    k = torch.arange(n, dtype=torch.int64, device=dev)
    j, r = k % world, k // world
    rank_of_pos = torch.where(r % 2 == 0, j, world - 1 - j).to(torch.int32)
    owner = torch.empty(n, dtype=torch.int32, device=dev)
    owner[order] = rank_of_pos
    counts = torch.bincount(owner.to(torch.int64), minlength=world).tolist()
    cuts = [0]
    for c in counts:
        cuts.append(cuts[-1] + int(c))
    nid = torch.empty(n, dtype=torch.int32, device=dev)
    for p in range(world):
        m = owner == p
        nid[m] = torch.arange(cuts[p], cuts[p + 1], dtype=torch.int32, device=dev)
    if scramble:
        nid = scramble_nid(owner, nid, cuts)
    return owner, nid, cuts


def contiguous_partition(weight, world):
    """The round-1 partition: contiguous original-id ranges of ~equal weight (kept for comparison, scripts/exp_halo.py)."""
    n = int(weight.numel())
    dev = weight.device
    cuts = balanced_cuts(weight, world)
    nid = torch.arange(n, dtype=torch.int32, device=dev)
    owner = (torch.searchsorted(torch.tensor(cuts[1:], dtype=torch.int64, device=dev), nid.to(torch.int64), right=True)
             .clamp_(max=world - 1).to(torch.int32))
    return owner, nid, cuts


ROUND_ROWS = 256 * 256   # one round of the transform's 256-row tiles over the 256 CUs


def chunk_bounds(n_rows, n_chunks):
    """Row-chunk boundaries [n_chunks + 1] of a rank's local rows: chunk k = rows [(n k) // K, (n (k+1)) // K) -- moved to the nearest
    multiple of ROUND_ROWS when chunks are at least two rounds long, so that a chunk's product is whole rounds of tiles (1.25 M rows in
    four chunks: 5 + 5 + 4 + 5.07 rounds instead of four times 4.77, each rounded up to 5).  A pure function of (n_rows, n_chunks):
    every rank derives every owner's boundaries from the partition's cuts."""
    n, K = int(n_rows), int(n_chunks)
    b = [(n * k) // K for k in range(K + 1)]
    if K > 1 and n // K >= 2 * ROUND_ROWS:
        b = [0] + [min(max((x + ROUND_ROWS // 2) // ROUND_ROWS * ROUND_ROWS, 0), n) for x in b[1:-1]] + [n]
    return b


class HaloSide:
    """One direction of the exchange (forward: columns of A; backward: columns of A^T).  Column ids are NEW ids.

    n_chunks = K > 1 (the pipelined exchange of the "training" schedule): every rank's local rows are cut into K row chunks
    (chunk_bounds), and both the halo tail of the receiver and the send buffer of the owner are laid out CHUNK-MAJOR --
    [chunk 0: from rank 0, rank 1, ... | chunk 1: ...], inside a (chunk, owner) segment ascending id -- so that the rows of chunk k can
    leave as soon as the producing GEMM has written chunk k: one all-to-all-v per chunk on contiguous segments of both buffers.  K = 1
    is the plain owner-major layout (what the C-ABI plan builds).  The [local | halo] renumbering of the columns follows the tail
    order; the order of a row's entries is untouched, so the bits are the same for every K."""

    def __init__(self, rowptr, colidx_nid, lo, hi, cuts, n_chunks=1):
        dev = colidx_nid.device
        P, K = len(cuts) - 1, int(n_chunks)
        self.n_local = hi - lo
        self.n_chunks = K
        self.rowptr = rowptr
        remote = (colidx_nid < lo) | (colidx_nid >= hi)
        halo_sorted = torch.unique(colidx_nid[remote])  # sorted ascending => grouped by owner
        self.halo_sorted = halo_sorted
        self.n_halo = int(halo_sorted.numel())
        # (owner, chunk at the owner) of every halo row: boundaries cuts[q] + chunk_bounds(n_q, K)[k], ascending over (q, k)
        bnd = torch.tensor([cuts[q] + b for q in range(P) for b in chunk_bounds(cuts[q + 1] - cuts[q], K)[:-1]], dtype=torch.int64, device=dev)
        j = torch.searchsorted(bnd, halo_sorted.to(torch.int64), right=True) - 1   # = owner * K + chunk
        owner_h, chunk_h = j // K, j % K
        order = torch.sort(chunk_h, stable=True).indices if K > 1 else torch.arange(self.n_halo, device=dev)
        self.halo = halo_sorted[order]                     # tail order: chunk-major, owner, ascending id
        inv = torch.empty_like(order)
        inv[order] = torch.arange(self.n_halo, device=dev)
        col = colidx_nid.to(torch.int64)
        local_id = col - lo
        pos_sorted = torch.searchsorted(halo_sorted.to(torch.int64), col).clamp_(max=max(self.n_halo - 1, 0))
        halo_id = (inv[pos_sorted] if self.n_halo else pos_sorted) + self.n_local
        self.colidx = torch.where(remote, halo_id, local_id).to(torch.int32)  # entry order untouched
        cnt = torch.bincount(chunk_h * P + owner_h, minlength=K * P).reshape(K, P) if self.n_halo else torch.zeros((K, P), dtype=torch.int64)
        self.recv_counts_k = [[int(v) for v in row] for row in cnt.tolist()]   # [chunk][owner]
        self.recv_counts = [sum(self.recv_counts_k[k][q] for k in range(K)) for q in range(P)]   # owner totals (requests, statistics)
        self.recv_off = [0]
        for k in range(K):
            self.recv_off.append(self.recv_off[-1] + sum(self.recv_counts_k[k]))
        self.send_counts = None     # peer totals
        self.send_counts_k = None   # [chunk][peer]
        self.send_off = None
        self.send_idx = None        # local row ids to pack, chunk-major then peer-major
        self.slot_table = None      # [n_local, 8] inverse of send_idx (ops.slot_table), made by a runner that packs from the producer's side

    def exchange_requests(self, dist, cuts, rank, world):
        """Tell every owner which of its rows this rank reads; learn which of mine the peers read."""
        dev = self.halo.device
        K = self.n_chunks
        rc = torch.tensor(self.recv_counts, dtype=torch.int64, device=dev)
        sc = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(sc, rc)
        self.send_counts = sc.tolist()
        want = torch.empty(int(sum(self.send_counts)), dtype=self.halo.dtype, device=dev)
        dist.all_to_all_single(want, self.halo_sorted.contiguous(), self.send_counts, self.recv_counts)   # owner-major, ascending id
        rows = (want - cuts[rank]).to(torch.int32)
        assert rows.numel() == 0 or (int(rows.min()) >= 0 and int(rows.max()) < self.n_local)
        self.set_send_lists(rows, self.send_counts)

    def set_send_lists(self, rows_peer_major, send_counts):
        """rows_peer_major: my local rows every peer reads, peer-major, ascending inside a peer (what the requests deliver)."""
        dev = rows_peer_major.device
        K, P = self.n_chunks, len(send_counts)
        self.send_counts = [int(c) for c in send_counts]
        if K == 1:
            self.send_idx = rows_peer_major
            self.send_counts_k = [list(self.send_counts)]
        else:
            b = torch.tensor(chunk_bounds(self.n_local, K)[1:-1], dtype=torch.int64, device=dev)
            chunk = torch.searchsorted(b, rows_peer_major.to(torch.int64), right=True)            # chunk of my own row
            peer = torch.repeat_interleave(torch.arange(P, device=dev), torch.tensor(self.send_counts, device=dev))
            order = torch.sort(chunk * P + peer, stable=True).indices                                # (chunk, peer), ascending row inside
            self.send_idx = rows_peer_major[order].contiguous()
            cnt = torch.bincount(chunk * P + peer, minlength=K * P).reshape(K, P)
            self.send_counts_k = [[int(v) for v in row] for row in cnt.tolist()]
        self.send_off = [0]
        for k in range(K):
            self.send_off.append(self.send_off[-1] + sum(self.send_counts_k[k]))


class HostStagedDist:
    """torch.distributed look-alike that moves device tensors through host memory over a CPU process group (gloo).

    REHEARSAL transport, never the measured one: it lets the whole multi-process job -- torch.distributed.run, one process per
    rank, rendezvous, partition, halo plan, both schedules, parameter all-reduce, the JSON line -- run on a box with ONE GPU
    (every rank on device 0), which is all the builder had; on a multi-GPU node the backend is nccl (= RCCL) and this class is
    not used.  Collectives are synchronous (the device stream is drained first); async_op returns a completed handle."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    class _Work:
        def wait(self):
            return True

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, async_op=False):
        torch.cuda.synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        self._d.all_to_all_single(o, inp.detach().cpu().contiguous(), output_split_sizes, input_split_sizes)
        out.copy_(o)
        torch.cuda.synchronize()
        return self._Work() if async_op else None

    def all_reduce(self, t, op=None):
        torch.cuda.synchronize()
        h = t.detach().cpu()
        self._d.all_reduce(h) if op is None else self._d.all_reduce(h, op=op)
        t.copy_(h)
        torch.cuda.synchronize()

    def broadcast(self, t, src):
        h = t.detach().cpu()
        self._d.broadcast(h, src)
        t.copy_(h)

    def barrier(self):
        torch.cuda.synchronize()
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


class _Done:
    """Handle of an exchange that has already completed on the caller's stream."""

    def wait(self):
        return True


class _All:
    """Handle of several exchanges: wait() waits for each."""

    def __init__(self, hs):
        self.hs = hs

    def wait(self):
        for h in self.hs:
            h.wait()
        return True


def exchange_rows(dist, side, buf, n_feat, pack, send_buf=None, native=None, async_op=False, chunk=None, scatter=None, prepacked=False):
    """buf: [n_local + n_halo, F]; rows [:n_local] are this rank's; fills rows [n_local:] from the owners.
    pack(src_rows_view, idx, out) gathers rows (gnnx_gather_rows_f32 on GPU).
    native: a NativeComm => the all-to-all-v runs through the C-ABI (gnnx_halo_exchange_f32, RCCL send/recv group)
    instead of torch.distributed.all_to_all_single (the same RCCL underneath).
    async_op: return a handle at once; handle.wait() orders the CALLER's current stream behind the exchange (the
    send buffer must then stay untouched until the wait).
    chunk: None = every row chunk of the plan, one after the other (a plan with one chunk: one all-to-all-v); k = only chunk k
    (HaloSide, n_chunks > 1): the local rows of chunk k must have been written, the others may still be in the making.
    scatter(src_rows_view, slot_table, out): the pack from the producer's side (gnnx_rows_to_slots_f32: every local row read once and
    written to each of its slots) -- used instead of `pack` when the side carries a slot table (one chunk); the same send buffer.
    prepacked: the caller has filled send_buf already (the pack rode in another pass: ShardedBench's dbias + pack of G).
    Returns (send_buf, handle)."""
    n_send = int(side.send_idx.numel())
    if send_buf is None or send_buf.shape[0] < n_send:
        send_buf = torch.empty((max(n_send, 1), n_feat), dtype=buf.dtype, device=buf.device)
    handles = []
    for k in (range(side.n_chunks) if chunk is None else [chunk]):
        s0, s1 = side.send_off[k], side.send_off[k + 1]
        r0, r1 = side.n_local + side.recv_off[k], side.n_local + side.recv_off[k + 1]
        out = send_buf[s0:s1]
        if s1 > s0 and not prepacked:
            if scatter is not None and chunk is None and side.n_chunks == 1 and getattr(side, "slot_table", None) is not None:
                scatter(buf[: side.n_local], side.slot_table, out)
            else:
                pack(buf[: side.n_local], side.send_idx[s0:s1], out)
        recv = buf[r0:r1]
        sc, rc = side.send_counts_k[k], side.recv_counts_k[k]
        if native is not None:
            h = native.halo_exchange(out, sc, recv, rc, n_feat, async_op=async_op)
        else:
            h = dist.all_to_all_single(recv, out, rc, sc, async_op=async_op) if async_op else dist.all_to_all_single(recv, out, rc, sc)
        if async_op and h is not None:
            handles.append(h)
    return send_buf, (_All(handles) if handles else _Done())


class NativeComm:
    """RCCL communicator owned by the C-ABI (gnnx_comm_*).  The 128-byte id is created on rank 0 and shipped with
    one torch.distributed broadcast (setup only); the per-step exchange and all-reduce then bypass torch.
    Asynchronous exchanges run on a communication stream of their own, ordered against the caller's stream with events."""

    def __init__(self, capi, ops, dist, rank, world, dev):
        import ctypes as C
        self.C, self.capi, self.ops, self.world = C, capi, ops, world
        idbuf = (C.c_char * 128)()
        if rank == 0:
            capi.call("gnnx_comm_unique_id", C.cast(idbuf, C.c_void_p))
        t = torch.frombuffer(bytearray(bytes(idbuf)), dtype=torch.uint8).to(dev)
        if world > 1:
            dist.broadcast(t, 0)
        raw = bytes(t.cpu().numpy().tobytes())
        self.h = C.c_void_p()
        capi.call("gnnx_comm_init", C.byref(self.h), world, rank, C.c_char_p(raw))
        self.comm_stream = torch.cuda.Stream(device=dev)

    class _Handle:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    def halo_exchange(self, send, send_counts, recv, recv_counts, n_feat, async_op=False):
        C = self.C
        sc = (C.c_int64 * self.world)(*[int(v) for v in send_counts])
        rc = (C.c_int64 * self.world)(*[int(v) for v in recv_counts])
        if not async_op:
            self.capi.call("gnnx_halo_exchange_f32", self.h, self.ops._ptr(send) if send.numel() else None, sc,
                           self.ops._ptr(recv) if recv.numel() else None, rc, int(n_feat), self.ops._stream())
            return None
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        self.capi.call("gnnx_halo_exchange_f32", self.h, self.ops._ptr(send) if send.numel() else None, sc,
                       self.ops._ptr(recv) if recv.numel() else None, rc, int(n_feat), C.c_void_p(self.comm_stream.cuda_stream))
        ev = torch.cuda.Event()
        ev.record(self.comm_stream)
        return NativeComm._Handle(ev)

    def allreduce(self, t):
        self.capi.call("gnnx_allreduce_sum_f32", self.h, self.ops._ptr(t), t.numel(), self.ops._stream())

    def __del__(self):
        try:
            self.capi.lib().gnnx_comm_destroy(self.h)
        except Exception:
            pass


def sharded_bn_stats(dist, ops, X_local, n_total):
    """Batch statistics of BatchNorm over the WHOLE graph from this rank's rows ("cross-shard statistics"):
    two [F] all-reduces -- the mean, then the centred squares against the GLOBAL mean (the exact two-pass biased variance of
    gnnx_bn_stats_f32 / reference nn.cpp:303,312, only summed in shard order).  Returns (mean, var) for the fused SpMM prologue."""
    inv_n = 1.0 / float(n_total)
    mean = ops.bn_partial(X_local, None, inv_n)
    dist.all_reduce(mean)
    var = ops.bn_partial(X_local, mean, inv_n)
    dist.all_reduce(var)
    return mean, var


class ShardPlan:
    """Everything rank `rank` needs for sharded forward + backward aggregation.

    csr_builder(src, dst, n_rows, n_cols_hint) -> (rowptr int32 [n_rows+1], colidx int32 [nnz]) must apply the
    reference's adjacency semantics (dedupe, self-loop strip, (src,dst) order): gnnx_csr_from_coo on the GPU.
    Rows are LOCAL ids, columns are ORIGINAL ids in the builder (so a row's entries are stored in the reference's order)
    and are mapped to new ids, then to [local | halo], afterwards.
    partition: "deal" (default: degree-sorted snake deal, rows spread inside every rank's range), "deal-ascending" (the same
    owners, a rank's rows in ascending original id), "contiguous", or a ready (owner, nid, cuts) triple.
    """

    def __init__(self, src, dst, n_nodes, rank, world, dist, csr_builder, partition="deal", row_weight=1, n_chunks=1):
        dev = src.device
        self.rank, self.world, self.n_nodes = rank, world, n_nodes
        self.n_chunks = int(n_chunks) if world > 1 else 1   # row chunks of the pipelined exchange (HaloSide)
        if isinstance(partition, str):
            # cost model of a vertex: `row_weight` for the dense work on its row (three GEMM passes, ~6 F^2 flop) plus one
            # unit per incident edge for the two aggregations (4 F bytes each): row_weight ~ 0.08 F on MI355X
            w = torch.bincount(src.to(torch.int64), minlength=n_nodes) + torch.bincount(dst.to(torch.int64), minlength=n_nodes) \
                + int(row_weight)
            if partition == "deal":
                partition = deal_partition(w, world)
            elif partition == "deal-ascending":
                partition = deal_partition(w, world, scramble=False)
            else:
                partition = contiguous_partition(w, world)
        self.owner, self.nid, self.cuts = partition
        cuts = self.cuts
        lo, hi = cuts[rank], cuts[rank + 1]
        self.lo, self.hi, self.n_local = lo, hi, hi - lo
        nid = self.nid
        self.verts = self.orig_ids(torch.arange(lo, hi, dtype=torch.int64, device=dev))   # original id of local row k
        # self loops must be dropped on GLOBAL ids (rows are renumbered below), duplicates collapse in the builder
        keep = src != dst
        mine = keep & (self.owner[src.long()] == rank)
        rp, ci = csr_builder((nid[src[mine].long()] - lo).to(torch.int32), dst[mine].to(torch.int32), self.n_local, n_nodes)
        self.row_chunks = chunk_bounds(self.n_local, self.n_chunks)
        self.fwd = HaloSide(rp, nid[ci.long()], lo, hi, cuts, self.n_chunks)
        mine_t = keep & (self.owner[dst.long()] == rank)
        rpt, cit = csr_builder((nid[dst[mine_t].long()] - lo).to(torch.int32), src[mine_t].to(torch.int32), self.n_local, n_nodes)
        self.bwd = HaloSide(rpt, nid[cit.long()], lo, hi, cuts, self.n_chunks)
        self.nnz_local = int(ci.numel())
        if world > 1 and dist is not None:
            self.fwd.exchange_requests(dist, cuts, rank, world)
            self.bwd.exchange_requests(dist, cuts, rank, world)
        elif world > 1:
            pass  # plan without a process group (tests / scripts fill the halo rows themselves)
        else:
            for s in (self.fwd, self.bwd):
                s.set_send_lists(torch.empty(0, dtype=torch.int32, device=dev), [0])
        self.s_ext = self.norm = self.norm_ext_bwd = None

    def orig_ids(self, new_ids):
        """Original vertex ids of new ids (e.g. of a HaloSide.halo list)."""
        if getattr(self, "_inv", None) is None:
            inv = torch.empty(self.n_nodes, dtype=torch.int64, device=self.nid.device)
            inv[self.nid.long()] = torch.arange(self.n_nodes, dtype=torch.int64, device=self.nid.device)
            self._inv = inv
        return self._inv[new_ids.long()]

    def compute_norm(self, dist, degree_norm, pack):
        """s for local rows from local degrees, s of halo columns by one exchange, then norm (graph.cpp:177-185);
        finally norm of the backward halo (colscale of the transposed aggregation)."""
        f, b = self.fwd, self.bwd
        dev = f.rowptr.device
        s_ext = torch.zeros((f.n_local + f.n_halo, 1), dtype=torch.float32, device=dev)
        degree_norm(f.rowptr, f.colidx, self.n_local, s_out=s_ext[: f.n_local], s_cols=None, norm_out=None)
        if self.world > 1:
            exchange_rows(dist, f, s_ext, 1, pack)
        self.norm = torch.zeros(self.n_local, dtype=torch.float32, device=dev)
        degree_norm(f.rowptr, f.colidx, self.n_local, s_out=None, s_cols=s_ext, norm_out=self.norm)
        self.s_ext = s_ext
        nb = torch.zeros((b.n_local + b.n_halo, 1), dtype=torch.float32, device=dev)
        nb[: b.n_local, 0] = self.norm
        if self.world > 1:
            exchange_rows(dist, b, nb, 1, pack)
        self.norm_ext_bwd = nb.reshape(-1).contiguous()
        return self.norm


# ---------------------------------------------------------------------------------------------------
class ShardedGcnStack:
    """The multi-layer training step (ops.GcnStack) on one rank's rows of a 1-D vertex partition: per layer one halo exchange forward
    (rows of H = h W^T) and one backward (rows of the upstream gradient), the loss over the global batch (each rank's term of the
    mean, 1 / N_total in the gradient), parameter gradients all-reduced, the same SGD step on every rank.  Per vertex the forward
    values are the unsharded stack's bit for bit (same fmaf chains in the dense products, same summation order in the
    aggregations); dW / db sum the ranks' contributions in rank order (rounding-level)."""

    def __init__(self, ops, dist, plan, dims, seed=0, chunk=1024, native=None):
        self.ops, self.dist, self.p, self.native = ops, dist, plan, native
        dev = plan.fwd.rowptr.device
        self.dims = list(dims)
        L = len(dims) - 1
        self.W = [ops.uniform_pm1(seed + 2 * l, (dims[l + 1], dims[l]), scale=dims[l] ** -0.5, device=dev) for l in range(L)]
        self.b = [torch.zeros(dims[l + 1], dtype=torch.float32, device=dev) for l in range(L)]
        self.dW = [torch.zeros_like(w) for w in self.W]
        self.db = [torch.zeros_like(b) for b in self.b]
        nl, p = plan.n_local, plan
        self.pack = lambda rows, idx, out: ops.gather_rows(rows, idx, out=out)
        if p.norm is None:
            raise ValueError("ShardPlan.compute_norm must have run")
        self.norm_nz_bwd = ops.gather_rows(p.norm_ext_bwd.reshape(-1, 1), p.bwd.colidx).reshape(-1)
        fmax = max(dims[1:])
        self.plan_f = ops.SpmmPlan(p.fwd.rowptr, chunk, fmax) if chunk > 0 else None
        self.plan_b = ops.SpmmPlan(p.bwd.rowptr, chunk, fmax) if chunk > 0 else None
        # [local | halo] buffers, one per layer output width and direction (re-used every step)
        self.Hext = [torch.empty((nl + p.fwd.n_halo, dims[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
        self.Gext = [torch.empty((nl + p.bwd.n_halo, dims[l + 1]), dtype=torch.float32, device=dev) for l in range(L)]
        ns = max(int(p.fwd.send_idx.numel()), int(p.bwd.send_idx.numel()), 1)
        self.send = torch.empty((ns, fmax), dtype=torch.float32, device=dev)
        self._saved = None
        # a layer's transform packs its send rows in its own epilogue (gnnx_gemm_nt_rows_to_slots_f32) wherever the output rows are
        # 16-byte pieces; the table does not depend on the width (chunked layouts too: a chunk's segment holds rows of that chunk only)
        self.table_f = self.table_b = None
        if p.world > 1 and hasattr(ops, "slot_table"):
            if int(p.fwd.send_idx.numel()):
                self.table_f = ops.slot_table(p.fwd.send_idx, nl)
            if int(p.bwd.send_idx.numel()):   # the gradient's rows are packed from the producer's side too (every local row read once)
                self.table_b = ops.slot_table(p.bwd.send_idx, nl)

    def _send_view(self, F):
        return self.send.view(-1)[: self.send.shape[0] * F].view(-1, F)

    def _packs_in_epilogue(self, F):
        return self.table_f is not None and F % 4 == 0 and F // 4 <= 256 and 256 % (F // 4) == 0

    def _exchange_gradient(self, buf, F):
        """The backward side's exchange of the rows at the head of `buf`: packed by gnnx_rows_to_slots_f32 where the rows are 16-byte
        pieces, by the send list otherwise; asynchronous."""
        if self.p.world > 1 and self.table_b is not None and F % 4 == 0 and F // 4 <= 256 and 256 % (F // 4) == 0:
            self.ops.rows_to_slots(buf[: self.p.n_local], self.table_b, self._send_view(F))
            return self._exchange(self.p.bwd, buf, F, async_op=True, prepacked=True)
        return self._exchange(self.p.bwd, buf, F, async_op=True)

    def _exchange(self, side, buf, F, async_op=False, chunk=None, prepacked=False):
        if self.p.world > 1:
            return exchange_rows(self.dist, side, buf, F, self.pack, self._send_view(F), self.native, async_op=async_op, chunk=chunk,
                                 prepacked=prepacked)[1]
        return _Done()

    def forward(self, X_local):
        """Per layer: the transform runs row chunk by row chunk (ShardPlan.row_chunks) and every finished chunk's rows leave at once
        (pack -> asynchronous all-to-all-v of that chunk) while the next chunk is being multiplied -- the one overlap a training step's
        forward allows, since the aggregation needs the whole halo.  One chunk (n_chunks = 1): GEMM -> exchange -> SpMM.  The chunked
        product is the same fmaf chain per row, the chunk-major tail is only another numbering: same bits."""
        ops, p, nl = self.ops, self.p, self.p.n_local
        L = len(self.W)
        saved, h = [], X_local
        for l in range(L):
            F = self.dims[l + 1]
            handles = []
            for k in range(p.n_chunks):
                r0, r1 = p.row_chunks[k], p.row_chunks[k + 1]
                packed = self._packs_in_epilogue(F)
                if r1 > r0 and packed:
                    ops.linear_fwd_rows_to_slots(h[r0:r1], self.W[l], self.Hext[l][r0:r1], self.table_f[r0:r1], self._send_view(F))
                elif r1 > r0:
                    ops.linear_fwd(h[r0:r1], self.W[l], out=self.Hext[l][r0:r1])
                handles.append(self._exchange(p.fwd, self.Hext[l], F, async_op=True, chunk=k, prepacked=packed))
            for hd in handles:
                hd.wait()
            Y = ops.spmm(p.fwd.rowptr, p.fwd.colidx, self.Hext[l], rowscale=p.norm, bias=self.b[l], plan=self.plan_f, n_rows=nl,
                         relu_out=l + 1 < L)
            saved.append((h, Y))
            h = Y
        self._saved = saved
        return h

    def loss_and_backward(self, logits_local, target_local, n_total):
        """softmax cross-entropy over the global batch + backward through the stack; returns the global mean loss (1-element
        tensor, all-reduced).  Parameter gradients are all-reduced: every rank ends with the global dW / db."""
        ops, p, nl, dist = self.ops, self.p, self.p.n_local, self.dist
        L = len(self.W)
        loss, G = ops.softmax_ce(logits_local, target_local, colsum_out=self.db[L - 1], n_total=n_total, grad_out=self.Gext[L - 1][:nl])
        pending = self._exchange_gradient(self.Gext[L - 1], self.dims[L])
        for l in reversed(range(L)):
            h, _ = self._saved[l]
            pending.wait()
            dH = ops.spmm(p.bwd.rowptr, p.bwd.colidx, self.Gext[l], vals=self.norm_nz_bwd, plan=self.plan_b, n_rows=nl)
            if l > 0:   # G_{l-1} = (dH . W_l) (.) (Y_{l-1} > 0) straight into the next exchange's buffer, db_{l-1} from the same epilogue;
                # its exchange starts at once and the layer's weight gradient is computed under it
                ops.gemm_relu_colsum(dH, self.W[l], h, out=self.Gext[l - 1][:nl], colsum_out=self.db[l - 1])
                pending = self._exchange_gradient(self.Gext[l - 1], self.dims[l])
            ops.gemm(dH, h, transA=True, out=self.dW[l])
        if p.world > 1:
            for t in self.dW + self.db + [loss]:
                if self.native is not None and t is not loss:
                    self.native.allreduce(t)
                else:
                    dist.all_reduce(t)
        return loss

    def step(self, lr, weight_decay=0.0):
        for prm, gr in zip(self.W + self.b, self.dW + self.db):
            self.ops.sgd_step(prm, gr, lr, weight_decay)


class ShardedTrain:
    """bench.py runner for `--train-layers L` on N > 1 ranks: the L-layer training step (forward, softmax-CE over the global batch,
    backward, parameter all-reduce, SGD) on the sharded graph -- ShardedGcnStack: the dependence forward -> loss -> backward is the
    real one, and what overlaps is what it leaves: a layer's transform is pipelined with the sending of its own rows (row chunks), a
    layer's weight-gradient product runs under the exchange of the gradient for the layer below."""

    def __init__(self, ops, capi, pkg, dist, dev, rank, world, n, e, F, abc, seed, chunk, layers, partition="deal", n_chunks=4):
        import ctypes as C
        self.C, self.ops, self.capi, self.dist, self.F = C, ops, capi, dist, F
        self.rank, self.world, self.layers = rank, world, layers
        if abc is None:
            s, d = pkg.synth.uniform_edges(seed, n, e)
            src, dst = torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)
        else:
            src, dst = ops.rmat_edges(seed, n, e, *abc, device=dev)

        def builder_once(s_, d_, n_rows, n_cols):
            rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
            return rp[: n_rows + 1].contiguous(), ci

        self.plan = p = ShardPlan(src, dst, n, rank, world, dist, builder_once, partition=partition, row_weight=max(1, round(0.078 * F)),
                                  n_chunks=n_chunks)
        del src, dst
        p.owner = p.nid = None
        ops._ws_cache.clear()
        torch.cuda.empty_cache()

        def degree_norm(rowptr, colidx, n_rows, s_out, s_cols, norm_out):
            capi.call("gnnx_degree_norm_f32", ops._ptr(rowptr), ops._ptr(colidx), n_rows, ops._ptr(s_out), ops._ptr(s_cols),
                      ops._ptr(norm_out), ops._stream())

        p.compute_norm(dist, degree_norm, lambda rows, idx, out: ops.gather_rows(rows, idx, out=out))
        t = torch.tensor([p.nnz_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        self.nnz_total = int(t.item())
        self.n_total = n
        self.net = ShardedGcnStack(ops, dist, p, [F] * (layers + 1), seed=seed + 100, chunk=chunk)
        nl = p.n_local
        self.X = ops.uniform_pm1(seed + 10 + 1000 * rank, (nl, F), device=dev)
        self.target = ((torch.arange(nl, device=dev, dtype=torch.int64) + p.lo) * 7 + 3).remainder(F).to(torch.int32)
        self.names = ["train_step"]
        self.ev = []

    def step(self, timed=False):
        stream = self.C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if timed:
            a, b = self.capi.Event(), self.capi.Event()
            a.record(stream)
        logits = self.net.forward(self.X)
        self.net.loss_and_backward(logits, self.target, self.n_total)
        self.net.step(lr=1e-3)
        if timed:
            b.record(stream)
            self.ev.append([(a, b)])

    def kernel_times(self):
        import numpy as np
        return {"train_step": float(np.mean([s[0][0].elapsed_ms(s[0][1]) for s in self.ev])) if self.ev else None}

    def aggregation_stats(self):
        p = self.plan
        return {"local_rows": p.n_local, "local_nnz": None, "features": None, "spmm_fwd_ms": None, "halo_rows_fwd": p.fwd.n_halo,
                "halo_rows_bwd": p.bwd.n_halo,
                "schedule": f"training step: per layer the transform in {p.n_chunks} row chunks, every chunk's rows sent while the next is "
                            "multiplied; backward exchanges under the weight-gradient products"}


class ShardedBench:
    """bench.py runner for N > 1 ranks: same synthetic graph as the single-GPU workload, sharded."""

    def __init__(self, ops, capi, pkg, dist, dev, rank, world, n, e, F, abc, seed, chunk, native_comm=False,
                 global_inputs=False, schedule="overlap", partition="deal", replicate_input_halo=False, n_chunks=None):
        import ctypes as C
        self.C = C
        if n_chunks is None:   # the pipelined exchange belongs to the training schedule; the other two send a whole side at once
            n_chunks = 4 if schedule == "training" else 1
        self.native = NativeComm(capi, ops, dist, rank, world, dev) if native_comm else None
        self.ops, self.capi, self.dist, self.F = ops, capi, dist, F
        self.rank, self.world = rank, world
        self.schedule = schedule
        if abc is None:
            s, d = pkg.synth.uniform_edges(seed, n, e)
            src, dst = torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)
        else:
            src, dst = ops.rmat_edges(seed, n, e, *abc, device=dev)

        def builder_once(s_, d_, n_rows, n_cols):
            rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)  # self loops handled on global ids
            return rp[: n_rows + 1].contiguous(), ci

        self.plan = p = ShardPlan(src, dst, n, rank, world, dist, builder_once, partition=partition,
                                  row_weight=max(1, round(0.078 * F)), n_chunks=n_chunks)
        del src, dst
        p.owner = p.nid = None  # 2 x 4 N bytes the step does not need
        ops._ws_cache.clear()
        torch.cuda.empty_cache()

        def degree_norm(rowptr, colidx, n_rows, s_out, s_cols, norm_out):
            capi.call("gnnx_degree_norm_f32", ops._ptr(rowptr), ops._ptr(colidx), n_rows, ops._ptr(s_out), ops._ptr(s_cols),
                      ops._ptr(norm_out), ops._stream())

        self.pack = lambda rows, idx, out: ops.gather_rows(rows, idx, out=out)
        p.compute_norm(dist, degree_norm, self.pack)
        # the per-step packs run from the producer's side (every local row read once, written to each of its send slots); the pack of
        # the upstream gradient rides in the pass that sums its columns (dbias).  One-chunk layouts of 16-byte row pieces only.
        # (chunked layouts too: a row's slots are positions in the chunk-major send buffer, and chunk k's segment holds rows of chunk k only)
        self.scatter = None
        self.tables = False
        if F % 4 == 0 and 256 % (F // 4) == 0 and F // 4 <= 256:
            for side in (p.fwd, p.bwd):
                side.slot_table = ops.slot_table(side.send_idx, p.n_local)
            self.tables = p.fwd.slot_table is not None and p.bwd.slot_table is not None
            if self.tables and n_chunks == 1:
                self.scatter = lambda rows, table, out: ops.rows_to_slots(rows, table, out)
        t = torch.tensor([p.nnz_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        self.nnz_total = int(t.item())
        # per-non-zero source scale of the backward aggregation (a coalesced stream; same arithmetic as colscale)
        self.norm_nz_bwd = ops.gather_rows(p.norm_ext_bwd.reshape(-1, 1), p.bwd.colidx).reshape(-1)
        self.plan_f = ops.SpmmPlan(p.fwd.rowptr, chunk, F) if chunk > 0 else None
        self.plan_b = ops.SpmmPlan(p.bwd.rowptr, chunk, F) if chunk > 0 else None
        nl = p.n_local
        if global_inputs:  # my rows of the SAME X / G a single-GPU run generates (tests compare against it)
            self.X = ops.uniform_pm1(seed + 10, (n, F), device=dev)[p.verts].clone()
        else:
            self.X = ops.uniform_pm1(seed + 10 + 1000 * rank, (nl, F), device=dev)
        self.W = ops.uniform_pm1(seed + 11, (F, F), scale=F ** -0.5, device=dev)
        self.bias = torch.zeros(F, dtype=torch.float32, device=dev)
        self.Hext = torch.empty((nl + p.fwd.n_halo, F), dtype=torch.float32, device=dev)   # [local | halo]
        self.Gext = torch.empty((nl + p.bwd.n_halo, F), dtype=torch.float32, device=dev)
        if global_inputs:
            self.Gext[:nl] = ops.uniform_pm1(seed + 12, (n, F), device=dev)[p.verts]
        else:
            self.Gext[:nl] = ops.uniform_pm1(seed + 12 + 1000 * rank, (nl, F), device=dev)
        self.out = torch.empty((nl, F), dtype=torch.float32, device=dev)
        self.dH = torch.empty((nl, F), dtype=torch.float32, device=dev)
        self.dX = torch.empty((nl, F), dtype=torch.float32, device=dev)
        self.dW = torch.empty((F, F), dtype=torch.float32, device=dev)
        self.dbias = torch.empty(F, dtype=torch.float32, device=dev)
        # one send buffer per direction: in the overlap schedule both exchanges are in flight at once
        self.send_f = torch.empty((max(int(p.fwd.send_idx.numel()), 1), F), dtype=torch.float32, device=dev)
        self.send_b = torch.empty((max(int(p.bwd.send_idx.numel()), 1), F), dtype=torch.float32, device=dev)
        # OPT-IN, first-layer only (the layer input is DATA, the same rows every step): the halo rows of X are fetched once, here,
        # and H = X . W^T is computed for [local | halo] rows on every rank -- the halo rows' products are the very fmaf chains
        # their owners compute (same bits) -- so the step has ONE exchange (the upstream gradient) instead of two.
        self.replicate = bool(replicate_input_halo)
        if self.replicate:
            self.Xext = torch.empty((nl + p.fwd.n_halo, F), dtype=torch.float32, device=dev)
            self.Xext[:nl] = self.X
            if world > 1:
                exchange_rows(dist, p.fwd, self.Xext, F, self.pack, self.send_f, self.native)
            torch.cuda.synchronize()
        self.set_schedule(schedule)
        self.ev = []

    def set_schedule(self, schedule):
        assert schedule in ("overlap", "sequential", "training")
        self.schedule = schedule
        if getattr(self, "replicate", False):
            self.names = ["pack_send_bwd", "gemm_xwT_local_and_halo", "spmm_fwd", "colsum", "wait_halo_bwd", "spmm_bwd", "gemm_dX", "gemm_dW",
                          "allreduce"]
        elif schedule == "sequential":
            self.names = ["gemm_xwT", "halo_fwd", "spmm_fwd", "colsum", "halo_bwd", "spmm_bwd", "gemm_dX", "gemm_dW", "allreduce"]
            if getattr(self, "scatter", None) is not None:
                self.names[3:5] = ["colsum_pack_halo_bwd"]
                self.names[0] = "gemm_xwT_pack_send_fwd"
        elif schedule == "training":
            self.names = ["gemm_xwT_pack_send_chunks", "wait_halo_fwd", "spmm_fwd", "pack_send_bwd_chunks", "colsum", "wait_halo_bwd", "spmm_bwd",
                          "gemm_dX", "gemm_dW", "allreduce"]
            if getattr(self, "tables", False):   # dbias and the pack of G in one pass, then the chunks' exchanges
                self.names[3:5] = ["colsum_pack_bwd", "send_bwd_chunks"]
        else:
            self.names = ["pack_send_bwd", "gemm_xwT", "pack_send_fwd", "colsum", "wait_halo_bwd", "spmm_bwd", "gemm_dX", "gemm_dW",
                          "wait_halo_fwd", "spmm_fwd", "allreduce"]
            if getattr(self, "scatter", None) is not None:   # dbias and the pack of G in one pass
                # ... and the pack of H in the epilogue of the product that makes it
                self.names = ["colsum_pack_send_bwd", "gemm_xwT_pack_send_fwd", "send_fwd", "wait_halo_bwd", "spmm_bwd", "gemm_dX", "gemm_dW",
                              "wait_halo_fwd", "spmm_fwd", "allreduce"]
        self.ev = []

    def _reduce_params(self):
        if self.native is not None:
            self.native.allreduce(self.dW)
            self.native.allreduce(self.dbias)
        else:
            self.dist.all_reduce(self.dW)
            self.dist.all_reduce(self.dbias)

    def step(self, timed=False):
        ops, p, dist, nl = self.ops, self.plan, self.dist, self.plan.n_local
        stream = self.C.c_void_p(torch.cuda.current_stream().cuda_stream)
        evs = []

        def run(fn):
            if timed:
                a, b = self.capi.Event(), self.capi.Event()
                a.record(stream)
                r = fn()
                b.record(stream)
                evs.append((a, b))
                return r
            return fn()

        Hl, Gl = self.Hext[:nl], self.Gext[:nl]
        spmm_f = lambda: ops.spmm(p.fwd.rowptr, p.fwd.colidx, self.Hext, out=self.out, rowscale=p.norm, bias=self.bias,  # noqa: E731
                                  plan=self.plan_f, n_rows=nl)
        spmm_b = lambda: ops.spmm(p.bwd.rowptr, p.bwd.colidx, self.Gext, out=self.dH, vals=self.norm_nz_bwd,  # noqa: E731
                                  plan=self.plan_b, n_rows=nl)
        def colsum_pack_exchange_b(async_op=False):
            # dbias = colsum(G) and the pack of G's rows in ONE pass over G (gnnx_rows_to_slots_f32), then the exchange of the packed rows
            if int(p.bwd.send_idx.numel()):
                ops.rows_to_slots(Gl, p.bwd.slot_table, self.send_b, colsum_out=self.dbias)
            else:
                ops.colsum(Gl, out=self.dbias)
            return exchange_rows(dist, p.bwd, self.Gext, self.F, self.pack, self.send_b, self.native, async_op=async_op, prepacked=True)

        def transform():
            # H = X . W^T; with a slot table the rows other ranks need leave for the send buffer from the product's epilogue
            if self.scatter is not None and int(p.fwd.send_idx.numel()):
                return ops.linear_fwd_rows_to_slots(self.X, self.W, Hl, p.fwd.slot_table, self.send_f)
            return ops.linear_fwd(self.X, self.W, out=Hl)

        if self.replicate:
            # one exchange (G, asynchronous); the whole forward chain -- transform of local AND halo rows, aggregation -- under it
            _, hb = run(lambda: exchange_rows(dist, p.bwd, self.Gext, self.F, self.pack, self.send_b, self.native, async_op=True))
            run(lambda: ops.linear_fwd(self.Xext, self.W, out=self.Hext))
            run(spmm_f)
            run(lambda: ops.colsum(Gl, out=self.dbias))
            run(hb.wait)
            run(spmm_b)
            run(lambda: ops.gemm(self.dH, self.W, out=self.dX))
            run(lambda: ops.gemm(self.dH, self.X, transA=True, out=self.dW))
            run(self._reduce_params)
        elif self.schedule == "training":
            # The dependence of a real training step -- forward -> loss -> backward (reference tensor.h:260-276: backward starts from the
            # loss) -- is honoured: nothing of the backward chain is issued before the forward aggregation, although G is a given input
            # here.  What overlaps is what that dependence leaves:
            #   forward:  the transform runs in row chunks and every finished chunk's rows are packed and sent at once, while the next
            #             chunk is multiplied (chunk-major halo tail: HaloSide);
            #   backward: G's rows leave chunk by chunk (pack k+1 while chunk k is on the links), dbias = colsum(G) under the exchange.
            # (A multi-layer step also has dW_l under the exchange of G_{l-1}: ShardedGcnStack.loss_and_backward.)
            packed_f = self.tables and int(p.fwd.send_idx.numel()) > 0
            packed_b = self.tables and int(p.bwd.send_idx.numel()) > 0

            def fwd_pipeline():
                hs = []
                for k in range(p.n_chunks):
                    r0, r1 = p.row_chunks[k], p.row_chunks[k + 1]
                    if r1 > r0 and packed_f:   # the chunk's send rows leave from the product's epilogue, into the chunk's segment
                        ops.linear_fwd_rows_to_slots(self.X[r0:r1], self.W, Hl[r0:r1], p.fwd.slot_table[r0:r1], self.send_f)
                    elif r1 > r0:
                        ops.linear_fwd(self.X[r0:r1], self.W, out=Hl[r0:r1])
                    hs.append(exchange_rows(dist, p.fwd, self.Hext, self.F, self.pack, self.send_f, self.native, async_op=True, chunk=k,
                                            prepacked=packed_f)[1])
                return _All(hs)

            hf = run(fwd_pipeline)
            run(hf.wait)
            run(spmm_f)
            if self.tables:   # G is read ONCE for dbias and for its pack; the chunks leave one after the other
                run(lambda: ops.rows_to_slots(Gl, p.bwd.slot_table, self.send_b, colsum_out=self.dbias) if packed_b
                    else ops.colsum(Gl, out=self.dbias))
            hb = run(lambda: _All([exchange_rows(dist, p.bwd, self.Gext, self.F, self.pack, self.send_b, self.native, async_op=True, chunk=k,
                                                 prepacked=packed_b)[1]
                                   for k in range(p.n_chunks)]))
            if not self.tables:
                run(lambda: ops.colsum(Gl, out=self.dbias))
            run(hb.wait)
            run(spmm_b)
            run(lambda: ops.gemm(self.dH, self.W, out=self.dX))
            run(lambda: ops.gemm(self.dH, self.X, transA=True, out=self.dW))
            run(self._reduce_params)
        elif self.schedule == "sequential":
            run(transform)
            run(lambda: exchange_rows(dist, p.fwd, self.Hext, self.F, self.pack, self.send_f, self.native, prepacked=self.scatter is not None))
            run(spmm_f)
            if self.scatter is not None:
                run(colsum_pack_exchange_b)
            else:
                run(lambda: ops.colsum(Gl, out=self.dbias))
                run(lambda: exchange_rows(dist, p.bwd, self.Gext, self.F, self.pack, self.send_b, self.native))
            run(spmm_b)
            run(lambda: ops.gemm(self.dH, self.W, out=self.dX))
            run(lambda: ops.gemm(self.dH, self.X, transA=True, out=self.dW))
            run(self._reduce_params)
        else:
            # backward chain's exchange first: it needs no compute in front of it, and the forward chain's GEMM + pack
            # then run under it; the backward chain's SpMM + two GEMMs run under the forward exchange; what is left
            # exposed is one exchange's head and the forward SpMM at the tail.
            if self.scatter is not None:
                _, hb = run(lambda: colsum_pack_exchange_b(async_op=True))
            else:
                _, hb = run(lambda: exchange_rows(dist, p.bwd, self.Gext, self.F, self.pack, self.send_b, self.native, async_op=True))
            run(transform)
            _, hf = run(lambda: exchange_rows(dist, p.fwd, self.Hext, self.F, self.pack, self.send_f, self.native, async_op=True,
                                              prepacked=self.scatter is not None))
            if self.scatter is None:
                run(lambda: ops.colsum(Gl, out=self.dbias))
            run(hb.wait)
            run(spmm_b)
            run(lambda: ops.gemm(self.dH, self.W, out=self.dX))
            run(lambda: ops.gemm(self.dH, self.X, transA=True, out=self.dW))
            run(hf.wait)
            run(spmm_f)
            run(self._reduce_params)
        if timed:
            self.ev.append(evs)

    def kernel_times(self):
        import numpy as np
        out = {}
        for i, nm in enumerate(self.names):
            out[nm] = float(np.mean([s[i][0].elapsed_ms(s[i][1]) for s in self.ev])) if self.ev else None
        return out

    def aggregation_stats(self):
        """This rank's forward aggregation as raw facts (bench.py turns them into its roofline object; the package knows nothing of
        the benchmark script)."""
        p = self.plan
        return {"local_rows": p.n_local, "local_nnz": p.nnz_local, "features": self.F, "spmm_fwd_ms": self.kernel_times()["spmm_fwd"],
                "halo_rows_fwd": p.fwd.n_halo, "halo_rows_bwd": p.bwd.n_halo, "send_rows_per_peer_fwd": p.fwd.send_counts,
                "schedule": self.schedule}
