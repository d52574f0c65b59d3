"""Pointer-level Python driver of the GCN hot path (tests / bench / multi-GPU harness).

torch tensors are used only as owners of device memory and for the current stream; every computation is
a C-ABI call into libgnnx_hip.so (include/gnnx.h).  No CPU fallback exists: a tensor that is not on a
CUDA (HIP) device is an error.

Mirrors the call structure of the reference's GCNConv (src/graph.cpp:170-212):
  CsrGraph.from_coo        <- add_self_loops + edge_to_adj_mat            graph.cpp:172,177
  CsrGraph.s / .norm       <- deg / pow / mm / *=                         graph.cpp:178-185
  linear_fwd               <- Linear::forward                             nn.cpp:205-211
  aggregate_fwd            <- aggregate_and_update (+ bias)               graph.cpp:204-212,188
  aggregate_bwd, colsum,   <- Add/Mul/MatMul/Transpose::_backward         operation.h:114-128,144-167,
  linear_bwd                                                              504-534,416-433
"""
import ctypes as C
import threading

import torch

from . import capi


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise capi.GnnxError(-1, "ops", "tensor is not on a HIP device (there is no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


_ws_cache = {}


def _workspace(nbytes, device, tag):
    """Grow-only device scratch buffer per (host thread, device, tag) -- allocated outside the timed/compute calls.
    Per thread because a scratch buffer is only safe to share between calls that are ordered on one stream by one
    submitter (the loopback test runs several ranks as threads of one process)."""
    key = (threading.get_ident(), str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class SpmmPlan:
    """gnnx_spmm_plan: hub rows (longer than `chunk`) to the sequential hub kernel, non-zero-balanced blocks for the rest; the
    planned aggregation has the same bits as the unplanned one."""

    def __init__(self, rowptr, chunk, max_feat):
        self.h = C.c_void_p()
        self._rowptr = rowptr
        capi.call("gnnx_spmm_plan_create", _ptr(rowptr), rowptr.numel() - 1, int(chunk), int(max_feat),
                  C.byref(self.h), _stream())
        a, b = C.c_int64(0), C.c_int64(0)
        capi.call("gnnx_spmm_plan_info", self.h, C.byref(a), C.byref(b))
        self.n_split_rows, self.n_hub_nnz = a.value, b.value   # hub rows (degree > chunk) and their non-zeros

    def hub_ids_structured(self):
        """gnnx_spmm_plan_hub_ids_structured: the hub rows sit on ids with few one-bits (a synthetic power-law graph as generated)."""
        v = C.c_int(0)
        capi.call("gnnx_spmm_plan_hub_ids_structured", self.h, C.byref(v))
        return bool(v.value)

    def set_big_row_threshold(self, threshold):
        """Hub rows longer than `threshold` take the producer / consumer hub kernel (< 0: the library picks them per call, the
        default; same bits for every value)."""
        capi.call("gnnx_spmm_plan_set_big_row_threshold", self.h, int(threshold))
        return self

    def __del__(self):
        try:
            capi.lib().gnnx_spmm_plan_destroy(self.h)
        except Exception:
            pass


class CsrGraph:
    """CSR of A (forward) and of A^T (backward) with the reference's adjacency semantics, plus s and norm."""

    def __init__(self, n_nodes, rowptr, colidx, rowptr_t=None, colidx_t=None):
        self.n = int(n_nodes)
        self.rowptr, self.colidx = rowptr, colidx
        self.rowptr_t, self.colidx_t = rowptr_t, colidx_t
        self.nnz = int(colidx.numel())
        self.s = self.norm = None
        self.plan = self.plan_t = None
        self.nid = None

    @staticmethod
    def csr_from_coo(src, dst, n_nodes, flags=0):
        """gnnx_csr_from_coo on device int32 tensors -> (rowptr, colidx)."""
        E = int(src.numel())
        dev = src.device
        rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
        colidx = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        ws = _workspace(capi.csr_from_coo_workspace(E, n_nodes), dev, "csr")
        nnz = C.c_int64(0)
        capi.call("gnnx_csr_from_coo", _ptr(src), _ptr(dst), E, int(n_nodes), int(flags), _ptr(rowptr), _ptr(colidx),
                  C.byref(nnz), _ptr(ws), ws.numel(), _stream())
        return rowptr, colidx[: nnz.value].clone()

    SCRAMBLE_MUL = 2654435761   # prime above any vertex count: v -> (v * SCRAMBLE_MUL) mod n is a bijection (gnnx_partition_scramble)

    @classmethod
    def from_coo(cls, src, dst, n_nodes, transpose=True, norm=True, relabel=None):
        """relabel: None (vertex v is row v), "scramble" (row nid[v] = (v * SCRAMBLE_MUL) mod n: spreads the hubs of a synthetic
        power-law graph, whose ids have few one-bits, over the cache sets) or an int32 [n] permutation nid.  With a relabelling the
        CSR is built exactly as a one-rank shard is (shard.ShardPlan / gnnx_shard_select_edges + gnnx_halo_plan_create): rows are
        new ids, a row's entries are sorted by ORIGINAL column id -- the reference's summation order -- and only then renumbered,
        so every vertex's result has the same bits as without the relabelling and is stored at row g.nid[v]."""
        if relabel is None:
            rowptr, colidx = cls.csr_from_coo(src, dst, n_nodes)
            g = cls(n_nodes, rowptr, colidx)
            if transpose:
                g.rowptr_t, g.colidx_t = cls.csr_from_coo(dst, src, n_nodes)
            g.nid = None
        else:
            dev = src.device
            if isinstance(relabel, str):
                assert relabel == "scramble"
                nid = ((torch.arange(n_nodes, dtype=torch.int64, device=dev) * cls.SCRAMBLE_MUL) % max(n_nodes, 1)).to(torch.int32)
            else:
                nid = relabel.to(torch.int32)
                if int(nid.numel()) != n_nodes or int(torch.unique(nid).numel()) != n_nodes or int(nid.min()) != 0 or \
                        int(nid.max()) != n_nodes - 1:
                    raise ValueError("relabel must be a permutation of 0..n_nodes-1")
            keep = src != dst                       # self loops are dropped on ORIGINAL ids (rows are renumbered below)
            s_, d_ = src[keep], dst[keep]
            rowptr, ci = cls.csr_from_coo(nid[s_.long()], d_, n_nodes, flags=1)        # flags = 1: keep the diagonal as given
            g = cls(n_nodes, rowptr, nid[ci.long()])
            if transpose:
                rowptr_t, cit = cls.csr_from_coo(nid[d_.long()], s_, n_nodes, flags=1)
                g.rowptr_t, g.colidx_t = rowptr_t, nid[cit.long()]
            g.nid = nid
        if norm:
            g.compute_norm()
        return g

    def to_new_order(self, X):
        """Rows of X (vertex v at row v) in this graph's row order (vertex v at row nid[v])."""
        if self.nid is None:
            return X
        out = torch.empty_like(X)
        out[self.nid.long()] = X
        return out

    def to_vertex_order(self, Y):
        """The inverse: Y has vertex v at row nid[v]; returns it with vertex v at row v."""
        return Y if self.nid is None else Y[self.nid.long()]

    def compute_norm(self):
        dev = self.rowptr.device
        self.s = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.norm = torch.empty(self.n, dtype=torch.float32, device=dev)
        capi.call("gnnx_degree_norm_f32", _ptr(self.rowptr), _ptr(self.colidx), self.n, _ptr(self.s), None,
                  _ptr(self.norm), _stream())
        return self.s, self.norm

    def norm_from_pow_table(self, pow_table):
        """OPT-IN "libm-exact" degree block: s_i = pow_table[1 + deg_i] with the table supplied by the CALLER -- pow_table[k] =
        powf((float)k, -0.5f) evaluated by the host's libm, which is literally what the reference computes (functional.h:253,
        std::pow on the host) -- instead of the device's correctly rounded rsqrt (1 ulp apart for a few degrees >= 1058), then
        norm = (A . s) (.) s through gnnx_degree_norm_f32 with the caller's s (its d_s_cols argument).  Returns (s, norm) in this
        graph's row order; nothing of the graph object changes.  With it the whole chain norm -> aggregation is bit-exact against
        the reference's arithmetic at any size (tests/test_gpu_parity.py::test_headline_config_whole_graph_vs_oracle)."""
        deg1 = (self.rowptr[1:] - self.rowptr[:-1] + 1).long()
        if int(deg1.max()) >= pow_table.numel():
            raise ValueError(f"pow_table has {pow_table.numel()} entries, the largest 1 + degree is {int(deg1.max())}")
        s = pow_table.to(torch.float32)[deg1].contiguous()
        norm = torch.empty(self.n, dtype=torch.float32, device=s.device)
        capi.call("gnnx_degree_norm_f32", _ptr(self.rowptr), _ptr(self.colidx), self.n, None, _ptr(s), _ptr(norm), _stream())
        return s, norm

    def make_plans(self, chunk, max_feat, big_rows=None):
        """Load-balancing plans for power-law rows (forward CSR and transposed CSR).  big_rows: the threshold above which a hub row
        takes the producer / consumer kernel (SpmmPlan.set_big_row_threshold; None = the library's default)."""
        self.plan = SpmmPlan(self.rowptr, chunk, max_feat)
        if self.rowptr_t is not None:
            self.plan_t = SpmmPlan(self.rowptr_t, chunk, max_feat)
        if big_rows is not None:
            for pl in (self.plan, self.plan_t):
                if pl is not None:
                    pl.set_big_row_threshold(big_rows)
        return self.plan, self.plan_t


def to_bf16(X, out=None):
    """f32 -> bf16 feature storage (torch.bfloat16 tensor, same shape) for spmm(..., X bf16)."""
    out = torch.empty(X.shape, dtype=torch.bfloat16, device=X.device) if out is None else out
    capi.call("gnnx_f32_to_bf16", _ptr(X), _ld(X), X.shape[0], X.shape[1], _ptr(out), _ld(out), _stream())
    return out


def linear_fwd_bf16(X, W, out=None):
    """OPT-IN (bf16 feature storage): H = X . W^T as a torch.bfloat16 tensor.  gnnx_gemm_nt_bf16out_f32 (the product's epilogue
    rounds and stores bf16) on the LDS-DMA kernel's shapes, else the product followed by to_bf16: the same bits either way."""
    M, K = X.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=X.device) if out is None else out
    ok = (K % 64 == 0 and N % 4 == 0 and N >= 64 and M >= 2048 and _ld(X) % 4 == 0 and _ld(out) % 4 == 0 and X.data_ptr() % 16 == 0
          and out.data_ptr() % 16 == 0)
    if not ok:
        return to_bf16(gemm(X, W, transB=True), out=out)
    wsb = C.c_size_t(0)
    capi.call("gnnx_gemm_nt_bf16out_workspace", M, N, K, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "gemm_bf16out")
    capi.call("gnnx_gemm_nt_bf16out_f32", M, N, K, _ptr(X), _ld(X), _ptr(W), _ld(W), _ptr(out), _ld(out), _ptr(ws), wsb.value, _stream())
    return out


DIAG_KEEP, DIAG_STRIP, DIAG_FILL = 0, 1, 2
CSR_KEEP_DUPLICATES, CSR_DROP_TRUNCATED_ZERO = 2, 4


def csr_from_coo_weighted(src, dst, w, n_nodes, diag_mode=DIAG_KEEP, diag_value=0.0, flags=0):
    """Weighted adjacency A[r][c] = w, last duplicate wins (edge_attr, reference graph.cpp:21-75) -> (rowptr, colidx, vals)."""
    E = int(src.numel())
    dev = src.device
    cap = max(E + n_nodes, 1)
    rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    colidx = torch.empty(cap, dtype=torch.int32, device=dev)
    vals = torch.empty(cap, dtype=torch.float32, device=dev)
    wsb = C.c_size_t(0)
    capi.call("gnnx_csr_from_coo_weighted_workspace", E, n_nodes, C.byref(wsb))
    ws = torch.empty(max(wsb.value, 1), dtype=torch.uint8, device=dev)
    nnz = C.c_int64(0)
    capi.call("gnnx_csr_from_coo_weighted", _ptr(src), _ptr(dst), _ptr(w), E, n_nodes, int(flags), int(diag_mode), float(diag_value),
              _ptr(rowptr), _ptr(colidx), _ptr(vals), C.byref(nnz), _ptr(ws), ws.numel(), _stream())
    return rowptr, colidx[: nnz.value].clone(), vals[: nnz.value].clone()


def csr_rowsum(rowptr, vals=None):
    n = int(rowptr.numel() - 1)
    out = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    capi.call("gnnx_csr_rowsum_f32", _ptr(rowptr), _ptr(vals), n, _ptr(out), _stream())
    return out


def spmm(rowptr, colidx, X, out=None, vals=None, colscale=None, rowscale=None, bias=None, beta=0.0, plan=None,
         n_rows=None, bn=None, relu_in=False, relu_out=False):
    """gnnx_spmm_csr_f32: Y = beta*Y + rowscale (.) (A . (colscale (.) X)) + bias.
    bn=(mean, var, gamma|None, beta|None, eps) / relu_in / relu_out: gnnx_spmm_csr_fused_f32 (BatchNorm / ReLU applied to
    every gathered row, ReLU on the stored row)."""
    n_rows = int(rowptr.numel() - 1) if n_rows is None else n_rows
    n_cols, F = X.shape
    if out is None:
        out = torch.empty((n_rows, F), dtype=torch.float32, device=X.device)
    if X.dtype == torch.bfloat16:  # opt-in bf16 feature storage: half the gather bytes, f32 accumulation
        capi.call("gnnx_spmm_csr_bf16_f32", n_rows, n_cols, F, _ptr(rowptr), _ptr(colidx), _ptr(vals), _ptr(colscale),
                  _ptr(rowscale), _ptr(bias), _ptr(X), _ld(X), float(beta), _ptr(out), _ld(out),
                  plan.h if plan is not None else None, _stream())
        return out
    if bn is not None or relu_in or relu_out:
        mean, var, gamma, bbeta, eps = bn if bn is not None else (None, None, None, None, 0.0)
        addr = lambda t: None if t is None else _ptr(t).value  # noqa: E731
        fu = capi.SpmmFusion(addr(mean), addr(var), addr(gamma), addr(bbeta), float(eps), int(relu_in), int(relu_out))
        capi.call("gnnx_spmm_csr_fused_f32", n_rows, n_cols, F, _ptr(rowptr), _ptr(colidx), _ptr(vals), _ptr(colscale),
                  _ptr(rowscale), _ptr(bias), _ptr(X), _ld(X), float(beta), _ptr(out), _ld(out), C.byref(fu),
                  plan.h if plan is not None else None, _stream())
        return out
    capi.call("gnnx_spmm_csr_f32", n_rows, n_cols, F, _ptr(rowptr), _ptr(colidx), _ptr(vals), _ptr(colscale),
              _ptr(rowscale), _ptr(bias), _ptr(X), _ld(X), float(beta), _ptr(out), _ld(out),
              plan.h if plan is not None else None, _stream())
    return out


def gemm(A, B, transA=False, transB=False, out=None, alpha=1.0, beta=0.0):
    """gnnx_gemm_f32: C = alpha * op(A) . op(B) + beta * C (row-major, no operand is transposed in memory)."""
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    Kb = B.shape[1] if transB else B.shape[0]
    if K != Kb:
        raise capi.GnnxError(-2, "gemm", f"inner dimensions differ: {K} vs {Kb}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    wsb = capi.gemm_workspace(transA, transB, M, N, K)
    ws = _workspace(wsb, A.device, "gemm") if wsb else None
    capi.call("gnnx_gemm_f32", int(transA), int(transB), M, N, K, float(alpha), _ptr(A), _ld(A), _ptr(B), _ld(B),
              float(beta), _ptr(out), _ld(out), _ptr(ws), wsb, _stream())
    return out


def gemm_relu_colsum(A, B, Ymask, out=None, colsum_out=None):
    """gnnx_gemm_relu_colsum_f32: C = (A . B) (.) (Ymask > 0), colsum = column sums of C -- the stacked layers' backward step
    G_{l-1} = (dH_l . W_l) (.) relu'(Y_{l-1}), db_{l-1} = colsum(G_{l-1}) in one pass over the output."""
    M, K = A.shape
    N = B.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=A.device) if out is None else out
    colsum_out = torch.empty(N, dtype=torch.float32, device=A.device) if colsum_out is None else colsum_out
    wsb = C.c_size_t(0)
    capi.call("gnnx_gemm_relu_colsum_workspace", M, N, K, C.byref(wsb))
    ws = _workspace(wsb.value, A.device, "gemm_fuse")
    capi.call("gnnx_gemm_relu_colsum_f32", M, N, K, _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(Ymask), _ld(Ymask), _ptr(out), _ld(out),
              _ptr(colsum_out), _ptr(ws), wsb.value, _stream())
    return out, colsum_out


def linear_fwd_bn_stats(X, W, out=None):
    """OPT-IN gnnx_gemm_bn_stats_f32: H = X . W^T and the BatchNorm batch statistics of H in one pass (single-pass shifted
    variance: within rounding of bn_stats(H), not bit-equal).  Returns (H, mean, var)."""
    M, K = X.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=X.device) if out is None else out
    mean = torch.empty(N, dtype=torch.float32, device=X.device)
    var = torch.empty(N, dtype=torch.float32, device=X.device)
    wsb = C.c_size_t(0)
    capi.call("gnnx_gemm_bn_stats_workspace", M, N, K, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "gemm_bn")
    capi.call("gnnx_gemm_bn_stats_f32", M, N, K, _ptr(X), _ld(X), _ptr(W), _ld(W), _ptr(out), _ld(out), _ptr(mean), _ptr(var), _ptr(ws),
              wsb.value, _stream())
    return out, mean, var


def gemm_split(A, B, transB=False, out=None):
    """OPT-IN split-precision GEMM (gnnx_gemm_split_bf16_f32): A[M,K] . op(B) on the bf16 matrix cores from exact 3-way bf16
    splits of the f32 operands, f32 accumulation; f32-level accuracy, not the reference's arithmetic."""
    M, K = A.shape
    N = B.shape[0] if transB else B.shape[1]
    out = torch.empty((M, N), dtype=torch.float32, device=A.device) if out is None else out
    wsb = C.c_size_t(0)
    capi.call("gnnx_gemm_split_workspace", M, N, K, C.byref(wsb))
    ws = _workspace(wsb.value, A.device, "gemm_split")
    capi.call("gnnx_gemm_split_bf16_f32", int(transB), M, N, K, _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(out), _ld(out), _ptr(ws),
              wsb.value, _stream())
    return out


def colsum(G, out=None, beta=0.0):
    N, F = G.shape
    if out is None:
        out = torch.empty(F, dtype=torch.float32, device=G.device)
    wsb = capi.colsum_workspace(N, F)
    ws = _workspace(wsb, G.device, "colsum")
    capi.call("gnnx_colsum_f32", _ptr(G), _ld(G), N, F, float(beta), _ptr(out), _ptr(ws), wsb, _stream())
    return out


SLOTS_PER_ROW = 8   # gnnx_rows_to_slots_f32: a row goes to at most world - 1 <= 7 peers


def slot_table(send_idx, n_rows):
    """[n_rows, 8] int32 for rows_to_slots: row r's positions in the send buffer (send_idx[slot] == r), ascending, packed to the
    front, -1 behind.  None when some row has more than 7 slots (a world of more than 8 ranks: use the gather pack)."""
    dev = send_idx.device
    table = torch.full((max(int(n_rows), 1), SLOTS_PER_ROW), -1, dtype=torch.int32, device=dev)
    if int(send_idx.numel()) == 0:
        return table
    order = torch.sort(send_idx.to(torch.int64), stable=True).indices        # slots grouped by row, ascending slot inside a row
    rows = send_idx.to(torch.int64)[order]
    first = torch.searchsorted(rows, rows, right=False)
    k = torch.arange(rows.numel(), device=dev) - first                          # rank of the slot among its row's slots
    if int(k.max()) >= SLOTS_PER_ROW - 1:
        return None
    table[rows, k] = order.to(torch.int32)
    return table


def rows_to_slots(X, table, send, colsum_out=None, beta=0.0):
    """gnnx_rows_to_slots_f32: the halo pack from the producer's side -- every row of X read once and written to each of its send
    slots (table = slot_table(send_idx, n_rows)); colsum_out: the column sums of all rows of X from the same pass (the bits of
    colsum())."""
    N, F = X.shape
    ws, wsb = None, 0
    if colsum_out is not None:
        wsb = capi.colsum_workspace(N, F)
        ws = _workspace(wsb, X.device, "colsum")
    capi.call("gnnx_rows_to_slots_f32", _ptr(X), _ld(X), N, F, _ptr(table), _ptr(send), _ld(send), _ptr(colsum_out) if colsum_out is not None else None,
              float(beta), _ptr(ws) if ws is not None else None, wsb, _stream())
    return send


def linear_fwd_rows_to_slots(X, W, out, table, send):
    """gnnx_gemm_nt_rows_to_slots_f32: H = X . W^T (linear_fwd's bits) with the halo pack in the product's epilogue -- every row of H
    listed in `table` (slot_table) also stored to its rows of `send`, from the registers H is stored from (rows_to_slots' bytes without
    the pass that reads H back)."""
    M, K = X.shape
    N = W.shape[0]
    wsb = C.c_size_t(0)
    capi.call("gnnx_gemm_nt_rows_to_slots_workspace", M, N, K, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "gemm")
    capi.call("gnnx_gemm_nt_rows_to_slots_f32", M, N, K, _ptr(X), _ld(X), _ptr(W), _ld(W), _ptr(out), _ld(out), _ptr(table), _ptr(send), _ld(send),
              _ptr(ws), wsb.value, _stream())
    return out


def gather_row_stride(n_rows, n_feat):
    """gnnx_gather_row_stride: the row pitch (floats) for a matrix whose rows the aggregation gathers (n_feat, or n_feat + 64 for large
    matrices of 512-byte-multiple rows: spreads the hub rows of a synthetic power-law graph over the memory channels)."""
    ld = C.c_int64(0)
    capi.call("gnnx_gather_row_stride", int(n_rows), int(n_feat), C.byref(ld))
    return ld.value


def empty_gathered(n_rows, n_feat, device="cuda"):
    """An uninitialised [n_rows, n_feat] view of a buffer on the gather pitch (gather_row_stride)."""
    ld = gather_row_stride(n_rows, n_feat)
    return torch.empty((n_rows, ld), dtype=torch.float32, device=device)[:, :n_feat]


def colsum_copy(G, copy, out=None, beta=0.0):
    """gnnx_colsum_copy_f32: column sums of G (as colsum, same bits) and, from the same pass, G's rows copied into `copy` (a view on
    another row stride: empty_gathered)."""
    N, F = G.shape
    if out is None:
        out = torch.empty(F, dtype=torch.float32, device=G.device)
    wsb = capi.colsum_workspace(N, F)
    ws = _workspace(wsb, G.device, "colsum")
    capi.call("gnnx_colsum_copy_f32", _ptr(G), _ld(G), N, F, float(beta), _ptr(out), _ptr(copy), _ld(copy), _ptr(ws), wsb, _stream())
    return out


def gather_rows(X, idx, out=None):
    n, F = int(idx.numel()), X.shape[1]
    if out is None:
        out = torch.empty((n, F), dtype=torch.float32, device=X.device)
    capi.call("gnnx_gather_rows_f32", _ptr(X), _ld(X), _ptr(idx), n, F, _ptr(out), _ld(out), _stream())
    return out


def scatter_add_rows(inp, idx, Y):
    n, F = int(idx.numel()), inp.shape[1]
    capi.call("gnnx_scatter_add_rows_f32", _ptr(inp), _ld(inp), _ptr(idx), n, F, _ptr(Y), _ld(Y), _stream())
    return Y


def rowscale(X, v, out=None):
    out = torch.empty_like(X) if out is None else out
    capi.call("gnnx_rowscale_f32", _ptr(X), _ld(X), _ptr(v), X.shape[0], X.shape[1], _ptr(out), _ld(out), _stream())
    return out


def bias_add(X, b, out=None):
    out = torch.empty_like(X) if out is None else out
    capi.call("gnnx_bias_add_f32", _ptr(X), _ld(X), _ptr(b), X.shape[0], X.shape[1], _ptr(out), _ld(out), _stream())
    return out


_BINARY_OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3}


def binary(op, A, B, out=None):
    """`A (op) B` with the API's 2-D broadcast (reference functional.h:163-239, utils.h:181-228): each operand is [N,F],
    [N,1], [1,F], [F] or a one-element tensor."""
    def as2d(t):
        return t.reshape(1, -1) if t.dim() <= 1 else t
    A2, B2 = as2d(A), as2d(B)
    n, f = max(A2.shape[0], B2.shape[0]), max(A2.shape[1], B2.shape[1])
    for t in (A2, B2):
        if t.shape[0] not in (1, n) or t.shape[1] not in (1, f) or not t.is_contiguous():
            raise ValueError("operands are not broadcastable 2-D contiguous tensors")
    strides = lambda t: (t.shape[1] if t.shape[0] == n and n > 1 else 0, 1 if t.shape[1] == f and f > 1 else 0)  # noqa: E731
    out = torch.empty((n, f), dtype=torch.float32, device=A.device) if out is None else out
    (ars, acs), (brs, bcs) = strides(A2), strides(B2)
    capi.call("gnnx_binary_bcast_f32", _BINARY_OPS[op], n, f, _ptr(A2), ars, acs, _ptr(B2), brs, bcs, _ptr(out), _ld(out), _stream())
    return out


def rowsum(X, out=None):
    out = torch.empty((X.shape[0],), dtype=torch.float32, device=X.device) if out is None else out
    capi.call("gnnx_rowsum_f32", _ptr(X), _ld(X), X.shape[0], X.shape[1], _ptr(out), _stream())
    return out


def axpy(a, x, y):
    capi.call("gnnx_axpy_f32", x.numel(), float(a), _ptr(x), _ptr(y), _stream())
    return y


def bn_stats(X):
    """Batch mean / biased variance per feature (x->mean(-2), x->var(-2, 0); reference nn.cpp:303,312)."""
    N, F = X.shape
    mean = torch.empty(F, dtype=torch.float32, device=X.device)
    var = torch.empty(F, dtype=torch.float32, device=X.device)
    wsb = C.c_size_t(0)
    capi.call("gnnx_bn_workspace", N, F, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "bn")
    capi.call("gnnx_bn_stats_f32", _ptr(X), _ld(X), N, F, _ptr(mean), _ptr(var), _ptr(ws), wsb.value, _stream())
    return mean, var


def bn_partial(X, mean=None, scale=1.0):
    """scale * column sums of X (mean None) or of (X - mean)^2: the per-shard halves of the batch statistics (gnnx_bn_partial_f32)."""
    N, F = X.shape
    out = torch.empty(F, dtype=torch.float32, device=X.device)
    wsb = C.c_size_t(0)
    capi.call("gnnx_bn_workspace", N, F, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "bn")
    capi.call("gnnx_bn_partial_f32", _ptr(X), _ld(X), N, F, _ptr(mean), float(scale), _ptr(out), _ptr(ws), wsb.value, _stream())
    return out


def bn_relu_fwd(X, mean=None, var=None, gamma=None, beta=None, eps=1e-5, relu=True, out=None):
    out = torch.empty_like(X) if out is None else out
    capi.call("gnnx_bn_relu_fwd_f32", _ptr(X), _ld(X), X.shape[0], X.shape[1], _ptr(mean), _ptr(var), float(eps), _ptr(gamma),
              _ptr(beta), int(relu), _ptr(out), _ld(out), _stream())
    return out


def bn_relu_bwd(X, Y, dY, mean=None, var=None, gamma=None, eps=1e-5, relu=True, beta=None, reference_quirk=False):
    """Y=None with relu: the forward output was never stored (fused forward); its sign is recomputed from X.
    reference_quirk: gnnx_bn_relu_bwd_quirk_f32, the gradient the REFERENCE's traversal delivers (statistics as constants)."""
    N, F = X.shape
    dX = torch.empty_like(X)
    dgamma = torch.empty(F, dtype=torch.float32, device=X.device) if mean is not None else None
    dbeta = torch.empty(F, dtype=torch.float32, device=X.device) if mean is not None else None
    wsb = C.c_size_t(0)
    capi.call("gnnx_bn_workspace", N, F, C.byref(wsb))
    ws = _workspace(wsb.value, X.device, "bn")
    capi.call("gnnx_bn_relu_bwd_quirk_f32" if reference_quirk else "gnnx_bn_relu_bwd_f32", _ptr(X), _ld(X), _ptr(Y),
              _ld(Y) if Y is not None else 0, _ptr(dY), _ld(dY), N, F, _ptr(mean),
              _ptr(var), float(eps), _ptr(gamma), _ptr(beta), int(relu), _ptr(dX), _ld(dX), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
              wsb.value, _stream())
    return dX, dgamma, dbeta


def rmat_edges(seed, n_nodes, n_edges, a=0.57, b=0.19, c=0.19, device="cuda", first_edge=0):
    src = torch.empty(n_edges, dtype=torch.int32, device=device)
    dst = torch.empty(n_edges, dtype=torch.int32, device=device)
    capi.call("gnnx_rmat_edges", int(seed), int(n_nodes), int(n_edges), int(first_edge), float(a), float(b), float(c),
              _ptr(src), _ptr(dst), _stream())
    return src, dst


def uniform_pm1(seed, shape, scale=1.0, device="cuda"):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    capi.call("gnnx_uniform_pm1_f32", int(seed), out.numel(), float(scale), _ptr(out), _stream())
    return out


# ---- the hot path, chained as the reference chains it ------------------------------------------------
def linear_fwd(X, W, out=None):
    """H = X . W^T  (nn.cpp:205-211; GCNConv's lin has no bias, graph.cpp:162)."""
    return gemm(X, W, transB=True, out=out)


def aggregate_fwd(g, H, bias=None, out=None, use_plan=True, bn=None, relu_in=False, relu_out=False):
    """out = norm (.) (A . H) (+ bias)  (graph.cpp:204-212, :188).  bn / relu_in: GCNConv's BatchNorm + ReLU between transform
    and aggregation (graph.cpp:174-175) folded into the gather; relu_out: the ReLU in front of the next layer."""
    return spmm(g.rowptr, g.colidx, H, out=out, rowscale=g.norm, bias=bias, plan=g.plan if use_plan else None, bn=bn,
                relu_in=relu_in, relu_out=relu_out)


def aggregate_bwd(g, G, out=None, beta=0.0, use_plan=True):
    """dH = A^T . (norm (.) G)  (operation.h:144-167 then :524-531).
    The per-source scale norm[i] is handed to the kernel per non-zero (vals_t[p] = norm[colidx_t[p]], gathered once per
    graph): a coalesced 4 B/edge stream instead of a random 4-byte gather per edge; the arithmetic (one rounded multiply
    per term) is identical."""
    if getattr(g, "norm_per_nz_t", None) is None:
        g.norm_per_nz_t = gather_rows(g.norm.reshape(-1, 1), g.colidx_t).reshape(-1)
    return spmm(g.rowptr_t, g.colidx_t, G, out=out, vals=g.norm_per_nz_t, beta=beta, plan=g.plan_t if use_plan else None)


def aggregate_bwd_bn_sums(g, G, H, mean, var, gamma=None, beta=None, eps=1e-5, relu=True, out=None, use_plan=True):
    """gnnx_spmm_csr_bn_sums_f32: dY = A^T . (norm (.) G) as aggregate_bwd computes it (same bits) plus BatchNorm's backward column
    sums dgamma / dbeta over g = dY masked by relu(BN(H)) -- accumulated by the wavefronts that store dY, so the separate sums
    pass over dY and H disappears.  Returns (dY, dgamma, dbeta); follow with gnnx_bn_relu_bwd_apply_f32."""
    if getattr(g, "norm_per_nz_t", None) is None:
        g.norm_per_nz_t = gather_rows(g.norm.reshape(-1, 1), g.colidx_t).reshape(-1)
    n, F = G.shape
    if out is None:
        out = torch.empty((n, F), dtype=torch.float32, device=G.device)
    dgamma = torch.empty(F, dtype=torch.float32, device=G.device)
    dbeta = torch.empty(F, dtype=torch.float32, device=G.device)
    plan = g.plan_t if use_plan else None
    ph = plan.h if plan is not None else None
    wsb = C.c_size_t(0)
    capi.call("gnnx_spmm_csr_bn_sums_workspace", n, F, ph, C.byref(wsb))
    ws = _workspace(wsb.value, G.device, "bn_sums")
    capi.call("gnnx_spmm_csr_bn_sums_f32", n, n, F, _ptr(g.rowptr_t), _ptr(g.colidx_t), _ptr(g.norm_per_nz_t), _ptr(G), _ld(G), _ptr(out),
              _ld(out), _ptr(H), _ld(H), _ptr(mean), _ptr(var), float(eps), _ptr(gamma), _ptr(beta), int(relu), _ptr(dgamma), _ptr(dbeta),
              _ptr(ws), wsb.value, ph, _stream())
    return out, dgamma, dbeta


def aggregate_fwd_sym(g, H, bias=None, out=None, self_term=False, use_plan=True):
    """Mode SYM, the textbook layer the north_star writes: out = D^-1/2 A D^-1/2 . H (+ bias), s = (1 + deg)^-1/2 as in the
    reference's degree block (graph.cpp:178,183); with self_term the D^-1/2 (A + I) D^-1/2 form.  Unlike Mode REF (the
    reference's factorised norm, graph.cpp:196-199) the scale is applied per edge: colscale = s, rowscale = s."""
    out = spmm(g.rowptr, g.colidx, H, out=out, colscale=g.s, rowscale=g.s, bias=bias, plan=g.plan if use_plan else None)
    if self_term:  # + s_i^2 * H_i
        s2 = g.s * g.s
        axpy(1.0, rowscale(H, s2), out)
    return out


def aggregate_bwd_sym(g, G, out=None, self_term=False, use_plan=True):
    """dH = D^-1/2 A^T D^-1/2 . G (+ s^2 (.) G): the same kernel on the transposed CSR."""
    out = spmm(g.rowptr_t, g.colidx_t, G, out=out, colscale=g.s, rowscale=g.s, plan=g.plan_t if use_plan else None)
    if self_term:
        axpy(1.0, rowscale(G, g.s * g.s), out)
    return out


def linear_bwd(dH, X, W, dX=None, dW=None, beta_dw=0.0):
    """dX = dH . W ; dW = dH^T . X  (operation.h:516-531, :416-433)."""
    dX = gemm(dH, W, out=dX)
    dW = gemm(dH, X, transA=True, out=dW, beta=beta_dw)
    return dX, dW


def gcn_layer_fwd(g, X, W, bias):
    H = linear_fwd(X, W)
    out = aggregate_fwd(g, H, bias)
    return H, out


def gcn_layer_bwd(g, X, W, G):
    dbias = colsum(G)
    dH = aggregate_bwd(g, G)
    dX, dW = linear_bwd(dH, X, W)
    return dict(dbias=dbias, dH=dH, dX=dX, dW=dW)


# ---- whole training step: multi-layer GCN + softmax cross-entropy + SGD ----------
def softmax_ce(logits, target, want_grad=True, colsum_out=None, n_total=None, grad_out=None):
    """(mean loss as a 1-element device tensor, dlogits or None): gnnx_softmax_ce_f32 (reference forward nn.cpp:442-453).
    colsum_out [C]: also the column sums of dlogits (the last layer's bias gradient), from the kernel that writes dlogits.
    n_total: the rows are one shard of a batch of n_total (gnnx_softmax_ce_partial_f32: loss = this rank's term of the mean).
    grad_out: where dlogits goes (e.g. the [:n_local] rows of a [local | halo] buffer)."""
    N, Cn = logits.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    d = (torch.empty_like(logits) if grad_out is None else grad_out) if want_grad else None
    wsb = C.c_size_t(0)
    if colsum_out is not None:
        assert want_grad and colsum_out.numel() == Cn and colsum_out.is_contiguous()
        capi.call("gnnx_softmax_ce_colsum_workspace", N, Cn, C.byref(wsb))
    else:
        capi.call("gnnx_softmax_ce_workspace", N, C.byref(wsb))
    ws = _workspace(wsb.value, logits.device, "ce")
    capi.call("gnnx_softmax_ce_partial_f32", _ptr(logits), _ld(logits), _ptr(target), N, Cn, int(N if n_total is None else n_total), _ptr(loss),
              _ptr(d), _ld(d) if want_grad else 0, _ptr(colsum_out), _ptr(ws), wsb.value, _stream())
    return loss, d


def sgd_step(param, grad, lr, weight_decay=0.0):
    capi.call("gnnx_sgd_step_f32", _ptr(param), _ptr(grad), param.numel(), float(lr), float(weight_decay), _stream())
    return param


class GcnStack:
    """L GCN layers on one graph: h_{l+1} = act( norm (.) (A . (h_l W_l^T)) + b_l ), ReLU between layers, none after the
    last.  forward / backward / SGD step, every op a C-ABI call; activations are kept for backward.

    Layout for widths off the 128 grid (pad_streamed; automatic for widths >= 64 on graphs of >= 100 000 nodes: the
    products-shaped F = 100): matrices whose rows are only STREAMED by the dense products -- layer inputs / outputs Y_l, dH_l,
    W_l, dW_l -- are stored with every width rounded up to 128 floats, pad columns zero (they stay zero: the aggregations write
    the logical columns only, the products of zero columns are zero), so the products run on the LDS-DMA kernels; matrices
    whose rows are GATHERED by the aggregations -- H_l = h_l W_l^T and the gradients G_l -- keep their own width (a 512-byte
    stride there costs the gather more than the products gain).  Zero columns add exact zeros at the end of
    every fmaf chain: the logical columns have the same bits in both layouts (dW: up to how split-K cuts the rows)."""

    def __init__(self, g, dims, seed=0, device="cuda", pad_streamed=None):
        self.g = g
        self.dims = list(dims)
        L = len(dims) - 1
        if pad_streamed is None:
            pad_streamed = g.n >= 100_000
        self.P = [-(-d // 128) * 128 if (pad_streamed and d % 128 and d >= 64) else d for d in dims]
        self.padded = self.P != self.dims
        self.Wp = [torch.zeros((self.P[l + 1], self.P[l]), dtype=torch.float32, device=device) for l in range(L)]
        self.dWp = [torch.zeros_like(w) for w in self.Wp]
        self.W = [self.Wp[l][:dims[l + 1], :dims[l]] for l in range(L)]     # the logical matrices (views)
        self.dW = [self.dWp[l][:dims[l + 1], :dims[l]] for l in range(L)]
        for l in range(L):
            self.W[l].copy_(uniform_pm1(seed + 2 * l, (dims[l + 1], dims[l]), scale=dims[l] ** -0.5, device=device))
        self.b = [torch.zeros(dims[l + 1], dtype=torch.float32, device=device) for l in range(L)]
        self.db = [torch.zeros_like(b) for b in self.b]
        self._saved = None
        self._buf = {}
        # The matrices the aggregations GATHER rows from -- H_l = h_l W_l^T forward, the gradients G_l backward -- sit on the padded
        # row pitch (gnnx_gather_row_stride) when the graph's hub ids call for it (a synthetic power-law graph in its as-generated
        # vertex order: gnnx_spmm_plan_hub_ids_structured).  Every one of them is written by a kernel of this stack (the products,
        # the loss: softmax_ce(grad_out=net.grad_buffer())), so the pitch costs nothing per step; same bits.
        self.gather_pitch = bool(g.plan is not None and g.plan_t is not None and (g.plan.hub_ids_structured() or g.plan_t.hub_ids_structured()))

    def _gathered(self, key, n, F, device):
        """persistent [n, F] buffer for a gathered matrix: on the gather pitch when the graph wants it"""
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != (n, F):
            t = self._buf[key] = empty_gathered(n, F, device=device) if self.gather_pitch else torch.empty((n, F), dtype=torch.float32, device=device)
        return t

    def grad_buffer(self, n=None):
        """Where the loss should write dlogits (softmax_ce(..., grad_out=net.grad_buffer())): the top gradient is gathered by the last
        layer's backward aggregation."""
        return self._gathered(("G", len(self.W)), self.g.n if n is None else n, self.dims[-1], self.Wp[0].device)

    def _zeros(self, key, shape, device):
        """persistent zero-initialised buffer (padded layout: pad columns are written once, here)"""
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[key] = torch.zeros(shape, dtype=torch.float32, device=device)
        return t

    def pad_input(self, X):
        """[n, P0] zero-padded copy of the layer-0 input (call once for static features; forward() accepts its [:, :d0] view)."""
        if self.P[0] == self.dims[0]:
            return X
        Xp = torch.zeros((X.shape[0], self.P[0]), dtype=torch.float32, device=X.device)
        Xp[:, :self.dims[0]] = X
        return Xp[:, :self.dims[0]]

    def forward(self, X):
        L = len(self.W)
        if not self.padded:
            saved, h = [], X
            for l in range(L):
                H = linear_fwd(h, self.W[l], out=self._gathered(("H", l), X.shape[0], self.dims[l + 1], X.device))
                # the ReLU between layers rides in the aggregation's epilogue: only relu(Z) is stored (its sign is the mask)
                Y = aggregate_fwd(self.g, H, self.b[l], relu_out=l + 1 < L)
                saved.append((h, Y))
                h = Y
            self._saved = saved
            return h
        n, d, P = X.shape[0], self.dims, self.P
        if P[0] != d[0]:
            if X.stride(0) != P[0] or X.stride(1) != 1:   # not a view of a padded buffer (pad_input): pad a copy
                X = self.pad_input(X)
            hp = torch.as_strided(X, (n, P[0]), (P[0], 1))
            if getattr(self, "_checked_input", None) != (X.data_ptr(), n):   # a caller's own wide buffer: its pad columns must be zero
                if bool(hp[:, d[0]:].any()):
                    raise ValueError("the columns behind the input's logical width must be zero (use GcnStack.pad_input)")
                self._checked_input = (X.data_ptr(), n)
        else:
            hp = X
        saved = []
        for l in range(L):
            H = linear_fwd(hp, self.Wp[l][:d[l + 1]])                 # K = P[l] (zero pads), N = d[l+1]: gathered, own width
            Yp = self._zeros(("Y", l), (n, P[l + 1]), X.device)
            aggregate_fwd(self.g, H, self.b[l], out=Yp[:, :d[l + 1]], relu_out=l + 1 < L)
            saved.append((hp, Yp))
            hp = Yp
        self._saved = saved
        return hp[:, :d[L]]

    def backward(self, dOut, fused=True, input_grad=True, have_last_bias_grad=False):
        """fused: the ReLU mask of the layer below and its bias gradient ride in the epilogue of dH . W
        (gnnx_gemm_relu_colsum_f32); fused=False runs them as their own passes (same G bits, db within rounding).
        input_grad=False: the stack's input is data (no requires_grad, as the reference's DataBatch features): dH . W of the
        first layer is not computed and None is returned.  have_last_bias_grad: db[L-1] was already written by the loss kernel
        (softmax_ce(..., colsum_out=net.db[-1]))."""
        G = dOut
        L = len(self.W)
        d, P = self.dims, self.P
        if not have_last_bias_grad:
            colsum(G, out=self.db[L - 1])
        for l in reversed(range(L)):
            h, Y = self._saved[l]
            if self.padded:
                # streamed: padded width.  One buffer per (padded, logical) width pair: the aggregation writes columns [:d] only, so two
                # layers whose widths differ but round to the same 128-float bucket must not share pad columns (a narrower layer
                # would inherit the wider one's values there, and dW / W pads would stop being zero)
                dH = self._zeros(("dH", P[l + 1], d[l + 1]), (G.shape[0], P[l + 1]), G.device)
                aggregate_bwd(self.g, G, out=dH[:, :d[l + 1]])
                Wl, hl = self.Wp[l][:, :d[l]], h[:, :d[l]]            # [P_out, d_in] (ld P_in); the layer input, logical width
                gemm(dH, h, transA=True, out=self.dWp[l])             # dW_l = dH^T . h on the padded widths
            else:
                dH = aggregate_bwd(self.g, G)
                Wl, hl = self.W[l], h
                gemm(dH, h, transA=True, out=self.dW[l])          # dW_l = dH^T . h
            if l == 0:
                G = gemm(dH, Wl) if input_grad else None             # dX of the first layer: no ReLU below it
            elif fused:
                # h = Y_{l-1} = relu output of the layer below; G_{l-1} is gathered by that layer's backward aggregation
                G, _ = gemm_relu_colsum(dH, Wl, hl, out=self._gathered(("G", l), dH.shape[0], Wl.shape[1], dH.device), colsum_out=self.db[l - 1])
            else:
                G = gemm(dH, Wl)
                G, _, _ = bn_relu_bwd(hl, hl, G, relu=True)
                colsum(G, out=self.db[l - 1])
        return G

    def step(self, lr, weight_decay=0.0):
        for p, gr in zip(self.Wp + self.b, self.dWp + self.db):   # padded storage: pads are 0 - lr * (0 + wd * 0) = 0
            sgd_step(p, gr, lr, weight_decay)
