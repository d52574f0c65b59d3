"""ctypes binding of oracle/libgcn_oracle.so (TEST INFRASTRUCTURE ONLY).

Each wrapper mirrors one function of gcn_oracle.c; see that file for the reference file:line each
one restates and for the summation-order argument that makes bit-exact parity possible.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgcn_oracle.so")
_lib = None

__all__ = ["build", "coo_to_csr", "csr_transpose", "degree_norm", "linear_fwd", "aggregate_fwd",
           "aggregate_bwd", "colsum", "linear_bwd", "dense_aggregate", "set_threads", "max_threads",
           "coo_to_csr_weighted", "spmm_vals", "rowsum_vals",
           "powf_table", "bn_relu_fwd", "bn_relu_bwd_quirk", "cross_entropy", "gcn_layer_fwd", "gcn_layer_bwd", "ref_driver_path"]


def build():
    src = os.path.join(_HERE, "gcn_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgcn_oracle.so"])
    return _SO


def ref_driver_path():
    """Path of the compiled real reference (oracle/_ref/ref_driver) or None."""
    p = os.path.join(_HERE, "_ref", "ref_driver")
    return p if os.path.exists(p) else None


def _L():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gcn_oracle_coo_to_csr.restype = C.c_int64
        _lib.gcn_oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    _L().gcn_oracle_set_threads(C.c_int(int(n)))


def max_threads():
    return int(_L().gcn_oracle_max_threads())


def coo_to_csr(src, dst, n_nodes):
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    E = src.shape[0]
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    colidx = np.zeros(max(E, 1), dtype=np.int32)
    nnz = _L().gcn_oracle_coo_to_csr(_p(src), _p(dst), C.c_int64(E), C.c_int32(n_nodes), _p(rowptr), _p(colidx))
    if nnz < 0:
        raise RuntimeError("edge index out of range")
    return rowptr, colidx[:nnz].copy()


def coo_to_csr_weighted(src, dst, w, n_nodes, diag_mode=0, diag_value=0.0, drop_truncated_zero=False):
    """(rowptr, colidx, vals) of the weighted adjacency; diag_mode 0 keep / 1 strip / 2 fill (see gcn_oracle.c)."""
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    E = src.shape[0]
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    colidx = np.zeros(E + n_nodes + 1, dtype=np.int32)
    vals = np.zeros(E + n_nodes + 1, dtype=np.float32)
    fn = _L().gcn_oracle_coo_to_csr_weighted
    fn.restype = C.c_int64
    nnz = fn(_p(src), _p(dst), _p(w), C.c_int64(E), C.c_int32(n_nodes), C.c_int(diag_mode), C.c_float(diag_value),
             C.c_int(int(drop_truncated_zero)), _p(rowptr), _p(colidx), _p(vals))
    if nnz < 0:
        raise RuntimeError("edge index out of range")
    return rowptr, colidx[:nnz].copy(), vals[:nnz].copy()


def spmm_vals(rowptr, colidx, vals, X):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    n = rowptr.shape[0] - 1
    out = np.empty((n, X.shape[1]), dtype=np.float32)
    _L().gcn_oracle_spmm_vals(_p(rowptr), _p(colidx), _p(vals), C.c_int32(n), C.c_int32(X.shape[1]), _p(X), _p(out))
    return out


def rowsum_vals(rowptr, vals):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    n = rowptr.shape[0] - 1
    out = np.empty(n, dtype=np.float32)
    _L().gcn_oracle_rowsum_vals(_p(rowptr), _p(vals), C.c_int32(n), _p(out))
    return out


def csr_transpose(rowptr, colidx, n_nodes):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    rT = np.zeros(n_nodes + 1, dtype=np.int64)
    cT = np.zeros(max(colidx.shape[0], 1), dtype=np.int32)
    _L().gcn_oracle_csr_transpose(_p(rowptr), _p(colidx), C.c_int32(n_nodes), _p(rT), _p(cT))
    return rT, cT[:colidx.shape[0]].copy()


def degree_norm(rowptr, colidx, n_nodes):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    s = np.zeros(n_nodes, dtype=np.float32)
    norm = np.zeros(n_nodes, dtype=np.float32)
    _L().gcn_oracle_degree_norm(_p(rowptr), _p(colidx), C.c_int32(n_nodes), _p(s), _p(norm))
    return s, norm


def linear_fwd(X, W):
    X, W = _f32(X), _f32(W)
    N, Fin = X.shape
    Fout = W.shape[0]
    H = np.zeros((N, Fout), dtype=np.float32)
    _L().gcn_oracle_linear_fwd(_p(X), _p(W), C.c_int64(N), C.c_int32(Fin), C.c_int32(Fout), _p(H))
    return H


def aggregate_fwd(rowptr, colidx, H, norm=None, bias=None, n_rows=None):
    """n_rows: number of output rows when the CSR is a rectangular row block (H has more rows: [local|halo])."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    H = _f32(H)
    N, F = H.shape
    if n_rows is not None:
        N = n_rows
    out = np.zeros((N, F), dtype=np.float32)
    norm = None if norm is None else _f32(norm)
    bias = None if bias is None else _f32(bias)
    _L().gcn_oracle_aggregate_fwd(_p(rowptr), _p(colidx), C.c_int32(N), C.c_int32(F), _p(H),
                                  _p(norm) if norm is not None else None,
                                  _p(bias) if bias is not None else None, _p(out))
    return out


def aggregate_bwd(rowptrT, colidxT, G, norm=None, n_rows=None):
    rowptrT = np.ascontiguousarray(rowptrT, dtype=np.int64)
    colidxT = np.ascontiguousarray(colidxT, dtype=np.int32)
    G = _f32(G)
    N, F = G.shape
    if n_rows is not None:
        N = n_rows
    dH = np.zeros((N, F), dtype=np.float32)
    norm = None if norm is None else _f32(norm)
    _L().gcn_oracle_aggregate_bwd(_p(rowptrT), _p(colidxT), C.c_int32(N), C.c_int32(F), _p(G),
                                  _p(norm) if norm is not None else None, _p(dH))
    return dH


def colsum(G):
    G = _f32(G)
    N, F = G.shape
    out = np.zeros(F, dtype=np.float32)
    _L().gcn_oracle_colsum(_p(G), C.c_int64(N), C.c_int32(F), _p(out))
    return out


def linear_bwd(dH, X, W, need_dx=True, need_dw=True):
    dH, X, W = _f32(dH), _f32(X), _f32(W)
    N, Fin = X.shape
    Fout = W.shape[0]
    dX = np.zeros((N, Fin), dtype=np.float32) if need_dx else None
    dW = np.zeros((Fout, Fin), dtype=np.float32) if need_dw else None
    _L().gcn_oracle_linear_bwd(_p(dH), _p(X), _p(W), C.c_int64(N), C.c_int32(Fin), C.c_int32(Fout),
                               _p(dX) if need_dx else None, _p(dW) if need_dw else None)
    return dX, dW


def dense_aggregate(src, dst, n_nodes, H, norm=None):
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    H = _f32(H)
    F = H.shape[1]
    out = np.zeros((n_nodes, F), dtype=np.float32)
    norm = None if norm is None else _f32(norm)
    _L().gcn_oracle_dense_aggregate(_p(src), _p(dst), C.c_int64(src.shape[0]), C.c_int32(n_nodes), C.c_int32(F),
                                    _p(H), _p(norm) if norm is not None else None, _p(out))
    return out


def bn_relu_fwd(X, gamma=None, beta=None, eps=1e-5, do_bn=True, do_relu=True):
    """BatchNorm (batch statistics) then ReLU, the reference's arithmetic (nn.cpp:301-330, 229-237)."""
    X = _f32(X)
    N, F = X.shape
    gamma = _f32(np.ones(F) if gamma is None else gamma).reshape(-1)
    beta = _f32(np.zeros(F) if beta is None else beta).reshape(-1)
    Y = np.zeros((N, F), dtype=np.float32)
    mean = np.zeros(F, dtype=np.float32)
    var = np.zeros(F, dtype=np.float32)
    _L().gcn_oracle_bn_relu_fwd(_p(X), C.c_int64(N), C.c_int32(F), _p(gamma), _p(beta), C.c_float(eps), C.c_int(int(do_bn)),
                                C.c_int(int(do_relu)), _p(Y), _p(mean), _p(var))
    return Y, mean, var



def bn_relu_bwd_quirk(X, dY, gamma=None, beta=None, eps=1e-5):
    """Backward through BatchNorm + ReLU as the REFERENCE's traversal delivers it (fan-in arrivals after the first are dropped,
    operation.h:80-88): returns (dX, dgamma, dbeta).  See gcn_oracle_bn_relu_bwd_quirk."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    dY = np.ascontiguousarray(dY, dtype=np.float32)
    N, F = X.shape
    gamma = np.ones(F, dtype=np.float32) if gamma is None else np.ascontiguousarray(gamma, dtype=np.float32).reshape(-1)
    beta = np.zeros(F, dtype=np.float32) if beta is None else np.ascontiguousarray(beta, dtype=np.float32).reshape(-1)
    dX = np.empty_like(X)
    dgamma = np.empty(F, dtype=np.float32)
    dbeta = np.empty(F, dtype=np.float32)
    _L().gcn_oracle_bn_relu_bwd_quirk(_p(X), C.c_int64(N), C.c_int32(F), _p(gamma), _p(beta), C.c_float(eps), _p(dY), _p(dX), _p(dgamma),
                                      _p(dbeta))
    return dX, dgamma, dbeta

def cross_entropy(logits, target):
    """Mean softmax cross-entropy, the reference's forward arithmetic (nn.cpp:442-453)."""
    logits = _f32(logits)
    target = np.ascontiguousarray(target, dtype=np.int32)
    L = _L()
    L.gcn_oracle_cross_entropy.restype = C.c_float
    return float(L.gcn_oracle_cross_entropy(_p(logits), _p(target), C.c_int64(logits.shape[0]), C.c_int32(logits.shape[1])))


def powf_table(n):
    out = np.zeros(n, dtype=np.float32)
    _L().gcn_oracle_powf_table(C.c_int32(n), _p(out))
    return out


def gcn_layer_fwd(src, dst, n_nodes, X, W, bias):
    """Hot-path forward as the reference chains it (graph.cpp:172-173,177-188, without BatchNorm/ReLU)."""
    rowptr, colidx = coo_to_csr(src, dst, n_nodes)
    s, norm = degree_norm(rowptr, colidx, n_nodes)
    H = linear_fwd(X, W)
    out = aggregate_fwd(rowptr, colidx, H, norm, bias)
    return dict(rowptr=rowptr, colidx=colidx, s=s, norm=norm, H=H, out=out)


def gcn_layer_bwd(fwd, X, W, G):
    """Backward of gcn_layer_fwd for upstream gradient G (see gcn_oracle.c header for file:line)."""
    n = fwd["rowptr"].shape[0] - 1
    rT, cT = csr_transpose(fwd["rowptr"], fwd["colidx"], n)
    dbias = colsum(G)
    dH = aggregate_bwd(rT, cT, G, fwd["norm"])
    dX, dW = linear_bwd(dH, X, W)
    return dict(dbias=dbias, dH=dH, dX=dX, dW=dW)
