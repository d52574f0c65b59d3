"""ctypes binding of the C-ABI in include/gnnx.h (libgnnx_hip.so).

This is the Python-side view of the drop-in boundary: every call takes plain device pointers and sizes.
torch is used ONLY as the owner of device memory / streams by the callers (tests, bench); nothing here
touches torch.  There is deliberately no CPU fallback: if the HIP library is missing or no GPU is
visible, calls raise GnnxError.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNNX_HIP_LIB=exp selects the measurement build (`make -C gnn.cpp_amd/csrc EXPERIMENTS=1`: kernel variants and environment switches
# for scripts/exp_*.py); tests, smoke and bench.py use the product library
LIB_PATH = os.path.join(_HERE, "libgnnx_hip_exp.so" if os.environ.get("GNNX_HIP_LIB") == "exp" else "libgnnx_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnx.h")

_lib = None


class GnnxError(RuntimeError):
    def __init__(self, status, where, msg):
        super().__init__(f"{where}: {msg} (status {status})")
        self.status = status


def lib():
    """Load libgnnx_hip.so (fails loudly if it was not built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnnxError(-100, "load", f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(LIB_PATH)
        _lib.gnnx_status_string.restype = C.c_char_p
        _lib.gnnx_last_error.restype = C.c_char_p
        _declare(_lib)
    return _lib


class SpmmFusion(C.Structure):
    """`gnnx_spmm_fusion` of include/gnnx.h."""
    _fields_ = [("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("bn_eps", C.c_float), ("relu_in", C.c_int), ("relu_out", C.c_int)]


def declared_symbols():
    """Every function name include/gnnx.h declares (used by the CPU test that the library exports them all)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnnx_[a-z0-9_]+)\s*\(", text)))


_i32, _i64, _u32, _u64, _f32, _f64, _vp, _sz = (C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double,
                                               C.c_void_p, C.c_size_t)

_SIGS = {
    "gnnx_device_count": [C.POINTER(C.c_int)],
    "gnnx_set_device": [C.c_int],
    "gnnx_device_name": [C.c_int, C.c_char_p, _sz],
    "gnnx_malloc": [C.POINTER(_vp), _sz],
    "gnnx_free": [_vp],
    "gnnx_memset": [_vp, C.c_int, _sz, _vp],
    "gnnx_memcpy_h2d": [_vp, _vp, _sz, _vp],
    "gnnx_memcpy_d2h": [_vp, _vp, _sz, _vp],
    "gnnx_memcpy_d2d": [_vp, _vp, _sz, _vp],
    "gnnx_memcpy2d_d2d": [_vp, _sz, _vp, _sz, _sz, _sz, _vp],
    "gnnx_stream_create": [C.POINTER(_vp)],
    "gnnx_stream_destroy": [_vp],
    "gnnx_stream_sync": [_vp],
    "gnnx_device_sync": [],
    "gnnx_event_create": [C.POINTER(_vp)],
    "gnnx_event_destroy": [_vp],
    "gnnx_event_record": [_vp, _vp],
    "gnnx_event_sync": [_vp],
    "gnnx_event_elapsed_ms": [_vp, _vp, C.POINTER(_f32)],
    "gnnx_stream_wait_event": [_vp, _vp],
    "gnnx_csr_from_coo_workspace": [_i64, _i32, C.POINTER(_sz)],
    "gnnx_csr_from_coo": [_vp, _vp, _i64, _i32, _u32, _vp, _vp, C.POINTER(_i64), _vp, _sz, _vp],
    "gnnx_csr_validate": [_vp, _vp, _i32, _i32, _vp],
    "gnnx_equal_i32": [_vp, _vp, _i64, C.POINTER(C.c_int), _vp],
    "gnnx_csr_from_coo_weighted_workspace": [_i64, _i32, C.POINTER(_sz)],
    "gnnx_csr_from_coo_weighted": [_vp, _vp, _vp, _i64, _i32, _u32, C.c_int, _f32, _vp, _vp, _vp, C.POINTER(_i64), _vp, _sz, _vp],
    "gnnx_degree_norm_f32": [_vp, _vp, _i32, _vp, _vp, _vp, _vp],
    "gnnx_spmm_plan_create": [_vp, _i32, _i32, _i32, C.POINTER(_vp), _vp],
    "gnnx_spmm_plan_destroy": [_vp],
    "gnnx_spmm_plan_status": [_vp],
    "gnnx_spmm_plan_info": [_vp, C.POINTER(_i64), C.POINTER(_i64)],
    "gnnx_spmm_plan_set_big_row_threshold": [_vp, _i32],
    "gnnx_spmm_plan_hub_ids_structured": [_vp, C.POINTER(C.c_int)],
    "gnnx_spmm_csr_bn_sums_workspace": [_i32, _i32, _vp, C.POINTER(_sz)],
    "gnnx_spmm_csr_bn_sums_f32": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _vp,
                                  _vp, _sz, _vp, _vp],
    "gnnx_spmm_csr_f32": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp, _i64, _vp, _vp],
    "gnnx_spmm_csr_fused_f32": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp, _i64, _vp, _vp, _vp],
    "gnnx_spmm_csr_bf16_f32": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp, _i64, _vp, _vp],
    "gnnx_f32_to_bf16": [_vp, _i64, _i64, _i32, _vp, _i64, _vp],
    "gnnx_gemm_workspace": [C.c_int, C.c_int, _i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_f32": [C.c_int, C.c_int, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _sz, _vp],
    "gnnx_gemm_relu_colsum_workspace": [_i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_relu_colsum_f32": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _sz, _vp],
    "gnnx_gemm_bn_stats_workspace": [_i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_bn_stats_f32": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _sz, _vp],
    "gnnx_gemm_split_workspace": [_i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_split_bf16_f32": [C.c_int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _sz, _vp],
    "gnnx_colsum_workspace": [_i64, _i32, C.POINTER(_sz)],
    "gnnx_colsum_f32": [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _sz, _vp],
    "gnnx_colsum_copy_f32": [_vp, _i64, _i64, _i32, _f32, _vp, _vp, _i64, _vp, _sz, _vp],
    "gnnx_rows_to_slots_f32": [_vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _f32, _vp, _sz, _vp],
    "gnnx_gemm_nt_rows_to_slots_workspace": [_i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_nt_rows_to_slots_f32": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _sz, _vp],
    "gnnx_gather_row_stride": [_i64, _i32, C.POINTER(_i64)],
    "gnnx_rowscale_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp],
    "gnnx_bias_add_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp],
    "gnnx_axpy_f32": [_i64, _f32, _vp, _vp, _vp],
    "gnnx_binary_bcast_f32": [C.c_int, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp],
    "gnnx_rowsum_f32": [_vp, _i64, _i64, _i32, _vp, _vp],
    "gnnx_fill_f32": [_vp, _i64, _f32, _vp],
    "gnnx_pow_f32": [_vp, _i64, _f32, _vp, _vp],
    "gnnx_csr_rowsum_f32": [_vp, _vp, _i32, _vp, _vp],
    "gnnx_transpose_f32": [_vp, _i64, _i64, _i64, _vp, _i64, _vp],
    "gnnx_bn_workspace": [_i64, _i32, C.POINTER(_sz)],
    "gnnx_bn_stats_f32": [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _sz, _vp],
    "gnnx_bn_relu_fwd_f32": [_vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _i64, _vp],
    "gnnx_bn_relu_bwd_f32": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _i64, _vp, _vp, _vp,
                             _sz, _vp],
    "gnnx_bn_relu_bwd_quirk_f32": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _i64, _vp, _vp, _vp,
                                   _sz, _vp],
    "gnnx_bn_partial_f32": [_vp, _i64, _i64, _i32, _vp, _f32, _vp, _vp, _sz, _vp],
    "gnnx_bn_relu_bwd_sums_f32": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _vp, _vp, _sz,
                                  _vp],
    "gnnx_bn_relu_bwd_apply_f32": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp, C.c_int, _vp, _vp, _i64, _vp,
                                   _i64, _vp, _sz, _vp],
    "gnnx_gemm_nt_bf16out_workspace": [_i64, _i64, _i64, C.POINTER(_sz)],
    "gnnx_gemm_nt_bf16out_f32": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _sz, _vp],
    "gnnx_softmax_ce_workspace": [_i64, C.POINTER(_sz)],
    "gnnx_softmax_ce_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _sz, _vp],
    "gnnx_softmax_ce_colsum_workspace": [_i64, _i32, C.POINTER(_sz)],
    "gnnx_softmax_ce_colsum_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _sz, _vp],
    "gnnx_softmax_ce_partial_f32": [_vp, _i64, _vp, _i64, _i32, _i64, _vp, _vp, _i64, _vp, _vp, _sz, _vp],
    "gnnx_sgd_step_f32": [_vp, _vp, _i64, _f32, _f32, _vp],
    "gnnx_comm_unique_id": [_vp],
    "gnnx_comm_init": [C.POINTER(_vp), C.c_int, C.c_int, _vp],
    "gnnx_comm_init_local": [C.POINTER(_vp), C.c_int],
    "gnnx_comm_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "gnnx_comm_destroy": [_vp],
    "gnnx_vertex_weights": [_vp, _vp, _i64, _i32, _i32, _vp, _vp],
    "gnnx_partition_deal": [_vp, _i32, C.c_int, _vp, _vp, C.POINTER(_i64), _vp],
    "gnnx_partition_scramble": [_vp, _i32, C.c_int, _vp, _vp, _vp],
    "gnnx_shard_select_edges": [_vp, _vp, _i64, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, C.POINTER(_i64), _vp],
    "gnnx_halo_plan_create": [_vp, _i64, _vp, _i32, C.c_int, C.c_int, C.POINTER(_i64), _vp, C.POINTER(_vp), _vp],
    "gnnx_halo_plan_destroy": [_vp],
    "gnnx_halo_plan_info": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                            C.POINTER(_vp), C.POINTER(_vp)],
    "gnnx_halo_plan_set_send_list": [_vp, _vp, C.POINTER(_i64), _vp],
    "gnnx_halo_plan_exchange_requests": [_vp, _vp, _vp],
    "gnnx_halo_exchange_rows_f32": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "gnnx_halo_plan_slot_table": [_vp, C.POINTER(_vp)],
    "gnnx_halo_exchange_packed_f32": [_vp, _vp, _vp, _i64, _i32, _vp, _vp],
    "gnnx_halo_exchange_f32": [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _i32, _vp],
    "gnnx_allreduce_sum_f32": [_vp, _vp, _i64, _vp],
    "gnnx_gather_rows_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp],
    "gnnx_scatter_add_rows_f32": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp],
    "gnnx_rmat_edges": [_u64, _i32, _i64, _i64, _f64, _f64, _f64, _vp, _vp, _vp],
    "gnnx_uniform_pm1_f32": [_u64, _i64, _f32, _vp, _vp],
}


def _declare(L):
    for name, argtypes in _SIGS.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int


def check(status, where):
    if status != 0:
        L = lib()
        msg = L.gnnx_last_error().decode() or L.gnnx_status_string(status).decode()
        raise GnnxError(status, where, msg)


def call(name, *args):
    """Call a C-ABI entry point by name and raise GnnxError on a non-zero status."""
    check(getattr(lib(), name)(*args), name)


# ---- convenience wrappers (still pointer-level) ------------------------------------------------------
def device_count():
    n = C.c_int(0)
    st = lib().gnnx_device_count(C.byref(n))
    return n.value if st == 0 else 0


def device_name(device=0):
    buf = C.create_string_buffer(256)
    call("gnnx_device_name", device, buf, 256)
    return buf.value.decode()


class Event:
    """hipEvent on an explicit stream (bench.py times kernels with these, not torch.cuda.Event)."""

    def __init__(self):
        self.h = _vp()
        call("gnnx_event_create", C.byref(self.h))

    def record(self, stream=None):
        call("gnnx_event_record", self.h, stream)

    def sync(self):
        call("gnnx_event_sync", self.h)

    def elapsed_ms(self, stop):
        ms = _f32(0)
        call("gnnx_event_elapsed_ms", self.h, stop.h, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            lib().gnnx_event_destroy(self.h)
        except Exception:
            pass


def csr_from_coo_workspace(n_edges, n_nodes):
    b = _sz(0)
    call("gnnx_csr_from_coo_workspace", n_edges, n_nodes, C.byref(b))
    return b.value


def gemm_workspace(transA, transB, M, N, K):
    b = _sz(0)
    call("gnnx_gemm_workspace", int(transA), int(transB), M, N, K, C.byref(b))
    return b.value


def colsum_workspace(n_rows, n_feat):
    b = _sz(0)
    call("gnnx_colsum_workspace", n_rows, n_feat, C.byref(b))
    return b.value
