// functional.h -- the "kernels" layer of the API: free functions on tensors, each one call into the C-ABI.
// This is the seam the reference left open: `// device::add(out_data, lhs_data, rhs_data);`
// (reference include/functional.h:174,180).  Only the shapes of the GCN hot path are implemented; anything
// else throws ERROR_BACKEND_UNSUPPORTED (there is no CPU fallback).
#ifndef GNNCPP_AMD_FUNCTIONAL_H
#define GNNCPP_AMD_FUNCTIONAL_H

#include "tensor.h"

namespace cyg {
namespace functional {

namespace impl {
inline tptr<float> new_out(const std::vector<size_t> &dims, bool req_grad)
{
    return std::make_shared<tensor<float>>(tensor<float>::device_tag{}, dims, req_grad);
}
inline bool is_col_of(const std::vector<size_t> &big, const std::vector<size_t> &small)  // [N,F] vs [N,1]
{
    return big.size() == 2 && small.size() == 2 && small[0] == big[0] && small[1] == 1;
}
inline bool is_row_of(const std::vector<size_t> &big, const std::vector<size_t> &small)  // [N,F] vs [F] or [1,F]
{
    return big.size() == 2 && ((small.size() == 1 && small[0] == big[1]) || (small.size() == 2 && small[0] == 1 && small[1] == big[1]));
}
// [N,F] / [N,1] / [1,F] / [F] / one element, as a (rows, cols) pair; false when the rank is above 2
inline bool as_2d(const std::vector<size_t> &d, size_t &r, size_t &c)
{
    if (d.size() == 1) { r = 1; c = d[0]; return true; }
    if (d.size() == 2) { r = d[0]; c = d[1]; return true; }
    return false;
}
// General 2-D broadcast through gnnx_binary_bcast_f32; the result takes the shape of the higher-rank / larger operand.
inline tptr<float> binary_bcast(int op, const tptr<float> &lhs, const tptr<float> &rhs, const char *what)
{
    auto ls = lhs->shape(), rs = rhs->shape();
    size_t lr, lc, rr, rc;
    if (ls == rs) {  // same shape, any rank: flat
        auto out = new_out(ls, lhs->requires_grad() || rhs->requires_grad());
        detail::gx(gnnx_binary_bcast_f32(op, 1, (int64_t)lhs->numel(), lhs->device_data(), 0, 1, rhs->device_data(), 0, 1,
                                         out->device_out(), (int64_t)lhs->numel(), detail::current_stream()), what);
        return out;
    }
    if (lhs->numel() == 1 || rhs->numel() == 1) {  // a one-element operand against any rank: flat, stride 0 on the scalar
        const bool lscalar = lhs->numel() == 1 && rhs->numel() != 1;
        const auto &big = lscalar ? rhs : lhs;
        auto out = new_out(big->shape().size() >= (lscalar ? ls : rs).size() ? big->shape() : (lscalar ? ls : rs),
                           lhs->requires_grad() || rhs->requires_grad());
        detail::gx(gnnx_binary_bcast_f32(op, 1, (int64_t)big->numel(), lhs->device_data(), 0, lhs->numel() == 1 ? 0 : 1,
                                         rhs->device_data(), 0, rhs->numel() == 1 ? 0 : 1, out->device_out(), (int64_t)big->numel(),
                                         detail::current_stream()), what);
        return out;
    }
    if (!as_2d(ls, lr, lc) || !as_2d(rs, rr, rc)) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    if (!is_broadcastable(ls, rs)) CHECK_EQUAL_SIZES(rs, ls);
    const size_t n = std::max(lr, rr), f = std::max(lc, rc);
    if ((lr != n && lr != 1) || (rr != n && rr != 1) || (lc != f && lc != 1) || (rc != f && rc != 1))
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    std::vector<size_t> od = (ls.size() == 2 || rs.size() == 2) ? std::vector<size_t>{n, f} : std::vector<size_t>{f};
    auto out = new_out(od, lhs->requires_grad() || rhs->requires_grad());
    auto stride = [&](size_t r, size_t c, int64_t &s_r, int64_t &s_c) {
        s_r = (r == n && n > 1) ? (int64_t)c : 0;
        s_c = (c == f && f > 1) ? 1 : 0;
    };
    int64_t ars, acs, brs, bcs;
    stride(lr, lc, ars, acs);
    stride(rr, rc, brs, bcs);
    detail::gx(gnnx_binary_bcast_f32(op, (int64_t)n, (int64_t)f, lhs->device_data(), ars, acs, rhs->device_data(), brs, bcs,
                                     out->device_out(), (int64_t)f, detail::current_stream()), what);
    return out;
}
}  // namespace impl

// [N,F]+[N,F] (same shape: copy + axpy) and [N,F]+[F] (bias broadcast, reference functional.h:163-187)
template <class T>
tptr<T> add(const tptr<T> &lhs, const tptr<T> &rhs)
{
    if constexpr (!std::is_same_v<T, float>) {
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    } else {
        const bool req = lhs->requires_grad() || rhs->requires_grad();
        void *st = detail::current_stream();
        auto ls = lhs->shape(), rs = rhs->shape();
        if (ls == rs) {
            auto out = impl::new_out(ls, req);
            float *o = out->device_out();
            detail::gx(gnnx_memcpy_d2d(o, lhs->device_data(), lhs->numel() * sizeof(float), st), "add");
            detail::gx(gnnx_axpy_f32((int64_t)lhs->numel(), 1.0f, rhs->device_data(), o, st), "add");
            return out;
        }
        if (impl::is_row_of(ls, rs)) {
            auto out = impl::new_out(ls, req);
            detail::gx(gnnx_bias_add_f32(lhs->device_data(), (int64_t)ls[1], rhs->device_data(), (int64_t)ls[0], (int32_t)ls[1],
                                         out->device_out(), (int64_t)ls[1], st), "add");
            return out;
        }
        if (impl::is_row_of(rs, ls)) return add(rhs, lhs);
        return impl::binary_bcast(GNNX_OP_ADD, lhs, rhs, "add");  // any other 2-D broadcast
    }
}

// [N,F]*[N,1] (row scale, reference functional.h:190-213 + utils.h:181-228) and same-shape [N,1]*[N,1]
template <class T>
tptr<T> mul(const tptr<T> &lhs, const tptr<T> &rhs)
{
    if constexpr (!std::is_same_v<T, float>) {
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    } else {
        const bool req = lhs->requires_grad() || rhs->requires_grad();
        void *st = detail::current_stream();
        auto ls = lhs->shape(), rs = rhs->shape();
        if (impl::is_col_of(ls, rs)) {
            auto out = impl::new_out(ls, req);
            detail::gx(gnnx_rowscale_f32(lhs->device_data(), (int64_t)ls[1], rhs->device_data(), (int64_t)ls[0], (int32_t)ls[1],
                                         out->device_out(), (int64_t)ls[1], st), "mul");
            return out;
        }
        if (impl::is_col_of(rs, ls)) return mul(rhs, lhs);
        if (ls == rs) {  // elementwise: one value per "row"
            auto out = impl::new_out(ls, req);
            detail::gx(gnnx_rowscale_f32(lhs->device_data(), 1, rhs->device_data(), (int64_t)lhs->numel(), 1, out->device_out(), 1, st),
                       "mul");
            return out;
        }
        return impl::binary_bcast(GNNX_OP_MUL, lhs, rhs, "mul");  // any other 2-D broadcast
    }
}

// elementwise quotient with the same broadcast (reference functional.h:216-239); off the hot path, here for operator/
template <class T>
tptr<T> div(const tptr<T> &lhs, const tptr<T> &rhs)
{
    if constexpr (!std::is_same_v<T, float>) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    else return impl::binary_bcast(GNNX_OP_DIV, lhs, rhs, "div");
}

// elementwise power with a scalar exponent (deg->pow(-0.5), reference graph.cpp:183 -> functional.h:242-264)
template <class T>
tptr<T> pow(const tptr<T> &base, float exponent)
{
    if constexpr (!std::is_same_v<T, float>) {
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    } else {
        auto out = impl::new_out(base->shape(), false);
        detail::gx(gnnx_pow_f32(base->device_data(), (int64_t)base->numel(), exponent, out->device_out(), detail::current_stream()),
                   "pow");
        return out;
    }
}

// sum(-1, keepdim) of a CSR adjacency = out-degrees (reference graph.cpp:178); sum(0) of [N,F] = column sums
// (sum_to_size in Add::_backward); reference functional.h:267-296
template <class T>
tptr<T> sum(const tptr<T> &base, int dim, bool keepdim)
{
    if constexpr (!std::is_same_v<T, float>) {
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    } else {
        void *st = detail::current_stream();
        auto shp = base->shape();
        const int rank = (int)shp.size();
        if (dim != INT_MAX && dim < 0) dim += rank;
        if (base->is_csr()) {
            if (dim != 1) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
            auto c = base->csr();
            std::vector<size_t> od = keepdim ? std::vector<size_t>{(size_t)c->n, 1} : std::vector<size_t>{(size_t)c->n};
            auto out = impl::new_out(od, false);
            if (base->csr_transposed()) c->ensure_transpose();
            detail::gx(gnnx_csr_rowsum_f32((const int32_t *)(base->csr_transposed() ? c->rowptr_t : c->rowptr),
                                           (const float *)(base->csr_transposed() ? c->vals_t : c->vals), c->n, out->device_out(), st),
                       "sum");
            return out;
        }
        if (rank == 2 && dim == 0) {
            std::vector<size_t> od = keepdim ? std::vector<size_t>{1, shp[1]} : std::vector<size_t>{shp[1]};
            auto out = impl::new_out(od, base->requires_grad());
            size_t wsb = 0;
            detail::gx(gnnx_colsum_workspace((int64_t)shp[0], (int32_t)shp[1], &wsb), "sum");
            detail::gx(gnnx_colsum_f32(base->device_data(), (int64_t)shp[1], (int64_t)shp[0], (int32_t)shp[1], 0.0f, out->device_out(),
                                       detail::workspace(wsb), wsb, st), "sum");
            return out;
        }
        if (rank == 2 && dim == 1) {  // row sums: undoes a column broadcast (sum_to_size to [N,1])
            std::vector<size_t> od = keepdim ? std::vector<size_t>{shp[0], 1} : std::vector<size_t>{shp[0]};
            auto out = impl::new_out(od, base->requires_grad());
            detail::gx(gnnx_rowsum_f32(base->device_data(), (int64_t)shp[1], (int64_t)shp[0], (int32_t)shp[1], out->device_out(), st),
                       "sum");
            return out;
        }
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    }
}

// [N,N](CSR) . [N,F] -> CSR SpMM ;  [M,K] . [K,N] -> MFMA GEMM, either operand possibly a transposed view
// (reference functional.h:399-441)
template <class T>
tptr<T> matmul(const tptr<T> &lhs, const tptr<T> &rhs)
{
    if constexpr (!std::is_same_v<T, float>) {
        throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    } else {
        void *st = detail::current_stream();
        const bool req = lhs->requires_grad() || rhs->requires_grad();
        auto ls = lhs->shape(), rs = rhs->shape();
        if (ls.size() != 2 || rs.size() != 2) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        if (ls[1] != rs[0]) throw std::runtime_error(ERROR_MM_COMPATIBLE);
        if (rhs->is_csr()) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        if (lhs->is_csr()) {
            auto c = lhs->csr();
            const bool tr = lhs->csr_transposed();
            if (tr) c->ensure_transpose();
            auto out = impl::new_out({ls[0], rs[1]}, req);
            detail::gx(gnnx_spmm_csr_f32(c->n, c->n, (int32_t)rs[1], (const int32_t *)(tr ? c->rowptr_t : c->rowptr),
                                         (const int32_t *)(tr ? c->colidx_t : c->colidx), (const float *)(tr ? c->vals_t : c->vals),
                                         nullptr, nullptr, nullptr, rhs->device_data(), (int64_t)rs[1], 0.0f, out->device_out(),
                                         (int64_t)rs[1], nullptr, st),
                       "matmul");
            return out;
        }
        bool ta = false, tb = false;
        const float *A = lhs->device_storage(ta);
        const float *B = rhs->device_storage(tb);
        const int64_t M = (int64_t)ls[0], K = (int64_t)ls[1], N = (int64_t)rs[1];
        const int64_t lda = ta ? M : K, ldb = tb ? K : N;  // leading dims of the buffers as stored
        auto out = impl::new_out({ls[0], rs[1]}, req);
        if (const size_t p = detail::PitchRequest::take((size_t)M, (size_t)N)) out->set_device_pitch(p);   // rows on the gather pitch
        int64_t ldc = N;
        float *C = out->device_out_pitched(ldc);
        size_t wsb = 0;
        if (!ta && tb) {  // x . W^T: an opt-in request for the batch statistics of the output's columns rides in the epilogue
            if (detail::BnStatsRequest *rq = detail::BnStatsRequest::take((size_t)M, (size_t)N)) {
                detail::gx(gnnx_gemm_bn_stats_workspace(M, N, K, &wsb), "matmul");
                detail::gx(gnnx_gemm_bn_stats_f32(M, N, K, A, lda, B, ldb, C, ldc, rq->mean, rq->var,
                                                  wsb ? detail::workspace(wsb) : nullptr, wsb, st), "matmul");
                rq->done = true;
                return out;
            }
            // the sharded layer's transform: rows other ranks need leave for the send buffer from the product's epilogue
            const bool pieces = N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0 && ldc % 4 == 0;
            if (detail::SendSlotsRequest *rq = pieces ? detail::SendSlotsRequest::take((size_t)M, (size_t)N) : nullptr) {
                detail::gx(gnnx_gemm_nt_rows_to_slots_workspace(M, N, K, &wsb), "matmul");
                detail::gx(gnnx_gemm_nt_rows_to_slots_f32(M, N, K, A, lda, B, ldb, C, ldc, rq->slots, rq->send, N,
                                                          wsb ? detail::workspace(wsb) : nullptr, wsb, st), "matmul");
                rq->done = true;
                return out;
            }
        }
        detail::gx(gnnx_gemm_workspace(ta, tb, M, N, K, &wsb), "matmul");
        detail::gx(gnnx_gemm_f32(ta, tb, M, N, K, 1.0f, A, lda, B, ldb, 0.0f, C, ldc, wsb ? detail::workspace(wsb) : nullptr,
                                 wsb, st), "matmul");
        return out;
    }
}

// global max of an index tensor (edge_index->max(), reference graph.cpp:25,89 -> functional.h:25-71); host side
template <class T>
std::tuple<tptr<T>, tptr<int>> max(const tensor<T> &t)
{
    auto &self = const_cast<tensor<T> &>(t);
    auto *h = self.data();
    size_t arg = 0;
    for (size_t i = 1; i < h->size(); i++)
        if ((*h)[i] > (*h)[arg]) arg = i;
    auto v = std::make_shared<tensor<T>>(std::vector<size_t>{1}, (*h)[arg], false);
    auto a = std::make_shared<tensor<int>>(std::vector<size_t>{1}, (int)arg, false);
    return {v, a};
}

}  // namespace functional
}  // namespace cyg

#endif
