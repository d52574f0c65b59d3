// Multi-GPU side of the graph layer over the C-ABI: dist::Comm, graph::Partition and the sharded GCNConv forward /
// backward (data flow as in gnn.cpp_amd/shard.py, which the tests compare it with).
// The reference has nothing to mirror here; what is preserved is its layer semantics
// (graph.cpp:170-212) on a rank's rows and the summation order of every row.
#include <algorithm>
#include <cstring>

#include "dist.h"
#include "graph.h"
#include "nn.h"

using namespace cyg;
using cyg::detail::current_stream;
using cyg::detail::dev_alloc;
using cyg::detail::dev_free;
using cyg::detail::gx;
using cyg::detail::workspace;

// ---------------------------------------------------------------------------------------------------- dist::Comm
namespace dist {

Comm::~Comm() { gnnx_comm_destroy(_h); }

std::vector<std::shared_ptr<Comm>> Comm::local_group(int world)
{
    std::vector<gnnx_comm *> hs((size_t)world, nullptr);
    gx(gnnx_comm_init_local(hs.data(), world), "comm");
    std::vector<std::shared_ptr<Comm>> out;
    for (int r = 0; r < world; r++) out.push_back(std::shared_ptr<Comm>(new Comm(hs[r], world, r)));
    return out;
}

void Comm::unique_id(unsigned char id_out[128]) { gx(gnnx_comm_unique_id(id_out), "comm"); }

std::shared_ptr<Comm> Comm::rccl(int world, int rank, const unsigned char id[128])
{
    gnnx_comm *h = nullptr;
    gx(gnnx_comm_init(&h, world, rank, id), "comm");
    return std::shared_ptr<Comm>(new Comm(h, world, rank));
}

void Comm::allreduce_sum(float *d_buf, int64_t n) { gx(gnnx_allreduce_sum_f32(_h, d_buf, n, current_stream()), "allreduce"); }

}  // namespace dist

// ---------------------------------------------------------------------------------------------------- graph::Partition
namespace graph {

namespace {
struct Scratch {  // pooled device scratch, returned on scope exit
    void *p;
    size_t bytes;
    explicit Scratch(size_t b) : p(dev_alloc(std::max<size_t>(b, 4))), bytes(std::max<size_t>(b, 4)) {}
    ~Scratch() { dev_free(p, bytes); }
    template <class U> U *as() { return static_cast<U *>(p); }
};
}  // namespace

void Partition::build_side(Side &s, const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, bool transpose)
{
    void *st = current_stream();
    const int rank = comm->rank(), world = comm->world();
    Scratch rows(sizeof(int32_t) * (size_t)n_edges), cols(sizeof(int32_t) * (size_t)n_edges);
    int64_t m = 0;
    gx(gnnx_shard_select_edges(d_src, d_dst, n_edges, (const int32_t *)_owner, (const int32_t *)_nid, rank, _lo, transpose ? 1 : 0,
                               rows.as<int32_t>(), cols.as<int32_t>(), &m, st), "partition");
    const int32_t n_csr = (int32_t)std::max<int64_t>(_n_local, (int64_t)_n);  // columns are original ids
    gx(gnnx_malloc(&s.rowptr, sizeof(int32_t) * ((size_t)n_csr + 1)), "partition");
    gx(gnnx_malloc(&s.colidx, sizeof(int32_t) * (size_t)std::max<int64_t>(m, 1)), "partition");
    size_t wsb = 0;
    gx(gnnx_csr_from_coo_workspace(m, n_csr, &wsb), "partition");
    // a local row id may equal an unrelated original column id: self loops were already dropped on original ids
    gx(gnnx_csr_from_coo(rows.as<int32_t>(), cols.as<int32_t>(), m, n_csr, GNNX_CSR_KEEP_SELF_LOOPS, (int32_t *)s.rowptr,
                         (int32_t *)s.colidx, &s.nnz, workspace(wsb), wsb, st), "partition");
    gx(gnnx_halo_plan_create((const int32_t *)s.colidx, s.nnz, (const int32_t *)_nid, (int32_t)_n, world, rank, _cuts.data(),
                             (int32_t *)s.colidx, &s.plan, st), "partition");
}

Partition::Partition(const tensor<int> &edge_index, size_t num_nodes, std::shared_ptr<dist::Comm> comm_, int row_weight)
    : comm(std::move(comm_)), _n(num_nodes)
{
    auto &ei = const_cast<tensor<int> &>(edge_index);
    if (ei.rank() != 2 || ei.shape()[0] != 2) throw std::runtime_error("invalid input for x, must be of 2D");
    const int64_t e = (int64_t)ei.shape()[1];
    const int32_t *d = ei.device_data();
    const int32_t *d_src = d, *d_dst = d + e;
    void *st = current_stream();
    const int world = comm->world(), rank = comm->rank();
    {
        Scratch w(sizeof(int32_t) * _n);
        gx(gnnx_vertex_weights(d_src, d_dst, e, (int32_t)_n, row_weight, w.as<int32_t>(), st), "partition");
        gx(gnnx_malloc(&_owner, sizeof(int32_t) * std::max<size_t>(_n, 1)), "partition");
        gx(gnnx_malloc(&_nid, sizeof(int32_t) * std::max<size_t>(_n, 1)), "partition");
        _cuts.assign((size_t)world + 1, 0);
        gx(gnnx_partition_deal(w.as<int32_t>(), (int32_t)_n, world, (int32_t *)_owner, (int32_t *)_nid, _cuts.data(), st), "partition");
        // spread every rank's rows inside its range: synthetic power-law hubs sit on ids with few one-bits, whose feature rows
        // pile onto a few memory channels (include/gnnx.h, gnnx_partition_scramble)
        gx(gnnx_partition_scramble((const int32_t *)_owner, (int32_t)_n, world, _cuts.data(), (int32_t *)_nid, st), "partition");
    }
    _lo = _cuts[rank];
    _n_local = _cuts[rank + 1] - _cuts[rank];
    build_side(fwd, d_src, d_dst, e, false);
    build_side(bwd, d_src, d_dst, e, true);
    for (Side *s : {&fwd, &bwd}) {
        gx(gnnx_halo_plan_exchange_requests(s->plan, comm->handle(), st), "partition");
        gx(gnnx_halo_plan_info(s->plan, nullptr, &s->n_halo, &s->n_send, nullptr, nullptr, nullptr, nullptr), "partition");
    }
    // degree block on the shard (reference graph.cpp:177-185): s of my rows from my degrees, s of the halo columns by one
    // exchange, norm in the reference's summation order; then norm of the backward halo, handed to the transposed
    // aggregation per non-zero
    const size_t nl = (size_t)_n_local;
    {
        const size_t n_ext = nl + (size_t)fwd.n_halo;
        s_ext = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{std::max<size_t>(n_ext, 1), 1}, false);
        float *sx = s_ext->device_out();
        gx(gnnx_memset(sx, 0, sizeof(float) * std::max<size_t>(n_ext, 1), st), "partition");
        gx(gnnx_degree_norm_f32((const int32_t *)fwd.rowptr, (const int32_t *)fwd.colidx, (int32_t)nl, sx, nullptr, nullptr, st), "partition");
        exchange(fwd, sx, 1);
        norm = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{std::max<size_t>(nl, 1), 1}, false);
        gx(gnnx_degree_norm_f32((const int32_t *)fwd.rowptr, (const int32_t *)fwd.colidx, (int32_t)nl, nullptr, sx, norm->device_out(), st),
           "partition");
        Scratch nb(sizeof(float) * (nl + (size_t)bwd.n_halo));
        gx(gnnx_memset(nb.p, 0, nb.bytes, st), "partition");
        if (nl) gx(gnnx_memcpy_d2d(nb.p, norm->device_data(), sizeof(float) * nl, st), "partition");
        exchange(bwd, nb.as<float>(), 1);
        gx(gnnx_malloc(&norm_nz_bwd, sizeof(float) * (size_t)std::max<int64_t>(bwd.nnz, 1)), "partition");
        gx(gnnx_gather_rows_f32(nb.as<float>(), 1, (const int32_t *)bwd.colidx, bwd.nnz, 1, (float *)norm_nz_bwd, 1, st), "partition");
        gx(gnnx_stream_sync(st), "partition");  // scratch goes back to the pool below
    }
}

std::vector<int> Partition::halo_new_ids(const Side &s) const
{
    std::vector<int> out((size_t)s.n_halo, -1);
    if (s.n_halo == 0) return out;
    const int32_t *d_ids = nullptr;
    gx(gnnx_halo_plan_info(s.plan, nullptr, nullptr, nullptr, nullptr, nullptr, &d_ids, nullptr), "partition");
    gx(gnnx_memcpy_d2h(out.data(), d_ids, sizeof(int32_t) * out.size(), current_stream()), "partition");
    return out;
}

Partition::~Partition()
{
    for (Side *s : {&fwd, &bwd}) {
        if (s->plan) gnnx_halo_plan_destroy(s->plan);
        if (s->spmm_plan) gnnx_spmm_plan_destroy(s->spmm_plan);
        if (s->rowptr) gnnx_free(s->rowptr);
        if (s->colidx) gnnx_free(s->colidx);
    }
    for (void *p : {_owner, _nid, _verts, norm_nz_bwd})
        if (p) gnnx_free(p);
}

void Partition::ensure_spmm_plans(int32_t n_feat)
{
    constexpr int32_t kChunk = 1024;  // rows longer than this go to the sequential hub kernel
    for (Side *s : {&fwd, &bwd}) {
        if (s->spmm_plan) continue;   // a plan does not depend on the feature width
        if (s->spmm_plan) gnnx_spmm_plan_destroy(s->spmm_plan);
        s->spmm_plan = nullptr;
        gx(gnnx_spmm_plan_create((const int32_t *)s->rowptr, (int32_t)_n_local, kChunk, n_feat, &s->spmm_plan, current_stream()), "plan");
        s->spmm_plan_feat = n_feat;
    }
}

void Partition::exchange(const Side &s, float *d_buf, int32_t n_feat)
{
    if (comm->world() == 1) return;
    Scratch send(sizeof(float) * (size_t)std::max<int64_t>(s.n_send, 1) * (size_t)std::max(n_feat, 1));
    gx(gnnx_halo_exchange_rows_f32(s.plan, comm->handle(), d_buf, n_feat, n_feat, send.as<float>(), current_stream()), "halo exchange");
    // the send buffer goes back to this thread's pool: the next kernel that takes it is ordered behind the exchange on the
    // same stream (RCCL) or the exchange has completed (local transport)
}

void Partition::exchange_packed(const Side &s, float *d_buf, int32_t n_feat, const float *d_send)
{
    if (comm->world() == 1) return;
    gx(gnnx_halo_exchange_packed_f32(s.plan, comm->handle(), d_buf, n_feat, n_feat, d_send, current_stream()), "halo exchange");
}

std::vector<int> Partition::local_vertices()
{
    std::vector<int32_t> owner(_n), nid(_n);
    if (_n) {
        gx(gnnx_memcpy_d2h(owner.data(), _owner, sizeof(int32_t) * _n, current_stream()), "partition");
        gx(gnnx_memcpy_d2h(nid.data(), _nid, sizeof(int32_t) * _n, current_stream()), "partition");
    }
    std::vector<int> out((size_t)_n_local, -1);
    const int rank = comm->rank();
    for (size_t v = 0; v < _n; v++)
        if (owner[v] == rank) out[(size_t)(nid[v] - _lo)] = (int)v;   // local row k holds the vertex whose new id is lo + k
    return out;
}

tptr<float> Partition::take_rows(const tptr<float> &full)
{
    if (full->rank() != 2 || full->shape()[0] != _n) throw std::runtime_error(ERROR_SIZE_MISMATCH);
    void *st = current_stream();
    if (!_verts) {
        auto v = local_vertices();
        gx(gnnx_malloc(&_verts, sizeof(int32_t) * std::max<size_t>(v.size(), 1)), "partition");
        if (!v.empty()) gx(gnnx_memcpy_h2d(_verts, v.data(), sizeof(int32_t) * v.size(), st), "partition");
    }
    const size_t f = full->shape()[1];
    auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{(size_t)_n_local, f}, false);
    gx(gnnx_gather_rows_f32(full->device_data(), (int64_t)f, (const int32_t *)_verts, _n_local, (int32_t)f, out->device_out(), (int64_t)f,
                            st), "partition");
    return out;
}

// ---------------------------------------------------------------------------------------------------- sharded layer
namespace {

// H [n_local, F] -> ( BatchNorm with GLOBAL batch statistics -> ReLU -> ) aggregation over [local | halo] + bias, as ONE
// autograd op.  Forward: H's rows go into the head of a [n_local + n_halo, F] buffer, one all-to-all-v fills the tail with
// the owners' RAW rows of H, and the SpMM gathers from it -- BatchNorm / ReLU applied to every gathered row from per-column
// constants (gnnx_spmm_csr_fused_f32), so the normalised activations are never stored or exchanged.  Backward pulls rows
// of G for the in-neighbours the same way through the transposed shard (no scatter-add, no atomics).  Parameter
// gradients stay LOCAL partial sums until GCNConv::allreduce_gradients().
class ShardedAggregateOp : public cyg::Operation<tensor<float>> {
public:
    std::shared_ptr<Partition> part;
    tptr<float> mean, var;  // global batch statistics (BatchNorm mode)
    float eps = 1e-5f;
    bool use_bn = false, has_beta = false;
    const float *packed_send = nullptr;   // forward only: the send buffer of h's rows, filled by the transform's epilogue (or null)
    ShardedAggregateOp() { name = "GCNShardedAggregate"; }

    tptr<float> forward(const tptr<float> &h, const tptr<float> &bias, const tptr<float> &gamma, const tptr<float> &beta)
    {
        const auto shp = h->shape();
        const int64_t nl = (int64_t)part->num_local();
        if (shp.size() != 2 || (int64_t)shp[0] != nl) throw std::runtime_error(ERROR_SIZE_MISMATCH);
        const int32_t f = (int32_t)shp[1];
        void *st = current_stream();
        part->ensure_spmm_plans(f);
        use_bn = (bool)gamma;
        has_beta = (bool)beta;
        if (use_bn) {  // two [F] all-reduces: global mean, then centred squares against it (exact two-pass variance)
            const float inv_n = 1.0f / (float)part->num_nodes();
            mean = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            var = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            size_t wsb = 0;
            gx(gnnx_bn_workspace(nl, f, &wsb), "BatchNorm");
            gx(gnnx_bn_partial_f32(h->device_data(), f, nl, f, nullptr, inv_n, mean->device_out(), workspace(wsb), wsb, st), "BatchNorm");
            part->comm->allreduce_sum(mean->device_inplace(), f);
            gx(gnnx_bn_partial_f32(h->device_data(), f, nl, f, mean->device_data(), inv_n, var->device_out(), workspace(wsb), wsb, st),
               "BatchNorm");
            part->comm->allreduce_sum(var->device_inplace(), f);
        }
        // [local | halo] buffer: the transform already wrote its rows at the head of one (GCNConv::forward_sharded reserves the halo
        // rows behind the product's output: the all-to-all-v receives straight into them, no copy); an `h` from elsewhere is copied
        const size_t halo_elems = (size_t)part->fwd.n_halo * (size_t)f;
        const bool in_place = h->device_tail_capacity() >= halo_elems;
        Scratch hext_buf(in_place ? 0 : sizeof(float) * ((size_t)nl * f + halo_elems));   // back to the pool on every exit path
        float *hext = in_place ? h->device_data() : hext_buf.as<float>();
        if (!in_place && nl) gx(gnnx_memcpy_d2d(hext, h->device_data(), sizeof(float) * (size_t)nl * f, st), "aggregate");
        if (packed_send) part->exchange_packed(part->fwd, hext, f, packed_send);   // the transform's epilogue filled the send buffer
        else part->exchange(part->fwd, hext, f);
        auto snapshot = [&](const float *src, size_t rows) {
            auto t = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{std::max<size_t>(rows, 1), (size_t)f}, false);
            if (rows) gx(gnnx_memcpy_d2d(t->device_out(), src, sizeof(float) * rows * (size_t)f, st), "trace");
            return t;
        };
        if (part->trace) part->trace->h_ext = snapshot(hext, (size_t)(nl + part->fwd.n_halo));
        const bool req = h->requires_grad() || (bias && bias->requires_grad()) || (use_bn && gamma->requires_grad()) ||
                         (has_beta && beta->requires_grad());
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, req);
        gnnx_spmm_fusion fu{};
        if (use_bn) {
            fu.bn_mean = mean->device_data();
            fu.bn_var = var->device_data();
            fu.bn_gamma = gamma->device_data();
            fu.bn_beta = has_beta ? beta->device_data() : nullptr;
            fu.bn_eps = eps;
            fu.relu_in = 1;
        }
        int rc = gnnx_spmm_csr_fused_f32((int32_t)nl, (int32_t)(nl + part->fwd.n_halo), f, (const int32_t *)part->fwd.rowptr,
                                         (const int32_t *)part->fwd.colidx, nullptr, nullptr, part->norm->device_data(),
                                         bias ? bias->device_data() : nullptr, hext, f, 0.0f, out->device_out(), f, use_bn ? &fu : nullptr,
                                         part->fwd.spmm_plan, st);
        gx(rc, "aggregate");
        if (part->trace) part->trace->out = snapshot(out->device_data(), (size_t)nl);
        has_bias = (bool)bias;
        if (req) context->save_for_backward({h, bias ? bias : h, use_bn ? gamma : h, has_beta ? beta : h});
        return out;
    }

    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        auto v = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(v, 4);
        auto h = v[0], bias = v[1], gamma = v[2], beta = v[3];
        const auto shp = h->shape();
        const int64_t nl = (int64_t)part->num_local();
        const int32_t f = (int32_t)shp[1];
        void *st = current_stream();
        if (has_bias && bias->requires_grad()) bias->backward(cyg::functional::sum(g, 0, bias->rank() == 2));  // local colsum(G)
        if (!(h->requires_grad() || (use_bn && gamma->requires_grad()) || (has_beta && beta->requires_grad()))) return;
        part->ensure_spmm_plans(f);
        Scratch gext_buf(sizeof(float) * (size_t)(nl + part->bwd.n_halo) * (size_t)f);
        float *gext = gext_buf.as<float>();
        if (nl) gx(gnnx_memcpy_d2d(gext, g->device_data(), sizeof(float) * (size_t)nl * f, st), "aggregate");
        part->exchange(part->bwd, gext, f);
        auto snapshot = [&](const float *src, size_t rows) {
            auto t = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{std::max<size_t>(rows, 1), (size_t)f}, false);
            if (rows) gx(gnnx_memcpy_d2d(t->device_out(), src, sizeof(float) * rows * (size_t)f, st), "trace");
            return t;
        };
        if (part->trace) part->trace->g_ext = snapshot(gext, (size_t)(nl + part->bwd.n_halo));
        auto dy = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        int rc = gnnx_spmm_csr_f32((int32_t)nl, (int32_t)(nl + part->bwd.n_halo), f, (const int32_t *)part->bwd.rowptr,
                                   (const int32_t *)part->bwd.colidx, (const float *)part->norm_nz_bwd, nullptr, nullptr, nullptr, gext, f,
                                   0.0f, dy->device_out(), f, part->bwd.spmm_plan, st);
        gx(rc, "aggregate");
        if (part->trace) part->trace->dy = snapshot(dy->device_data(), (size_t)nl);
        if (!use_bn) {
            if (h->requires_grad()) h->backward(dy);
            return;
        }
        // BatchNorm + ReLU backward over the GLOBAL batch: local sums -> all-reduce of two [F] vectors -> apply with N_global
        size_t wsb = 0;
        gx(gnnx_bn_workspace(nl, f, &wsb), "BatchNorm");
        auto dgamma = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        auto dbeta = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        gx(gnnx_bn_relu_bwd_sums_f32(h->device_data(), f, nullptr, 0, dy->device_data(), f, nl, f, mean->device_data(), var->device_data(), eps,
                                     gamma->device_data(), has_beta ? beta->device_data() : nullptr, 1, dgamma->device_out(),
                                     dbeta->device_out(), workspace(wsb), wsb, st), "BatchNorm");
        auto dgamma_all = dgamma->clone(false), dbeta_all = dbeta->clone(false);
        part->comm->allreduce_sum(dgamma_all->device_inplace(), f);
        part->comm->allreduce_sum(dbeta_all->device_inplace(), f);
        auto dh = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        gx(gnnx_bn_relu_bwd_apply_f32(h->device_data(), f, nullptr, 0, dy->device_data(), f, nl, f, mean->device_data(), var->device_data(),
                                      eps, gamma->device_data(), has_beta ? beta->device_data() : nullptr, 1, dgamma_all->device_data(),
                                      dbeta_all->device_data(), (int64_t)part->num_nodes(), dh->device_out(), f, workspace(wsb), wsb, st),
           "BatchNorm");
        if (h->requires_grad()) h->backward(dh);
        if (gamma->requires_grad()) gamma->backward(dgamma);  // local partial sums, like every parameter gradient
        if (has_beta && beta->requires_grad()) beta->backward(dbeta);
    }
    bool has_bias = false;
};

}  // namespace

void GCNConv::shard(std::shared_ptr<Partition> part)
{
    _part = std::move(part);
    invalidate_graph_cache();
}

// the reference's layer (graph.cpp:170-191) on this rank's rows: transform (row-parallel, W replicated: no communication)
// -> [BatchNorm -> ReLU] -> normalised aggregation with one halo exchange -> + bias
tptr<float> GCNConv::forward_sharded(const tptr<float> &x)
{
    if (x->rank() != 2 || x->shape()[0] != _part->num_local()) throw std::runtime_error(ERROR_SIZE_MISMATCH);
    tptr<float> h;
    // the rows other ranks need leave for the send buffer from the transform's epilogue (tensor.h: SendSlotsRequest); the buffer lives
    // until the exchange inside the aggregation op below has been issued, then goes back to this thread's pool (same stream)
    const int32_t *slots = nullptr;
    if (_part->comm->world() > 1) gx(gnnx_halo_plan_slot_table(_part->fwd.plan, &slots), "halo plan");
    Scratch send(slots ? sizeof(float) * (size_t)std::max<int64_t>(_part->fwd.n_send, 1) * (size_t)_out_channels : 0);
    bool packed = false;
    {   // the product's output [n_local, F_out] gets room for the halo rows behind it (tensor.h: TailReservation)
        cyg::detail::TailReservation halo_rows(_part->num_local() * _out_channels, (size_t)_part->fwd.n_halo * _out_channels);
        cyg::detail::SendSlotsRequest send_rows(_part->num_local(), _out_channels, slots, slots ? send.as<float>() : nullptr);
        h = (*get_module("lin"))(x);
        packed = send_rows.done;
    }
    if (packed) _part->packed_transforms++;
    auto op = std::make_unique<ShardedAggregateOp>();
    op->part = _part;
    op->packed_send = packed ? send.as<float>() : nullptr;
    tptr<float> gamma, beta;
    if (!hot_path_only) {
        auto *bn = dynamic_cast<nn::BatchNorm *>(get_module("bnorm").get());
        if (!bn || !bn->uses_batch_stats()) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);  // eval-mode BN: not sharded yet
        op->eps = bn->_eps;
        gamma = bn->get_parameter("gammas");
        if (bn->_affine) beta = bn->get_parameter("betas");
    }
    auto res = op->forward(h, get_parameter("bias"), gamma, beta);
    if (res->requires_grad()) res->grad_fn = std::move(op);
    return res;
}

// data-parallel reduction of the parameter gradients (W, bias, BatchNorm gammas / betas): [F_out x F_in] + O(F) floats
void GCNConv::allreduce_gradients()
{
    if (!_part) return;
    for (auto &p : parameters()) {
        float *g = p->device_grad_inplace();
        if (g) _part->comm->allreduce_sum(g, (int64_t)p->numel());
    }
}

}  // namespace graph
