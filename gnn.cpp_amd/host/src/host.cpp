// Backend glue of the cyg / nn / graph API: C-ABI status -> exception, stream + workspace, CSR storage,
// the module registry, Linear, and the graph layer (reference src/utils.cpp, src/tensor.cpp, src/nn.cpp:12-211,
// src/graph.cpp re-implemented over include/gnnx.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iterator>
#include <mutex>
#include <random>
#include <unordered_map>
#include <vector>

#include "graph.h"
#include "nn.h"
#include "tensor.h"

using namespace cyg;

// ---------------------------------------------------------------------------------------------------- detail
namespace cyg {
namespace detail {

// Per-THREAD runtime state.  Contract: a rank (dist.h) = one host thread = one in-order stream of its own.
//   * The stream is created on the thread's first use of the backend (a non-blocking stream on the device current at that moment;
//     no thread ever issues on the legacy NULL stream, whose implicit cross-stream synchronisation would couple the rank threads of
//     an in-process group) and destroyed, drained, when the thread ends.  set_current_stream() lets an embedding application
//     supply its own (not owned, never destroyed here).
//   * Scratch and the allocator cache belong to the thread: a cached block is only ever re-issued to work queued behind its last
//     user on the same stream -- no events needed.  Every block carries its owner: a block released by ANOTHER thread (a tensor that
//     outlived its rank thread, or was handed across threads) is not cached but given back to the driver, whose free synchronises the
//     device -- so the contract "tensors are used by the thread that made them" is a performance rule, not a correctness one.
//   * Everything that crosses threads is ordered explicitly: the in-process collectives (gnnx_comm.hip: stream synchronisation on
//     both sides of a host barrier) and the driver's own device-wide synchronisation in hipFree.
struct ThreadRuntime;
static ThreadRuntime &rt();
static std::mutex g_owner_mu;
static std::unordered_map<void *, ThreadRuntime *> g_owner;   // live block -> the thread runtime whose stream last used it

struct ThreadRuntime {
    void *stream = nullptr;
    bool stream_owned = false, stream_set = false;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    std::unordered_map<size_t, std::vector<void *>> blocks;

    void *get_stream()
    {
        if (!stream_set) {
            // diagnostic switch (tests/cpp/test_host_sharded_gpu: round 4's configuration under round 5's stage-by-stage diagnosis):
            // every thread on the legacy NULL stream, as all rank threads were before round 5.  Never set in normal use.
            static const bool shared_null = std::getenv("GNNCPP_SHARED_NULL_STREAM") != nullptr;
            if (shared_null) {
                stream = nullptr;
                stream_owned = false;
            } else {
                gx(gnnx_stream_create(&stream), "stream");
                stream_owned = true;
            }
            stream_set = true;
        }
        return stream;
    }
    void drain()
    {
        if (stream_set) (void)gnnx_stream_sync(stream);
    }
    void release_blocks()
    {
        {
            std::lock_guard<std::mutex> lk(g_owner_mu);
            for (auto &kv : blocks)
                for (void *p : kv.second) g_owner.erase(p);
        }
        for (auto &kv : blocks) {
            for (void *p : kv.second) gnnx_free(p);
            kv.second.clear();
        }
    }
    // A rank thread of dist::Comm::local_group that exits must not strand its blocks, its scratch or its stream.  Blocks of this
    // thread that are still alive inside tensors (owned by someone else now) lose their owner: their release frees them directly.
    ~ThreadRuntime()
    {
        drain();
        release_blocks();
        if (ws) gnnx_free(ws);
        ws = nullptr;
        ws_bytes = 0;
        {
            std::lock_guard<std::mutex> lk(g_owner_mu);
            for (auto it = g_owner.begin(); it != g_owner.end();) it = it->second == this ? g_owner.erase(it) : std::next(it);
        }
        if (stream_owned && stream) (void)gnnx_stream_destroy(stream);
        stream = nullptr;
        stream_set = stream_owned = false;
    }
};
static ThreadRuntime &rt()
{
    static thread_local ThreadRuntime r;
    return r;
}

void gx(int status, const char *where)
{
    if (status == GNNX_OK) return;
    std::string msg;
    switch (status) {
    case GNNX_ERR_SHAPE: msg = ERROR_MM_COMPATIBLE; break;
    case GNNX_ERR_INDEX_RANGE:
        msg = "invalid input, max value in edge_index should be less than the number of nodes from x";  // graph.cpp:90
        break;
    default: msg = std::string(where) + ": " + gnnx_last_error() + " [" + gnnx_status_string(status) + "]";
    }
    throw std::runtime_error(msg);
}

void *current_stream() { return rt().get_stream(); }
void set_current_stream(void *stream)
{
    ThreadRuntime &r = rt();
    if (r.stream_set && stream == r.stream) return;
    if (r.stream_set) {
        gx(gnnx_stream_sync(r.stream), "set_current_stream");  // cached blocks / scratch may still be in use on the old stream
        if (r.stream_owned && r.stream) gx(gnnx_stream_destroy(r.stream), "set_current_stream");
    }
    r.stream = stream;
    r.stream_owned = false;
    r.stream_set = true;
}

void *workspace(size_t bytes)
{
    if (bytes == 0) return nullptr;
    ThreadRuntime &r = rt();
    if (bytes > r.ws_bytes) {
        if (r.ws) {
            gx(gnnx_stream_sync(r.get_stream()), "workspace");
            gnnx_free(r.ws);
        }
        r.ws = nullptr;
        r.ws_bytes = 0;
        gx(gnnx_malloc(&r.ws, bytes), "workspace");
        r.ws_bytes = bytes;
    }
    return r.ws;
}

static size_t bucket(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

void *dev_alloc(size_t bytes)
{
    if (bytes == 0) return nullptr;
    const size_t b = bucket(bytes);
    ThreadRuntime &r = rt();
    auto &fl = r.blocks[b];
    if (!fl.empty()) {
        void *p = fl.back();
        fl.pop_back();
        return p;   // (still registered to this thread)
    }
    void *p = nullptr;
    int st = gnnx_malloc(&p, b);
    if (st != GNNX_OK) {  // out of memory: give the cached blocks back and retry once
        cyg::empty_cache();
        gx(gnnx_malloc(&p, b), "alloc");
    }
    std::lock_guard<std::mutex> lk(g_owner_mu);
    g_owner[p] = &r;
    return p;
}

void dev_free(void *ptr, size_t bytes)
{
    if (!ptr) return;
    ThreadRuntime &r = rt();
    bool mine;
    {
        std::lock_guard<std::mutex> lk(g_owner_mu);
        auto it = g_owner.find(ptr);
        mine = it != g_owner.end() && it->second == &r;
        if (!mine && it != g_owner.end()) g_owner.erase(it);
    }
    if (mine) r.blocks[bucket(bytes)].push_back(ptr);
    else gnnx_free(ptr);   // another thread's block (or an orphan): hipFree synchronises the device before the memory is reused
}

Csr::~Csr()
{
    drop_plans();
    for (void *p : {rowptr, colidx, rowptr_t, colidx_t, coo_src /* coo_dst lives in the same buffer */, norm_per_nz_t, coo_w, vals, vals_t})
        if (p) gnnx_free(p);
}

void Csr::drop_plans()
{
    if (plan) gnnx_spmm_plan_destroy(plan);
    if (plan_t) gnnx_spmm_plan_destroy(plan_t);
    if (plan_pro) gnnx_spmm_plan_destroy(plan_pro);
    plan = plan_t = plan_pro = nullptr;
    plan_feat = 0;
}

void Csr::ensure_plans(int32_t n_feat)
{
    if (plan) return;   // a plan does not depend on the feature width (gnnx_spmm_plan_create ignores max_feat)
    drop_plans();
    ensure_transpose();
    constexpr int32_t kChunk = 1024;  // rows longer than this go to the sequential hub kernel
    gx(gnnx_spmm_plan_create((const int32_t *)rowptr, n, kChunk, n_feat, &plan, current_stream()), "plan");
    gx(gnnx_spmm_plan_create((const int32_t *)rowptr_t, n, kChunk, n_feat, &plan_t, current_stream()), "plan");
    plan_feat = n_feat;
}

void Csr::ensure_prologue_plan(int32_t n_feat)
{
    if (plan_pro) return;
    // With the BatchNorm / ReLU prologue the hub kernel is bound by its vector ALU (five separately rounded operations per gathered
    // element on one wavefront per SIMD) while the streaming kernel carries the prologue at its plain speed: rows of up to 4096
    // non-zeros stay with the streaming kernel (RMAT 10 M / 100 M, F = 256: 15.3 -> 14.65 ms; without a prologue 1024 is the better
    // threshold, 13.6 vs 13.7 ms; scripts/exp_hub_prologue.py).  Same bits: a row's order of additions does not depend on the kernel.
    constexpr int32_t kChunkPrologue = 4096;
    static const int32_t chunk = [] { const char *e = std::getenv("GNNCPP_PROLOGUE_CHUNK"); return e && atoi(e) > 0 ? atoi(e) : kChunkPrologue; }();   // A/B
    gx(gnnx_spmm_plan_create((const int32_t *)rowptr, n, chunk, n_feat, &plan_pro, current_stream()), "plan");
}

static void build_one(const void *src, const void *dst, int64_t n_edges, int32_t n, uint32_t flags, void **rowptr, void **colidx,
                      int64_t *nnz)
{
    if (*rowptr) gnnx_free(*rowptr);
    if (*colidx) gnnx_free(*colidx);
    *rowptr = *colidx = nullptr;
    gx(gnnx_malloc(rowptr, sizeof(int32_t) * ((size_t)n + 1)), "csr");
    gx(gnnx_malloc(colidx, sizeof(int32_t) * (size_t)std::max<int64_t>(n_edges, 1)), "csr");
    size_t wsb = 0;
    gx(gnnx_csr_from_coo_workspace(n_edges, n, &wsb), "csr");
    gx(gnnx_csr_from_coo((const int32_t *)src, (const int32_t *)dst, n_edges, n, flags, (int32_t *)*rowptr, (int32_t *)*colidx, nnz,
                         workspace(wsb), wsb, current_stream()), "csr");
}

// weighted: A[r][c] = w, last duplicate wins; capacity n_edges (+ n for a filled diagonal)
static void build_one_weighted(const void *src, const void *dst, const void *w, int64_t n_edges, int32_t n, uint32_t flags, int diag_mode,
                               float diag_value, void **rowptr, void **colidx, void **vals, int64_t *nnz)
{
    for (void **p : {rowptr, colidx, vals}) {
        if (*p) gnnx_free(*p);
        *p = nullptr;
    }
    const size_t cap = (size_t)std::max<int64_t>(n_edges + n, 1);
    gx(gnnx_malloc(rowptr, sizeof(int32_t) * ((size_t)n + 1)), "csr");
    gx(gnnx_malloc(colidx, sizeof(int32_t) * cap), "csr");
    gx(gnnx_malloc(vals, sizeof(float) * cap), "csr");
    size_t wsb = 0;
    gx(gnnx_csr_from_coo_weighted_workspace(n_edges, n, &wsb), "csr");
    gx(gnnx_csr_from_coo_weighted((const int32_t *)src, (const int32_t *)dst, (const float *)w, n_edges, n,
                                  flags & ~GNNX_CSR_KEEP_SELF_LOOPS, diag_mode, diag_value, (int32_t *)*rowptr, (int32_t *)*colidx,
                                  (float *)*vals, nnz, workspace(wsb), wsb, current_stream()), "csr");
}

void Csr::make_weighted()
{
    if (coo_w) return;
    gx(gnnx_malloc(&coo_w, sizeof(float) * (size_t)std::max<int64_t>(n_edges, 1)), "csr");
    gx(gnnx_fill_f32((float *)coo_w, n_edges, 1.0f, current_stream()), "csr");
}

void Csr::build()
{
    if (weighted())
        build_one_weighted(coo_src, coo_dst, coo_w, n_edges, n, flags, diag_mode, diag_value, &rowptr, &colidx, &vals, &nnz);
    else
        build_one(coo_src, coo_dst, n_edges, n, flags, &rowptr, &colidx, &nnz);
    nnz_t = -1;  // transpose and plans are stale
    drop_plans();
    if (norm_per_nz_t) gnnx_free(norm_per_nz_t);
    norm_per_nz_t = nullptr;
}

void Csr::ensure_transpose()
{
    if (nnz_t >= 0) return;
    if (weighted())
        build_one_weighted(coo_dst, coo_src, coo_w, n_edges, n, flags, diag_mode, diag_value, &rowptr_t, &colidx_t, &vals_t, &nnz_t);
    else
        build_one(coo_dst, coo_src, n_edges, n, flags, &rowptr_t, &colidx_t, &nnz_t);
}

}  // namespace detail

void empty_cache()
{
    detail::rt().drain();
    detail::rt().release_blocks();
}

// One generator for the process, like the reference's global engine (utils.cpp:6) -- but rank threads construct their layers
// concurrently (nn::Linear draws its initial weights), so every draw takes a lock.
static std::mutex g_rng_mu;
static std::mt19937_64 &engine()
{
    static std::mt19937_64 e(0x5eed5eedull);
    return e;
}
void manual_seed(unsigned long long seed)
{
    std::lock_guard<std::mutex> lk(g_rng_mu);
    engine().seed(seed);
}
float generate_random(const float &low, const float &high)
{
    std::lock_guard<std::mutex> lk(g_rng_mu);
    return std::uniform_real_distribution<float>(low, high)(engine());
}
tptr<float> randn(std::vector<size_t> dims, int low, int high, bool requires_grad)
{
    if (low >= high) throw std::runtime_error("pls check input params, low must be lower than high");
    auto t = std::make_shared<tensor<float>>(dims, 0.0f, requires_grad);
    t->uniform((float)low, (float)high);
    return t;
}
}  // namespace cyg

// ---------------------------------------------------------------------------------------------------- nn
namespace nn {

void Module::register_module(std::string n, Module *module)
{
    module->name = n;
    _modules.push_back({n, std::shared_ptr<Module>(module)});
}
void Module::register_parameter(std::string n, tptr<float> p) { _parameters[n] = std::move(p); }
void Module::register_buffer(std::string n, tptr<float> p) { _buffers[n] = std::move(p); }

void Module::zero_grad()
{
    for (auto &p : parameters()) p->zero_grad();
}
void Module::train(const bool &isTrain)
{
    training = isTrain;
    for (auto &[n, m] : _modules) m->train(isTrain);
}
std::vector<std::shared_ptr<Module>> Module::modules(const bool &recurse)
{
    std::vector<std::shared_ptr<Module>> out;
    for (auto &[n, m] : _modules) {
        out.push_back(m);
        if (recurse)
            for (auto &c : m->modules(true)) out.push_back(c);
    }
    return out;
}
std::unordered_map<std::string, std::shared_ptr<Module>> Module::named_modules(const bool &recurse)
{
    std::unordered_map<std::string, std::shared_ptr<Module>> out;
    if (!recurse || _modules.empty()) {
        out[name] = shared_from_this();
        return out;
    }
    for (auto &[cname, m] : _modules)
        for (auto &[k, v] : m->named_modules(true)) out[out.count(k) ? cname + "_" + k : k] = v;
    return out;
}
std::unordered_map<std::string, tptr<float>> Module::named_parameters(const bool &recurse)
{
    auto out = _parameters;
    if (recurse)
        for (auto &[cname, m] : _modules)
            for (auto &[k, v] : m->named_parameters(true)) out[out.count(k) ? cname + "_" + k : k] = v;
    return out;
}
std::unordered_map<std::string, tptr<float>> Module::named_buffers(const bool &recurse)
{
    auto out = _buffers;
    if (recurse)
        for (auto &[cname, m] : _modules)
            for (auto &[k, v] : m->named_buffers(true)) out[out.count(k) ? cname + "_" + k : k] = v;
    return out;
}
std::vector<tptr<float>> Module::parameters(const bool &recurse)
{
    std::vector<tptr<float>> out;
    for (auto &[k, v] : named_parameters(recurse)) out.push_back(v);
    return out;
}
tptr<float> Module::get_parameter(std::string n)
{
    auto all = named_parameters();
    auto it = all.find(n);
    if (it == all.end()) throw std::runtime_error("no parameter named " + n);
    return it->second;
}
tptr<float> Module::get_buffer(std::string n)
{
    auto all = named_buffers();
    auto it = all.find(n);
    if (it == all.end()) throw std::runtime_error("no buffer named " + n);
    return it->second;
}
std::shared_ptr<Module> Module::get_module(std::string n)
{
    for (auto &[k, m] : _modules)
        if (k == n) return m;
    for (auto &[k, m] : _modules) {
        try {
            return m->get_module(n);
        } catch (const std::runtime_error &) {
        }
    }
    throw std::runtime_error("no module named " + n);
}

Linear::Linear(const size_t &in_features, const size_t &out_features, const bool &bias, const std::string &n)
    : Module(n), _bias(bias), _in_features(in_features), _out_features(out_features)
{
    register_parameter("weight", std::make_shared<tensor<float>>(std::vector<size_t>{out_features, in_features}, 1.0f, true));
    if (_bias) register_parameter("bias", std::make_shared<tensor<float>>(std::vector<size_t>{out_features}, 1.0f, true));
    reset_parameters();
}
void Linear::reset_parameters()
{
    const float bound = 1.0f / std::sqrt((float)_in_features);
    _parameters["weight"]->uniform(-bound, bound);
    if (_bias) _parameters["bias"]->uniform(-bound, bound);
}
// the dense feature transform of the hot path: x.mm(W.t(-1,-2)) (+ b) -- reference nn.cpp:205-211; t() is a view,
// so this is ONE MFMA GEMM with transB = 1
tptr<float> Linear::forward(const tptr<float> &input_tensor)
{
    auto output = input_tensor->mm(get_parameter("weight")->t(-1, -2));
    if (_bias) return output + get_parameter("bias");
    return output;
}

// ---- BatchNorm / ReLU as autograd ops over the fused device kernels
namespace {
// GNNCPP_REFERENCE_QUIRKS=1: BatchNorm's backward reproduces what the reference's traversal delivers (only the first arrival
// at BatchNorm's input reaches the transform, operation.h:80-88) instead of the mathematical gradient -- for comparing
// through-layer gradients with the reference's own (tests/golden ref_full_*).  Off by default.
using bn_bwd_fn = decltype(&gnnx_bn_relu_bwd_f32);
bn_bwd_fn bn_backward()
{
    static const bool quirks = std::getenv("GNNCPP_REFERENCE_QUIRKS") != nullptr;  // read once, not on every backward
    return quirks ? &gnnx_bn_relu_bwd_quirk_f32 : &gnnx_bn_relu_bwd_f32;
}
}  // namespace
decltype(&gnnx_bn_relu_bwd_f32) bn_backward_fn() { return bn_backward(); }
namespace {
class BatchNormOp : public cyg::Operation<tensor<float>> {
public:
    float eps = 1e-5f;
    bool batch_stats = true;
    tptr<float> mean, var;
    BatchNormOp() { name = "BatchNorm"; }
    tptr<float> forward(const tptr<float> &x, const tptr<float> &gamma, const tptr<float> &beta)
    {
        const auto shp = x->shape();
        if (shp.size() != 2) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        const int64_t n = (int64_t)shp[0];
        const int32_t f = (int32_t)shp[1];
        void *st = detail::current_stream();
        size_t wsb = 0;
        detail::gx(gnnx_bn_workspace(n, f, &wsb), "BatchNorm");
        if (batch_stats) {
            mean = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            var = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            detail::gx(gnnx_bn_stats_f32(x->device_data(), f, n, f, mean->device_out(), var->device_out(), detail::workspace(wsb), wsb, st),
                       "BatchNorm");
        }
        const bool req = x->requires_grad() || gamma->requires_grad() || (beta && beta->requires_grad());
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, req);
        detail::gx(gnnx_bn_relu_fwd_f32(x->device_data(), f, n, f, mean->device_data(), var->device_data(), eps, gamma->device_data(),
                                        beta ? beta->device_data() : nullptr, 0, out->device_out(), f, st), "BatchNorm");
        if (req) context->save_for_backward({x, gamma, beta ? beta : gamma});
        has_beta = (bool)beta;
        return out;
    }
    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        if (!batch_stats) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);  // eval-mode backward is not on any path
        auto var_ = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(var_, 3);
        auto x = var_[0], gamma = var_[1], beta = var_[2];
        const auto shp = x->shape();
        const int64_t n = (int64_t)shp[0];
        const int32_t f = (int32_t)shp[1];
        size_t wsb = 0;
        detail::gx(gnnx_bn_workspace(n, f, &wsb), "BatchNorm");
        auto dx = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        auto dgamma = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        auto dbeta = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        detail::gx(bn_backward()(x->device_data(), f, nullptr, 0, g->device_data(), f, n, f, mean->device_data(), var->device_data(),
                                        eps, gamma->device_data(), nullptr, 0, dx->device_out(), f, dgamma->device_out(), dbeta->device_out(),
                                        detail::workspace(wsb), wsb, detail::current_stream()), "BatchNorm");
        if (x->requires_grad()) x->backward(dx);
        if (gamma->requires_grad()) gamma->backward(dgamma);
        if (has_beta && beta->requires_grad()) beta->backward(dbeta);
    }
    bool has_beta = true;
};

class ReluOp : public cyg::Operation<tensor<float>> {
public:
    ReluOp() { name = "ReLU"; }
    tptr<float> forward(const tptr<float> &x)
    {
        const auto shp = x->shape();
        if (shp.size() != 2) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, x->requires_grad());
        detail::gx(gnnx_bn_relu_fwd_f32(x->device_data(), (int64_t)shp[1], (int64_t)shp[0], (int32_t)shp[1], nullptr, nullptr, 0.f, nullptr,
                                        nullptr, 1, out->device_out(), (int64_t)shp[1], detail::current_stream()), "ReLU");
        if (out->requires_grad()) {
            context->save_for_backward({x});
            y = out.get();  // the output owns this op (grad_fn), so a raw back-pointer cannot dangle
        }
        return out;
    }
    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        auto var_ = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(var_, 1);
        auto x = var_[0];
        const auto shp = x->shape();
        auto dx = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        detail::gx(gnnx_bn_relu_bwd_f32(x->device_data(), (int64_t)shp[1], y->device_data(), (int64_t)shp[1], g->device_data(),
                                        (int64_t)shp[1], (int64_t)shp[0], (int32_t)shp[1], nullptr, nullptr, 0.f, nullptr, nullptr, 1,
                                        dx->device_out(), (int64_t)shp[1], nullptr, nullptr, nullptr, 0, detail::current_stream()), "ReLU");
        if (x->requires_grad()) x->backward(dx);
    }
    tensor<float> *y = nullptr;
};
}  // namespace

BatchNorm::BatchNorm(const size_t &num_features, const float &eps, const float &momentum, const bool &affine,
                     const bool &track_running_stats, const std::string &n)
    : Module(n), _num_features(num_features), _eps(eps), _momentum(momentum), _affine(affine), _tracking_running_stats(track_running_stats)
{
    std::vector<size_t> dims = {1, num_features};
    register_parameter("gammas", std::make_shared<tensor<float>>(dims, 1.0f, true));
    if (affine) register_parameter("betas", std::make_shared<tensor<float>>(dims, 0.0f, true));
    if (_tracking_running_stats) {
        register_buffer("running_mean", std::make_shared<tensor<float>>(dims, 0.0f, false));
        register_buffer("running_var", std::make_shared<tensor<float>>(dims, 0.0f, false));
    }
    training = true;
}

tptr<float> BatchNorm::forward(const tptr<float> &x)
{
    auto op = std::make_unique<BatchNormOp>();
    op->eps = _eps;
    op->batch_stats = training || !_tracking_running_stats;
    if (!op->batch_stats) {
        op->mean = get_buffer("running_mean");
        op->var = get_buffer("running_var");
    }
    auto out = op->forward(x, _parameters["gammas"], _affine ? _parameters["betas"] : nullptr);
    if (out->requires_grad()) out->grad_fn = std::move(op);
    return out;
}

tptr<float> ReLU::forward(const tptr<float> &input_tensor)
{
    auto op = std::make_unique<ReluOp>();
    auto out = op->forward(input_tensor);
    if (out->requires_grad()) out->grad_fn = std::move(op);
    return out;
}

// ---- loss + optimiser
namespace {
class CrossEntropyOp : public cyg::Operation<tensor<float>> {
public:
    tptr<int> target;
    int64_t n_total = 0;   // > 0: the rows are one shard of a batch of n_total (sharded training: the caller sums the ranks' losses)
    CrossEntropyOp() { name = "CrossEntropy"; }
    tptr<float> forward(const tptr<float> &logits, const tptr<int> &tgt)
    {
        if (logits->rank() != 2 || tgt->rank() != 1)
            throw std::runtime_error("invalid input, logits must be of rank 2 and targets must be 1D tensor");
        const auto shp = logits->shape();
        if (tgt->numel() != shp[0]) throw std::runtime_error(ERROR_SIZE_MISMATCH);
        target = tgt;
        size_t wsb = 0;
        detail::gx(gnnx_softmax_ce_workspace((int64_t)shp[0], &wsb), "cross_entropy");
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1}, logits->requires_grad());
        detail::gx(gnnx_softmax_ce_partial_f32(logits->device_data(), (int64_t)shp[1], tgt->device_data(), (int64_t)shp[0], (int32_t)shp[1],
                                               n_total > 0 ? n_total : (int64_t)shp[0], out->device_out(), nullptr, 0, nullptr,
                                               detail::workspace(wsb), wsb, detail::current_stream()),
                   "cross_entropy");
        if (out->requires_grad()) context->save_for_backward({logits});
        return out;
    }
    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        auto var = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(var, 1);
        auto logits = var[0];
        if (!logits->requires_grad()) return;
        const auto shp = logits->shape();
        size_t wsb = 0;
        detail::gx(gnnx_softmax_ce_workspace((int64_t)shp[0], &wsb), "cross_entropy");
        auto d = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        void *st = detail::current_stream();
        detail::gx(gnnx_softmax_ce_partial_f32(logits->device_data(), (int64_t)shp[1], target->device_data(), (int64_t)shp[0], (int32_t)shp[1],
                                               n_total > 0 ? n_total : (int64_t)shp[0], nullptr, d->device_out(), (int64_t)shp[1], nullptr,
                                               detail::workspace(wsb), wsb, st), "cross_entropy");
        const float up = g->item();  // upstream scalar (1 for loss->backward())
        if (up != 1.0f) {
            auto scaled = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
            detail::gx(gnnx_fill_f32(scaled->device_out(), (int64_t)scaled->numel(), 0.0f, st), "cross_entropy");
            detail::gx(gnnx_axpy_f32((int64_t)scaled->numel(), up, d->device_data(), scaled->device_inplace(), st), "cross_entropy");
            d = scaled;
        }
        logits->backward(d);
    }
};
}  // namespace

tptr<float> cross_entropy_loss(const tptr<float> logits, const tptr<int> target, size_t n_total)
{
    auto op = std::make_unique<CrossEntropyOp>();
    op->n_total = (int64_t)n_total;
    if (n_total != 0 && n_total < logits->shape()[0]) throw std::runtime_error(ERROR_SIZE_MISMATCH);
    auto out = op->forward(logits, target);
    if (out->requires_grad()) out->grad_fn = std::move(op);
    return out;
}

tptr<float> cross_entropy_loss(const tptr<float> logits, const tptr<int> target)
{
    auto op = std::make_unique<CrossEntropyOp>();
    auto out = op->forward(logits, target);
    if (out->requires_grad()) out->grad_fn = std::move(op);
    return out;
}

void Optimizer::zero_grad()
{
    for (auto &p : _parameters) p->zero_grad();
}

void SGD::step()
{
    if (_momentum != 0) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);  // plain SGD only (momentum: next)
    for (auto &p : _parameters) {
        float *g = p->device_grad();
        if (!g) continue;
        detail::gx(gnnx_sgd_step_f32(p->device_inplace(), g, (int64_t)p->numel(), _lr, _weight_decay, detail::current_stream()), "SGD");
    }
}

}  // namespace nn

// ---------------------------------------------------------------------------------------------------- graph
namespace graph {

tptr<int> vec_to_edge_list(std::vector<int> source, std::vector<int> destination)
{
    if (source.size() != destination.size()) throw std::runtime_error("input vectors must be of same length");
    const size_t e = source.size();
    auto *v = new std::valarray<int>(2 * e);
    for (size_t i = 0; i < e; i++) {
        (*v)[i] = source[i];
        (*v)[e + i] = destination[i];
    }
    if (e == 0) throw std::runtime_error(ERROR_INVALID_DIMS);
    return std::make_shared<tensor<int>>(std::vector<size_t>{2, e}, v, false);
}

// COO -> CSR on the device; the returned tensor says shape [N,N] but holds no dense data
tptr<float> edge_to_adj_mat(const tensor<int> &edge_index, tensor<float> *edge_attr, size_t n_nodes)
{
    auto &ei = const_cast<tensor<int> &>(edge_index);
    if (ei.rank() != 2 || ei.shape()[0] != 2) throw std::runtime_error("invalid input for x, must be of 2D");
    const size_t e = ei.shape()[1];
    if (edge_attr != nullptr) {
        if (e != edge_attr->shape()[0])
            throw std::runtime_error("invalid inputs, number of edges in edge_index must be equal to size of edge_attr");
    }
    size_t n = n_nodes;
    if (n == 0) n = (size_t)std::get<0>(ei.max())->item() + 1;
    auto c = std::make_shared<detail::Csr>();
    c->n = (int32_t)n;
    c->n_edges = (int64_t)e;
    const int32_t *d = ei.device_data();  // rows 0 (sources) and 1 (destinations), contiguous
    // one [2, E] buffer like the edge_index tensor itself (coo_dst points into it): the graph cache compares both rows in one pass
    detail::gx(gnnx_malloc(&c->coo_src, 2 * e * sizeof(int32_t)), "adj");
    c->coo_dst = static_cast<int32_t *>(c->coo_src) + e;
    detail::gx(gnnx_memcpy_d2d(c->coo_src, d, 2 * e * sizeof(int32_t), detail::current_stream()), "adj");
    if (edge_attr != nullptr) {  // A[r][c] = w (graph.cpp:38-40)
        detail::gx(gnnx_malloc(&c->coo_w, e * sizeof(float)), "adj");
        detail::gx(gnnx_memcpy_d2d(c->coo_w, edge_attr->device_data(), e * sizeof(float), detail::current_stream()), "adj");
    }
    c->build();
    return std::make_shared<tensor<float>>(c, false);
}

// CSR -> COO, row-major order (what the reference's scan of the dense matrix yields, graph.cpp:46-67)
std::tuple<tptr<int>, tptr<float>> adj_to_edge_list(tensor<float> &adj_mat)
{
    if (!adj_mat.is_csr()) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
    auto c = adj_mat.csr();
    const size_t nnz = (size_t)c->nnz;
    std::vector<int32_t> rp((size_t)c->n + 1), ci(std::max<size_t>(nnz, 1));
    std::vector<float> wv(std::max<size_t>(nnz, 1), 1.0f);
    detail::gx(gnnx_memcpy_d2h(rp.data(), c->rowptr, rp.size() * sizeof(int32_t), detail::current_stream()), "edge list");
    if (nnz) detail::gx(gnnx_memcpy_d2h(ci.data(), c->colidx, nnz * sizeof(int32_t), detail::current_stream()), "edge list");
    if (nnz && c->vals) detail::gx(gnnx_memcpy_d2h(wv.data(), c->vals, nnz * sizeof(float), detail::current_stream()), "edge list");
    // the reference keeps the entries with `int(value) != 0` (graph.cpp:54): weights of magnitude below 1 vanish here
    size_t kept = 0;
    for (size_t p = 0; p < nnz; p++) kept += (int)wv[p] != 0;
    if (kept == 0) throw std::runtime_error(ERROR_INVALID_DIMS);  // the reference cannot represent an empty [2,0] tensor either
    auto *v = new std::valarray<int>(2 * kept);
    auto *a = new std::valarray<float>(kept);
    size_t k = 0;
    for (int32_t r = 0; r < c->n; r++)
        for (int32_t p = rp[r]; p < rp[r + 1]; p++) {
            if ((int)wv[p] == 0) continue;
            (*v)[k] = r;
            (*v)[kept + k] = ci[p];
            (*a)[k] = wv[p];
            k++;
        }
    auto edge_index = std::make_shared<tensor<int>>(std::vector<size_t>{2, kept}, v, false);
    auto edge_attr = std::make_shared<tensor<float>>(std::vector<size_t>{kept}, a, false);
    return {edge_index, edge_attr};
}

std::tuple<tptr<int>, tptr<float>> add_self_loops(const tensor<int> &edge_index, tensor<float> *edge_attr, const float &fillValue,
                                                  const int &num_nodes)
{
    auto mat = edge_to_adj_mat(edge_index, edge_attr, (size_t)num_nodes);
    mat->fill_diagonal_(fillValue);
    return adj_to_edge_list(*mat);
}

Data::Data(const tptr<float> &x, tensor<int> *edge_index, tptr<float> edge_attr, tensor<float> *y)
    : _num_nodes(x->shape()[0]), _num_node_features(x->rank() > 1 ? x->shape()[1] : 0), _edge_index(edge_index), _y(y), _x(x),
      _edge_attr(edge_attr)
{
    if (x->rank() != 2) throw std::runtime_error("invalid input for x, must be 2D");
    if (edge_index != nullptr) {
        if (edge_index->rank() != 2 || edge_index->shape()[0] != 2) throw std::runtime_error("invalid input for x, must be of 2D");
        _num_edges = edge_index->shape()[1];
        if ((int)x->shape()[0] <= std::get<0>(edge_index->max())->item())
            throw std::runtime_error("invalid input, max value in edge_index should be less than the number of nodes from x");
        if (edge_attr != nullptr) {
            if (edge_attr->rank() != 2) throw std::runtime_error("pls check input tensors, must of 2D for x, edge_index and edge_attr");
            if (edge_index->shape()[1] != edge_attr->shape()[0])
                throw std::runtime_error("invalid edge_index and/or edge_attr input, edge_index should of [2, num_edges] and edge_attr "
                                         "should be of [num_edges, num_edge_feature]");
            _num_edge_features = edge_attr->shape()[1];
        }
    }
}
tensor<int> *Data::edge_index()
{
    if (_edge_index == nullptr) throw std::runtime_error("pls provide adj matr or edge");
    return _edge_index;
}
void Data::set_edge_index(tensor<int> *edge_index, tptr<float> edge_attr)
{
    _edge_index = edge_index;  // not owned (the reference deletes the previous one, graph.cpp:114; callers own theirs here)
    _edge_attr = std::move(edge_attr);
}
tptr<float> Data::to_adj()
{
    if (_edge_index == nullptr) throw std::runtime_error("pls provide adj matr or edge");
    return edge_to_adj_mat(*_edge_index, _edge_attr.get(), _num_nodes);
}

tptr<float> MessagePassing::propagate(const tensor<int> &edge_index, const tptr<float> &x, const tptr<float> *)
{
    return aggregate_and_update(x, edge_index, nullptr);
}

namespace {
// pooled device scratch, returned on scope exit (the next kernel that takes the block is ordered behind this thread's stream)
struct DevScratch {
    void *p = nullptr;
    size_t bytes = 0;
    void alloc(size_t b)
    {
        bytes = std::max<size_t>(b, 4);
        p = cyg::detail::dev_alloc(bytes);
    }
    ~DevScratch()
    {
        if (p) cyg::detail::dev_free(p, bytes);
    }
};

// The row pitch (floats) the aggregation wants the matrix it GATHERS rows from to sit on: padded (gnnx_gather_row_stride) when the
// rows it gathers most -- the hub rows of the OTHER direction's CSR (`gathered_by`: plan_t for the forward aggregation, plan for
// the backward one) -- sit on vertex ids with few one-bits (gnnx_spmm_plan_hub_ids_structured); a data set whose ids are already
// spread keeps its rows on their own width and pays nothing.
int64_t gather_pitch(int64_t n_rows, int32_t n_feat, const gnnx_spmm_plan *gathered_by)
{
    int64_t ld = n_feat;
    int structured = 0;
    static const bool off = std::getenv("GNNCPP_NO_GATHER_PITCH") != nullptr;   // A/B switch of the tests (same bits either way), read once
    if (off) return ld;
    if (gathered_by) cyg::detail::gx(gnnx_spmm_plan_hub_ids_structured(gathered_by, &structured), "aggregate");
    if (structured) cyg::detail::gx(gnnx_gather_row_stride(n_rows, n_feat, &ld), "aggregate");
    return ld;
}

// The upstream gradient as the backward aggregation gathers it, and the bias gradient.  g is the caller's tensor (rows on their
// own width): when the gather pitch of its shape is wider, the rows are copied onto it by the pass that sums the columns anyway
// (gnnx_colsum_copy_f32; a plain strided copy when no bias gradient is wanted) into `pad`; else g is gathered as it lies.
// Returns the matrix to gather from and its pitch; *dbias (when bias wants a gradient) = column sums of g.
const float *gathered_gradient(const tptr<float> &g, int64_t n, int32_t f, const gnnx_spmm_plan *gathered_by, bool want_dbias,
                               const tptr<float> &bias, DevScratch &pad, int64_t &ldg, tptr<float> *dbias)
{
    void *st = cyg::detail::current_stream();
    const int64_t ld = gather_pitch(n, f, gathered_by);
    if (ld == f) {
        if (want_dbias) *dbias = cyg::functional::sum(g, 0, bias->rank() == 2);  // dbias = column sums of G, straight from G (no N x F clone)
        return g->device_pitched(ldg);
    }
    int64_t lds = f;
    const float *gs = g->device_pitched(lds);
    if (lds != f) {   // already on a pitch (a product's output asked for it)
        if (want_dbias) *dbias = cyg::functional::sum(g, 0, bias->rank() == 2);
        ldg = lds;
        return g->device_pitched(ldg);
    }
    pad.alloc(sizeof(float) * (size_t)n * (size_t)ld);
    if (want_dbias) {
        *dbias = std::make_shared<tensor<float>>(tensor<float>::device_tag{},
                                                 bias->rank() == 2 ? std::vector<size_t>{1, (size_t)f} : std::vector<size_t>{(size_t)f}, false);
        size_t wsb = 0;
        cyg::detail::gx(gnnx_colsum_workspace(n, f, &wsb), "sum");
        cyg::detail::gx(gnnx_colsum_copy_f32(gs, f, n, f, 0.0f, (*dbias)->device_out(), (float *)pad.p, ld, cyg::detail::workspace(wsb), wsb, st),
                        "sum");
    } else {
        cyg::detail::gx(gnnx_memcpy2d_d2d(pad.p, sizeof(float) * (size_t)ld, gs, sizeof(float) * (size_t)f, sizeof(float) * (size_t)f, (size_t)n, st),
                        "aggregate");
    }
    ldg = ld;
    return (const float *)pad.p;
}

// out = norm (.) (A . x) (+ bias) as ONE SpMM with fused epilogue; backward dX = A^T . (norm (.) G), dbias = colsum(G).
// Same arithmetic, in the same order, as the MatMul -> Mul -> Add chain it replaces (reference graph.cpp:208-209,188).
class AggregateOp : public cyg::Operation<tensor<float>> {
public:
    std::shared_ptr<cyg::detail::Csr> csr;
    tptr<float> norm;
    AggregateOp() { name = "GCNAggregate"; }
    tptr<float> forward(const tptr<float> &adj, const tptr<float> &x, const tptr<float> &norm_, const tptr<float> &bias)
    {
        csr = adj->csr();
        norm = norm_;
        const auto shp = x->shape();
        const int32_t f = (int32_t)shp[1];
        csr->ensure_plans(f);
        const bool req = x->requires_grad() || (bias && bias->requires_grad());
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, req);
        int64_t ldx = f;
        const float *xp = x->device_pitched(ldx);   // (the transform's output sits on the gather pitch when GCNConv asked for it)
        cyg::detail::gx(gnnx_spmm_csr_f32(csr->n, csr->n, f, (const int32_t *)csr->rowptr, (const int32_t *)csr->colidx, nullptr, nullptr,
                                          norm->device_data(), bias ? bias->device_data() : nullptr, xp, ldx, 0.0f,
                                          out->device_out(), f, csr->plan, cyg::detail::current_stream()), "aggregate");
        if (req) context->save_for_backward({x, bias ? bias : x});
        has_bias = (bool)bias;
        return out;
    }
    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        auto var = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(var, 2);
        auto x = var[0], bias = var[1];
        const auto shp = x->shape();
        const int32_t f = (int32_t)shp[1];
        void *st = cyg::detail::current_stream();
        const bool want_db = has_bias && bias->requires_grad();
        if (!x->requires_grad()) {
            if (want_db) bias->backward(cyg::functional::sum(g, 0, bias->rank() == 2));  // dbias = column sums of G, straight from G
            return;
        }
        DevScratch gpad;
        int64_t ldg = f;
        tptr<float> db;
        csr->ensure_plans(f);
        const float *gp = gathered_gradient(g, csr->n, f, csr->plan, want_db, bias, gpad, ldg, &db);   // the backward gathers what A's hub rows read
        if (want_db) bias->backward(db);
        if (!csr->norm_per_nz_t) {  // norm[colidx_t[p]] once per graph
            cyg::detail::gx(gnnx_malloc(&csr->norm_per_nz_t, sizeof(float) * (size_t)std::max<int64_t>(csr->nnz_t, 1)), "aggregate");
            cyg::detail::gx(gnnx_gather_rows_f32(norm->device_data(), 1, (const int32_t *)csr->colidx_t, csr->nnz_t, 1,
                                                 (float *)csr->norm_per_nz_t, 1, st), "aggregate");
        }
        auto dx = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        cyg::detail::gx(gnnx_spmm_csr_f32(csr->n, csr->n, f, (const int32_t *)csr->rowptr_t, (const int32_t *)csr->colidx_t,
                                          (const float *)csr->norm_per_nz_t, nullptr, nullptr, nullptr, gp, ldg, 0.0f,
                                          dx->device_out(), f, csr->plan_t, st), "aggregate");
        x->backward(dx);
    }
    bool has_bias = false;
};

// transform output H -> [BatchNorm (batch statistics) -> ReLU] -> aggregation + bias as ONE op: the two modules of
// graph.cpp:174-175 ride in the SpMM's gather (gnnx_spmm_csr_fused_f32), so neither BN(H) nor relu(BN(H)) is written to HBM;
// backward recomputes the ReLU mask from H.  Same arithmetic, same bits as running the three ops one after the other.
class BnReluAggregateOp : public cyg::Operation<tensor<float>> {
public:
    std::shared_ptr<cyg::detail::Csr> csr;
    tptr<float> norm, mean, var;
    float eps = 1e-5f;
    bool has_beta = false;
    bool have_stats = false;   // mean / var were delivered by the transform's epilogue (GCNConv::fuse_bn_stats)
    BnReluAggregateOp() { name = "GCNBnReluAggregate"; }
    tptr<float> forward(const tptr<float> &adj, const tptr<float> &h, const tptr<float> &norm_, const tptr<float> &bias,
                        const tptr<float> &gamma, const tptr<float> &beta)
    {
        csr = adj->csr();
        norm = norm_;
        const auto shp = h->shape();
        const int64_t n = (int64_t)shp[0];
        const int32_t f = (int32_t)shp[1];
        void *st = cyg::detail::current_stream();
        csr->ensure_plans(f);
        csr->ensure_prologue_plan(f);
        if (!have_stats) {
            mean = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            var = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, shp[1]}, false);
            size_t wsb = 0;
            cyg::detail::gx(gnnx_bn_workspace(n, f, &wsb), "BatchNorm");
            int64_t ldh0 = f;
            const float *hp0 = h->device_pitched(ldh0);
            cyg::detail::gx(gnnx_bn_stats_f32(hp0, ldh0, n, f, mean->device_out(), var->device_out(), cyg::detail::workspace(wsb),
                                              wsb, st), "BatchNorm");
        }
        int64_t ldh = f;
        const float *hp = h->device_pitched(ldh);   // (on the gather pitch when GCNConv asked the transform for it)
        const bool req = h->requires_grad() || bias->requires_grad() || gamma->requires_grad() || (beta && beta->requires_grad());
        auto out = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, req);
        gnnx_spmm_fusion fu{};
        fu.bn_mean = mean->device_data();
        fu.bn_var = var->device_data();
        fu.bn_gamma = gamma->device_data();
        fu.bn_beta = beta ? beta->device_data() : nullptr;
        fu.bn_eps = eps;
        fu.relu_in = 1;
        cyg::detail::gx(gnnx_spmm_csr_fused_f32(csr->n, csr->n, f, (const int32_t *)csr->rowptr, (const int32_t *)csr->colidx, nullptr,
                                                nullptr, norm->device_data(), bias->device_data(), hp, ldh, 0.0f,
                                                out->device_out(), f, &fu, csr->plan_pro, st), "aggregate");
        has_beta = (bool)beta;
        if (req) context->save_for_backward({h, bias, gamma, beta ? beta : gamma});
        return out;
    }
    void _backward(std::shared_ptr<tensor<float>> g) override
    {
        auto v = context->get_variables();
        CHECK_BACKWARD<tensor<float>>(v, 4);
        auto h = v[0], bias = v[1], gamma = v[2], beta = v[3];
        const auto shp = h->shape();
        const int64_t n = (int64_t)shp[0];
        const int32_t f = (int32_t)shp[1];
        void *st = cyg::detail::current_stream();
        if (!(h->requires_grad() || gamma->requires_grad() || (has_beta && beta->requires_grad()))) {
            if (bias->requires_grad()) bias->backward(cyg::functional::sum(g, 0, bias->rank() == 2));
            return;
        }
        DevScratch gpad;
        int64_t ldg = f, ldh = f;
        tptr<float> db;
        csr->ensure_plans(f);
        const float *gp = gathered_gradient(g, n, f, csr->plan, bias->requires_grad(), bias, gpad, ldg, &db);
        if (bias->requires_grad()) bias->backward(db);
        const float *hp = h->device_pitched(ldh);
        if (!csr->norm_per_nz_t) {
            cyg::detail::gx(gnnx_malloc(&csr->norm_per_nz_t, sizeof(float) * (size_t)std::max<int64_t>(csr->nnz_t, 1)), "aggregate");
            cyg::detail::gx(gnnx_gather_rows_f32(norm->device_data(), 1, (const int32_t *)csr->colidx_t, csr->nnz_t, 1,
                                                 (float *)csr->norm_per_nz_t, 1, st), "aggregate");
        }
        auto dy = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);  // d relu(BN(H))
        size_t wsb = 0;
        cyg::detail::gx(gnnx_bn_workspace(n, f, &wsb), "BatchNorm");
        auto dh = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, shp, false);
        auto dgamma = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        auto dbeta = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, gamma->shape(), false);
        const float *beta_d = has_beta ? beta->device_data() : nullptr;
        // BatchNorm's backward needs dbeta = sum g and dgamma = sum g xhat before it can write dX: the aggregation's own store
        // epilogue accumulates them (gnnx_spmm_csr_bn_sums_f32), so dY and H are not re-read by a sums pass of their own; widths
        // the fused form does not cover (and the reference-quirk mode) take the two separate kernels
        size_t sums_ws = 0;
        const bool fused_sums = nn::bn_backward_fn() == &gnnx_bn_relu_bwd_f32 &&
                                gnnx_spmm_csr_bn_sums_workspace(csr->n, f, csr->plan_t, &sums_ws) == GNNX_OK && f % 4 == 0 && f > 64;
        if (fused_sums) {
            cyg::detail::gx(gnnx_spmm_csr_bn_sums_f32(csr->n, csr->n, f, (const int32_t *)csr->rowptr_t, (const int32_t *)csr->colidx_t,
                                                      (const float *)csr->norm_per_nz_t, gp, ldg, dy->device_out(), f,
                                                      hp, ldh, mean->device_data(), var->device_data(), eps,
                                                      gamma->device_data(), beta_d, 1, dgamma->device_out(), dbeta->device_out(),
                                                      cyg::detail::workspace(sums_ws), sums_ws, csr->plan_t, st), "aggregate");
            cyg::detail::gx(gnnx_bn_relu_bwd_apply_f32(hp, ldh, nullptr, 0, dy->device_data(), f, n, f, mean->device_data(),
                                                       var->device_data(), eps, gamma->device_data(), beta_d, 1, dgamma->device_data(),
                                                       dbeta->device_data(), n, dh->device_out(), f, cyg::detail::workspace(wsb), wsb, st),
                            "BatchNorm");
        } else {
            cyg::detail::gx(gnnx_spmm_csr_f32(csr->n, csr->n, f, (const int32_t *)csr->rowptr_t, (const int32_t *)csr->colidx_t,
                                              (const float *)csr->norm_per_nz_t, nullptr, nullptr, nullptr, gp, ldg, 0.0f,
                                              dy->device_out(), f, csr->plan_t, st), "aggregate");
            cyg::detail::gx(nn::bn_backward_fn()(hp, ldh, nullptr, 0, dy->device_data(), f, n, f, mean->device_data(),
                                                 var->device_data(), eps, gamma->device_data(), beta_d, 1, dh->device_out(), f,
                                                 dgamma->device_out(), dbeta->device_out(), cyg::detail::workspace(wsb), wsb, st),
                            "BatchNorm");
        }
        if (h->requires_grad()) h->backward(dh);
        if (gamma->requires_grad()) gamma->backward(dgamma);
        if (has_beta && beta->requires_grad()) beta->backward(dbeta);
    }
};
}  // namespace

GCNConv::GCNConv(size_t in_channels, size_t out_channels, float dropout)
    : MessagePassing(), _in_channels(in_channels), _out_channels(out_channels), _dropout(dropout)
{
    register_module("lin", new nn::Linear(in_channels, out_channels, false));
    register_module("bnorm", new nn::BatchNorm(out_channels));
    register_module("drop", new nn::Dropout(dropout));  // registered, never applied: as in the reference
    register_module("relu", new nn::ReLU());
    register_parameter("bias", std::make_shared<tensor<float>>(std::vector<size_t>{out_channels}, 0.0f, true));
    if (std::getenv("GNNCPP_UNFUSED")) fused = false;  // run the op-by-op path (tests compare both)
}

// Same sequence of API calls as the reference layer (graph.cpp:170-191); each lands on the device:
//   add_self_loops          -> CSR build (dedupe, diagonal stripped), never a dense N x N
//   lin                     -> MFMA GEMM, W^T as a view
//   sum(-1,true) + 1, pow   -> degrees from rowptr, s = the host libm's powf(deg, -0.5f) from the process-wide table (gnnx_degree_norm_f32)
//   adj.mm(deg), norm *= deg-> CSR SpMV in the reference's summation order + elementwise
//   propagate, + bias       -> CSR SpMM, row scale, bias broadcast
tptr<float> GCNConv::forward(Data &&input)
{
    if (_part) return forward_sharded(input.x());  // this rank's rows of a partitioned graph (dist.h)
    if (fused) {
        // static-graph cache: adjacency (dedupe + diagonal strip) and norm are built once per edge list.  The key is the CONTENT
        // identity of the edge_index tensor's device copy -- (storage id, content version), tensor.h detail::Store: the version moves
        // when the list is uploaded again after the host side was handed out for writing (data()) or set_data() / a kernel rewrote
        // it, the id is unique per storage (a new tensor at a freed one's address is another id) -- so a forward costs two integer
        // compares: no pass over the list, no host synchronisation.  (The reference rebuilds on every
        // forward, graph.cpp:172-185.)
        tensor<int> *ei = input.edge_index();
        (void)ei->device_data();   // brings the device copy up to date (uploads, and bumps the version, if the host side was written)
        const bool hit = _cache_adj && _cache_ei_id == ei->storage_id() && _cache_ei_version == ei->storage_version() &&
                         _cache_nodes == input.num_nodes();
        if (!hit) {
            graph_cache_builds++;
            auto adj = edge_to_adj_mat(*ei, nullptr, input.num_nodes());
            adj->fill_diagonal_(0);  // == add_self_loops(..., fillValue 0): self loops removed (graph.cpp:172)
            auto deg = adj->sum(-1, true) + 1;
            deg = deg->pow(-0.5);
            auto norm = adj->mm(deg);
            norm *= deg;
            _cache_adj = adj;
            _cache_norm = norm;
            _cache_ei_id = ei->storage_id();
            _cache_ei_version = ei->storage_version();
            _cache_nodes = input.num_nodes();
        }
        tptr<float> out, st_mean, st_var;
        bool have_stats = false;
        // the transform's output is gathered row by row by the aggregation below: it is asked to write its rows on the gather pitch
        // (a power-of-two pitch piles the hub rows of a power-law graph onto a few memory channels: gnnx_gather_row_stride)
        const size_t n_rows = input.x()->shape()[0];
        _cache_adj->csr()->ensure_plans((int32_t)_out_channels);
        gathered_row_pitch = (size_t)gather_pitch((int64_t)n_rows, (int32_t)_out_channels, _cache_adj->csr()->plan_t);
        cyg::detail::PitchRequest on_pitch(n_rows, _out_channels, gathered_row_pitch);
        if (fuse_bn_stats && !hot_path_only) {   // opt-in: the statistics ride in the transform's epilogue
            st_mean = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, _out_channels}, false);
            st_var = std::make_shared<tensor<float>>(tensor<float>::device_tag{}, std::vector<size_t>{1, _out_channels}, false);
            cyg::detail::BnStatsRequest rq(input.x()->shape()[0], _out_channels, st_mean->device_out(), st_var->device_out());
            out = (*get_module("lin"))(input.x());
            have_stats = rq.done;
        } else {
            out = (*get_module("lin"))(input.x());
        }
        if (!hot_path_only) {
            auto *bn = dynamic_cast<nn::BatchNorm *>(get_module("bnorm").get());
            static const bool no_prologue_fusion = std::getenv("GNNCPP_NO_PROLOGUE_FUSION") != nullptr;  // A/B switch of the tests, read once
            if (bn && bn->uses_batch_stats() && !no_prologue_fusion) {
                auto op = std::make_unique<BnReluAggregateOp>();
                op->eps = bn->_eps;
                if (have_stats) {
                    op->mean = st_mean;
                    op->var = st_var;
                    op->have_stats = true;
                }
                auto res = op->forward(_cache_adj, out, _cache_norm, get_parameter("bias"), bn->get_parameter("gammas"),
                                       bn->_affine ? bn->get_parameter("betas") : nullptr);
                if (res->requires_grad()) res->grad_fn = std::move(op);
                return res;
            }
            out = (*get_module("bnorm"))(out);
            out = (*get_module("relu"))(out);
        }
        auto op = std::make_unique<AggregateOp>();
        auto res = op->forward(_cache_adj, out, _cache_norm, get_parameter("bias"));
        if (res->requires_grad()) res->grad_fn = std::move(op);
        return res;
    }
    auto [edge_index, _] = add_self_loops(*input.edge_index(), nullptr, 0, (int)input.num_nodes());
    auto out = (*get_module("lin"))(input.x());
    if (!hot_path_only) {
        out = (*get_module("bnorm"))(out);
        out = (*get_module("relu"))(out);
    }
    auto adj_mat = edge_to_adj_mat(*edge_index, nullptr, input.num_nodes());
    auto deg = adj_mat->sum(-1, true) + 1;
    deg = deg->pow(-0.5);
    auto norm = adj_mat->mm(deg);
    norm *= deg;
    out = propagate(*edge_index, out, &norm);
    out = out + get_parameter("bias");
    return out;
}

tptr<float> GCNConv::propagate(const tensor<int> &edge_index, const tptr<float> &x, const tptr<float> *norm)
{
    return aggregate_and_update(x, edge_index, norm);
}

// norm (.) (A . x): SpMM then row scale, recorded as MatMul + Mul so that backward is A^T . (norm (.) G)
// (reference graph.cpp:204-212)
tptr<float> GCNConv::aggregate_and_update(const tptr<float> &x, const tensor<int> &edge_index, const tptr<float> *norm)
{
    auto adj_mat = edge_to_adj_mat(edge_index, nullptr, x->shape()[0]);
    if (fused) {
        auto op = std::make_unique<AggregateOp>();
        auto res = op->forward(adj_mat, x, *norm, nullptr);
        if (res->requires_grad()) res->grad_fn = std::move(op);
        return res;
    }
    auto agg_x = adj_mat->mm(x);
    agg_x = agg_x * *norm;
    return agg_x;
}

}  // namespace graph
