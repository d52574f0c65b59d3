// tensor.h -- cyg::tensor<T> / tptr<T> / Operation / Context: the reference's autograd tensor API
// (reference include/tensor.h, include/operation.h, include/functional.h) re-implemented over device memory.
//
// What is kept: type names, the shared_ptr ownership model, the op protocol
// (`forward` saves inputs in a Context, the output owns the op through `grad_fn`, `backward(G)` does
// `_grad += G` then recurses, reference tensor.h:260-276, operation.h:20-100), constructor signatures
// (the pointer ctor ADOPTS the valarray, tensor.h:122-130), error texts, and the member functions the GCN
// hot path uses: mm, t, sum, pow, mul, add, clone, sum_to_size, fill_diagonal_, backward, grad, data,
// set_data, shape/numel/rank, item, operator()(i,j), max, uniform and the arithmetic operators.
//
// What is different (by design, MI355X-first):
//   * storage lives in HBM; `data()` / `grad()` hand out a host std::valarray synchronised on demand;
//   * every arithmetic op of the path is one call into the C-ABI (include/gnnx.h): matmul -> MFMA GEMM or CSR
//     SpMM, mul/add broadcasts -> fused row-scale / bias kernels, sum_to_size -> column sum;
//   * an adjacency "matrix" (graph::edge_to_adj_mat) is a tensor whose shape() is [N,N] but whose storage is
//     CSR (+ lazily its transpose): mm, sum(-1,true), fill_diagonal_ dispatch on that, N x N is never built;
//   * t() of a 2-D tensor is a zero-copy view (a flag the GEMM consumes as transA/transB);
//   * ops outside the hot path throw ERROR_BACKEND_UNSUPPORTED -- there is no CPU fallback;
//   * gradients fan-in correctly (the reference drops the 2nd arrival at an op, operation.h:82-86; never
//     triggered on the hot path itself).
#ifndef GNNCPP_AMD_TENSOR_H
#define GNNCPP_AMD_TENSOR_H

#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <tuple>
#include <typeinfo>
#include <unordered_map>
#include <atomic>
#include <valarray>
#include <vector>

#include "gnnx.h"
#include "utils.h"

namespace cyg {

template <class T>
class tensor;
template <class A>
using tptr = std::shared_ptr<tensor<A>>;

namespace detail {

// One logical buffer with a host copy and/or a device copy; whichever was written last is authoritative.
// A backend op may ask that the NEXT op result of a given size carries spare device elements behind its data (the sharded
// layer: the transform's output [n_local, F] gets room for the halo rows, so the all-to-all-v receives straight behind the rows the
// GEMM wrote and the aggregation gathers from one [local | halo] buffer without a copy).  Per thread, consumed by the first
// device result whose element count matches, dropped by the guard's destructor if nothing matched.
struct TailReservation {
    size_t match_numel, tail_elems;
    static TailReservation *&pending()
    {
        static thread_local TailReservation *p = nullptr;
        return p;
    }
    TailReservation(size_t match_numel_, size_t tail_elems_) : match_numel(match_numel_), tail_elems(tail_elems_) { pending() = this; }
    ~TailReservation()
    {
        if (pending() == this) pending() = nullptr;
    }
    static size_t take(size_t numel)
    {
        TailReservation *p = pending();
        if (!p || p->match_numel != numel) return 0;
        pending() = nullptr;
        return p->tail_elems;
    }
};

// OPT-IN (graph::GCNConv::fuse_bn_stats): the next product X[M,K] . W[N,K]^T with M x N outputs is asked to leave the BatchNorm
// batch statistics of its columns in mean / var (gnnx_gemm_bn_stats_f32: one pass, single-pass variance finished in double --
// within rounding of, not bit-equal to, the exact two-pass statistics that stay the default).  Same per-thread one-shot pattern as
// TailReservation; `done` tells the requester whether a product took it.
struct BnStatsRequest {
    size_t rows, cols;
    float *mean, *var;   // device [cols]
    bool done = false;
    static BnStatsRequest *&pending()
    {
        static thread_local BnStatsRequest *p = nullptr;
        return p;
    }
    BnStatsRequest(size_t rows_, size_t cols_, float *mean_, float *var_) : rows(rows_), cols(cols_), mean(mean_), var(var_) { pending() = this; }
    ~BnStatsRequest()
    {
        if (pending() == this) pending() = nullptr;
    }
    static BnStatsRequest *take(size_t rows, size_t cols)
    {
        BnStatsRequest *p = pending();
        if (!p || p->rows != rows || p->cols != cols) return nullptr;
        pending() = nullptr;
        return p;
    }
};

// The sharded layer (graph::GCNConv::forward_sharded): the next product X[M,K] . W[N,K]^T with M x N outputs is asked to store the rows
// other ranks need into the halo plan's send buffer as well, from its epilogue (gnnx_gemm_nt_rows_to_slots_f32; slots = the plan's
// [M][8] table, send = [n_send][N] dense).  Same per-thread one-shot pattern; `done` tells the requester whether a product took it --
// if not (rows that are no 16-byte pieces), the exchange packs as before.
struct SendSlotsRequest {
    size_t rows, cols;
    const int32_t *slots;
    float *send;
    bool done = false;
    static SendSlotsRequest *&pending()
    {
        static thread_local SendSlotsRequest *p = nullptr;
        return p;
    }
    SendSlotsRequest(size_t rows_, size_t cols_, const int32_t *slots_, float *send_) : rows(rows_), cols(cols_), slots(slots_), send(send_)
    {
        if (slots && send) pending() = this;
    }
    ~SendSlotsRequest()
    {
        if (pending() == this) pending() = nullptr;
    }
    static SendSlotsRequest *take(size_t rows, size_t cols)
    {
        SendSlotsRequest *p = pending();
        if (!p || p->rows != rows || p->cols != cols) return nullptr;
        pending() = nullptr;
        return p;
    }
};

inline uint64_t next_store_id()
{
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

// The next product with rows x cols outputs is asked to store them on the GATHER row pitch (gnnx_gather_row_stride): GCNConv asks
// for it around `lin(x)` when its aggregation -- a pitch-aware consumer -- gathers the product's rows.  Same per-thread one-shot
// pattern as TailReservation.
struct PitchRequest {
    size_t rows, cols, ld;
    static PitchRequest *&pending()
    {
        static thread_local PitchRequest *p = nullptr;
        return p;
    }
    PitchRequest(size_t rows_, size_t cols_, size_t ld_) : rows(rows_), cols(cols_), ld(ld_) { pending() = this; }
    ~PitchRequest()
    {
        if (pending() == this) pending() = nullptr;
    }
    static size_t take(size_t rows, size_t cols)
    {
        PitchRequest *p = pending();
        if (!p || p->rows != rows || p->cols != cols) return 0;
        pending() = nullptr;
        return p->ld;
    }
};

template <class T>
struct Store {
    size_t n = 0;
    size_t tail = 0;  // spare device elements behind the n logical ones (TailReservation)
    std::valarray<T> *host = nullptr;
    void *dev = nullptr;
    bool host_ok = false, dev_ok = false;
    // identity of the CONTENT of the device copy: `id` is unique per Store for the life of the process (never an address, so a new
    // tensor allocated where a freed one lived is a different id), `version` moves whenever the device copy is (re)written -- an
    // upload after the host side was handed out for writing (data()), a kernel output (d_out), an adopted array.  A consumer that
    // caches something derived from a tensor (GCNConv's static-graph cache) keys it on (id, version): no device pass, no host
    // synchronisation per call.
    uint64_t id = next_store_id();
    uint64_t version = 0;

    // Device ROW PITCH (elements): 0 = the n elements are contiguous; else the device copy is prow rows of pcol elements, row r at
    // r * pitch (gnnx_gather_row_stride: a matrix the aggregation gathers rows from).  Only pitch-aware consumers see it
    // (d_pitched): every generic accessor (d, d_out, h) first makes the storage contiguous again, so code that knows nothing of
    // pitches stays correct and merely pays one strided copy.
    size_t pitch = 0, prow = 0, pcol = 0;

    explicit Store(size_t n_) : n(n_) {}
    Store(const Store &) = delete;
    Store &operator=(const Store &) = delete;
    size_t dev_elems() const { return (pitch ? prow * pitch : n) + tail; }
    ~Store()
    {
        delete host;
        if (dev) dev_free(dev, dev_elems() * sizeof(T));
    }
    void adopt_host(std::valarray<T> *h)
    {
        delete host;
        host = h;
        host_ok = true;
        dev_ok = false;
        version++;
    }
    // a fresh (unallocated or dead) device copy gets a row pitch
    void set_pitch(size_t rows, size_t cols, size_t ld)
    {
        if (dev) {
            dev_free(dev, dev_elems() * sizeof(T));
            dev = nullptr;
            dev_ok = false;
        }
        if (ld > cols && rows * cols == n) {
            pitch = ld;
            prow = rows;
            pcol = cols;
        } else {
            pitch = prow = pcol = 0;
        }
    }
    // back to contiguous rows (contents kept when valid)
    void make_contiguous()
    {
        if (!pitch) return;
        if constexpr (std::is_same_v<T, bool>) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        void *old = dev;
        const size_t old_bytes = dev_elems() * sizeof(T), ld = pitch, rows = prow, cols = pcol;
        pitch = prow = pcol = 0;
        dev = nullptr;
        if (old && dev_ok && n) {
            dev = dev_alloc((n + tail) * sizeof(T));
            gx(gnnx_memcpy2d_d2d(dev, cols * sizeof(T), old, ld * sizeof(T), cols * sizeof(T), rows, current_stream()), "compact");
        } else {
            dev_ok = false;
        }
        if (old) dev_free(old, old_bytes);   // (a pooled block: whoever takes it next is ordered behind the copy on this thread's stream)
    }
    // host view, valid contents
    std::valarray<T> *h()
    {
        if (!host) host = new std::valarray<T>(T(), n);
        if (!host_ok) {
            if (dev_ok && n) {
                if constexpr (std::is_same_v<T, bool>) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
                else {
                    make_contiguous();
                    gx(gnnx_memcpy_d2h(&(*host)[0], dev, n * sizeof(T), current_stream()), "download");
                }
            }
            host_ok = true;
        }
        return host;
    }
    void ensure_dev_alloc()
    {
        if (!dev && dev_elems()) dev = dev_alloc(dev_elems() * sizeof(T));   // n == 0 with a reserved tail (a rank without local rows) still allocates
    }
    // device pointer, valid contents
    T *d()
    {
        if constexpr (std::is_same_v<T, bool>) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        make_contiguous();
        ensure_dev_alloc();
        if (!dev_ok) {
            if (host_ok && n) gx(gnnx_memcpy_h2d(dev, &(*host)[0], n * sizeof(T), current_stream()), "upload");
            else if (n) gx(gnnx_memset(dev, 0, n * sizeof(T), current_stream()), "memset");
            dev_ok = true;
            version++;
        }
        return static_cast<T *>(dev);
    }
    // device pointer about to be fully overwritten by a kernel
    T *d_out()
    {
        if (pitch) set_pitch(0, 0, 0);   // a generic writer knows nothing of pitches
        ensure_dev_alloc();
        dev_ok = true;
        host_ok = false;
        version++;
        return static_cast<T *>(dev);
    }
    // pitch-aware accessors: the device copy as it is laid out; ld = the row pitch in elements (cols when contiguous)
    T *d_pitched(size_t cols, size_t &ld)
    {
        if (!pitch || !dev_ok) {   // contiguous, or not on the device yet: the generic path
            T *p = d();
            ld = cols;
            return p;
        }
        ld = pitch;
        return static_cast<T *>(dev);
    }
    T *d_out_pitched(size_t cols, size_t &ld)
    {
        ensure_dev_alloc();
        dev_ok = true;
        host_ok = false;
        version++;
        ld = pitch ? pitch : cols;
        return static_cast<T *>(dev);
    }
    void host_written()
    {
        host_ok = true;
        dev_ok = false;
    }
    // never written since it was created: logically all zeros (zero_grad() allocates nothing)
    bool untouched() const { return !host_ok && !dev_ok; }
};

// CSR adjacency living on the device (+ the COO it came from, so that A^T can be built on demand).
struct Csr {
    int32_t n = 0;
    int64_t nnz = 0, nnz_t = -1;
    void *rowptr = nullptr, *colidx = nullptr;      // A
    void *rowptr_t = nullptr, *colidx_t = nullptr;  // A^T (lazy)
    void *coo_src = nullptr, *coo_dst = nullptr;    // device copies of the edge list
    // weighted adjacency (edge_attr): per-edge weights of the list, per-entry values of A / A^T (last duplicate wins)
    void *coo_w = nullptr, *vals = nullptr, *vals_t = nullptr;
    int diag_mode = GNNX_DIAG_KEEP;                 // fill_diagonal_(0) -> STRIP, fill_diagonal_(v) -> FILL with diag_value
    float diag_value = 0.0f;
    bool weighted() const { return coo_w != nullptr; }
    void make_weighted();                           // attach unit weights (fill_diagonal_(v != 0) on a 0/1 adjacency)
    int64_t n_edges = 0;
    uint32_t flags = GNNX_CSR_KEEP_SELF_LOOPS;       // edge_to_adj_mat keeps the diagonal; fill_diagonal_(0) strips it
    // load-balancing plans of the SpMM (power-law rows), built on demand for a feature width
    gnnx_spmm_plan *plan = nullptr, *plan_t = nullptr;
    gnnx_spmm_plan *plan_pro = nullptr;   // forward plan of the aggregation that applies BatchNorm / ReLU to every gathered row (ensure_prologue_plan)
    int32_t plan_feat = 0;
    void *norm_per_nz_t = nullptr;  // norm[colidx_t[p]]: the backward's per-source scale as a coalesced stream
    ~Csr();
    void build();            // (re)build A from the COO with `flags`
    void ensure_transpose();
    void ensure_plans(int32_t n_feat);
    void ensure_prologue_plan(int32_t n_feat);
    void drop_plans();
};

}  // namespace detail

// ---------------------------------------------------------------------------------------------------
// autograd protocol (reference operation.h:20-100)
template <class T>
class Context {
public:
    std::vector<std::shared_ptr<T>> cache;
    std::unordered_map<std::string, int> saved_data;
    void save_for_backward(std::vector<std::shared_ptr<T>> tensors)
    {
        for (const auto &t : tensors) cache.push_back(t);
    }
    std::vector<std::shared_ptr<T>> get_variables() { return cache; }
};

template <class T>
class Operation {
public:
    std::string name = "Operation";
    std::unique_ptr<Context<T>> context = std::make_unique<Context<T>>();
    virtual ~Operation() = default;
    // Unlike the reference (operation.h:80-88) a second arrival is not dropped: every arrival propagates,
    // which sums gradients correctly on fan-out.  The hot path is a chain, where both behave the same.
    void backward(std::shared_ptr<tensor<float>> incoming_grad) { _backward(std::move(incoming_grad)); }
    virtual void _backward(std::shared_ptr<tensor<float>> incoming_grad) = 0;
};

template <class T>
void CHECK_BACKWARD(const std::vector<std::shared_ptr<T>> &var, size_t expected = 1)
{
    if (var.size() != expected) throw std::runtime_error("cannot backprop without a executing a forward computation first");
}

namespace functional {
template <class T>
tptr<T> add(const tptr<T> &lhs, const tptr<T> &rhs);
template <class T>
tptr<T> mul(const tptr<T> &lhs, const tptr<T> &rhs);
template <class T>
tptr<T> div(const tptr<T> &lhs, const tptr<T> &rhs);
template <class T>
tptr<T> pow(const tptr<T> &base, float exponent);
template <class T>
tptr<T> sum(const tptr<T> &base, int dim, bool keepdim);
template <class T>
tptr<T> matmul(const tptr<T> &lhs, const tptr<T> &rhs);
template <class T>
std::tuple<tptr<T>, tptr<int>> max(const tensor<T> &t);
}  // namespace functional

template <class T>
class Add;
template <class T>
class Mul;
template <class T>
class Div;
template <class T>
class MatMul;
template <class T>
class Transpose;
template <class T>
class Sum;

// ---------------------------------------------------------------------------------------------------
template <class T>
class tensor : public std::enable_shared_from_this<tensor<T>> {
    // arithmetic operators of the API (reference tensor.h:30-95)
    friend tptr<T> operator+(tptr<T> lhs, const tptr<T> &rhs) { return lhs->add(rhs); }
    friend tptr<T> operator+(tptr<T> lhs, const float &rhs) { return lhs->add(rhs); }
    friend tptr<T> operator+(tptr<T> lhs, const int &rhs) { return lhs->add((float)rhs); }
    friend tptr<T> operator-(const tptr<T> &lhs) { return lhs->mul(-1.0f); }
    friend tptr<T> operator-(tptr<T> lhs, const float &rhs) { return lhs->add(-rhs); }
    friend tptr<T> operator-(tptr<T> lhs, const tptr<T> &rhs) { return lhs->add(-rhs); }  // add(-rhs), as the reference spells it
    friend tptr<T> operator/(tptr<T> lhs, const tptr<T> &rhs) { return lhs->div(rhs); }
    friend tptr<T> operator/(tptr<T> lhs, const float &rhs) { return lhs->div(rhs); }
    friend tptr<T> operator*(tptr<T> lhs, const tptr<T> &rhs) { return lhs->mul(rhs); }
    friend tptr<T> operator*(tptr<T> lhs, const float &rhs) { return lhs->mul(rhs); }
    friend tptr<T> operator*=(tptr<T> lhs, const tptr<T> &rhs)
    {
        lhs->check_in_place();
        auto out = lhs * rhs;
        lhs->assign_from(*out);
        return lhs;
    }
    friend tptr<T> operator+=(tptr<T> lhs, const tptr<T> &rhs)
    {
        lhs->check_in_place();
        auto out = lhs + rhs;
        lhs->assign_from(*out);
        return lhs;
    }
    friend tptr<T> operator+=(tptr<T> lhs, const float &rhs)
    {
        lhs->check_in_place();
        auto out = lhs + rhs;
        lhs->assign_from(*out);
        return lhs;
    }
    friend tptr<T> operator*=(tptr<T> lhs, const float &rhs)
    {
        lhs->check_in_place();
        auto out = lhs * rhs;
        lhs->assign_from(*out);
        return lhs;
    }
    template <class A>
    friend tptr<T> operator-=(tptr<T> lhs, const A &rhs)
    {
        lhs->check_in_place();
        auto out = lhs - rhs;
        lhs->assign_from(*out);
        return lhs;
    }
    template <class A>
    friend tptr<T> operator/=(tptr<T> lhs, const A &rhs)
    {
        lhs->check_in_place();
        auto out = lhs / rhs;
        lhs->assign_from(*out);
        return lhs;
    }

public:
    typedef T value_type;
    std::unique_ptr<Operation<tensor<T>>> grad_fn;

    explicit tensor(std::vector<size_t> dims, T value = 0, bool requires_grad = false)
        : _dims(dims), _requires_grad(requires_grad)
    {
        CHECK_VALID_DIMS(dims);
        _st = std::make_shared<detail::Store<T>>(numel_of(dims));
        _st->adopt_host(new std::valarray<T>(value, _st->n));
        init_grad();
    }
    // ADOPTS `data` (reference tensor.h:122-130)
    explicit tensor(std::vector<size_t> dims, std::valarray<T> *data, bool requires_grad = false)
        : _dims(dims), _requires_grad(requires_grad)
    {
        CHECK_VALID_DIMS(dims);
        _st = std::make_shared<detail::Store<T>>(numel_of(dims));
        if (data != nullptr) {
            CHECK_SIZE(dims, data->size());
            _st->adopt_host(data);
        } else {
            _st->adopt_host(new std::valarray<T>(T(), _st->n));
        }
        init_grad();
    }
    // backend-side constructors
    struct device_tag {};
    // op results: never leaves, so no gradient buffer is attached (backward() only accumulates into leaves)
    tensor(device_tag, std::vector<size_t> dims, bool requires_grad) : _dims(dims), _requires_grad(requires_grad), _backend_temp(true)
    {
        CHECK_VALID_DIMS(dims);
        _st = std::make_shared<detail::Store<T>>(numel_of(dims));
        _st->tail = detail::TailReservation::take(_st->n);
    }
    // spare device elements behind the data (detail::TailReservation); 0 for views and ordinary results
    size_t device_tail_capacity() const { return (_st && !_tview) ? _st->tail : 0; }
    tensor(std::shared_ptr<detail::Csr> csr, bool) : _dims({(size_t)csr->n, (size_t)csr->n}), _requires_grad(false), _csr(csr) {}

    // ---- plain accessors
    std::vector<size_t> shape() const { return _dims; }
    size_t numel() const { return numel_of(_dims); }
    int rank() const { return (int)_dims.size(); }
    bool requires_grad() const { return _requires_grad; }
    bool is_csr() const { return (bool)_csr; }
    bool is_transposed_view() const { return _tview; }
    const std::shared_ptr<detail::Csr> &csr() const { return _csr; }

    // host view of the values; the caller may write through it, so the host copy becomes authoritative
    std::valarray<T> *data()
    {
        materialize();
        auto *h = _st->h();
        _st->host_written();
        return h;
    }
    // read-only host view: the device copy stays authoritative (no re-upload, no new content version -- a consumer that caches on
    // (storage id, version), such as GCNConv's static-graph cache, is not invalidated by a READ of the edge list)
    const std::valarray<T> *cdata()
    {
        materialize();
        return _st->h();
    }
    // device pointers for the backend (valid contents / to be overwritten)
    T *device_data()
    {
        materialize();
        return _st->d();
    }
    T *device_out() { return _st->d_out(); }
    // pitch-aware access to a 2-D tensor's device copy (detail::Store::pitch): ld = row pitch in elements
    T *device_pitched(int64_t &ld)
    {
        materialize();
        size_t l = 0;
        T *p = _st->d_pitched(_dims.empty() ? 1 : _dims.back(), l);
        ld = (int64_t)l;
        return p;
    }
    T *device_out_pitched(int64_t &ld)
    {
        size_t l = 0;
        T *p = _st->d_out_pitched(_dims.empty() ? 1 : _dims.back(), l);
        ld = (int64_t)l;
        return p;
    }
    void set_device_pitch(size_t ld)
    {
        if (_dims.size() == 2 && !_tview) _st->set_pitch(_dims[0], _dims[1], ld);
    }
    // (id, version) of the device copy's content (detail::Store): call after device_data(), which brings the copy up to date
    uint64_t storage_id() const { return _st ? _st->id : 0; }
    uint64_t storage_version() const { return _st ? _st->version : 0; }
    // valid device contents that a kernel is about to update in place (optimiser step)
    T *device_inplace()
    {
        materialize();
        T *p = _st->d();
        _st->host_ok = false;
        _st->version++;   // the content is about to change: whatever was derived from (id, version) is stale
        return p;
    }
    // storage as it is laid out ([rows, cols] of the UNtransposed buffer) + whether this tensor views it transposed
    T *device_storage(bool &transposed)
    {
        transposed = _tview;
        return _st->d();
    }

    template <class A>
    void set_data(std::valarray<A> *data)
    {
        if (data->size() != numel()) throw std::runtime_error(ERROR_SIZE_MISMATCH);
        require_dense();
        _tview = false;
        auto *h = new std::valarray<T>(numel());
        for (size_t i = 0; i < numel(); i++) (*h)[i] = static_cast<T>((*data)[i]);
        _st = std::make_shared<detail::Store<T>>(numel());
        _st->adopt_host(h);
        if (_requires_grad) zero_grad();
    }

    std::valarray<float> *grad()
    {
        if (grad_fn != nullptr) throw std::runtime_error(WARNING_GRAD_NOT_LEAF);
        if (!_grad) throw std::runtime_error("invalid op, pls enable grad on this tensor");
        auto *h = _grad->h();
        _grad->host_written();
        return h;
    }
    float *device_grad() { return _grad ? _grad->d() : nullptr; }
    // the gradient buffer about to be updated in place by the backend (all-reduce over the ranks): a host copy goes stale
    float *device_grad_inplace()
    {
        if (!_grad) return nullptr;
        float *p = _grad->d();
        _grad->host_ok = false;
        return p;
    }

    void requires_grad_(bool requires_grad)
    {
        _requires_grad = requires_grad;
        if (requires_grad) zero_grad();
        else _grad.reset();
    }
    void zero_grad()
    {
        if (typeid(T) != typeid(float)) throw std::runtime_error(ERROR_GRAD_DTYPE);
        // lazily zero: the first device use memsets HBM, the first host use creates a zero valarray -- no host
        // allocation of N x F zeros per call
        _grad = std::make_shared<detail::Store<float>>(numel());
    }

    // drop / insert size-1 dimensions (reference tensor.h:232-252); metadata only
    void squeeze()
    {
        require_dense();
        _dims.erase(std::remove(_dims.begin(), _dims.end(), (size_t)1), _dims.end());
        if (_dims.empty()) _dims.push_back(1);
    }
    tptr<T> unsqueeze(int dim)
    {
        require_dense();
        CHECK_VALID_RANGE(dim, rank() + 1, -rank() - 1);
        _dims.insert(_dims.begin() + (dim < 0 ? dim + rank() + 1 : dim), 1);
        return this->shared_from_this();
    }

    // ---- autograd entry (reference tensor.h:260-276)
    void backward(std::shared_ptr<tensor<float>> incoming_gradient = nullptr)
    {
        if (incoming_gradient == nullptr && numel() != 1) throw std::runtime_error(ERROR_NON_SCALAR_BACKPROP);
        if (incoming_gradient == nullptr) incoming_gradient = std::make_shared<tensor<float>>(_dims, 1.0f, false);
        if (incoming_gradient->numel() != numel()) throw std::runtime_error(ERROR_GRAD_MISMATCH);
        if (_grad && !grad_fn) {  // `_grad += G`; only leaves keep it (a non-leaf's grad() is refused anyway)
            if (_grad->untouched()) {
                // first gradient into a zeroed buffer: 0 + G is G.  A temporary made by a backend op that nobody else holds
                // (the calling op's local + this parameter) simply BECOMES the gradient buffer -- no memset, no N x F
                // read-modify-write; anything else is copied once.
                if constexpr (std::is_same_v<T, float>) {
                    auto &g = *incoming_gradient;
                    if (g._backend_temp && !g._tview && !g._csr && incoming_gradient.use_count() <= 2 && g._st.use_count() == 1 &&
                        g._st->n == numel() && g._st->dev_ok) {
                        _grad = g._st;
                    } else {
                        detail::gx(gnnx_memcpy_d2d(_grad->d_out(), g.device_data(), numel() * sizeof(float), detail::current_stream()),
                                   "grad accumulate");
                    }
                } else {
                    detail::gx(gnnx_memcpy_d2d(_grad->d_out(), incoming_gradient->device_data(), numel() * sizeof(float),
                                               detail::current_stream()), "grad accumulate");
                }
            } else {
                detail::gx(gnnx_axpy_f32((int64_t)numel(), 1.0f, incoming_gradient->device_data(), _grad->d(),
                                         detail::current_stream()), "grad accumulate");
            }
            _grad->host_ok = false;
        }
        if (grad_fn) grad_fn->backward(incoming_gradient);
    }

    // ---- ops of the hot path (each builds the reference's op object and records it)
    tptr<T> add(const tptr<T> &other)
    {
        CHECK_ARGS_OPS_BROADCAST(shape(), other->shape());
        auto op = std::make_unique<Add<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), other);
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    tptr<T> add(const float &other) { return add(std::make_shared<tensor<T>>(std::vector<size_t>{1}, static_cast<T>(other), false)); }
    tptr<T> mul(const tptr<T> &other)
    {
        CHECK_ARGS_OPS_BROADCAST(shape(), other->shape());
        auto op = std::make_unique<Mul<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), other);
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    tptr<T> mul(const float &other) { return mul(std::make_shared<tensor<T>>(std::vector<size_t>{1}, static_cast<T>(other), false)); }
    tptr<T> div(const tptr<T> &other)
    {
        CHECK_ARGS_OPS_BROADCAST(shape(), other->shape());
        auto op = std::make_unique<Div<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), other);
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    tptr<T> div(const float &other) { return div(std::make_shared<tensor<T>>(std::vector<size_t>{1}, static_cast<T>(other), false)); }
    tptr<T> mm(const tptr<T> &other)
    {
        CHECK_MM_DIMS(shape(), other->shape());
        auto op = std::make_unique<MatMul<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), other);
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    template <class A>
    tptr<T> pow(const A &exponent)
    {
        if (_requires_grad) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);  // only the (grad-free) degree scaling is on the path
        return functional::pow(this->shared_from_this(), (float)exponent);
    }
    tptr<T> sum(int dim = INT_MAX, const bool &keepdim = false, const bool &inplace = false)
    {
        CHECK_VALID_RANGE(dim, rank(), -rank());
        auto op = std::make_unique<Sum<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), dim, keepdim);
        if (inplace) {
            assign_from(*out);
            return this->shared_from_this();
        }
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    // transpose of two adjacent dims; for 2-D tensors a zero-copy view (the GEMM takes it as a flag)
    tptr<T> t(int d1 = -1, int d2 = -2, const bool &inplace = false)
    {
        CHECK_TRANSPOSE(shape(), d1, d2);
        if (rank() != 2) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        if (inplace) {
            check_in_place();
            require_dense();
            std::swap(_dims[0], _dims[1]);
            _tview = !_tview;
            return this->shared_from_this();
        }
        auto op = std::make_unique<Transpose<tensor<T>>>();
        auto out = op->forward(this->shared_from_this(), d1, d2);
        if (out->requires_grad()) out->grad_fn = std::move(op);
        return out;
    }
    tptr<T> clone(const bool &require_grad = false, const T fillValue = INT_MAX) const
    {
        auto self = const_cast<tensor<T> *>(this);
        self->require_dense();
        if (fillValue != static_cast<T>(INT_MAX)) return std::make_shared<tensor<T>>(_dims, fillValue, require_grad);
        auto out = std::make_shared<tensor<T>>(device_tag{}, _dims, require_grad);
        if constexpr (std::is_same_v<T, bool>) {
            *out->_st->h() = *self->_st->h();
            out->_st->host_written();
        } else {
            detail::gx(gnnx_memcpy_d2d(out->device_out(), self->device_data(), numel() * sizeof(T), detail::current_stream()),
                       "clone");
        }
        return out;
    }
    // reduce (in place) to `dims` by summing the broadcast dimensions (reference tensor.h:618-638)
    void sum_to_size(std::vector<size_t> dims)
    {
        if ((size_t)rank() < dims.size()) throw std::runtime_error("not expandable");
        if (!is_broadcastable(_dims, dims)) throw std::runtime_error("dims is not broacastable to this tensor's size");
        for (int i = -1; i >= -rank(); i--) {
            if ((size_t)std::abs(i) > dims.size()) {
                this->sum(0, false, true);
            } else {
                size_t mine = _dims[_dims.size() + i], want = dims[dims.size() + i];
                if (mine < want) throw std::runtime_error(" not expandable");
                if (mine != want) this->sum(i, true, true);
            }
        }
    }
    template <class A>
    void fill_diagonal_(const A &value = 0)
    {
        if (rank() != 2 || _dims[0] != _dims[1]) throw std::runtime_error("all dimensions must be of same length and tensor must be 2D");
        if (_csr) {
            if ((float)value == 0.0f) {  // 0 strips self loops (graph.cpp:72): a zero entry adds nothing to any product
                _csr->flags &= ~GNNX_CSR_KEEP_SELF_LOOPS;
                _csr->diag_mode = GNNX_DIAG_STRIP;
            } else {                     // every (i, i) becomes `value`, whether the list had that self loop or not
                _csr->make_weighted();
                _csr->diag_mode = GNNX_DIAG_FILL;
                _csr->diag_value = (float)value;
            }
            _csr->build();
            return;
        }
        auto *h = data();
        for (size_t i = 0; i < _dims[0]; i++) (*h)[i * _dims[1] + i] = static_cast<T>(value);
    }
    T item()
    {
        if (numel() != 1) throw std::runtime_error("invalid op, item() is only valid for tensors with one element");
        return (*const_cast<tensor<T> *>(this)->_st->h())[0];
    }
    // element access (scalar tensor, as the reference returns: tensor.h:282-293)
    tptr<T> operator()(size_t i, size_t j)
    {
        if (rank() != 2) throw std::runtime_error(ERROR_RANK_MISMATCH);
        if (i >= _dims[0] || j >= _dims[1]) throw std::runtime_error(ERROR_OUT_OF_RANGE);
        materialize();
        return std::make_shared<tensor<T>>(std::vector<size_t>{1}, (*_st->h())[i * _dims[1] + j], false);
    }
    T operator[](size_t i)
    {
        if (i >= numel()) throw std::runtime_error(ERROR_OUT_OF_RANGE);
        materialize();
        return (*_st->h())[i];
    }
    // the reference declares both accessors const (tensor.h:282,295) and calls them on const tensors (graph.cpp:35-36)
    tptr<T> operator()(size_t i, size_t j) const { return (*const_cast<tensor<T> *>(this))(i, j); }
    T operator[](size_t i) const { return (*const_cast<tensor<T> *>(this))[i]; }
    std::tuple<tptr<T>, tptr<int>> max() { return functional::max<T>(*this); }
    // U(low, high) initialisation (reference tensor.h:693-705; the reference seeds from time(), utils.cpp:6 --
    // here the engine is seedable through cyg::manual_seed for reproducible runs)
    tptr<T> uniform(const float &low, const float &high);

    // ---- backend helpers
    void assign_from(tensor<T> &other)
    {
        _st = other._st;
        _dims = other._dims;
        _tview = other._tview;
        _csr = other._csr;
        _csr_t = other._csr_t;
    }
    void check_in_place() const
    {
        if (_requires_grad && grad_fn == nullptr) throw std::runtime_error(ERROR_IN_PLACE_OP_LEAF);
    }
    void require_dense() const
    {
        if (_csr) throw std::runtime_error("this tensor is a CSR adjacency (graph::edge_to_adj_mat); it has no dense data");
    }
    // turn a transposed view into real row-major storage
    void materialize()
    {
        require_dense();
        if (!_tview) return;
        if constexpr (std::is_same_v<T, float>) {
            auto fresh = std::make_shared<detail::Store<T>>(numel());
            // stored buffer is [dims1, dims0]; this tensor is its transpose [dims0, dims1]
            detail::gx(gnnx_transpose_f32(_st->d(), (int64_t)_dims[0], (int64_t)_dims[1], (int64_t)_dims[0], fresh->d_out(),
                                          (int64_t)_dims[1], detail::current_stream()), "transpose");
            _st = fresh;
            _tview = false;
        } else {
            throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        }
    }
    // grad-free transposed alias of this tensor: no copy (dense: flips the view flag; CSR: selects A^T)
    tptr<T> clone_view_t()
    {
        if (rank() != 2) throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED);
        if (_csr) {
            auto out = std::make_shared<tensor<T>>(_csr, false);
            out->_csr_t = !_csr_t;
            return out;
        }
        auto out = std::make_shared<tensor<T>>(device_tag{}, std::vector<size_t>{_dims[1], _dims[0]}, false);
        out->set_storage(_st, !_tview);
        return out;
    }
    bool csr_transposed() const { return _csr_t; }
    std::shared_ptr<detail::Store<T>> storage() const { return _st; }
    void set_storage(std::shared_ptr<detail::Store<T>> st, bool tview)
    {
        _st = std::move(st);
        _tview = tview;
    }

private:
    void init_grad()
    {
        if (_requires_grad) {
            if (typeid(T) != typeid(float)) throw std::runtime_error(ERROR_GRAD_DTYPE);
            zero_grad();
        }
    }
    std::vector<size_t> _dims;
    bool _requires_grad = false;
    bool _tview = false;
    bool _csr_t = false;
    bool _backend_temp = false;  // created by a backend op (device_tag): its storage may be adopted as a leaf's gradient
    std::shared_ptr<detail::Store<T>> _st;
    std::shared_ptr<detail::Store<float>> _grad;
    std::shared_ptr<detail::Csr> _csr;
};

void manual_seed(unsigned long long seed);
float generate_random(const float &low, const float &high);
tptr<float> randn(std::vector<size_t> dims, int low = -1, int high = 1, bool requires_grad = false);

template <class T>
tptr<T> tensor<T>::uniform(const float &low, const float &high)
{
    auto *h = data();
    for (size_t i = 0; i < numel(); i++) (*h)[i] = static_cast<T>(generate_random(low, high));
    return this->shared_from_this();
}

// ---------------------------------------------------------------------------------------------------
// op classes (reference operation.h:103-168, 256-292, 399-434, 490-535): same protocol, device arithmetic
template <class T>
class Add : public Operation<T> {
public:
    Add() { this->name = "Add"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &lhs, const std::shared_ptr<T> &rhs)
    {
        auto output = functional::add(lhs, rhs);
        if (output->requires_grad()) this->context->save_for_backward({lhs, rhs});
        return output;
    }
    void _backward(std::shared_ptr<tensor<float>> incoming_gradient) override
    {
        auto var = this->context->get_variables();
        CHECK_BACKWARD<T>(var, 2);
        for (const auto &t : var)
            if (t->requires_grad()) {
                auto g = incoming_gradient->clone(false);
                g->sum_to_size(t->shape());  // [N,F] -> [F] for the bias: column sum on the device
                t->backward(g);
            }
    }
};

template <class T>
class Mul : public Operation<T> {
public:
    Mul() { this->name = "Mul"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &lhs, const std::shared_ptr<T> &rhs)
    {
        auto output = functional::mul(lhs, rhs);
        if (output->requires_grad()) this->context->save_for_backward({lhs, rhs});
        return output;
    }
    void _backward(std::shared_ptr<tensor<float>> incoming_grad) override
    {
        auto var = this->context->get_variables();
        CHECK_BACKWARD<T>(var, 2);
        auto lhs = var[0], rhs = var[1];
        if (rhs->requires_grad()) {
            auto g = functional::mul(incoming_grad, lhs);
            g->sum_to_size(rhs->shape());
            rhs->backward(g);
        }
        if (lhs->requires_grad()) {
            auto g = functional::mul(incoming_grad, rhs);
            g->sum_to_size(lhs->shape());
            lhs->backward(g);
        }
    }
};

// y = a / b:  dy/da = G / b ;  dy/db = G * (-a / b^2)   (reference operation.h:170-208)
template <class T>
class Div : public Operation<T> {
public:
    Div() { this->name = "Div"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &numerator, const std::shared_ptr<T> &denominator)
    {
        auto output = functional::div(numerator, denominator);
        if (output->requires_grad()) this->context->save_for_backward({numerator, denominator});
        return output;
    }
    void _backward(std::shared_ptr<tensor<float>> incoming_grad) override
    {
        auto var = this->context->get_variables();
        CHECK_BACKWARD<T>(var, 2);
        auto num = var[0], den = var[1];
        if (num->requires_grad()) {
            auto g = functional::div(incoming_grad, den);
            g->requires_grad_(false);
            g->sum_to_size(num->shape());
            num->backward(g);
        }
        if (den->requires_grad()) {
            auto neg = std::make_shared<tensor<float>>(std::vector<size_t>{1}, -1.0f, false);
            auto cn = functional::mul(num, neg);                       // -a
            auto cd = functional::mul(den, den);                       // b^2
            cn->requires_grad_(false);
            cd->requires_grad_(false);
            auto g = functional::mul(incoming_grad, functional::div(cn, cd));
            g->requires_grad_(false);
            g->sum_to_size(den->shape());
            den->backward(g);
        }
    }
};

template <class T>
class Sum : public Operation<T> {
public:
    Sum() { this->name = "Sum"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &base, int dim, bool keepdim)
    {
        auto output = functional::sum(base, dim, keepdim);
        if (output->requires_grad()) this->context->save_for_backward({base});
        return output;
    }
    void _backward(std::shared_ptr<tensor<float>>) override { throw std::runtime_error(ERROR_BACKEND_UNSUPPORTED); }
};

template <class T>
class Transpose : public Operation<T> {
public:
    Transpose() { this->name = "Transpose"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &lhs, int d1 = -1, int d2 = -2)
    {
        lhs->require_dense();
        auto dims = lhs->shape();
        std::swap(dims[0], dims[1]);
        auto output = std::make_shared<T>(typename T::device_tag{}, dims, lhs->requires_grad());
        output->set_storage(lhs->storage(), !lhs->is_transposed_view());  // zero-copy view
        if (output->requires_grad()) {
            this->context->save_for_backward({lhs});
            this->context->saved_data["d1"] = d1;
            this->context->saved_data["d2"] = d2;
        }
        return output;
    }
    void _backward(std::shared_ptr<tensor<float>> incoming_grad) override
    {
        auto var = this->context->get_variables();
        CHECK_BACKWARD<T>(var, 1);
        auto base = var[0];
        if (base->requires_grad()) {
            auto g = incoming_grad->clone(false);
            g->t(this->context->saved_data["d1"], this->context->saved_data["d2"], true);
            base->backward(g);
        }
    }
};

template <class T>
class MatMul : public Operation<T> {
public:
    MatMul() { this->name = "MatMul"; }
    std::shared_ptr<T> forward(const std::shared_ptr<T> &lhs, const std::shared_ptr<T> &rhs)
    {
        auto output = functional::matmul(lhs, rhs);
        if (output->requires_grad()) this->context->save_for_backward({lhs, rhs});
        return output;
    }
    // dL = G . R^T ; dR = L^T . G  (reference operation.h:504-534) -- transposes are views, and for a CSR
    // lhs (the adjacency) L^T . G runs on the transposed CSR instead of a dense N x N transpose.
    void _backward(std::shared_ptr<tensor<float>> incoming_grad) override
    {
        auto var = this->context->get_variables();
        CHECK_BACKWARD<T>(var, 2);
        auto lhs = var[0], rhs = var[1];
        if (lhs->requires_grad()) {
            auto g = functional::matmul(incoming_grad, rhs->clone_view_t());
            g->sum_to_size(lhs->shape());
            lhs->backward(g);
        }
        if (rhs->requires_grad()) {
            auto g = functional::matmul(lhs->clone_view_t(), incoming_grad);
            g->sum_to_size(rhs->shape());
            rhs->backward(g);
        }
    }
};

}  // namespace cyg

#include "functional.h"

#endif
