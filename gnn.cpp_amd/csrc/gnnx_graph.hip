// Graph build on the device: COO -> CSR with the reference's adjacency semantics, and the
// degree / normalisation block of GCNConv::forward.
//
// Replaces (reference src/graph.cpp): edge_to_adj_mat :21-44 (dense N x N scatter by assignment =>
// duplicates collapse), add_self_loops :68-75 with fillValue 0 (diagonal zeroed => self loops removed),
// adj_to_edge_list :46-67 (row-major scan => (src,dst) order), and :177-185 (deg, pow(-0.5), A.s, *= s).
//
// Pipeline: pack (src,dst) into one u64 key per edge (self loops / out-of-range -> sentinel) ->
// rocPRIM radix sort of the keys (only the bits that can be set) -> head flags + exclusive scan ->
// compact unique keys into colidx -> rowptr by per-row binary search on the sorted unique keys.
// Integer work, HBM-bound, deterministic.  rocPRIM is used for the sort/scan primitives only.
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <mutex>
#include <vector>

#include "gnnx_common.h"

// Parity depends on separately rounded fp32 mul / add (the reference has no FMA): never contract.
#pragma clang fp contract(off)

using namespace gnnx;

namespace {

constexpr uint64_t kSentinel = ~0ull;

__global__ void pack_keys_kernel(const int32_t *src, const int32_t *dst, int64_t n_edges, int32_t n_nodes,
                                 uint32_t flags, uint64_t *keys, int32_t *bad)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    int32_t r = src[e], c = dst[e];
    uint64_t key;
    if (r < 0 || c < 0 || r >= n_nodes || c >= n_nodes) {
        key = kSentinel;
        atomicOr(bad, 1);
    } else if (r == c && !(flags & GNNX_CSR_KEEP_SELF_LOOPS)) {
        key = kSentinel;
    } else {
        key = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
    }
    keys[e] = key;
}

// flag[i] = 1 if sorted key i is a kept entry (not sentinel, and first of its run unless duplicates are kept)
__global__ void head_flags_kernel(const uint64_t *keys, int64_t n, uint32_t flags, int32_t *flag)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = keys[i];
    bool keep = k != kSentinel;
    if (keep && i > 0 && !(flags & GNNX_CSR_KEEP_DUPLICATES)) keep = keys[i - 1] != k;
    flag[i] = keep ? 1 : 0;
}

__global__ void compact_kernel(const uint64_t *keys, const int32_t *flag, const int32_t *pos, int64_t n,
                               uint64_t *ukeys, int32_t *colidx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    uint64_t k = keys[i];
    int32_t p = pos[i];
    ukeys[p] = k;
    colidx[p] = (int32_t)(k & 0xffffffffu);
}

// rowptr[r] = first position whose key >= (r << 32)   (lower bound on the sorted unique keys)
__global__ void rowptr_kernel(const uint64_t *ukeys, int32_t nnz, int32_t n_nodes, int32_t *rowptr)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_nodes) return;
    uint64_t target = (uint64_t)(uint32_t)r << 32;
    int32_t lo = 0, hi = nnz;
    while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (ukeys[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    rowptr[r] = lo;
}

// s_i = (1 + deg_i)^(-1/2).  The reference calls the HOST libm's powf(deg, -0.5f) (functional.h:253, std::pow on a float) -- glibc's
// powf is 1 ulp away from the correctly rounded value for 9 685 of the 2^24 degrees, the smallest 1058
// (tests/test_oracle_vs_reference.py::test_pow_minus_half_vs_double_rsqrt) -- so s is LOOKED UP in a table of that very call,
// evaluated on the host for 1 .. 1 + max degree by gnnx_degree_norm_f32 (a once-per-graph build call): s, norm and through them the
// whole layer carry the reference's bits at every size, by default.
__global__ void max_degree_kernel(const int32_t *rowptr, int32_t n_rows, int32_t *out)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t d = i < n_rows ? rowptr[i + 1] - rowptr[i] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(d, off, 64);
        d = o > d ? o : d;
    }
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, d);
}

#ifdef GNNX_EXPERIMENTS
__global__ void deg_rsqrt_exp_kernel(const int32_t *rowptr, int32_t n_rows, float *s)   // A/B only: the correctly rounded value, no table
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    s[i] = (float)(1.0 / sqrt((double)((float)(rowptr[i + 1] - rowptr[i]) + 1.0f)));
}
#endif

__global__ void deg_pow_table_kernel(const int32_t *rowptr, int32_t n_rows, const float *table, int32_t table_len, float *s, int32_t *bad)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int32_t k = rowptr[i + 1] - rowptr[i] + 1;   // table[k] = powf((float)k, -0.5f)
    if (k < 1 || k >= table_len) {   // never on a consistent rowptr (the table covers 1 + max degree): an error, not a wild read
        atomicOr(bad, 1);
        s[i] = 0.f;
        return;
    }
    s[i] = table[k];
}

// norm_i = fl(fl(sum_{j desc} s_j) * s_i), strictly sequential in the reference's matmul order (descending column)
// => bit-identical.  N x 1 SpMV, once per graph (not on the per-step path).
//   short rows: one lane per row (norm_kernel);
//   rows of >= kLongRow non-zeros (power-law hubs, a 100k-degree row would keep one lane busy for tens of ms) are summed by a
//   whole wavefront each (norm_long_kernel): 64 values are fetched in parallel (the next 64 already in flight), then added one by
//   one IN ORDER through v_readlane -- same additions, same order, ~50x the load parallelism.
// The two kernels share NOTHING but the degree test: each finds its rows from rowptr by itself (round 4 handed the long rows over in
// a device-side list with an atomic counter -- a memset, an appending kernel and a consuming kernel chained through a temporary;
// when that chain broke, on a stream shared by several host threads, the consumer read a count of zero and the hubs' norm was
// never written: DESIGN.md section 2).
constexpr int kLongRow = 128;

__global__ void norm_kernel(const int32_t *rowptr, const int32_t *colidx, int32_t n_rows, const float *s_rows,
                            const float *s_cols, float *norm)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    int32_t b = rowptr[i], e = rowptr[i + 1];
    if (e - b >= kLongRow) return;   // norm_long_kernel's
    float acc = 0.f;
    for (int32_t p = e - 1; p >= b; p--) acc = __fadd_rn(acc, s_cols[colidx[p]]);
    norm[i] = __fmul_rn(acc, s_rows[i]);
}

__global__ __launch_bounds__(256) void norm_long_kernel(const int32_t *rowptr, const int32_t *colidx, int32_t n_rows, const float *s_rows,
                                                         const float *s_cols, float *norm)
{
    const int lane = threadIdx.x & 63;
    const int32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    // a wavefront looks at 64 consecutive rows at a time (one coalesced rowptr read), then sums the long ones among them one by one
    for (int64_t base = (int64_t)wave * 64; base < n_rows; base += (int64_t)n_waves * 64) {
        const int64_t r = base + lane;
        int32_t my_b = 0, my_e = 0;
        if (r < n_rows) {
            my_b = rowptr[r];
            my_e = rowptr[r + 1];
        }
        uint64_t todo = __ballot(my_e - my_b >= kLongRow);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int32_t row = (int32_t)(base + src);
            const int32_t b = __shfl(my_b, src, 64), e = __shfl(my_e, src, 64);
            float acc = 0.f;
            int32_t q = e - 1 - lane;
            float v = s_cols[colidx[q >= b ? q : b]];  // unpredicated; out-of-range lanes are never added
            for (int32_t hi = e; hi > b; hi -= 64) {
                const int32_t qn = hi - 64 - 1 - lane;  // next chunk, in flight during the serial adds below
                const float vn = s_cols[colidx[qn >= b ? qn : b]];
                const int cnt = hi - b < 64 ? hi - b : 64;
                for (int j = 0; j < cnt; j++) acc = __fadd_rn(acc, __shfl(v, j, 64));
                v = vn;
            }
            if (lane == 0) norm[row] = __fmul_rn(acc, s_rows[row]);
        }
    }
}

}  // namespace

// ---- the host libm's powf(k, -0.5f), k = 0 .. len-1, on the device --------------------------------------------------------------
// One table per device for the whole process: built on the host with the very call the reference makes (functional.h:253 -> glibc
// powf; 1 ulp from the correctly rounded value for 9 685 of the 2^24 integers, so no device-side rsqrt reproduces it), uploaded
// synchronously under a mutex, then IMMUTABLE: a longer table is a new allocation, the old one stays alive (and valid: a prefix of
// the new one) for kernels that are still reading it -- growth is geometric, so the retired copies sum to less than the live one.
// Callers on any thread and any stream may use the returned pointer without further ordering.
namespace gnnx {
int libm_pow_m05_table(size_t need_len, const float **d_table_out, size_t *len_out)
{
    constexpr int kMaxDev = 64;
    constexpr size_t kMaxLen = ((size_t)1 << 24) + 2;   // floats hold integers exactly up to 2^24
    struct Table {
        float *d = nullptr;
        size_t len = 0;
    };
    static std::mutex mu;
    static Table tables[kMaxDev];
    GNNX_REQUIRE(d_table_out && need_len <= kMaxLen, GNNX_ERR_UNSUPPORTED, "degree table of %zu entries (limit %zu)", need_len, kMaxLen);
    int dev = 0;
    GNNX_HIP_CHECK(hipGetDevice(&dev));
    GNNX_REQUIRE(dev >= 0 && dev < kMaxDev, GNNX_ERR_UNSUPPORTED, "device index %d", dev);
    std::lock_guard<std::mutex> lk(mu);
    Table &t = tables[dev];
    if (t.len < need_len) {
        size_t len = need_len > 2 * t.len ? need_len : 2 * t.len;
        if (len < 4096) len = 4096;
        if (len > kMaxLen) len = kMaxLen;
        std::vector<float> h(len);
        volatile float expo = -0.5f;   // a run-time exponent: the call stays a libm powf call whatever the optimiser knows about -0.5
        for (size_t k = 0; k < len; k++) h[k] = powf((float)k, expo);
        float *d = nullptr;
        GNNX_HIP_CHECK(hipMalloc((void **)&d, sizeof(float) * len));
        const hipError_t e = hipMemcpy(d, h.data(), sizeof(float) * len, hipMemcpyHostToDevice);   // complete on return
        if (e != hipSuccess) {
            (void)hipFree(d);
            GNNX_HIP_CHECK(e);
        }
        t.d = d;   // (the previous table, if any, is retired, not freed: a kernel of another thread may be reading it)
        t.len = len;
    }
    *d_table_out = t.d;
    if (len_out) *len_out = t.len;
    return GNNX_OK;
}
}  // namespace gnnx

namespace {
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct CsrWorkspace {
    uint64_t *keys_in, *keys_out;
    int32_t *flag, *pos, *bad;
    void *prim;
    size_t prim_bytes;
    size_t total;
};

int key_bits(int32_t n_nodes)
{
    int bits = 1;
    while (bits < 32 && (1ll << bits) < (long long)n_nodes) bits++;
    return bits;
}

hipError_t plan_workspace(int64_t n_edges, int32_t n_nodes, char *base, CsrWorkspace &w)
{
    size_t n = (size_t)(n_edges > 0 ? n_edges : 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, n, 0, 64);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, n, rocprim::plus<int32_t>());
    if (e != hipSuccess) return e;
    (void)n_nodes;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return base + o; };
    w.keys_in = (uint64_t *)take(n * 8);
    w.keys_out = (uint64_t *)take(n * 8);
    w.flag = (int32_t *)take(n * 4);
    w.pos = (int32_t *)take(n * 4);
    w.bad = (int32_t *)take(256);
    w.prim_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    w.prim = take(w.prim_bytes);
    w.total = off;
    return hipSuccess;
}


// ---- weighted build (edge_attr): A[r][c] = w by assignment, so of duplicate edges the LAST one in the list wins --------
// (reference graph.cpp:40).  Keys are sorted together with their position in the list by a stable radix sort; the last
// element of every run of equal keys is the winner.  GNNX_DIAG_FILL appends one (i, i) key per node behind the list, i.e.
// a later assignment that overrides any self loop given -- fill_diagonal_(value) of add_self_loops (graph.cpp:72).
__global__ void pack_keys_weighted_kernel(const int32_t *src, const int32_t *dst, int64_t n_edges, int32_t n_nodes, int diag_mode,
                                          uint64_t *keys, int32_t *idx, int32_t *bad)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n_edges + (diag_mode == GNNX_DIAG_FILL ? n_nodes : 0);
    if (e >= total) return;
    uint64_t key;
    if (e >= n_edges) {
        const uint32_t i = (uint32_t)(e - n_edges);
        key = ((uint64_t)i << 32) | i;
    } else {
        int32_t r = src[e], c = dst[e];
        if (r < 0 || c < 0 || r >= n_nodes || c >= n_nodes) {
            key = kSentinel;
            atomicOr(bad, 1);
        } else if (r == c && diag_mode != GNNX_DIAG_KEEP) {
            key = kSentinel;  // stripped, or replaced by the appended diagonal entry
        } else {
            key = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
        }
    }
    keys[e] = key;
    idx[e] = (int32_t)e;
}

__global__ void tail_flags_weighted_kernel(const uint64_t *keys, const int32_t *idx, int64_t n, int64_t n_edges, const float *w,
                                           float diag_value, uint32_t flags, int32_t *flag)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    bool keep = k != kSentinel;
    if (keep && i + 1 < n && !(flags & GNNX_CSR_KEEP_DUPLICATES)) keep = keys[i + 1] != k;  // last of its run
    if (keep && (flags & GNNX_CSR_DROP_TRUNCATED_ZERO)) {
        const int32_t e = idx[i];
        const float v = e < n_edges ? w[e] : diag_value;
        keep = (int)v != 0;  // adj_to_edge_list: `if (int(data[i]) != 0)` (graph.cpp:54) -- |w| < 1 vanishes
    }
    flag[i] = keep ? 1 : 0;
}

__global__ void compact_weighted_kernel(const uint64_t *keys, const int32_t *idx, const int32_t *flag, const int32_t *pos, int64_t n,
                                        int64_t n_edges, const float *w, float diag_value, uint64_t *ukeys, int32_t *colidx,
                                        float *vals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t k = keys[i];
    const int32_t p = pos[i], e = idx[i];
    ukeys[p] = k;
    colidx[p] = (int32_t)(k & 0xffffffffu);
    vals[p] = e < n_edges ? w[e] : diag_value;
}

struct CsrWorkspaceW {
    uint64_t *keys_in, *keys_out;
    int32_t *idx_in, *idx_out, *flag, *pos, *bad;
    void *prim;
    size_t prim_bytes;
    size_t total;
};

hipError_t plan_workspace_weighted(int64_t n_keys, char *base, CsrWorkspaceW &w)
{
    size_t n = (size_t)(n_keys > 0 ? n_keys : 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr,
                                             (int32_t *)nullptr, n, 0, 64);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, n, rocprim::plus<int32_t>());
    if (e != hipSuccess) return e;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return base + o; };
    w.keys_in = (uint64_t *)take(n * 8);
    w.keys_out = (uint64_t *)take(n * 8);
    w.idx_in = (int32_t *)take(n * 4);
    w.idx_out = (int32_t *)take(n * 4);
    w.flag = (int32_t *)take(n * 4);
    w.pos = (int32_t *)take(n * 4);
    w.bad = (int32_t *)take(256);
    w.prim_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    w.prim = take(w.prim_bytes);
    w.total = off;
    return hipSuccess;
}

}  // namespace

__global__ void differ_kernel(const int32_t *a, const int32_t *b, int64_t n, int32_t *differ)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) *differ = 1;  // same value from every writer
}

GNNX_API int gnnx_equal_i32(const int32_t *d_a, const int32_t *d_b, int64_t n, int *equal_out, void *stream)
{
    GNNX_REQUIRE(equal_out && n >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *equal_out = 1;
    if (n == 0 || d_a == d_b) return GNNX_OK;
    GNNX_REQUIRE(d_a && d_b, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    DeviceFreeSync flag_g;   // plain hipMalloc / hipFree, as every temporary of the build-time calls (see gnnx_degree_norm_f32)
    GNNX_HIP_CHECK(hipMalloc(&flag_g.p, sizeof(int32_t)));
    int32_t *flag = static_cast<int32_t *>(flag_g.p);
    GNNX_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(differ_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, st, d_a, d_b, n, flag);
    GNNX_LAUNCH_CHECK();
    int32_t h = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h, flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    *equal_out = h ? 0 : 1;
    return GNNX_OK;
}

__global__ void csr_validate_kernel(const int32_t *rowptr, const int32_t *colidx, int32_t n_rows, int32_t n_cols, int32_t *bad)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    if (b < 0 || e < b || (r == 0 && b != 0)) {
        *bad = 1;
        return;
    }
    for (int32_t p = b; p < e; p++) {
        const int32_t c = colidx[p];
        if (c < 0 || c >= n_cols) *bad = 1;
    }
}

GNNX_API int gnnx_csr_validate(const int32_t *d_rowptr, const int32_t *d_colidx, int32_t n_rows, int32_t n_cols, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0) return GNNX_OK;
    GNNX_REQUIRE(d_rowptr, GNNX_ERR_INVALID_ARG, "rowptr is null");
    hipStream_t st = as_stream(stream);
    int32_t h[2] = {0, 0};
    GNNX_HIP_CHECK(hipMemcpyAsync(&h[0], d_rowptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(h[0] >= 0 && (h[0] == 0 || d_colidx), GNNX_ERR_INVALID_ARG, "colidx is null but the graph has %d entries", h[0]);
    DeviceFreeSync bad_g;
    GNNX_HIP_CHECK(hipMalloc(&bad_g.p, sizeof(int32_t)));
    int32_t *bad = static_cast<int32_t *>(bad_g.p);
    GNNX_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(csr_validate_kernel, dim3((uint32_t)ceil_div(n_rows, 256)), dim3(256), 0, st, d_rowptr, d_colidx, n_rows, n_cols, bad);
    GNNX_LAUNCH_CHECK();
    GNNX_HIP_CHECK(hipMemcpyAsync(&h[1], bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h[1], GNNX_ERR_INDEX_RANGE, "CSR is malformed: rowptr not monotone from 0, or a column id outside [0, %d)", n_cols);
    return GNNX_OK;
}

GNNX_API int gnnx_csr_from_coo_workspace(int64_t n_edges, int32_t n_nodes, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_edges >= 0 && n_nodes >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    CsrWorkspace w;
    GNNX_HIP_CHECK(plan_workspace(n_edges, n_nodes, nullptr, w));
    *bytes = w.total;
    return GNNX_OK;
}

GNNX_API int gnnx_csr_from_coo(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, int32_t n_nodes,
                               uint32_t flags, int32_t *d_rowptr, int32_t *d_colidx, int64_t *nnz_out,
                               void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(n_edges >= 0 && n_nodes >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    GNNX_REQUIRE(n_edges < (1ll << 31), GNNX_ERR_UNSUPPORTED, "n_edges must be < 2^31 (int32 CSR offsets)");
    GNNX_REQUIRE(d_rowptr && nnz_out, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    *nnz_out = 0;
    if (n_edges == 0) {
        GNNX_HIP_CHECK(hipMemsetAsync(d_rowptr, 0, sizeof(int32_t) * ((size_t)n_nodes + 1), st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        return GNNX_OK;
    }
    GNNX_REQUIRE(d_src && d_dst && d_colidx && d_workspace, GNNX_ERR_INVALID_ARG, "null pointer");
    CsrWorkspace w;
    GNNX_HIP_CHECK(plan_workspace(n_edges, n_nodes, (char *)d_workspace, w));
    GNNX_REQUIRE(workspace_bytes >= w.total, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.total);

    const int T = 256;
    dim3 egrid((uint32_t)ceil_div(n_edges, T));
    GNNX_HIP_CHECK(hipMemsetAsync(w.bad, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(pack_keys_kernel, egrid, dim3(T), 0, st, d_src, d_dst, n_edges, n_nodes, flags, w.keys_in, w.bad);
    GNNX_LAUNCH_CHECK();
    // sort all 64 bits when sentinels may be present (they must go last); the row/col fields only need
    // key_bits(n_nodes) bits each, so skip the dead bits [bits, 32) by sorting in two ranges is not possible
    // with one call -- a full 64-bit sort is 8 passes over 8 B keys, fine for a one-off build.
    size_t prim_bytes = w.prim_bytes;
    GNNX_HIP_CHECK(rocprim::radix_sort_keys(w.prim, prim_bytes, w.keys_in, w.keys_out, (size_t)n_edges, 0, 64, st));
    hipLaunchKernelGGL(head_flags_kernel, egrid, dim3(T), 0, st, w.keys_out, n_edges, flags, w.flag);
    GNNX_LAUNCH_CHECK();
    prim_bytes = w.prim_bytes;
    GNNX_HIP_CHECK(rocprim::exclusive_scan(w.prim, prim_bytes, w.flag, w.pos, 0, (size_t)n_edges,
                                           rocprim::plus<int32_t>(), st));
    // unique keys land in keys_in (its content is dead after the sort)
    hipLaunchKernelGGL(compact_kernel, egrid, dim3(T), 0, st, w.keys_out, w.flag, w.pos, n_edges, w.keys_in, d_colidx);
    GNNX_LAUNCH_CHECK();
    int32_t h_last[2] = {0, 0}, h_bad = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_last[0], w.pos + (n_edges - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_last[1], w.flag + (n_edges - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, w.bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE,
                 "invalid input, max value in edge_index should be less than the number of nodes from x");
    int32_t nnz = h_last[0] + h_last[1];
    dim3 rgrid((uint32_t)ceil_div((int64_t)n_nodes + 1, T));
    hipLaunchKernelGGL(rowptr_kernel, rgrid, dim3(T), 0, st, w.keys_in, nnz, n_nodes, d_rowptr);
    GNNX_LAUNCH_CHECK();
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    *nnz_out = nnz;
    return GNNX_OK;
}

GNNX_API int gnnx_csr_from_coo_weighted_workspace(int64_t n_edges, int32_t n_nodes, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_edges >= 0 && n_nodes >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    CsrWorkspaceW w;
    GNNX_HIP_CHECK(plan_workspace_weighted(n_edges + n_nodes, nullptr, w));
    *bytes = w.total;
    return GNNX_OK;
}

GNNX_API int gnnx_csr_from_coo_weighted(const int32_t *d_src, const int32_t *d_dst, const float *d_weights, int64_t n_edges,
                                        int32_t n_nodes, uint32_t flags, int diag_mode, float diag_value, int32_t *d_rowptr,
                                        int32_t *d_colidx, float *d_vals, int64_t *nnz_out, void *d_workspace, size_t workspace_bytes,
                                        void *stream)
{
    GNNX_REQUIRE(n_edges >= 0 && n_nodes >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    GNNX_REQUIRE(diag_mode == GNNX_DIAG_KEEP || diag_mode == GNNX_DIAG_STRIP || diag_mode == GNNX_DIAG_FILL, GNNX_ERR_INVALID_ARG,
                 "unknown diag_mode %d", diag_mode);
    const int64_t n_keys = n_edges + (diag_mode == GNNX_DIAG_FILL ? n_nodes : 0);
    GNNX_REQUIRE(n_keys < (1ll << 31), GNNX_ERR_UNSUPPORTED, "n_edges (+ n_nodes) must be < 2^31 (int32 CSR offsets)");
    GNNX_REQUIRE(d_rowptr && nnz_out, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    *nnz_out = 0;
    if (n_keys == 0) {
        GNNX_HIP_CHECK(hipMemsetAsync(d_rowptr, 0, sizeof(int32_t) * ((size_t)n_nodes + 1), st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        return GNNX_OK;
    }
    GNNX_REQUIRE((n_edges == 0 || (d_src && d_dst && d_weights)) && d_colidx && d_vals && d_workspace, GNNX_ERR_INVALID_ARG,
                 "null pointer");
    CsrWorkspaceW w;
    GNNX_HIP_CHECK(plan_workspace_weighted(n_edges + n_nodes, (char *)d_workspace, w));
    GNNX_REQUIRE(workspace_bytes >= w.total, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, w.total);
    const int T = 256;
    dim3 egrid((uint32_t)ceil_div(n_keys, T));
    GNNX_HIP_CHECK(hipMemsetAsync(w.bad, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(pack_keys_weighted_kernel, egrid, dim3(T), 0, st, d_src, d_dst, n_edges, n_nodes, diag_mode, w.keys_in, w.idx_in,
                       w.bad);
    GNNX_LAUNCH_CHECK();
    size_t prim_bytes = w.prim_bytes;
    // radix sort is stable: inside a run of equal keys the list positions stay ascending
    GNNX_HIP_CHECK(rocprim::radix_sort_pairs(w.prim, prim_bytes, w.keys_in, w.keys_out, w.idx_in, w.idx_out, (size_t)n_keys, 0, 64, st));
    hipLaunchKernelGGL(tail_flags_weighted_kernel, egrid, dim3(T), 0, st, w.keys_out, w.idx_out, n_keys, n_edges, d_weights, diag_value,
                       flags, w.flag);
    GNNX_LAUNCH_CHECK();
    prim_bytes = w.prim_bytes;
    GNNX_HIP_CHECK(rocprim::exclusive_scan(w.prim, prim_bytes, w.flag, w.pos, 0, (size_t)n_keys, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(compact_weighted_kernel, egrid, dim3(T), 0, st, w.keys_out, w.idx_out, w.flag, w.pos, n_keys, n_edges, d_weights,
                       diag_value, w.keys_in, d_colidx, d_vals);
    GNNX_LAUNCH_CHECK();
    int32_t h_last[2] = {0, 0}, h_bad = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_last[0], w.pos + (n_keys - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_last[1], w.flag + (n_keys - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, w.bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE,
                 "invalid input, max value in edge_index should be less than the number of nodes from x");
    int32_t nnz = h_last[0] + h_last[1];
    dim3 rgrid((uint32_t)ceil_div((int64_t)n_nodes + 1, T));
    hipLaunchKernelGGL(rowptr_kernel, rgrid, dim3(T), 0, st, w.keys_in, nnz, n_nodes, d_rowptr);
    GNNX_LAUNCH_CHECK();
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    *nnz_out = nnz;
    return GNNX_OK;
}

GNNX_API int gnnx_degree_norm_f32(const int32_t *d_rowptr, const int32_t *d_colidx, int32_t n_rows, float *d_s,
                                  const float *d_s_cols, float *d_norm, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0) return GNNX_OK;
    GNNX_REQUIRE(d_rowptr, GNNX_ERR_INVALID_ARG, "rowptr is null");
    GNNX_REQUIRE(d_s || d_s_cols, GNNX_ERR_INVALID_ARG, "need d_s or d_s_cols");
    hipStream_t st = as_stream(stream);
    const int T = 256;
    dim3 grid((uint32_t)ceil_div(n_rows, T));
#ifdef GNNX_EXPERIMENTS
    static const int deg_exp = [] { const char *e = experiment_env("GNNX_DEG_RSQRT"); return e ? atoi(e) : 0; }();
    if (d_s && deg_exp) {
        hipLaunchKernelGGL(deg_rsqrt_exp_kernel, grid, dim3(T), 0, st, d_rowptr, n_rows, d_s);
        GNNX_LAUNCH_CHECK();
    } else
#endif
    if (d_s) {
        // s = table[deg + 1], table[k] = the host libm's powf((float)k, -0.5f): what the reference evaluates per vertex
        // (functional.h:253).  The table is the process-wide, immutable, grow-only one of libm_pow_m05_table() -- rank threads of an
        // in-process group that build their shards concurrently share it and never build, upload or free a table of their own.
        // One reduction + one host synchronisation per call (a once-per-graph call).
        DeviceFreeSync max_g;
        GNNX_HIP_CHECK(hipMalloc(&max_g.p, 2 * sizeof(int32_t)));
        int32_t *d_max = static_cast<int32_t *>(max_g.p);   // [0] max degree, [1] error flag of the lookup kernel
        GNNX_HIP_CHECK(hipMemsetAsync(d_max, 0, 2 * sizeof(int32_t), st));
        hipLaunchKernelGGL(max_degree_kernel, grid, dim3(T), 0, st, d_rowptr, n_rows, d_max);
        GNNX_LAUNCH_CHECK();
        int32_t h_max = 0;
        GNNX_HIP_CHECK(hipMemcpyAsync(&h_max, d_max, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        GNNX_REQUIRE(h_max >= 0, GNNX_ERR_INVALID_ARG, "rowptr is not monotone");
        const float *d_table = nullptr;
        size_t table_len = 0;
        const int rc = libm_pow_m05_table((size_t)h_max + 2, &d_table, &table_len);
        if (rc != GNNX_OK) return rc;
        hipLaunchKernelGGL(deg_pow_table_kernel, grid, dim3(T), 0, st, d_rowptr, n_rows, d_table, (int32_t)(table_len < 0x7fffffffu ? table_len : 0x7fffffffu),
                           d_s, d_max + 1);
        GNNX_LAUNCH_CHECK();
        int32_t h_bad = 0;
        GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, d_max + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));   // (d_max is freed behind the kernel)
        GNNX_REQUIRE(!h_bad, GNNX_ERR_HIP, "degree table lookup out of range (max degree read back as %d): rowptr changed under the call?", h_max);
    }
    if (d_norm) {
        if (!d_colidx) {  // only a graph without entries may come without colidx (once-per-graph call: the sync is fine)
            int32_t h_nnz = 0;
            GNNX_HIP_CHECK(hipMemcpyAsync(&h_nnz, d_rowptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            GNNX_HIP_CHECK(hipStreamSynchronize(st));
            GNNX_REQUIRE(h_nnz == 0, GNNX_ERR_INVALID_ARG, "colidx is null but the graph has %d entries", h_nnz);
        }
        // rows' own s: d_s when written here, else the caller's column-indexed vector is also row-indexed
        const float *s_rows = d_s ? d_s : d_s_cols;
        const float *s_cols = d_s_cols ? d_s_cols : d_s;
        hipLaunchKernelGGL(norm_kernel, grid, dim3(T), 0, st, d_rowptr, d_colidx, n_rows, s_rows, s_cols, d_norm);
        GNNX_LAUNCH_CHECK();
        hipLaunchKernelGGL(norm_long_kernel, dim3(1024), dim3(256), 0, st, d_rowptr, d_colidx, n_rows, s_rows, s_cols, d_norm);
        GNNX_LAUNCH_CHECK();   // (no temporary, no count handed from kernel to kernel: stream-asynchronous, nothing to free)
    }
    return GNNX_OK;
}
