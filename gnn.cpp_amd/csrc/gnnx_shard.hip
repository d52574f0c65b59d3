// 1-D vertex partition of the graph over the GPUs of one node and the halo plan of a shard, on the device
// (SURVEY.md section 8(b) `gnnx_halo_plan`, 8(e)).  Integer streaming work, HBM-bound, deterministic; rocPRIM supplies
// the sort / scan primitives.  The same logic, as torch index ops, is gnn.cpp_amd/shard.py (the two are compared array for
// array in tests/test_gpu_sharded_loopback.py); the reference has no counterpart (single process, dense N x N).
//
//   gnnx_vertex_weights      w[v] = out-degree + in-degree + row_weight          (cost model of a vertex)
//   gnnx_partition_deal      stable sort by descending weight, snake deal to the ranks, rank-contiguous new ids
//   gnnx_shard_select_edges  the edges a rank owns: (local row, ORIGINAL column), self loops dropped on global ids
//   gnnx_halo_plan_*         remote columns -> sorted halo list, [local | halo] renumbering that leaves the order of a
//                            row's entries (= the summation order) untouched, per-peer receive / send lists
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <memory>
#include <vector>

#include "gnnx_common.h"

using namespace gnnx;

namespace gnnx {
int comm_alltoallv_bytes(gnnx_comm *comm, const void *d_send, const int64_t *send_bytes, void *d_recv, const int64_t *recv_bytes,
                         void *stream);
int comm_alltoall_i64(gnnx_comm *comm, const int64_t *h_send, int64_t *h_recv, void *stream);
}  // namespace gnnx

namespace {

constexpr int T = 256;
inline dim3 grid_for(int64_t n) { return dim3((uint32_t)ceil_div(n > 0 ? n : 1, T)); }
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ void degree_count_kernel(const int32_t *src, const int32_t *dst, int64_t n_edges, int32_t n_nodes, int32_t *w, int32_t *bad)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int32_t r = src[e], c = dst[e];
    if (r < 0 || c < 0 || r >= n_nodes || c >= n_nodes) {
        atomicOr(bad, 1);
        return;
    }
    atomicAdd(&w[r], 1);  // integer adds: order-independent
    atomicAdd(&w[c], 1);
}

__global__ void fill_i32_kernel(int32_t *x, int64_t n, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

__global__ void max_i32_kernel(const int32_t *x, int64_t n, int32_t *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t v = i < n ? x[i] : 0;
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(out, v);
}

// key = wmax - w (ascending stable sort == descending weight, ties by ascending id); value = vertex id
__global__ void deal_keys_kernel(const int32_t *w, int32_t n, const int32_t *wmax, uint32_t *keys, int32_t *ids)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (uint32_t)(*wmax - w[i]);
    ids[i] = i;
}

// position k of the sorted order -> rank: j = k % world in even rounds k / world, world-1-j in odd rounds
__global__ void deal_owner_kernel(const int32_t *sorted_ids, int32_t n, int world, int32_t *owner)
{
    int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int j = k % world, r = k / world;
    owner[sorted_ids[k]] = (r & 1) ? world - 1 - j : j;
}

// one-hot of the owner, `world` planes of n flags (flag[p*n + v] = owner[v] == p): an exclusive scan over the
// concatenation numbers rank 0's vertices 0.., then rank 1's, ... in ascending original id == the new ids
__global__ void owner_planes_kernel(const int32_t *owner, int32_t n, int world, int32_t *flag)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * world) return;
    const int32_t v = (int32_t)(i % n), p = (int32_t)(i / n);
    flag[i] = owner[v] == p ? 1 : 0;
}

__global__ void nid_from_planes_kernel(const int32_t *owner, const int32_t *pos, int32_t n, int32_t *nid)
{
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    nid[v] = pos[(int64_t)owner[v] * n + v];
}

__global__ void select_flags_kernel(const int32_t *src, const int32_t *dst, int64_t n_edges, const int32_t *owner, int rank,
                                    int transpose, int32_t *flag)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int32_t r = src[e], c = dst[e];
    flag[e] = (r != c && owner[transpose ? c : r] == rank) ? 1 : 0;
}

__global__ void select_scatter_kernel(const int32_t *src, const int32_t *dst, int64_t n_edges, const int32_t *flag, const int32_t *pos,
                                      const int32_t *nid, int32_t lo, int transpose, int32_t *rows, int32_t *cols)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges || !flag[e]) return;
    const int32_t r = transpose ? dst[e] : src[e], c = transpose ? src[e] : dst[e];
    const int32_t p = pos[e];
    rows[p] = nid[r] - lo;
    cols[p] = c;
}

// columns to new ids (values only: the order of a row's entries is the original-id order the CSR build gave them);
// remote columns are marked in a table over all new ids
__global__ void map_mark_kernel(const int32_t *colidx_orig, int64_t nnz, const int32_t *nid, int32_t lo, int32_t hi, int32_t *col_nid,
                                int32_t *mark)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int32_t c = nid[colidx_orig[p]];
    col_nid[p] = c;
    if (c < lo || c >= hi) mark[c] = 1;  // same value from every writer
}

__global__ void halo_list_kernel(const int32_t *mark, const int32_t *slot, int32_t n, int32_t *halo)
{
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n && mark[v]) halo[slot[v]] = v;
}

__global__ void renumber_kernel(const int32_t *col_nid, int64_t nnz, const int32_t *slot, int32_t lo, int32_t hi, int32_t n_local,
                                int32_t *colidx_local)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int32_t c = col_nid[p];
    colidx_local[p] = (c < lo || c >= hi) ? n_local + slot[c] : c - lo;
}

__global__ void send_idx_kernel(const int32_t *want_nid, int64_t n, int32_t lo, int32_t n_local, int32_t *send_idx, int32_t *bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = want_nid[i] - lo;
    if (v < 0 || v >= n_local) atomicOr(bad, 1);
    send_idx[i] = v;
}

hipError_t scan_i32(void *tmp, size_t &tmp_bytes, const int32_t *in, int32_t *out, size_t n, hipStream_t st)
{
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0, n, rocprim::plus<int32_t>(), st);
}

struct DevBuf {  // scratch that lives for one build call
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class U> U *as() { return static_cast<U *>(p); }
};

}  // namespace

struct gnnx_halo_plan {
    int world = 1, rank = 0;
    int64_t lo = 0, n_local = 0, n_halo = 0, n_send = -1;
    int32_t *d_halo = nullptr;      // [n_halo] new ids, ascending => grouped by owner
    int32_t *d_send_idx = nullptr;  // [n_send] local row ids, peer-major
    int32_t *d_slots = nullptr;     // [n_local][8] inverse of d_send_idx (gnnx_rows_to_slots_f32), or nullptr (a row with more than 7 slots)
    std::vector<int64_t> recv_rows, send_rows;
    gnnx_halo_plan() = default;
    gnnx_halo_plan(const gnnx_halo_plan &) = delete;
    gnnx_halo_plan &operator=(const gnnx_halo_plan &) = delete;
    ~gnnx_halo_plan()   // owns its device lists: an error path that drops the plan frees them too
    {
        if (d_halo) hipFree(d_halo);
        if (d_send_idx) hipFree(d_send_idx);
        if (d_slots) hipFree(d_slots);
    }
};

GNNX_API int gnnx_vertex_weights(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, int32_t n_nodes, int32_t row_weight,
                                 int32_t *d_weight, void *stream)
{
    GNNX_REQUIRE(n_edges >= 0 && n_nodes >= 0 && row_weight >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_nodes == 0) return GNNX_OK;
    GNNX_REQUIRE(d_weight && (n_edges == 0 || (d_src && d_dst)), GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    DevBuf bad;
    GNNX_HIP_CHECK(bad.alloc(sizeof(int32_t)));
    GNNX_HIP_CHECK(hipMemsetAsync(bad.p, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(fill_i32_kernel, grid_for(n_nodes), dim3(T), 0, st, d_weight, (int64_t)n_nodes, row_weight);
    GNNX_LAUNCH_CHECK();
    if (n_edges) {
        hipLaunchKernelGGL(degree_count_kernel, grid_for(n_edges), dim3(T), 0, st, d_src, d_dst, n_edges, n_nodes, d_weight, bad.as<int32_t>());
        GNNX_LAUNCH_CHECK();
    }
    int32_t h_bad = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE, "invalid input, max value in edge_index should be less than the number of nodes from x");
    return GNNX_OK;
}

GNNX_API int gnnx_partition_deal(const int32_t *d_weight, int32_t n_nodes, int world, int32_t *d_owner, int32_t *d_nid, int64_t *cuts_out,
                                 void *stream)
{
    GNNX_REQUIRE(n_nodes >= 0 && world >= 1 && cuts_out, GNNX_ERR_INVALID_ARG, "bad arguments");
    GNNX_REQUIRE((int64_t)n_nodes * world < (1ll << 31), GNNX_ERR_UNSUPPORTED, "n_nodes * world must be < 2^31");
    hipStream_t st = as_stream(stream);
    cuts_out[0] = 0;
    if (n_nodes == 0) {
        for (int p = 1; p <= world; p++) cuts_out[p] = 0;
        return GNNX_OK;
    }
    GNNX_REQUIRE(d_weight && d_owner && d_nid, GNNX_ERR_INVALID_ARG, "null pointer");
    const size_t n = (size_t)n_nodes, nw = n * (size_t)world;
    size_t sort_bytes = 0, scan_bytes = 0;
    GNNX_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr,
                                             (int32_t *)nullptr, n, 0, 32));
    GNNX_HIP_CHECK(scan_i32(nullptr, scan_bytes, nullptr, nullptr, nw, st));
    DevBuf keys, keys2, ids, ids2, wmax, flag, pos, tmp;
    GNNX_HIP_CHECK(keys.alloc(n * 4));
    GNNX_HIP_CHECK(keys2.alloc(n * 4));
    GNNX_HIP_CHECK(ids.alloc(n * 4));
    GNNX_HIP_CHECK(ids2.alloc(n * 4));
    GNNX_HIP_CHECK(wmax.alloc(4));
    GNNX_HIP_CHECK(flag.alloc(nw * 4));
    GNNX_HIP_CHECK(pos.alloc(nw * 4));
    GNNX_HIP_CHECK(tmp.alloc(sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    GNNX_HIP_CHECK(hipMemsetAsync(wmax.p, 0, 4, st));
    hipLaunchKernelGGL(max_i32_kernel, grid_for(n_nodes), dim3(T), 0, st, d_weight, (int64_t)n_nodes, wmax.as<int32_t>());
    GNNX_LAUNCH_CHECK();
    hipLaunchKernelGGL(deal_keys_kernel, grid_for(n_nodes), dim3(T), 0, st, d_weight, n_nodes, wmax.as<int32_t>(), keys.as<uint32_t>(),
                       ids.as<int32_t>());
    GNNX_LAUNCH_CHECK();
    size_t tb = sort_bytes;
    GNNX_HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, keys.as<uint32_t>(), keys2.as<uint32_t>(), ids.as<int32_t>(), ids2.as<int32_t>(), n,
                                             0, 32, st));  // stable: ties keep ascending id
    hipLaunchKernelGGL(deal_owner_kernel, grid_for(n_nodes), dim3(T), 0, st, ids2.as<int32_t>(), n_nodes, world, d_owner);
    GNNX_LAUNCH_CHECK();
    hipLaunchKernelGGL(owner_planes_kernel, grid_for((int64_t)nw), dim3(T), 0, st, d_owner, n_nodes, world, flag.as<int32_t>());
    GNNX_LAUNCH_CHECK();
    tb = scan_bytes;
    GNNX_HIP_CHECK(scan_i32(tmp.p, tb, flag.as<int32_t>(), pos.as<int32_t>(), nw, st));
    hipLaunchKernelGGL(nid_from_planes_kernel, grid_for(n_nodes), dim3(T), 0, st, d_owner, pos.as<int32_t>(), n_nodes, d_nid);
    GNNX_LAUNCH_CHECK();
    // cuts: the scan value at the start of every plane
    std::vector<int32_t> h((size_t)world);
    for (int p = 0; p < world; p++)
        GNNX_HIP_CHECK(hipMemcpyAsync(&h[p], pos.as<int32_t>() + (size_t)p * n, 4, hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    for (int p = 0; p < world; p++) cuts_out[p] = h[p];
    cuts_out[world] = n_nodes;
    return GNNX_OK;
}

// Spread a rank's rows over its new-id range: position k (ascending original id) of rank p's n_p vertices moves to
// (k * 2654435761) mod n_p.  2654435761 is prime and larger than any n_p, so the map is a bijection of [0, n_p).
namespace {
constexpr uint64_t kScrambleMul = 2654435761ull;
__global__ __launch_bounds__(256) void scramble_nid_kernel(const int32_t *owner, int32_t n, const int64_t *cuts, int32_t *nid)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const int p = owner ? owner[v] : 0;
    const int64_t lo = cuts[p], cnt = cuts[p + 1] - lo;
    const uint64_t k = (uint64_t)((int64_t)nid[v] - lo);
    nid[v] = (int32_t)(lo + (int64_t)((k * kScrambleMul) % (uint64_t)cnt));
}
}  // namespace

GNNX_API int gnnx_partition_scramble(const int32_t *d_owner, int32_t n_nodes, int world, const int64_t *cuts, int32_t *d_nid, void *stream)
{
    GNNX_REQUIRE(n_nodes >= 0 && world >= 1 && cuts && (n_nodes == 0 || d_nid), GNNX_ERR_INVALID_ARG, "bad arguments");
    GNNX_REQUIRE(world == 1 || d_owner, GNNX_ERR_INVALID_ARG, "owner is required for more than one rank");
    if (n_nodes == 0) return GNNX_OK;
    for (int p = 0; p < world; p++)
        GNNX_REQUIRE(cuts[p] <= cuts[p + 1] && cuts[0] == 0 && cuts[world] == n_nodes, GNNX_ERR_INVALID_ARG, "cuts are not a partition of [0, n)");
    hipStream_t st = as_stream(stream);
    DevBuf dcuts;
    GNNX_HIP_CHECK(dcuts.alloc(sizeof(int64_t) * (size_t)(world + 1)));
    GNNX_HIP_CHECK(hipMemcpyAsync(dcuts.p, cuts, sizeof(int64_t) * (size_t)(world + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scramble_nid_kernel, dim3((uint32_t)ceil_div((int64_t)n_nodes, (int64_t)256)), dim3(256), 0, st, world == 1 ? nullptr : d_owner,
                       n_nodes, dcuts.as<int64_t>(), d_nid);
    GNNX_LAUNCH_CHECK();
    GNNX_HIP_CHECK(hipStreamSynchronize(st));   // dcuts is released on return
    return GNNX_OK;
}

GNNX_API int gnnx_shard_select_edges(const int32_t *d_src, const int32_t *d_dst, int64_t n_edges, const int32_t *d_owner,
                                     const int32_t *d_nid, int rank, int64_t lo, int transpose, int32_t *d_rows, int32_t *d_cols,
                                     int64_t *n_selected, void *stream)
{
    GNNX_REQUIRE(n_edges >= 0 && n_selected && rank >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    GNNX_REQUIRE(n_edges < (1ll << 31), GNNX_ERR_UNSUPPORTED, "n_edges must be < 2^31");
    *n_selected = 0;
    if (n_edges == 0) return GNNX_OK;
    GNNX_REQUIRE(d_src && d_dst && d_owner && d_nid && d_rows && d_cols, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    size_t scan_bytes = 0;
    GNNX_HIP_CHECK(scan_i32(nullptr, scan_bytes, nullptr, nullptr, (size_t)n_edges, st));
    DevBuf flag, pos, tmp;
    GNNX_HIP_CHECK(flag.alloc((size_t)n_edges * 4));
    GNNX_HIP_CHECK(pos.alloc((size_t)n_edges * 4));
    GNNX_HIP_CHECK(tmp.alloc(scan_bytes));
    hipLaunchKernelGGL(select_flags_kernel, grid_for(n_edges), dim3(T), 0, st, d_src, d_dst, n_edges, d_owner, rank, transpose,
                       flag.as<int32_t>());
    GNNX_LAUNCH_CHECK();
    GNNX_HIP_CHECK(scan_i32(tmp.p, scan_bytes, flag.as<int32_t>(), pos.as<int32_t>(), (size_t)n_edges, st));
    hipLaunchKernelGGL(select_scatter_kernel, grid_for(n_edges), dim3(T), 0, st, d_src, d_dst, n_edges, flag.as<int32_t>(),
                       pos.as<int32_t>(), d_nid, (int32_t)lo, transpose, d_rows, d_cols);
    GNNX_LAUNCH_CHECK();
    int32_t last[2] = {0, 0};
    GNNX_HIP_CHECK(hipMemcpyAsync(&last[0], pos.as<int32_t>() + (n_edges - 1), 4, hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&last[1], flag.as<int32_t>() + (n_edges - 1), 4, hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    *n_selected = (int64_t)last[0] + last[1];
    return GNNX_OK;
}

GNNX_API int gnnx_halo_plan_create(const int32_t *d_colidx_orig, int64_t nnz, const int32_t *d_nid, int32_t n_nodes, int world, int rank,
                                   const int64_t *cuts, int32_t *d_colidx_local, gnnx_halo_plan **plan_out, void *stream)
{
    GNNX_REQUIRE(plan_out && cuts && world >= 1 && rank >= 0 && rank < world && nnz >= 0 && n_nodes >= 0, GNNX_ERR_INVALID_ARG,
                 "bad arguments");
    GNNX_REQUIRE(cuts[0] == 0 && cuts[world] == n_nodes, GNNX_ERR_INVALID_ARG, "cuts must run from 0 to n_nodes");
    for (int p = 0; p < world; p++) GNNX_REQUIRE(cuts[p] <= cuts[p + 1], GNNX_ERR_INVALID_ARG, "cuts must be monotone");
    GNNX_REQUIRE(nnz == 0 || (d_colidx_orig && d_nid && d_colidx_local), GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    auto plan = std::make_unique<gnnx_halo_plan>();
    plan->world = world;
    plan->rank = rank;
    plan->lo = cuts[rank];
    plan->n_local = cuts[rank + 1] - cuts[rank];
    plan->recv_rows.assign((size_t)world, 0);
    const int32_t lo = (int32_t)cuts[rank], hi = (int32_t)cuts[rank + 1];
    const size_t n1 = (size_t)n_nodes + 1;
    size_t scan_bytes = 0;
    GNNX_HIP_CHECK(scan_i32(nullptr, scan_bytes, nullptr, nullptr, n1, st));
    DevBuf mark, slot, col_nid, tmp;
    GNNX_HIP_CHECK(mark.alloc(n1 * 4));
    GNNX_HIP_CHECK(slot.alloc(n1 * 4));
    GNNX_HIP_CHECK(col_nid.alloc((size_t)nnz * 4));
    GNNX_HIP_CHECK(tmp.alloc(scan_bytes));
    GNNX_HIP_CHECK(hipMemsetAsync(mark.p, 0, n1 * 4, st));
    if (nnz) {
        hipLaunchKernelGGL(map_mark_kernel, grid_for(nnz), dim3(T), 0, st, d_colidx_orig, nnz, d_nid, lo, hi, col_nid.as<int32_t>(),
                           mark.as<int32_t>());
        GNNX_LAUNCH_CHECK();
    }
    GNNX_HIP_CHECK(scan_i32(tmp.p, scan_bytes, mark.as<int32_t>(), slot.as<int32_t>(), n1, st));  // slot[n_nodes] = n_halo
    std::vector<int32_t> at((size_t)world + 1);
    for (int p = 0; p <= world; p++)
        GNNX_HIP_CHECK(hipMemcpyAsync(&at[p], slot.as<int32_t>() + cuts[p], 4, hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    plan->n_halo = at[world];
    for (int p = 0; p < world; p++) plan->recv_rows[p] = at[p + 1] - at[p];
    GNNX_HIP_CHECK(hipMalloc((void **)&plan->d_halo, sizeof(int32_t) * (size_t)(plan->n_halo ? plan->n_halo : 1)));
    if (n_nodes) {
        hipLaunchKernelGGL(halo_list_kernel, grid_for(n_nodes), dim3(T), 0, st, mark.as<int32_t>(), slot.as<int32_t>(), n_nodes, plan->d_halo);
        GNNX_LAUNCH_CHECK();
    }
    if (nnz) {
        hipLaunchKernelGGL(renumber_kernel, grid_for(nnz), dim3(T), 0, st, col_nid.as<int32_t>(), nnz, slot.as<int32_t>(), lo, hi,
                           (int32_t)plan->n_local, d_colidx_local);
        GNNX_LAUNCH_CHECK();
    }
    GNNX_HIP_CHECK(hipStreamSynchronize(st));  // the scratch buffers are freed on return
    if (world == 1) {
        plan->send_rows.assign(1, 0);
        plan->n_send = 0;
    }
    *plan_out = plan.release();
    return GNNX_OK;
}

GNNX_API int gnnx_halo_plan_destroy(gnnx_halo_plan *plan)
{
    delete plan;
    return GNNX_OK;
}

GNNX_API int gnnx_halo_plan_info(const gnnx_halo_plan *plan, int64_t *n_local, int64_t *n_halo, int64_t *n_send, int64_t *recv_rows,
                                 int64_t *send_rows, const int32_t **d_halo_ids, const int32_t **d_send_idx)
{
    GNNX_REQUIRE(plan, GNNX_ERR_INVALID_ARG, "plan is null");
    if (n_local) *n_local = plan->n_local;
    if (n_halo) *n_halo = plan->n_halo;
    if (n_send) *n_send = plan->n_send;
    if (recv_rows)
        for (int p = 0; p < plan->world; p++) recv_rows[p] = plan->recv_rows[p];
    if (send_rows) {
        GNNX_REQUIRE(plan->n_send >= 0, GNNX_ERR_INVALID_ARG, "the send list has not been set yet");
        for (int p = 0; p < plan->world; p++) send_rows[p] = plan->send_rows[p];
    }
    if (d_halo_ids) *d_halo_ids = plan->d_halo;
    if (d_send_idx) *d_send_idx = plan->d_send_idx;
    return GNNX_OK;
}

GNNX_API int gnnx_halo_plan_set_send_list(gnnx_halo_plan *plan, const int32_t *d_want_new_ids, const int64_t *send_rows, void *stream)
{
    GNNX_REQUIRE(plan && send_rows, GNNX_ERR_INVALID_ARG, "bad arguments");
    int64_t total = 0;
    for (int p = 0; p < plan->world; p++) {
        GNNX_REQUIRE(send_rows[p] >= 0, GNNX_ERR_INVALID_ARG, "negative row count");
        total += send_rows[p];
    }
    GNNX_REQUIRE(send_rows[plan->rank] == 0, GNNX_ERR_INVALID_ARG, "a rank sends nothing to itself");
    GNNX_REQUIRE(total == 0 || d_want_new_ids, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    if (plan->d_send_idx) hipFree(plan->d_send_idx);
    plan->d_send_idx = nullptr;
    plan->n_send = -1;
    GNNX_HIP_CHECK(hipMalloc((void **)&plan->d_send_idx, sizeof(int32_t) * (size_t)(total ? total : 1)));
    if (total) {
        DevBuf bad;
        GNNX_HIP_CHECK(bad.alloc(4));
        GNNX_HIP_CHECK(hipMemsetAsync(bad.p, 0, 4, st));
        hipLaunchKernelGGL(send_idx_kernel, grid_for(total), dim3(T), 0, st, d_want_new_ids, total, (int32_t)plan->lo, (int32_t)plan->n_local,
                           plan->d_send_idx, bad.as<int32_t>());
        GNNX_LAUNCH_CHECK();
        int32_t h_bad = 0;
        GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE, "a peer requested a row this rank does not own");
    }
    plan->send_rows.assign(send_rows, send_rows + plan->world);
    plan->n_send = total;
    // the inverse of the send list for the pack from the producer's side (gnnx_rows_to_slots_f32): row -> its slots, ascending, packed
    // to the front of 8 entries.  Built on the host, once per plan; a row that more than 7 peers want keeps the gather pack.
    if (plan->d_slots) hipFree(plan->d_slots);
    plan->d_slots = nullptr;
    if (total > 0 && plan->n_local > 0) {
        constexpr int kSlots = 8;
        std::vector<int32_t> h_idx((size_t)total);
        GNNX_HIP_CHECK(hipMemcpyAsync(h_idx.data(), plan->d_send_idx, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<int32_t> table((size_t)plan->n_local * kSlots, -1);
        std::vector<uint8_t> fill((size_t)plan->n_local, 0);
        bool fits = true;
        for (int64_t slot = 0; slot < total && fits; slot++) {   // ascending slot: a row's entries come out ascending
            const int32_t r = h_idx[(size_t)slot];
            if (fill[(size_t)r] >= kSlots - 1) fits = false;
            else table[(size_t)r * kSlots + fill[(size_t)r]++] = (int32_t)slot;
        }
        if (fits) {
            GNNX_HIP_CHECK(hipMalloc((void **)&plan->d_slots, sizeof(int32_t) * table.size()));
            GNNX_HIP_CHECK(hipMemcpyAsync(plan->d_slots, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice, st));
            GNNX_HIP_CHECK(hipStreamSynchronize(st));   // `table` is pageable host memory: read by the copy until here
        }
    }
    return GNNX_OK;
}

GNNX_API int gnnx_halo_plan_exchange_requests(gnnx_halo_plan *plan, gnnx_comm *comm, void *stream)
{
    GNNX_REQUIRE(plan && comm, GNNX_ERR_INVALID_ARG, "bad arguments");
    int world = 0, rank = 0;
    gnnx_comm_info(comm, &world, &rank);
    GNNX_REQUIRE(world == plan->world && rank == plan->rank, GNNX_ERR_INVALID_ARG, "the communicator does not match the plan");
    std::vector<int64_t> sc((size_t)world), rb((size_t)world), sb((size_t)world);
    int st = comm_alltoall_i64(comm, plan->recv_rows.data(), sc.data(), stream);  // what I receive from p is what p sends me
    if (st != GNNX_OK) return st;
    int64_t total = 0;
    for (int p = 0; p < world; p++) {
        rb[p] = plan->recv_rows[p] * 4;
        sb[p] = sc[p] * 4;
        total += sc[p];
    }
    DevBuf want;
    GNNX_HIP_CHECK(want.alloc((size_t)total * 4));
    // my halo list (grouped by owner) goes out as the requests; the requests of the peers come in
    st = comm_alltoallv_bytes(comm, plan->d_halo, rb.data(), want.p, sb.data(), stream);
    if (st != GNNX_OK) return st;
    return gnnx_halo_plan_set_send_list(plan, want.as<int32_t>(), sc.data(), stream);
}

GNNX_API int gnnx_halo_exchange_rows_f32(const gnnx_halo_plan *plan, gnnx_comm *comm, float *d_buf, int64_t ldb, int32_t n_feat,
                                         float *d_send_buf, void *stream)
{
    GNNX_REQUIRE(plan && comm && n_feat >= 0 && ldb >= n_feat, GNNX_ERR_INVALID_ARG, "bad arguments");
    GNNX_REQUIRE(plan->n_send >= 0, GNNX_ERR_INVALID_ARG, "the send list has not been set yet");
    GNNX_REQUIRE(ldb == n_feat || plan->n_halo == 0, GNNX_ERR_UNSUPPORTED, "the halo tail must be densely packed (ldb == n_feat)");
    if (plan->world == 1) return GNNX_OK;
    GNNX_REQUIRE(d_buf && (plan->n_send == 0 || d_send_buf), GNNX_ERR_INVALID_ARG, "null buffer");
    if (plan->n_send) {
        // rows of 16-byte pieces: the pack from the producer's side (every local row read once); else the gather by the send list
        const bool vec = plan->d_slots && n_feat % 4 == 0 && ldb % 4 == 0 && n_feat / 4 <= 256 && 256 % (n_feat / 4) == 0 &&
                         (reinterpret_cast<uintptr_t>(d_buf) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_send_buf) & 15u) == 0;
        int st = vec ? gnnx_rows_to_slots_f32(d_buf, ldb, plan->n_local, n_feat, plan->d_slots, d_send_buf, n_feat, nullptr, 0.f, nullptr, 0, stream)
                     : gnnx_gather_rows_f32(d_buf, ldb, plan->d_send_idx, plan->n_send, n_feat, d_send_buf, n_feat, stream);
        if (st != GNNX_OK) return st;
    }
    return gnnx_halo_exchange_f32(comm, d_send_buf, plan->send_rows.data(), d_buf + plan->n_local * ldb, plan->recv_rows.data(), n_feat,
                                  stream);
}

// The plan's slot table (gnnx_rows_to_slots_f32's [n_local][8] inverse of the send list; NULL: a world of more than 8 ranks, or no send
// list yet) for producers that pack while they produce (gnnx_gemm_nt_rows_to_slots_f32), and the exchange of a send buffer such a
// producer has filled.
GNNX_API int gnnx_halo_plan_slot_table(const gnnx_halo_plan *plan, const int32_t **d_slots)
{
    GNNX_REQUIRE(plan && d_slots, GNNX_ERR_INVALID_ARG, "bad arguments");
    *d_slots = plan->n_send > 0 ? plan->d_slots : nullptr;
    return GNNX_OK;
}

GNNX_API int gnnx_halo_exchange_packed_f32(const gnnx_halo_plan *plan, gnnx_comm *comm, float *d_buf, int64_t ldb, int32_t n_feat,
                                           const float *d_send_buf, void *stream)
{
    GNNX_REQUIRE(plan && comm && n_feat >= 0 && ldb >= n_feat, GNNX_ERR_INVALID_ARG, "bad arguments");
    GNNX_REQUIRE(plan->n_send >= 0, GNNX_ERR_INVALID_ARG, "the send list has not been set yet");
    GNNX_REQUIRE(ldb == n_feat || plan->n_halo == 0, GNNX_ERR_UNSUPPORTED, "the halo tail must be densely packed (ldb == n_feat)");
    if (plan->world == 1) return GNNX_OK;
    GNNX_REQUIRE(d_buf && (plan->n_send == 0 || d_send_buf), GNNX_ERR_INVALID_ARG, "null buffer");
    return gnnx_halo_exchange_f32(comm, d_send_buf, plan->send_rows.data(), d_buf + plan->n_local * ldb, plan->recv_rows.data(), n_feat,
                                  stream);
}
