// Small HBM-bound ops on the GCN path: dbias column sum, unfused row-scale / bias / axpy, and the halo
// pack / unpack (row gather, row scatter-add).  All are streaming kernels: 16 B per lane where alignment
// allows, grid capped at 2048 workgroups with a grid-stride loop.
#include <cmath>
#include <vector>

#include "gnnx_common.h"

// Parity depends on separately rounded fp32 mul / add (the reference has no FMA): never contract.
#pragma clang fp contract(off)

using namespace gnnx;

namespace {

constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 blocks/CU

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- colsum: stage 1 -- each workgroup reduces a contiguous slab of rows to one partial row ---------
// Vector form (F % 4 == 0, 16-B aligned): a row is covered by L = F/4 lanes (16 B each, coalesced); the 256
// threads form 256/L row groups that walk the slab interleaved, 4 rows in flight per thread; the groups'
// partials are combined through LDS in group order.  Fixed grid and fixed order => deterministic.
// COPY: every row that is read is also written to `copy` (row stride ldc) -- gnnx_colsum_copy_f32: the upstream gradient on the
// padded row stride the backward aggregation gathers from, made by the pass that reads it anyway.
template <int UNROLL, bool COPY = false>
__global__ __launch_bounds__(256) void colsum_stage1_vec(const float *G, int64_t ldg, int64_t n_rows, int32_t n_feat,
                                                          int64_t rows_per_range, int ranges_per_block, float *partial, float *copy = nullptr,
                                                          int64_t ldc = 0)
{
    // partial[v] = sums of row range v (rows_per_range consecutive rows); a workgroup takes ranges_per_block consecutive ranges one
    // after the other, so the partials -- and with them every bit of the result -- do not depend on how many workgroups are launched
    __shared__ float4 red[256];
    const int L = n_feat / 4;            // lanes per row (<= 256)
    const int groups = 256 / L;          // row groups per pass
    const int li = threadIdx.x % L, grp = threadIdx.x / L;
    for (int v = 0; v < ranges_per_block; v++) {
        const int64_t range = (int64_t)blockIdx.x * ranges_per_block + v;
        int64_t r0 = range * rows_per_range;
        int64_t r1 = r0 + rows_per_range < n_rows ? r0 + rows_per_range : n_rows;
        for (int f0 = 0; f0 < n_feat; f0 += 4 * L) {  // one iteration unless F > 1024
            float4 acc[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grp < groups && f0 + 4 * li < n_feat) {
                const float *base = G + f0 + 4 * li;
                int64_t r = r0 + grp;
                for (; r + (int64_t)(UNROLL - 1) * groups < r1; r += (int64_t)UNROLL * groups) {
                    float4 w[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) w[u] = *reinterpret_cast<const float4 *>(base + (r + (int64_t)u * groups) * ldg);
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        acc[u].x += w[u].x; acc[u].y += w[u].y; acc[u].z += w[u].z; acc[u].w += w[u].w;
                        if constexpr (COPY) *reinterpret_cast<float4 *>(copy + f0 + 4 * li + (r + (int64_t)u * groups) * ldc) = w[u];
                    }
                }
                for (; r < r1; r += groups) {
                    float4 w = *reinterpret_cast<const float4 *>(base + r * ldg);
                    acc[0].x += w.x; acc[0].y += w.y; acc[0].z += w.z; acc[0].w += w.w;
                    if constexpr (COPY) *reinterpret_cast<float4 *>(copy + f0 + 4 * li + r * ldc) = w;
                }
            }
            float4 t = acc[0];
#pragma unroll
            for (int u = 1; u < UNROLL; u++) { t.x += acc[u].x; t.y += acc[u].y; t.z += acc[u].z; t.w += acc[u].w; }
            red[threadIdx.x] = t;
            __syncthreads();
            if (grp == 0 && f0 + 4 * li < n_feat) {
                float4 s4 = red[li];
                for (int k = 1; k < groups; k++) {
                    float4 o = red[k * L + li];
                    s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
                }
                *reinterpret_cast<float4 *>(partial + range * n_feat + f0 + 4 * li) = s4;
            }
            __syncthreads();
        }
    }
}

// ---- halo pack from the PRODUCER's side: rows -> send slots (+ the column sums of the same pass) ------------------------------
// The halo pack of the sharded step used to be a gather driven by the send list (slot -> row): every row a peer needs is read once per
// peer.  Here the rows are walked in order, each read ONCE, and written to every slot that wants it: slots[row][kSlotsPerRow] holds the
// row's positions in the send buffer (-1: none; a row goes to at most world - 1 <= 7 peers) -- one 32-byte load per row, issued with
// the row itself, no dependent index chain.  SUM: the pass also produces the column sums of ALL rows, with the partials of
// colsum_stage1_vec (fixed row ranges, group order): the same bits as gnnx_colsum_f32 -- dbias of the layer and the pack of the
// upstream gradient in one read of G.
constexpr int kSlotsPerRow = 8;
template <int UNROLL, bool SUM>
__global__ __launch_bounds__(256) void rows_to_slots_vec(const float *X, int64_t ldx, int64_t n_rows, int32_t n_feat, const int32_t *slots,
                                                         float *send, int64_t lds, int64_t rows_per_range, int ranges_per_block, float *partial)
{
    __shared__ float4 red[SUM ? 256 : 1];
    const int L = n_feat / 4;            // lanes per row (<= 256)
    const int groups = 256 / L;          // row groups per pass
    const int li = threadIdx.x % L, grp = threadIdx.x / L;
    auto scatter = [&](const float4 &w, const int4 &s0, const int4 &s1, int f) {
        const int sl[kSlotsPerRow] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int t = 0; t < kSlotsPerRow; t++) {
            if (sl[t] < 0) break;   // slots of a row are packed to the front
            *reinterpret_cast<float4 *>(send + (int64_t)sl[t] * lds + f) = w;
        }
    };
    for (int v = 0; v < ranges_per_block; v++) {
        const int64_t range = (int64_t)blockIdx.x * ranges_per_block + v;
        int64_t r0 = range * rows_per_range;
        int64_t r1 = r0 + rows_per_range < n_rows ? r0 + rows_per_range : n_rows;
        for (int f0 = 0; f0 < n_feat; f0 += 4 * L) {  // one iteration unless F > 1024
            float4 acc[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grp < groups && f0 + 4 * li < n_feat) {
                const float *base = X + f0 + 4 * li;
                int64_t r = r0 + grp;
                for (; r + (int64_t)(UNROLL - 1) * groups < r1; r += (int64_t)UNROLL * groups) {
                    float4 w[UNROLL];
                    int4 sa[UNROLL], sb[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        const int64_t row = r + (int64_t)u * groups;
                        sa[u] = *reinterpret_cast<const int4 *>(slots + row * kSlotsPerRow);
                        sb[u] = *reinterpret_cast<const int4 *>(slots + row * kSlotsPerRow + 4);
                        if (SUM || sa[u].x >= 0) w[u] = *reinterpret_cast<const float4 *>(base + row * ldx);   // (without the sums a row nobody wants is not read)
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; u++) {
                        if constexpr (SUM) { acc[u].x += w[u].x; acc[u].y += w[u].y; acc[u].z += w[u].z; acc[u].w += w[u].w; }
                        scatter(w[u], sa[u], sb[u], f0 + 4 * li);
                    }
                }
                for (; r < r1; r += groups) {
                    const int4 sa = *reinterpret_cast<const int4 *>(slots + r * kSlotsPerRow);
                    const int4 sb = *reinterpret_cast<const int4 *>(slots + r * kSlotsPerRow + 4);
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (SUM || sa.x >= 0) w = *reinterpret_cast<const float4 *>(base + r * ldx);
                    if constexpr (SUM) { acc[0].x += w.x; acc[0].y += w.y; acc[0].z += w.z; acc[0].w += w.w; }
                    scatter(w, sa, sb, f0 + 4 * li);
                }
            }
            if constexpr (SUM) {
                float4 t = acc[0];
#pragma unroll
                for (int u = 1; u < UNROLL; u++) { t.x += acc[u].x; t.y += acc[u].y; t.z += acc[u].z; t.w += acc[u].w; }
                red[threadIdx.x] = t;
                __syncthreads();
                if (grp == 0 && f0 + 4 * li < n_feat) {
                    float4 s4 = red[li];
                    for (int k = 1; k < groups; k++) {
                        float4 o = red[k * L + li];
                        s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
                    }
                    *reinterpret_cast<float4 *>(partial + range * n_feat + f0 + 4 * li) = s4;
                }
                __syncthreads();
            }
        }
    }
}

// Scalar form: thread t owns feature columns {t, t+256, ...}
__global__ __launch_bounds__(256) void colsum_stage1(const float *G, int64_t ldg, int64_t n_rows, int32_t n_feat,
                                                      int64_t rows_per_block, float *partial)
{
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    for (int32_t f = threadIdx.x; f < n_feat; f += 256) {
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            acc0 += G[r * ldg + f];
            acc1 += G[(r + 1) * ldg + f];
            acc2 += G[(r + 2) * ldg + f];
            acc3 += G[(r + 3) * ldg + f];
        }
        for (; r < r1; r++) acc0 += G[r * ldg + f];
        partial[(int64_t)blockIdx.x * n_feat + f] = (acc0 + acc1) + (acc2 + acc3);
    }
}

// For narrow matrices (n_feat < 256) pack several rows per workgroup pass: thread t -> (row t / fw, col t % fw)
__global__ __launch_bounds__(256) void colsum_stage1_narrow(const float *G, int64_t ldg, int64_t n_rows, int32_t n_feat,
                                                             int32_t fw, int64_t rows_per_block, float *partial)
{
    __shared__ float red[256];
    const int rpp = 256 / fw;  // rows per pass
    const int f = threadIdx.x % fw, rr = threadIdx.x / fw;
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    float acc = 0.f;
    if (f < n_feat && rr < rpp)
        for (int64_t r = r0 + rr; r < r1; r += rpp) acc += G[r * ldg + f];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < fw && threadIdx.x < n_feat) {
        float s = 0.f;
        for (int k = 0; k < rpp; k++) s += red[k * fw + threadIdx.x];
        partial[(int64_t)blockIdx.x * n_feat + threadIdx.x] = s;
    }
}

// stage 2: out[f] = beta*out[f] + sum_b partial[b][f].  64 columns per workgroup, 4 interleaved row parts per column
// (each walks every 4th partial row with 4 loads in flight), combined in part order through LDS: deterministic.
__global__ __launch_bounds__(256) void colsum_stage2(const float *partial, int32_t n_blocks, int32_t n_feat, float beta, float *out)
{
    __shared__ float red[256];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int32_t f = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (f < n_feat) {
        int32_t b = part;
        for (; b + 12 < n_blocks; b += 16) {
            a0 += partial[(int64_t)b * n_feat + f];
            a1 += partial[(int64_t)(b + 4) * n_feat + f];
            a2 += partial[(int64_t)(b + 8) * n_feat + f];
            a3 += partial[(int64_t)(b + 12) * n_feat + f];
        }
        for (; b < n_blocks; b += 4) a0 += partial[(int64_t)b * n_feat + f];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && f < n_feat) {
        float acc = ((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c];
        out[f] = beta != 0.f ? out[f] + acc : acc;
    }
}

int colsum_blocks(int64_t n_rows)
{
    // two resident workgroups per CU, each streaming ONE contiguous range of rows: 10 M x 256 sums in 1.65 ms (6.2 TB/s; 1.73 with
    // 2048 workgroups, 1.91 with 8192), sums + copy in 3.46 ms (5.9 TB/s read + written; 3.98 / 4.08) -- scripts/exp_colsum_copy.py
    int64_t b = ceil_div(n_rows, 64);
    static const int cap_env = [] { const char *e = experiment_env("GNNX_COLSUM_BLOCKS"); return e ? atoi(e) : 0; }();   // A/B
    const int64_t cap = cap_env > 0 ? cap_env : 2 * kNumCU;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- row-wise elementwise -------------------------------------------------------------------------
template <int OP>  // 0: Y = X * v[row]   1: Y = X + b[col]
__global__ __launch_bounds__(256) void rowwise_kernel(const float *X, int64_t ldx, const float *v, int64_t n_rows,
                                                       int32_t n_feat, float *Y, int64_t ldy)
{
    int64_t total = n_rows * n_feat;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i / n_feat;
        int32_t f = (int32_t)(i - r * n_feat);
        float x = X[r * ldx + f];
        Y[r * ldy + f] = OP == 0 ? __fmul_rn(x, v[r]) : __fadd_rn(x, v[f]);
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(int64_t n, float a, const float *x, float *y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = __fadd_rn(y[i], __fmul_rn(a, x[i]));
}

__global__ __launch_bounds__(256) void fill_kernel(float *x, int64_t n, float v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

__global__ __launch_bounds__(256) void pow_kernel(const float *x, int64_t n, float e, float *y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = e == -0.5f ? (float)(1.0 / sqrt(x[i] == 0.f ? 0.0 : (double)x[i])) : powf(x[i], e);   // pow(-0, -0.5) = +inf (C99), not 1 / sqrt(-0)
}

// info[0] = max over i of x[i] as an integer, info[1] = 1 when some x[i] is not a non-negative integer <= kPowTableMax
constexpr int32_t kPowTableMax = 1 << 24;
__global__ __launch_bounds__(256) void pow_scan_kernel(const float *x, int64_t n, int32_t *info)
{
    int32_t mx = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        if (!(v >= 0.f && v <= (float)kPowTableMax && v == truncf(v)) || (v == 0.f && signbit(v))) bad = 1;   // pow(-0, e < 0) = -inf: not the table's +inf
        else mx = (int32_t)v > mx ? (int32_t)v : mx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
        bad |= __shfl_xor(bad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx > 0) atomicMax(&info[0], mx);
        if (bad) atomicOr(&info[1], 1);
    }
}

__global__ __launch_bounds__(256) void pow_table_kernel(const float *x, int64_t n, const float *table, float *y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = table[(int32_t)x[i]];
}

__global__ __launch_bounds__(256) void csr_rowsum_kernel(const int32_t *rowptr, const float *vals, int32_t n, float *out)
{
    int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t b = rowptr[i], e = rowptr[i + 1];
    if (!vals) {
        out[i] = (float)(e - b);  // ascending sum of 1.0f's, exact below 2^24
    } else {
        float acc = 0.f;
        for (int32_t p = b; p < e; p++) acc = __fadd_rn(acc, vals[p]);  // functional::sum walks UP
        out[i] = acc;
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const float *X, int64_t ldx, int64_t R, int64_t Cc, float *Y, int64_t ldy)
{
    __shared__ float tile[32][33];
    int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < Cc) tile[k][tx] = X[(r0 + k) * ldx + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < Cc && r0 + tx < R) Y[(c0 + k) * ldy + r0 + tx] = tile[tx][k];
}


// General 2-D broadcast elementwise op (reference functional.h:163-239 add/mul/div over utils.h:181-228 broadcast):
// Y[r][c] = A[r*ars + c*acs] (op) B[r*brs + c*bcs]; a stride of 0 broadcasts that dimension.  One rounding per element.
template <int OP>
__global__ __launch_bounds__(256) void binary_bcast_kernel(const float *A, int64_t ars, int64_t acs, const float *B, int64_t brs,
                                                            int64_t bcs, int64_t n_rows, int64_t n_cols, float *Y, int64_t ldy)
{
    int64_t total = n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i / n_cols, c = i - r * n_cols;
        float a = A[r * ars + c * acs], b = B[r * brs + c * bcs];
        float y = OP == 0 ? __fadd_rn(a, b) : OP == 1 ? __fsub_rn(a, b) : OP == 2 ? __fmul_rn(a, b) : __fdiv_rn(a, b);
        Y[r * ldy + c] = y;
    }
}

// The same, 4 columns (16 bytes) per thread: every operand is either column-contiguous (col stride 1, rows and base 16-byte
// aligned) or column-broadcast (col stride 0: one value per row, or one value in all).  Same single rounding per element.
template <int OP>
__global__ __launch_bounds__(256) void binary_bcast_vec_kernel(const float *A, int64_t ars, int acs, const float *B, int64_t brs, int bcs,
                                                                int64_t n_rows, int32_t quads, float *Y, int64_t ldy)
{
    const int64_t total = n_rows * quads;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / quads;
        const int32_t c = 4 * (int32_t)(i - r * quads);
        float4 a, b, y;
        if (acs) a = *reinterpret_cast<const float4 *>(A + r * ars + c);
        else { const float t = A[r * ars]; a = make_float4(t, t, t, t); }
        if (bcs) b = *reinterpret_cast<const float4 *>(B + r * brs + c);
        else { const float t = B[r * brs]; b = make_float4(t, t, t, t); }
#define GNNX_B1(u) y.u = OP == 0 ? __fadd_rn(a.u, b.u) : OP == 1 ? __fsub_rn(a.u, b.u) : OP == 2 ? __fmul_rn(a.u, b.u) : __fdiv_rn(a.u, b.u)
        GNNX_B1(x); GNNX_B1(y); GNNX_B1(z); GNNX_B1(w);
#undef GNNX_B1
        *reinterpret_cast<float4 *>(Y + r * ldy + c) = y;
    }
}

// out[r] = sum_c X[r][c], ascending c (functional::sum walks UP, reference functional.h:267-296); one wavefront per row
template <bool VEC>
__global__ __launch_bounds__(256) void rowsum_kernel(const float *X, int64_t ldx, int64_t n_rows, int32_t n_cols, float *out)
{
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    // lane l owns the contiguous slice [l*w, (l+1)*w): partial sums combine left to right, so a row of <= 64 columns
    // is summed in exactly the reference's order.  VEC (w % 4 == 0, 16-byte aligned rows): the slice is read 16 bytes at a time
    // and added in the same ascending order.
    const int32_t w = (n_cols + 63) / 64;
    float acc = 0.f;
    if (VEC) {
        for (int32_t c = lane * w; c < (lane + 1) * w && c < n_cols; c += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(X + r * ldx + c);
            acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, v.x), v.y), v.z), v.w);
        }
    } else {
        for (int32_t c = lane * w; c < (lane + 1) * w && c < n_cols; c++) acc = __fadd_rn(acc, X[r * ldx + c]);
    }
    __shared__ float part[4][64];
    part[threadIdx.x >> 6][lane] = acc;
    __syncthreads();
    if (lane == 0) {
        float s = 0.f;
        for (int l = 0; l < 64; l++) s = __fadd_rn(s, part[threadIdx.x >> 6][l]);
        out[r] = s;
    }
}


// f32 -> bf16 (round to nearest even; a plain cast so that hipcc emits v_cvt_pk_bf16_f32, which keeps NaNs NaNs): the opt-in
// bf16 feature storage in front of gnnx_spmm_csr_bf16_f32.  4 elements per thread.
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *X, int64_t ldx, int64_t n_rows, int32_t n_cols, uint16_t *Y,
                                                       int64_t ldy)
{
    const int32_t quads = n_cols / 4;
    const int64_t total = n_rows * quads;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / quads;
        const int32_t q = (int32_t)(i - r * quads);
        const float4 v = *reinterpret_cast<const float4 *>(X + r * ldx + 4 * q);
        union { __bf16 h[4]; uint2 u; } o;
        o.h[0] = (__bf16)v.x;
        o.h[1] = (__bf16)v.y;
        o.h[2] = (__bf16)v.z;
        o.h[3] = (__bf16)v.w;
        *reinterpret_cast<uint2 *>(Y + r * ldy + 4 * q) = o.u;
    }
}
__global__ __launch_bounds__(256) void to_bf16_scalar_kernel(const float *X, int64_t ldx, int64_t n_rows, int32_t n_cols, uint16_t *Y,
                                                              int64_t ldy)
{
    const int64_t total = n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / n_cols;
        const int32_t c = (int32_t)(i - r * n_cols);
        union { __bf16 h; uint16_t u; } o;
        o.h = (__bf16)X[r * ldx + c];
        Y[r * ldy + c] = o.u;
    }
}

// ---- halo pack / unpack: one G-lane group per row, 16 B per lane ---------------------------------------
template <int VEC, bool SCATTER_ADD>
__global__ __launch_bounds__(256) void rows_kernel(const float *in, int64_t ldi, const int32_t *idx, int64_t n_idx,
                                                    int32_t n_feat, float *out, int64_t ldo, int lanes_per_row)
{
    const int rows_per_block = 256 / lanes_per_row;
    const int li = threadIdx.x % lanes_per_row, rr = threadIdx.x / lanes_per_row;
    for (int64_t k = (int64_t)blockIdx.x * rows_per_block + rr; k < n_idx; k += (int64_t)gridDim.x * rows_per_block) {
        int64_t src_row = SCATTER_ADD ? k : idx[k];
        int64_t dst_row = SCATTER_ADD ? idx[k] : k;
        for (int32_t f = li * VEC; f < n_feat; f += lanes_per_row * VEC) {
            if constexpr (VEC == 4) {
                float4 v = *reinterpret_cast<const float4 *>(in + src_row * ldi + f);
                float4 *d = reinterpret_cast<float4 *>(out + dst_row * ldo + f);
                if constexpr (SCATTER_ADD) {
                    float4 o = *d;
                    v = make_float4(__fadd_rn(o.x, v.x), __fadd_rn(o.y, v.y), __fadd_rn(o.z, v.z), __fadd_rn(o.w, v.w));
                }
                *d = v;
            } else {
                float v = in[src_row * ldi + f];
                float *d = out + dst_row * ldo + f;
                *d = SCATTER_ADD ? __fadd_rn(*d, v) : v;
            }
        }
    }
}

int pick_lanes(int32_t n_feat, int vec)
{
    int need = (n_feat + vec - 1) / vec;
    int l = 4;
    while (l < 64 && l < need) l <<= 1;
    return l;
}

template <bool SCATTER_ADD>
int launch_rows(const float *in, int64_t ldi, const int32_t *idx, int64_t n_idx, int32_t n_feat, float *out, int64_t ldo,
                hipStream_t st)
{
    if (n_idx == 0 || n_feat == 0) return GNNX_OK;
    const bool vec4 = n_feat % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0 && aligned16(in) && aligned16(out);
    int lanes = pick_lanes(n_feat, vec4 ? 4 : 1);
    int rows_per_block = 256 / lanes;
    int64_t blocks = ceil_div(n_idx, rows_per_block);
    if (blocks > 16 * kMaxBlocks) blocks = 16 * kMaxBlocks;
    if (vec4)
        hipLaunchKernelGGL((rows_kernel<4, SCATTER_ADD>), dim3((uint32_t)blocks), dim3(256), 0, st, in, ldi, idx, n_idx,
                           n_feat, out, ldo, lanes);
    else
        hipLaunchKernelGGL((rows_kernel<1, SCATTER_ADD>), dim3((uint32_t)blocks), dim3(256), 0, st, in, ldi, idx, n_idx,
                           n_feat, out, ldo, lanes);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

}  // namespace

GNNX_API int gnnx_colsum_workspace(int64_t n_rows, int32_t n_feat, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(float) * (size_t)colsum_blocks(n_rows) * (size_t)(n_feat > 0 ? n_feat : 1);
    return GNNX_OK;
}

namespace {
int colsum_impl(const float *d_G, int64_t ldg, int64_t n_rows, int32_t n_feat, float beta, float *d_out, float *d_copy, int64_t ldc,
                void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_out, GNNX_ERR_INVALID_ARG, "out is null");
    hipStream_t st = as_stream(stream);
    int nb = colsum_blocks(n_rows);
    size_t need = sizeof(float) * (size_t)nb * n_feat;
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu",
                 workspace_bytes, need);
    GNNX_REQUIRE(n_rows == 0 || (d_G && ldg >= n_feat), GNNX_ERR_INVALID_ARG, "G null or ld < n_feat");
    float *partial = static_cast<float *>(d_workspace);
    int64_t rpb = ceil_div(n_rows > 0 ? n_rows : 1, nb);
    const bool vec = n_feat % 4 == 0 && ldg % 4 == 0 && aligned16(d_G) && aligned16(partial) && n_feat / 4 <= 256 &&
                     256 % (n_feat / 4) == 0;
    if (d_copy) {   // the copy rides in the 16-byte kernel only; other shapes: the plain sums and a strided copy behind them
        GNNX_REQUIRE(ldc >= n_feat && d_copy != d_G, GNNX_ERR_INVALID_ARG, "copy: ld < n_feat or aliasing");
        if (vec && ldc % 4 == 0 && aligned16(d_copy)) {
            // read + write streams: ONE workgroup per CU (two row ranges each, the same partials as the plain sums) -- 10 M x 256 inside
            // the layer step: 3.40 ms, against 4.04 with two workgroups per CU and 4.10 with eight (scripts/exp_colsum_copy.py, bench.py)
            static const int rpb_env = [] { const char *e = experiment_env("GNNX_COLSUM_COPY_RANGES"); return e ? atoi(e) : 0; }();   // A/B
            int per = rpb_env > 0 ? rpb_env : 2;
            if (nb % per != 0) per = 1;
            hipLaunchKernelGGL((colsum_stage1_vec<4, true>), dim3(nb / per), dim3(256), 0, st, d_G, ldg, n_rows, n_feat, rpb, per, partial, d_copy, ldc);
            GNNX_LAUNCH_CHECK();
            hipLaunchKernelGGL(colsum_stage2, dim3((uint32_t)ceil_div(n_feat, 64)), dim3(256), 0, st, partial, nb, n_feat, beta, d_out);
            GNNX_LAUNCH_CHECK();
            return GNNX_OK;
        }
        if (n_rows > 0)
            GNNX_HIP_CHECK(hipMemcpy2DAsync(d_copy, sizeof(float) * (size_t)ldc, d_G, sizeof(float) * (size_t)ldg, sizeof(float) * (size_t)n_feat,
                                            (size_t)n_rows, hipMemcpyDeviceToDevice, st));
    }
    if (vec) {
        hipLaunchKernelGGL(colsum_stage1_vec<4>, dim3(nb), dim3(256), 0, st, d_G, ldg, n_rows, n_feat, rpb, 1, partial, nullptr, 0);
    } else if (n_feat >= 128) {
        hipLaunchKernelGGL(colsum_stage1, dim3(nb), dim3(256), 0, st, d_G, ldg, n_rows, n_feat, rpb, partial);
    } else {
        int fw = 1;
        while (fw < n_feat) fw <<= 1;
        hipLaunchKernelGGL(colsum_stage1_narrow, dim3(nb), dim3(256), 0, st, d_G, ldg, n_rows, n_feat, fw, rpb, partial);
    }
    GNNX_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2, dim3((uint32_t)ceil_div(n_feat, 64)), dim3(256), 0, st, partial, nb, n_feat, beta,
                       d_out);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}
}  // namespace

GNNX_API int gnnx_colsum_f32(const float *d_G, int64_t ldg, int64_t n_rows, int32_t n_feat, float beta, float *d_out,
                             void *d_workspace, size_t workspace_bytes, void *stream)
{
    return colsum_impl(d_G, ldg, n_rows, n_feat, beta, d_out, nullptr, 0, d_workspace, workspace_bytes, stream);
}

GNNX_API int gnnx_colsum_copy_f32(const float *d_G, int64_t ldg, int64_t n_rows, int32_t n_feat, float beta, float *d_out, float *d_copy,
                                  int64_t ldc, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(d_copy, GNNX_ERR_INVALID_ARG, "copy is null");
    return colsum_impl(d_G, ldg, n_rows, n_feat, beta, d_out, d_copy, ldc, d_workspace, workspace_bytes, stream);
}

GNNX_API int gnnx_rows_to_slots_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_feat, const int32_t *d_slots, float *d_send,
                                    int64_t ld_send, float *d_colsum, float beta, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_feat > 0, GNNX_ERR_INVALID_ARG, "bad sizes");
    if (n_rows == 0 && !d_colsum) return GNNX_OK;
    GNNX_REQUIRE((n_rows == 0 || (d_X && d_slots && d_send)) && ldx >= n_feat && ld_send >= n_feat, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_feat");
    const bool vec = n_feat % 4 == 0 && ldx % 4 == 0 && ld_send % 4 == 0 && aligned16(d_X) && aligned16(d_send) && aligned16(d_slots) &&
                     n_feat / 4 <= 256 && 256 % (n_feat / 4) == 0;
    GNNX_REQUIRE(vec, GNNX_ERR_UNSUPPORTED, "rows of 16-byte pieces only (n_feat %% 4 == 0, n_feat / 4 a divisor of 256): use the gather pack");
    hipStream_t st = as_stream(stream);
    // the partials are those of gnnx_colsum_f32 (same fixed row ranges, same order): the sums are the same bits
    const int nb = colsum_blocks(n_rows);
    const int64_t rpb = ceil_div(n_rows > 0 ? n_rows : 1, nb);
    if (d_colsum) {
        const size_t need = sizeof(float) * (size_t)nb * n_feat;
        GNNX_REQUIRE(d_workspace && workspace_bytes >= need && aligned16(d_workspace), GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
        float *partial = static_cast<float *>(d_workspace);
        hipLaunchKernelGGL((rows_to_slots_vec<4, true>), dim3(nb), dim3(256), 0, st, d_X, ldx, n_rows, n_feat, d_slots, d_send, ld_send, rpb, 1, partial);
        GNNX_LAUNCH_CHECK();
        hipLaunchKernelGGL(colsum_stage2, dim3((uint32_t)ceil_div(n_feat, 64)), dim3(256), 0, st, partial, nb, n_feat, beta, d_colsum);
        GNNX_LAUNCH_CHECK();
        return GNNX_OK;
    }
    // no sums: nothing depends on the ranges -- many short ranges, eight workgroups per CU keep the stores coming
    const int64_t rows_per = 256;
    const int64_t blocks = ceil_div(n_rows, rows_per);
    hipLaunchKernelGGL((rows_to_slots_vec<4, false>), dim3((uint32_t)blocks), dim3(256), 0, st, d_X, ldx, n_rows, n_feat, d_slots, d_send, ld_send, rows_per, 1,
                       nullptr);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_gather_row_stride(int64_t n_rows, int32_t n_feat, int64_t *ld_out)
{
    GNNX_REQUIRE(ld_out && n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    // Rows of k * 512 bytes gathered by index from a big matrix: with a power-of-two row pitch the rows of a synthetic power-law
    // graph's hubs (vertex ids with few one-bits) have addresses with few one-bits and pile onto a few memory channels /
    // Infinity-Cache slices.  64 floats of padding per row spread them: RMAT 10 M / 100 M, F = 256, vertices as generated, forward
    // aggregation 17.75 ms on stride 256, 14.64 / 14.35 / 14.13 / 13.88 ms on 264 / 272 / 288 / 320 (13.76 with the vertices
    // relabelled by a multiplicative hash; scripts/exp_spmm_stride.py).  Small matrices (cache-resident) keep their width.
    *ld_out = (n_feat > 0 && (n_feat * 4) % 512 == 0 && n_rows * (int64_t)n_feat >= (int64_t)1 << 24) ? (int64_t)n_feat + 64 : (int64_t)n_feat;
    return GNNX_OK;
}

GNNX_API int gnnx_rowscale_f32(const float *d_X, int64_t ldx, const float *d_v, int64_t n_rows, int32_t n_feat,
                               float *d_Y, int64_t ldy, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_v && d_Y && ldx >= n_feat && ldy >= n_feat, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_feat");
    if (n_feat > 1)   // Y = X * v[row]: the broadcast kernel (16 bytes per thread when the rows allow it)
        return gnnx_binary_bcast_f32(GNNX_OP_MUL, n_rows, n_feat, d_X, ldx, 1, d_v, 1, 0, d_Y, ldy, stream);
    int64_t blocks = ceil_div(n_rows * n_feat, 256);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(rowwise_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_X, ldx, d_v, n_rows,
                       n_feat, d_Y, ldy);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_bias_add_f32(const float *d_X, int64_t ldx, const float *d_b, int64_t n_rows, int32_t n_feat,
                               float *d_Y, int64_t ldy, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_b && d_Y && ldx >= n_feat && ldy >= n_feat, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_feat");
    if (n_feat > 1 && n_rows > 1)   // Y = X + b[col]
        return gnnx_binary_bcast_f32(GNNX_OP_ADD, n_rows, n_feat, d_X, ldx, 1, d_b, 0, 1, d_Y, ldy, stream);
    int64_t blocks = ceil_div(n_rows * n_feat, 256);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(rowwise_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_X, ldx, d_b, n_rows,
                       n_feat, d_Y, ldy);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_binary_bcast_f32(int op, int64_t n_rows, int64_t n_cols, const float *d_A, int64_t a_row_stride,
                                   int64_t a_col_stride, const float *d_B, int64_t b_row_stride, int64_t b_col_stride, float *d_Y,
                                   int64_t ldy, void *stream)
{
    GNNX_REQUIRE(op >= GNNX_OP_ADD && op <= GNNX_OP_DIV, GNNX_ERR_INVALID_ARG, "unknown op %d", op);
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_cols == 0) return GNNX_OK;
    GNNX_REQUIRE(d_A && d_B && d_Y && ldy >= n_cols, GNNX_ERR_INVALID_ARG, "null pointer or ldy < n_cols");
    GNNX_REQUIRE(a_row_stride >= 0 && a_col_stride >= 0 && b_row_stride >= 0 && b_col_stride >= 0, GNNX_ERR_INVALID_ARG,
                 "negative stride");
    hipStream_t st = as_stream(stream);
    auto vec_operand = [](const float *ptr, int64_t rs, int64_t cs) {   // column-contiguous and 16-byte aligned rows, or column-broadcast
        return cs == 0 || (cs == 1 && rs % 4 == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0);
    };
    if (n_cols % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(d_Y) & 15u) == 0 && vec_operand(d_A, a_row_stride, a_col_stride) &&
        vec_operand(d_B, b_row_stride, b_col_stride)) {
        const int32_t quads = (int32_t)(n_cols / 4);
        int64_t vb = ceil_div(n_rows * quads, 256);
        if (vb > 16384) vb = 16384;
        dim3 vg((uint32_t)vb), vt(256);
#define GNNX_BINV(OPV) hipLaunchKernelGGL(binary_bcast_vec_kernel<OPV>, vg, vt, 0, st, d_A, a_row_stride, (int)a_col_stride, d_B, \
                                          b_row_stride, (int)b_col_stride, n_rows, quads, d_Y, ldy)
        switch (op) {
        case GNNX_OP_ADD: GNNX_BINV(0); break;
        case GNNX_OP_SUB: GNNX_BINV(1); break;
        case GNNX_OP_MUL: GNNX_BINV(2); break;
        default: GNNX_BINV(3); break;
        }
#undef GNNX_BINV
        GNNX_LAUNCH_CHECK();
        return GNNX_OK;
    }
    int64_t blocks = ceil_div(n_rows * n_cols, 256);
    if (blocks > 8192) blocks = 8192;
    dim3 g((uint32_t)blocks), b(256);
#define GNNX_BIN(OPV) hipLaunchKernelGGL(binary_bcast_kernel<OPV>, g, b, 0, st, d_A, a_row_stride, a_col_stride, d_B, b_row_stride, \
                                         b_col_stride, n_rows, n_cols, d_Y, ldy)
    switch (op) {
    case GNNX_OP_ADD: GNNX_BIN(0); break;
    case GNNX_OP_SUB: GNNX_BIN(1); break;
    case GNNX_OP_MUL: GNNX_BIN(2); break;
    default: GNNX_BIN(3); break;
    }
#undef GNNX_BIN
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_rowsum_f32(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_cols, float *d_out, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_out && ldx >= n_cols, GNNX_ERR_INVALID_ARG, "null pointer or ldx < n_cols");
    const bool vec = ((n_cols + 63) / 64) % 4 == 0 && n_cols % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(d_X) & 15u) == 0;
    if (vec) hipLaunchKernelGGL(rowsum_kernel<true>, dim3((uint32_t)ceil_div(n_rows, 4)), dim3(256), 0, as_stream(stream), d_X, ldx, n_rows, n_cols, d_out);
    else hipLaunchKernelGGL(rowsum_kernel<false>, dim3((uint32_t)ceil_div(n_rows, 4)), dim3(256), 0, as_stream(stream), d_X, ldx, n_rows, n_cols, d_out);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_f32_to_bf16(const float *d_X, int64_t ldx, int64_t n_rows, int32_t n_cols, uint16_t *d_Y, int64_t ldy, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_cols == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_Y && ldx >= n_cols && ldy >= n_cols, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_cols");
    const bool vec = n_cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(d_X) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(d_Y) & 7u) == 0;
    int64_t blocks = ceil_div(vec ? n_rows * (n_cols / 4) : n_rows * n_cols, 256);
    if (blocks > 16384) blocks = 16384;
    if (vec) hipLaunchKernelGGL(to_bf16_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_X, ldx, n_rows, n_cols, d_Y, ldy);
    else hipLaunchKernelGGL(to_bf16_scalar_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_X, ldx, n_rows, n_cols, d_Y, ldy);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_axpy_f32(int64_t n, float a, const float *d_x, float *d_y, void *stream)
{
    GNNX_REQUIRE(n >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_x && d_y, GNNX_ERR_INVALID_ARG, "null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(axpy_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), n, a, d_x, d_y);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_gather_rows_f32(const float *d_X, int64_t ldx, const int32_t *d_idx, int64_t n_idx, int32_t n_feat,
                                  float *d_out, int64_t ldo, void *stream)
{
    GNNX_REQUIRE(n_idx >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_idx == 0 || n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_idx && d_out && ldx >= n_feat && ldo >= n_feat, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_feat");
    return launch_rows<false>(d_X, ldx, d_idx, n_idx, n_feat, d_out, ldo, as_stream(stream));
}

GNNX_API int gnnx_scatter_add_rows_f32(const float *d_in, int64_t ldi, const int32_t *d_idx, int64_t n_idx,
                                       int32_t n_feat, float *d_Y, int64_t ldy, void *stream)
{
    GNNX_REQUIRE(n_idx >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_idx == 0 || n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_in && d_idx && d_Y && ldi >= n_feat && ldy >= n_feat, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_feat");
    return launch_rows<true>(d_in, ldi, d_idx, n_idx, n_feat, d_Y, ldy, as_stream(stream));
}

GNNX_API int gnnx_fill_f32(float *d_x, int64_t n, float value, void *stream)
{
    GNNX_REQUIRE(n >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_x, GNNX_ERR_INVALID_ARG, "null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    hipLaunchKernelGGL(fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_x, n, value);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_pow_f32(const float *d_x, int64_t n, float exponent, float *d_y, void *stream)
{
    GNNX_REQUIRE(n >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_x && d_y, GNNX_ERR_INVALID_ARG, "null pointer");
    hipStream_t st = as_stream(stream);
    int64_t blocks = ceil_div(n, 256);
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    // The reference evaluates std::pow on the HOST (functional.h:253), i.e. the host libm's powf, and the one call on the hot path is
    // deg->pow(-0.5) on the degrees 1 + rowsum(A): non-negative integers.  For exponent -0.5 and such an argument vector (every element
    // an integer in [0, 2^24], no negative zero) the result is looked up in the process-wide table of that very libm call
    // (libm_pow_m05_table, gnnx_graph.hip) -- the reference's bits, where a device pow would be 1 ulp off for some k.  One reduction and
    // one host synchronisation: a graph-build call, not a per-step one.  Every other exponent or argument vector is evaluated on the
    // device without any synchronisation (tolerance-level against the host libm; stream-asynchronous, capturable).
    if (exponent == -0.5f) {
        DeviceFreeSync info_g;
        GNNX_HIP_CHECK(hipMalloc(&info_g.p, 2 * sizeof(int32_t)));
        int32_t *d_info = static_cast<int32_t *>(info_g.p);
        GNNX_HIP_CHECK(hipMemsetAsync(d_info, 0, 2 * sizeof(int32_t), st));
        hipLaunchKernelGGL(pow_scan_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_x, n, d_info);
        GNNX_LAUNCH_CHECK();
        int32_t info[2] = {0, 1};
        GNNX_HIP_CHECK(hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        if (!info[1]) {
            const float *d_table = nullptr;
            const int rc = libm_pow_m05_table((size_t)info[0] + 1, &d_table, nullptr);
            if (rc != GNNX_OK) return rc;
            hipLaunchKernelGGL(pow_table_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_x, n, d_table, d_y);
            GNNX_LAUNCH_CHECK();
            return GNNX_OK;   // (the table is immutable and never freed: nothing to wait for)
        }
    }
    hipLaunchKernelGGL(pow_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_x, n, exponent, d_y);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_csr_rowsum_f32(const int32_t *d_rowptr, const float *d_vals, int32_t n_rows, float *d_out, void *stream)
{
    GNNX_REQUIRE(n_rows >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0) return GNNX_OK;
    GNNX_REQUIRE(d_rowptr && d_out, GNNX_ERR_INVALID_ARG, "null pointer");
    hipLaunchKernelGGL(csr_rowsum_kernel, dim3((uint32_t)ceil_div(n_rows, 256)), dim3(256), 0, as_stream(stream), d_rowptr,
                       d_vals, n_rows, d_out);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_transpose_f32(const float *d_X, int64_t ldx, int64_t n_rows, int64_t n_cols, float *d_Y, int64_t ldy,
                                void *stream)
{
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_cols == 0) return GNNX_OK;
    GNNX_REQUIRE(d_X && d_Y && d_X != d_Y && ldx >= n_cols && ldy >= n_rows, GNNX_ERR_INVALID_ARG, "bad pointers or ld");
    dim3 grid((uint32_t)ceil_div(n_cols, 32), (uint32_t)ceil_div(n_rows, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, as_stream(stream), d_X, ldx, n_rows, n_cols, d_Y, ldy);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}
