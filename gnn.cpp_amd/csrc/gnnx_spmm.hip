// CSR SpMM for gfx950 (MI355X): the aggregation step of the GCN hot path.
//
// Replaces reference functional::matmul on the dense N x N adjacency (functional.h:399-441 called from
// graph.cpp:208 and, in backward, operation.h:524-531) plus the broadcast multiply by norm
// (graph.cpp:209) and the bias add (graph.cpp:188).
//
// Design (HBM-bound; DESIGN.md section "SpMM"):
//   * one output row per G-lane group, G = ceil(F/4) rounded to a power of two (64 lanes = one wavefront
//     at F = 256, two rows per wavefront at F = 128/100, ...); each lane owns 4 consecutive features and
//     reads its 16-byte piece of every neighbour row with one global_load_dwordx4, so a neighbour row is
//     one fully coalesced 4*F-byte burst (1 KiB per wave-instruction at F = 256);
//   * U neighbour rows in flight per group (register staging; no LDS: a row is used by exactly one
//     group, there is nothing to share), accumulated strictly in DESCENDING column order with separately
//     rounded fp32 adds => bit-identical to the reference's sequential dense dot product;
//   * at G = 64 the row is wave-uniform: rowptr/colidx are fetched with scalar loads, the row base
//     address lives in SGPRs and the VMEM unit sees only the feature traffic;
//   * power-law rows: an optional plan cuts rows longer than `chunk` into chunks that are gathered by
//     separate wavefronts into a partial slab and combined in fixed chunk order by a second tiny kernel
//     (deterministic; no float atomics).  Chunk items are placed first in the grid.
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "gnnx_common.h"

// Parity depends on separately rounded fp32 mul / add (the reference has no FMA): never contract.
#pragma clang fp contract(off)

using namespace gnnx;

struct gnnx_spmm_plan {
    int32_t n_rows = 0;
    int32_t chunk = 0;
    int32_t max_feat = 0;
    int32_t n_split_rows = 0;   // rows with degree > chunk
    int32_t n_chunks = 0;       // total chunk items
    int4 *d_items = nullptr;    // [n_chunks] {row, first nz, end nz, partial slot}
    int4 *d_rows = nullptr;     // [n_split_rows] {row, first slot, n_chunks, 0}
    float *d_partial = nullptr; // [n_chunks, max_feat]
    int32_t *d_counters = nullptr;
    // non-zero-balanced row blocks for the streaming kernel: block k owns rows [d_block_starts[k], [k+1])
    int32_t block_nnz = 0;
    int32_t n_blocks = 0;
    int32_t *d_block_starts = nullptr;
};

namespace {

struct SpmmArgs {
    int32_t n_rows;
    int32_t n_feat;
    const int32_t *rowptr;
    const int32_t *colidx;
    const float *vals;
    const float *colscale;
    const float *rowscale;
    const float *bias;
    const float *X;
    int64_t ldx;
    float *Y;
    int64_t ldy;
    int32_t beta;           // 0 or 1
    int32_t split_threshold; // rows with degree > this are left to the chunk items (0 = none)
    // chunk items (plan)
    const int4 *items;
    int32_t n_items;
    float *partial;
    int32_t partial_ld;
    // row blocks of the streaming kernel (plan): nullptr => fixed blocks of StreamCfg<G>::R rows
    const int32_t *block_starts;
    int32_t n_blocks;
    // fusion (gnnx_spmm_csr_fused_f32): ReLU on the stored row; BatchNorm / ReLU applied to every gathered row of X
    int32_t relu_out;
    int32_t x_bf16;          // X holds bf16 (gnnx_spmm_csr_bf16_f32)
    const float *pro_mean, *pro_var, *pro_gamma, *pro_beta;
    float pro_eps;
};

template <int VEC> struct Vec;
template <> struct Vec<4> { using type = float4; };
template <> struct Vec<1> { using type = float; };

__device__ __forceinline__ float4 ld_vec(const float4 *p) { return *p; }
__device__ __forceinline__ float ld_vec(const float *p) { return *p; }

// A neighbour row's slice for this lane: VEC features stored as f32, or as bf16 (opt-in feature storage, half the gather
// bytes; widened exactly -- a bf16 is the top half of an f32 -- and accumulated in f32 like the f32 rows).
typedef uint16_t bf16_t;
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::type ld_x(const float *p)
{
    return ld_vec(reinterpret_cast<const typename Vec<VEC>::type *>(p));
}
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::type ld_x(const bf16_t *p)
{
    if constexpr (VEC == 4) {
        const uint2 h = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16),
                           __uint_as_float(h.y & 0xffff0000u));
    } else {
        return __uint_as_float((uint32_t)*p << 16);
    }
}

// Separately rounded ops: hipcc contracts a*b+c into an fma by default; the reference rounds the product
// and the sum separately (x86-64 baseline, no FMA), so spell the roundings out.
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float4 add_rn(float4 a, float4 b)
{
    return make_float4(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y), __fadd_rn(a.z, b.z), __fadd_rn(a.w, b.w));
}
__device__ __forceinline__ float4 mul_rn(float4 a, float s)
{
    return make_float4(__fmul_rn(a.x, s), __fmul_rn(a.y, s), __fmul_rn(a.z, s), __fmul_rn(a.w, s));
}
__device__ __forceinline__ void zero(float4 &v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void zero(float &v) { v = 0.f; }

__device__ __forceinline__ float relu1(float v) { return v > 0.f ? v : 0.f; }  // where(x > 0, x, 0), nn.cpp:229-237
__device__ __forceinline__ float relu_v(float v) { return relu1(v); }
__device__ __forceinline__ float4 relu_v(float4 v) { return make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)); }

// Prologue on a gathered row (MODE 3..5): the modules GCNConv::forward runs between transform and aggregation
// (graph.cpp:174-175) folded into the gather, so the normalised / rectified H is never written to HBM.  Same separately
// rounded op order as gnnx_bn_relu_fwd_f32 (sub, div, mul, add, where): fused and unfused results are the same bits.
constexpr bool has_sc(int MODE) { return MODE == 1 || MODE == 6; }    // per-column scale (colscale)
constexpr bool has_val(int MODE) { return MODE == 2 || MODE == 6; }   // per-entry value (vals)
constexpr bool has_pro(int MODE) { return MODE >= 3 && MODE <= 5; }   // prologue on the gathered row
constexpr bool pro_bn(int MODE) { return MODE == 4 || MODE == 5; }
constexpr bool pro_relu(int MODE) { return MODE == 3 || MODE == 5; }
template <int VEC> struct ProConst { typename Vec<VEC>::type mean, sd, gamma, beta; };

template <int MODE>
__device__ __forceinline__ float pro1(float v, float mean, float sd, float gamma, float beta)
{
    if constexpr (pro_bn(MODE)) {
        v = __fdiv_rn(__fsub_rn(v, mean), sd);
        v = __fmul_rn(v, gamma);
        v = __fadd_rn(v, beta);
    }
    if constexpr (pro_relu(MODE)) v = relu1(v);
    return v;
}
template <int MODE>
__device__ __forceinline__ float pro_apply(float v, const ProConst<1> &c) { return pro1<MODE>(v, c.mean, c.sd, c.gamma, c.beta); }
template <int MODE>
__device__ __forceinline__ float4 pro_apply(float4 v, const ProConst<4> &c)
{
    return make_float4(pro1<MODE>(v.x, c.mean.x, c.sd.x, c.gamma.x, c.beta.x), pro1<MODE>(v.y, c.mean.y, c.sd.y, c.gamma.y, c.beta.y),
                       pro1<MODE>(v.z, c.mean.z, c.sd.z, c.gamma.z, c.beta.z), pro1<MODE>(v.w, c.mean.w, c.sd.w, c.gamma.w, c.beta.w));
}
__device__ __forceinline__ float sd_of(float var, float eps) { return sqrtf(__fadd_rn(var, eps)); }  // (var + eps)->pow(0.5)
__device__ __forceinline__ float4 sd_of(float4 var, float eps)
{
    return make_float4(sd_of(var.x, eps), sd_of(var.y, eps), sd_of(var.z, eps), sd_of(var.w, eps));
}
__device__ __forceinline__ void splat(float &v, float x) { v = x; }
__device__ __forceinline__ void splat(float4 &v, float x) { v = make_float4(x, x, x, x); }

template <int VEC, int MODE, class ARGS>
__device__ __forceinline__ ProConst<VEC> pro_load(const ARGS &a, int32_t f0, bool active)
{
    using V = typename Vec<VEC>::type;
    ProConst<VEC> c;
    splat(c.mean, 0.f);
    splat(c.sd, 1.f);
    splat(c.gamma, 1.f);
    splat(c.beta, 0.f);
    if constexpr (pro_bn(MODE)) {
        if (active) {
            c.mean = ld_vec(reinterpret_cast<const V *>(a.pro_mean + f0));
            c.sd = sd_of(ld_vec(reinterpret_cast<const V *>(a.pro_var + f0)), a.pro_eps);
            if (a.pro_gamma) c.gamma = ld_vec(reinterpret_cast<const V *>(a.pro_gamma + f0));
            if (a.pro_beta) c.beta = ld_vec(reinterpret_cast<const V *>(a.pro_beta + f0));
        }
    }
    return c;
}

// MODE 0: plain gather-add (forward).  MODE 1: gathered row scaled by colscale[c] (backward: norm (.) G).
// MODE 2: per-entry values (vals).  MODE 6: vals and colscale.  MODE 3 / 4 / 5: forward with a ReLU / BatchNorm /
// BatchNorm+ReLU prologue.
//
// Broadcast lane `src` (index inside the G-lane row group) of v to the whole group.  At G == 64 the source
// lane is wave-uniform => v_readlane into an SGPR, so the neighbour row's base address is scalar and the
// VMEM instruction is the saddr form; below 64 it is a ds_bpermute (LDS crossbar, no memory traffic).
template <int G>
__device__ __forceinline__ int32_t bcast(int32_t v, int src, int gbase)
{
    if constexpr (G == 64) return __builtin_amdgcn_readlane(v, src);
    else return __shfl(v, gbase + src, 64);
}

// One batch of B neighbour rows: all B loads are issued before the first add (B rows in flight per group),
// adds strictly in order k, k+1, ... (= descending column).  No load is predicated: hipcc turns a
// conditional load into branch + s_waitcnt vmcnt(0) per element, which serialises the gather.
template <int G, int VEC, int B, int MODE, class XT>
__device__ __forceinline__ void gather_batch(typename Vec<VEC>::type &acc, int k, int32_t myc, float mysc, float myval,
                                             int gbase, const XT *xf, const SpmmArgs &a, const ProConst<VEC> &pc)
{
    using V = typename Vec<VEC>::type;
    int32_t c[B];
    V v[B];
#pragma unroll
    for (int u = 0; u < B; u++) c[u] = bcast<G>(myc, k + u, gbase);
#pragma unroll
    for (int u = 0; u < B; u++) v[u] = ld_x<VEC>(xf + (int64_t)c[u] * a.ldx);
#pragma unroll
    for (int u = 0; u < B; u++) {
        V t = v[u];
        if constexpr (has_pro(MODE)) t = pro_apply<MODE>(t, pc);
        if constexpr (has_sc(MODE)) t = mul_rn(t, __int_as_float(bcast<G>(__float_as_int(mysc), k + u, gbase)));
        if constexpr (has_val(MODE)) t = mul_rn(t, __int_as_float(bcast<G>(__float_as_int(myval), k + u, gbase)));
        acc = add_rn(acc, t);
    }
}

// Gather-accumulate nz range [b, e) of one row in DESCENDING column order (`s = p[n-1]; s += p[n-2]; ...`;
// starting from 0 is the same thing except for the sign of a zero).
// Indices are fetched G at a time with ONE coalesced vector load per group (lane li takes the li-th
// neighbour from the top), together with their colscale / vals, then handed out by bcast().
template <int G, int VEC, int U, int MODE, class XT>
__device__ __forceinline__ typename Vec<VEC>::type gather_range(int32_t b, int32_t e, const XT *xf, int li,
                                                                const SpmmArgs &a, const ProConst<VEC> &pc)
{
    using V = typename Vec<VEC>::type;
    constexpr int UE = U < G ? U : G;
    V acc;
    zero(acc);
    const int gbase = (threadIdx.x & 63) - li;
    for (int32_t hi = e; hi > b; hi -= G) {
        int32_t q = hi - 1 - li;
        q = q >= b ? q : b;
        const int32_t myc = a.colidx[q];
        float mysc = 1.f, myval = 1.f;
        if constexpr (has_sc(MODE)) mysc = a.colscale[myc];
        if constexpr (has_val(MODE)) myval = a.vals[q];
        const int n = (hi - b) < G ? (hi - b) : G;
        int k = 0;
        for (; k + UE <= n; k += UE) gather_batch<G, VEC, UE, MODE>(acc, k, myc, mysc, myval, gbase, xf, a, pc);
        if constexpr (UE >= 8) if (k + 4 <= n) { gather_batch<G, VEC, 4, MODE>(acc, k, myc, mysc, myval, gbase, xf, a, pc); k += 4; }
        if constexpr (UE >= 4) if (k + 2 <= n) { gather_batch<G, VEC, 2, MODE>(acc, k, myc, mysc, myval, gbase, xf, a, pc); k += 2; }
        if constexpr (UE >= 2) if (k + 1 <= n) { gather_batch<G, VEC, 1, MODE>(acc, k, myc, mysc, myval, gbase, xf, a, pc); k += 1; }
    }
    return acc;
}

template <int VEC>
__device__ __forceinline__ void epilogue_store(typename Vec<VEC>::type acc, int32_t row, int32_t f0, const SpmmArgs &a)
{
    using V = typename Vec<VEC>::type;
    if (a.rowscale) acc = mul_rn(acc, a.rowscale[row]);
    if (a.bias) acc = add_rn(acc, ld_vec(reinterpret_cast<const V *>(a.bias + f0)));
    V *dst = reinterpret_cast<V *>(a.Y + (int64_t)row * a.ldy + f0);
    if (a.beta) acc = add_rn(*dst, acc);
    if (a.relu_out) acc = relu_v(acc);
    *dst = acc;
}

// Same epilogue with the per-row scale and the bias already in registers.  Inside the streaming loop a fresh vector load
// (bias[f], rowscale[row]) would have to wait behind the whole batch of neighbour rows just put in flight -- VMEM returns
// in order -- i.e. drain the pipeline at every row boundary; so both are fetched ahead (bias once per group, rowscale
// one entry per lane for the block's rows).
template <int VEC>
__device__ __forceinline__ void epilogue_store_pre(typename Vec<VEC>::type acc, int32_t row, int32_t f0, const SpmmArgs &a,
                                                   bool has_rs, float rs, bool has_bias, typename Vec<VEC>::type bias)
{
    using V = typename Vec<VEC>::type;
    if (has_rs) acc = mul_rn(acc, rs);
    if (has_bias) acc = add_rn(acc, bias);
    V *dst = reinterpret_cast<V *>(a.Y + (int64_t)row * a.ldy + f0);
    if (a.beta) acc = add_rn(*dst, acc);
    if (a.relu_out) acc = relu_v(acc);
    *dst = acc;
}

// grid.x = n_item_blocks + n_row_blocks ; grid.y = feature tiles of G*VEC features.
template <int G, int VEC, int U, int MODE, class XT>
__global__ __launch_bounds__(256) void spmm_kernel(SpmmArgs a, int32_t n_item_blocks)
{
    constexpr int GROUPS = 256 / G;
    const int tid = threadIdx.x;
    const int li = tid % G;         // lane inside the row group
    int grp = tid / G;              // group inside the block
    if constexpr (G == 64) grp = __builtin_amdgcn_readfirstlane(grp);
    const int32_t f0 = (blockIdx.y * G + li) * VEC;
    const bool active = f0 + VEC <= a.n_feat;
    const XT *xf = reinterpret_cast<const XT *>(a.X) + (active ? f0 : 0);  // lanes past n_feat read feature 0 and never store
    const ProConst<VEC> pc = pro_load<VEC, MODE>(a, f0, active);

    if ((int32_t)blockIdx.x < n_item_blocks) {
        // ---- chunk item of a long row: partial sum into the plan's slab
        int32_t it = blockIdx.x * GROUPS + grp;
        if (it >= a.n_items) return;
        int4 item = a.items[it];
        if constexpr (G == 64) {
            item.x = __builtin_amdgcn_readfirstlane(item.x);
            item.y = __builtin_amdgcn_readfirstlane(item.y);
            item.z = __builtin_amdgcn_readfirstlane(item.z);
            item.w = __builtin_amdgcn_readfirstlane(item.w);
        }
        auto acc = gather_range<G, VEC, U, MODE>(item.y, item.z, xf, li, a, pc);
        if (active) *reinterpret_cast<typename Vec<VEC>::type *>(a.partial + (int64_t)item.w * a.partial_ld + f0) = acc;
        return;
    }

    int32_t row = (blockIdx.x - n_item_blocks) * GROUPS + grp;
    if (row >= a.n_rows) return;
    int32_t b = a.rowptr[row], e = a.rowptr[row + 1];
    if constexpr (G == 64) {
        b = __builtin_amdgcn_readfirstlane(b);
        e = __builtin_amdgcn_readfirstlane(e);
    }
    if (a.split_threshold > 0 && e - b > a.split_threshold) return;  // handled by chunk items + combine
    auto acc = gather_range<G, VEC, U, MODE>(b, e, xf, li, a, pc);
    if (active) epilogue_store<VEC>(acc, row, f0, a);
}

// ---- streaming kernel (the hot one at F > 64) -------------------------------------------------------------
// One G-lane group owns a BLOCK of R consecutive rows and streams the block's whole non-zero range
// [rowptr[r0], rowptr[r0+R)) from the top down (so inside every row the columns come in DESCENDING order,
// the reference's order), B neighbour rows per batch, double-buffered: batch j+1 is in flight from HBM
// while batch j is added.  Row boundaries are crossed inside the stream (store the finished row, clear the
// accumulator, go on), so short rows -- most rows of a power-law graph -- cost no pipeline drain, no extra
// wavefront launch and no dependent rowptr -> colidx -> feature latency chain of their own.
//   * rowptr of the block: one coalesced load, kept one entry per lane, read back with bcast();
//   * colidx (+ colscale / vals): one coalesced load per G non-zeros, the next chunk prefetched;
//   * no load is predicated (see gather_batch): the last batch of a block re-reads an in-range row;
//   * rows longer than the plan's threshold are skipped here (chunk items + combine kernel own them): the
//     block is cut into segments of consecutive non-hub rows with a ballot mask.
// At G == 64 every control decision is wave-uniform (SALU + s_cbranch); at G == 32 the two half-waves
// stream independent blocks under the EXEC mask.
template <int G> struct StreamCfg { static constexpr int R = G / 2; };

template <int G, int VEC, int B, int MODE, class XT>
struct Stream {
    using V = typename Vec<VEC>::type;
    const SpmmArgs &a;
    const XT *xf;
    int li, gbase, f0;
    bool active;
    int32_t r0;      // first row of the block
    int32_t rp_l;    // lane l holds rowptr[r0 + min(l, nr)]
    float rs_l;      // lane l holds rowscale[r0 + min(l, nr - 1)] (if any)
    V bias_v;        // this lane's slice of the bias (if any)
    ProConst<VEC> pc;  // this lane's slice of the prologue constants (MODE >= 3)

    __device__ __forceinline__ int32_t rp(int l) const { return bcast<G>(rp_l, l, gbase); }

    __device__ __forceinline__ void flush(V &acc, int r) const
    {
        const float rs = __int_as_float(bcast<G>(__float_as_int(rs_l), r, gbase));
        if (active) epilogue_store_pre<VEC>(acc, r0 + r, f0, a, a.rowscale != nullptr, rs, a.bias != nullptr, bias_v);
        zero(acc);
    }

    struct Chunk { int32_t c; float sc; float val; };  // per lane: one non-zero of the current G-chunk

    __device__ __forceinline__ Chunk fetch_chunk(int32_t hi, int32_t lo, int cidx) const
    {
        int32_t q = hi - 1 - (cidx * G + li);
        q = q < lo ? lo : q;  // clamped: still a valid non-zero of this segment
        Chunk ch;
        ch.c = a.colidx[q];
        ch.sc = 1.f;
        ch.val = 1.f;
        if constexpr (has_sc(MODE)) ch.sc = a.colscale[ch.c];
        if constexpr (has_val(MODE)) ch.val = a.vals[q];
        return ch;
    }

    struct Batch { V v[B]; float sc[B]; float val[B]; };

    __device__ __forceinline__ void issue(Batch &b, const Chunk &ch, int k0) const
    {
        int32_t c[B];
#pragma unroll
        for (int u = 0; u < B; u++) c[u] = bcast<G>(ch.c, k0 + u, gbase);
#pragma unroll
        for (int u = 0; u < B; u++) b.v[u] = ld_x<VEC>(xf + (int64_t)c[u] * a.ldx);
        if constexpr (has_sc(MODE)) {
#pragma unroll
            for (int u = 0; u < B; u++) b.sc[u] = __int_as_float(bcast<G>(__float_as_int(ch.sc), k0 + u, gbase));
        }
        if constexpr (has_val(MODE)) {
#pragma unroll
            for (int u = 0; u < B; u++) b.val[u] = __int_as_float(bcast<G>(__float_as_int(ch.val), k0 + u, gbase));
        }
    }

    // add entries e .. e+B-1 of the stream (entry e is non-zero hi-1-e), crossing row boundaries as they come
    __device__ __forceinline__ void consume(const Batch &b, int32_t e, int32_t total, int32_t hi, V &acc, int &r,
                                            int32_t &rs) const
    {
#pragma unroll
        for (int u = 0; u < B; u++) {
            if (e + u < total) {
                const int32_t q = hi - 1 - (e + u);
                while (q < rs) {  // every non-zero of row r (and of the empty rows above q) is in: store them
                    flush(acc, r);
                    r--;
                    rs = rp(r);
                }
                V t = b.v[u];
                if constexpr (has_pro(MODE)) t = pro_apply<MODE>(t, pc);
                if constexpr (has_sc(MODE)) t = mul_rn(t, b.sc[u]);
                if constexpr (has_val(MODE)) t = mul_rn(t, b.val[u]);
                acc = add_rn(acc, t);
            }
        }
    }

    // rows [sa, sb) of the block (local indices), none of them a hub
    __device__ __forceinline__ void segment(int sa, int sb) const
    {
        const int32_t lo = rp(sa), hi = rp(sb);
        const int32_t total = hi - lo;
        int r = sb - 1;
        int32_t rs = rp(r);
        V acc;
        zero(acc);
        if (total > 0) {
            Chunk cur = fetch_chunk(hi, lo, 0);
            Chunk nxt = fetch_chunk(hi, lo, 1);
            int cidx = 1;
            Batch ba, bb;
            issue(ba, cur, 0);
            int32_t e = 0;
            while (true) {
                {   // batch e is in ba; put batch e+B in flight into bb, then add ba
                    const int kn = (e + B) % G;
                    if (kn == 0) { cur = nxt; cidx++; nxt = fetch_chunk(hi, lo, cidx); }
                    issue(bb, cur, kn);
                    consume(ba, e, total, hi, acc, r, rs);
                    e += B;
                    if (e >= total) break;
                }
                {
                    const int kn = (e + B) % G;
                    if (kn == 0) { cur = nxt; cidx++; nxt = fetch_chunk(hi, lo, cidx); }
                    issue(ba, cur, kn);
                    consume(bb, e, total, hi, acc, r, rs);
                    e += B;
                    if (e >= total) break;
                }
            }
        }
        while (r >= sa) {  // the current row and any empty rows below it
            flush(acc, r);
            r--;
        }
    }
};

template <int G, int VEC, int B, int MODE, int TPB, class XT>
__global__ __launch_bounds__(TPB) void spmm_stream_kernel(SpmmArgs a, int32_t n_item_blocks)
{
    constexpr int GROUPS = TPB / G;
    constexpr int R = StreamCfg<G>::R;
    const int tid = threadIdx.x;
    const int li = tid % G;
    int grp = tid / G;
    if constexpr (G == 64) grp = __builtin_amdgcn_readfirstlane(grp);
    const int32_t f0 = (blockIdx.y * G + li) * VEC;
    const bool active = f0 + VEC <= a.n_feat;
    const XT *xf = reinterpret_cast<const XT *>(a.X) + (active ? f0 : 0);
    const ProConst<VEC> pc = pro_load<VEC, MODE>(a, f0, active);

    if ((int32_t)blockIdx.x < n_item_blocks) {  // chunk item of a hub row -> partial slab (same as spmm_kernel)
        int32_t it = blockIdx.x * GROUPS + grp;
        if (it >= a.n_items) return;
        int4 item = a.items[it];
        if constexpr (G == 64) {
            item.y = __builtin_amdgcn_readfirstlane(item.y);
            item.z = __builtin_amdgcn_readfirstlane(item.z);
            item.w = __builtin_amdgcn_readfirstlane(item.w);
        }
        auto acc = gather_range<G, VEC, 8, MODE>(item.y, item.z, xf, li, a, pc);
        if (active) *reinterpret_cast<typename Vec<VEC>::type *>(a.partial + (int64_t)item.w * a.partial_ld + f0) = acc;
        return;
    }

    const int32_t blk = (blockIdx.x - n_item_blocks) * GROUPS + grp;
    int32_t r0, nr;
    if (a.block_starts) {  // non-zero-balanced blocks of the plan (at most kPlanBlockRows <= R rows each)
        if (blk >= a.n_blocks) return;
        r0 = a.block_starts[blk];
        nr = a.block_starts[blk + 1] - r0;
        if constexpr (G == 64) {
            r0 = __builtin_amdgcn_readfirstlane(r0);
            nr = __builtin_amdgcn_readfirstlane(nr);
        }
    } else {
        const int64_t r0l = (int64_t)blk * R;
        if (r0l >= a.n_rows) return;
        r0 = (int32_t)r0l;
        nr = a.n_rows - r0 < R ? a.n_rows - r0 : R;
    }
    const int gbase = (tid & 63) - li;
    typename Vec<VEC>::type bias_v;
    zero(bias_v);
    if (a.bias && active) bias_v = ld_vec(reinterpret_cast<const typename Vec<VEC>::type *>(a.bias + f0));
    const float rs_l = a.rowscale ? a.rowscale[r0 + (li < nr ? li : nr - 1)] : 1.f;
    Stream<G, VEC, B, MODE, XT> st{a, xf, li, gbase, f0, active, r0, a.rowptr[r0 + (li < nr ? li : nr)], rs_l, bias_v, pc};

    uint64_t hub = 0;  // bit l: local row l is a hub (left to the chunk items)
    if (a.split_threshold > 0) {
        const int32_t nxt = __shfl(st.rp_l, (tid & 63) + 1, 64);  // lane li+1 (li < nr < G: inside the group)
        const uint64_t m = __ballot(li < nr && nxt - st.rp_l > a.split_threshold);
        hub = G == 64 ? m : ((m >> gbase) & ((1ull << (G & 63)) - 1));
    }
    int sa = 0;
    while (sa < nr) {
        const uint64_t rest = hub >> sa;
        if (rest & 1) { sa++; continue; }
        int run = rest ? __builtin_ctzll(rest) : 64;
        int sb = sa + run < nr ? sa + run : nr;
        st.segment(sa, sb);
        sa = sb;
    }
}

#ifdef GNNX_EXPERIMENTS
// ---- LDS-staged variant (measurement only: GNNX_SPMM_VARIANT=lds, F = 256, forward mode) ---------------------------
// Same streaming structure, but the neighbour rows land in LDS through LDS-DMA (global_load_lds_dwordx4: one 1-KiB
// row per wave-instruction, per-lane source address = a row gather) and are summed from there, B = 16 rows per batch,
// two batches of slots per wavefront (the BASELINE north_star's "LDS staging of feature tiles").  No VGPRs are spent on
// rows in flight; a row is still consumed by one wavefront only, i.e. LDS adds a write + a read per byte and shares
// nothing.  Kept to put a number on that choice.  LDS reads go through inline asm: behind a
// pending LDS-DMA hipcc would otherwise insert s_waitcnt vmcnt(0) in front of every ds_read and drain the pipeline.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef float f32x4_lds __attribute__((ext_vector_type(4)));

template <int B>
__global__ __launch_bounds__(64) void spmm_lds_kernel(SpmmArgs a, int32_t n_item_blocks)
{
    constexpr int G = 64, VEC = 4;
    __shared__ float lds[2 * B * 256];
    const int li = threadIdx.x;
    const int32_t f0 = li * VEC;
    const float *xf = a.X + f0;
    if ((int32_t)blockIdx.x < n_item_blocks) {
        int32_t it = blockIdx.x;
        if (it >= a.n_items) return;
        int4 item = a.items[it];
        item.y = __builtin_amdgcn_readfirstlane(item.y);
        item.z = __builtin_amdgcn_readfirstlane(item.z);
        item.w = __builtin_amdgcn_readfirstlane(item.w);
        auto acc = gather_range<G, VEC, 8, 0>(item.y, item.z, xf, li, a, ProConst<VEC>{});
        *reinterpret_cast<float4 *>(a.partial + (int64_t)item.w * a.partial_ld + f0) = acc;
        return;
    }
    const int32_t blk = blockIdx.x - n_item_blocks;
    int32_t r0, nr;
    if (a.block_starts) {
        if (blk >= a.n_blocks) return;
        r0 = __builtin_amdgcn_readfirstlane(a.block_starts[blk]);
        nr = __builtin_amdgcn_readfirstlane(a.block_starts[blk + 1]) - r0;
    } else {
        constexpr int R = StreamCfg<G>::R;
        const int64_t r0l = (int64_t)blk * R;
        if (r0l >= a.n_rows) return;
        r0 = (int32_t)r0l;
        nr = a.n_rows - r0 < R ? a.n_rows - r0 : R;
    }
    const int32_t rp_l = a.rowptr[r0 + (li < nr ? li : nr)];
    const float rs_l = a.rowscale ? a.rowscale[r0 + (li < nr ? li : nr - 1)] : 1.f;
    float4 bias_v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias_v = *reinterpret_cast<const float4 *>(a.bias + f0);
    auto rp = [&](int l) { return __builtin_amdgcn_readlane(rp_l, l); };
    auto flush = [&](float4 &acc, int r) {
        const float rs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs_l), r));
        epilogue_store_pre<VEC>(acc, r0 + r, f0, a, a.rowscale != nullptr, rs, a.bias != nullptr, bias_v);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    uint64_t hub = 0;
    if (a.split_threshold > 0) {
        const int32_t nxt = __shfl(rp_l, li + 1, 64);
        hub = __ballot(li < nr && nxt - rp_l > a.split_threshold);
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void_t *)lds + (uint32_t)f0 * 4u;  // this lane's 16 bytes of slot 0
    auto issue = [&](int buf, int32_t chunk_c, int k0) {
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int32_t c = __builtin_amdgcn_readlane(chunk_c, k0 + u);
            __builtin_amdgcn_global_load_lds(xf + (int64_t)c * a.ldx, (lds_void_t *)(lds + (buf * B + u) * 256), 16, 0, 0);
        }
    };
    int sa = 0;
    while (sa < nr) {
        const uint64_t rest = hub >> sa;
        if (rest & 1) { sa++; continue; }
        const int run = rest ? __builtin_ctzll(rest) : 64;
        const int sb = sa + run < nr ? sa + run : nr;
        const int32_t lo = rp(sa), hi = rp(sb);
        const int32_t total = hi - lo;
        int r = sb - 1;
        int32_t rs = rp(r);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (total > 0) {
            auto fetch = [&](int cidx) {
                int32_t q = hi - 1 - (cidx * G + li);
                return a.colidx[q < lo ? lo : q];
            };
            int32_t cur = fetch(0), nxt = fetch(1);
            int cidx = 1, buf = 0;
            issue(0, cur, 0);
            for (int32_t e = 0; e < total; e += B) {
                const int kn = (e + B) % G;
                if (kn == 0) { cur = nxt; cidx++; nxt = fetch(cidx); }
                issue(buf ^ 1, cur, kn);  // batch e+B in flight (clamped past the end: an in-range row, never added)
                // all but the B DMAs just issued have completed: batch e is in LDS (and older stores have left)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
                for (int u0 = 0; u0 < B; u0 += 4) {
                    f32x4_lds v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(lds_base + (uint32_t)((buf * B + u0 + j) * 1024)) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int u = u0 + j;
                        if (e + u < total) {
                            const int32_t q = hi - 1 - (e + u);
                            while (q < rs) {
                                flush(acc, r);
                                r--;
                                rs = rp(r);
                            }
                            acc = add_rn(acc, make_float4(v[j].x, v[j].y, v[j].z, v[j].w));
                        }
                    }
                }
                buf ^= 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued tail batch must land before its slots are reused
        }
        while (r >= sa) {
            flush(acc, r);
            r--;
        }
        sa = sb;
    }
}

#endif  // GNNX_EXPERIMENTS

// Combine the partial slabs of split rows in chunk order (chunk 0 holds the HIGHEST columns), then the
// same epilogue as the main kernel.  One G-lane group per split row.
template <int G, int VEC>
__global__ __launch_bounds__(256) void spmm_combine_kernel(SpmmArgs a, const int4 *rows, int32_t n_split)
{
    constexpr int GROUPS = 256 / G;
    const int li = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    const int32_t f0 = (blockIdx.y * G + li) * VEC;
    int32_t k = blockIdx.x * GROUPS + grp;
    if (k >= n_split || f0 + VEC > a.n_feat) return;
    using V = typename Vec<VEC>::type;
    int4 r = rows[k];
    V acc = ld_vec(reinterpret_cast<const V *>(a.partial + (int64_t)r.y * a.partial_ld + f0));
    for (int32_t c = 1; c < r.z; c++)
        acc = add_rn(acc, ld_vec(reinterpret_cast<const V *>(a.partial + (int64_t)(r.y + c) * a.partial_ld + f0)));
    epilogue_store<VEC>(acc, r.x, f0, a);
}

// ---- plan construction -------------------------------------------------------------------------
__global__ void plan_count_kernel(const int32_t *rowptr, int32_t n_rows, int32_t chunk, int32_t *counters)
{
    int32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    int32_t deg = rowptr[row + 1] - rowptr[row];
    if (deg > chunk) {
        atomicAdd(&counters[0], 1);
        atomicAdd(&counters[1], (deg + chunk - 1) / chunk);
    }
}

__global__ void plan_fill_kernel(const int32_t *rowptr, int32_t n_rows, int32_t chunk, int32_t *counters, int4 *items,
                                 int4 *rows)
{
    int32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    int32_t b = rowptr[row], e = rowptr[row + 1];
    int32_t deg = e - b;
    if (deg <= chunk) return;
    int32_t nc = (deg + chunk - 1) / chunk;
    int32_t k = atomicAdd(&counters[2], 1);
    int32_t slot = atomicAdd(&counters[3], nc);
    rows[k] = make_int4(row, slot, nc, 0);
    // chunk 0 = the highest columns, so that chunk order == descending column order
    int32_t hi = e;
    for (int32_t c = 0; c < nc; c++) {
        int32_t lo = hi - chunk < b ? b : hi - chunk;
        items[slot + c] = make_int4(row, lo, hi, slot + c);
        hi = lo;
    }
}

// GNNX_SPMM_VARIANT=rows forces the one-row-per-group kernel everywhere (A/B measurements); default: the
// streaming kernel for G >= 32 (F > 64), the row kernel below that (cache-resident widths).
bool use_stream_kernel(int G)
{
    static const int forced = [] {
        const char *v = experiment_env("GNNX_SPMM_VARIANT");
        if (!v) return 0;
        return strcmp(v, "rows") == 0 ? 1 : (strcmp(v, "stream") == 0 ? 2 : 0);
    }();
    if (forced == 1) return false;
    if (forced == 2) return G >= 8;
    return G >= 32;
}

bool use_lds_variant()
{
    static const bool v = [] { const char *e = experiment_env("GNNX_SPMM_VARIANT"); return e && strcmp(e, "lds") == 0; }();
    return v;
}

// Row r opens a new block when it starts a new group of kPlanBlockRows rows or when its first non-zero falls
// into another block_nnz-sized bucket than the previous row's: a block then streams < block_nnz + (its last
// row's degree) non-zeros, whatever the degree skew, and clustered heavy rows end up in blocks of their own.
constexpr int kPlanBlockRows = 16;  // <= StreamCfg<G>::R for every G that uses plan blocks (G >= 32)

__global__ void plan_block_flags_kernel(const int32_t *rowptr, int32_t n_rows, int32_t block_nnz, int32_t *flag)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    bool cut = r % kPlanBlockRows == 0;
    if (!cut) cut = rowptr[r] / block_nnz != rowptr[r - 1] / block_nnz;
    flag[r] = cut ? 1 : 0;
}

__global__ void plan_block_fill_kernel(const int32_t *flag, const int32_t *pos, int32_t n_rows, int32_t n_blocks,
                                       int32_t *starts)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) starts[n_blocks] = n_rows;
    if (r >= n_rows || !flag[r]) return;
    starts[pos[r]] = r;
}

template <int G, int VEC, int U, int TPB>
void launch_stream(int mode, dim3 grid, hipStream_t st, const SpmmArgs &a, int32_t n_item_blocks)
{
#define GNNX_STREAM(M, XT) hipLaunchKernelGGL((spmm_stream_kernel<G, VEC, U, M, TPB, XT>), grid, dim3(TPB), 0, st, a, n_item_blocks)
    if (a.x_bf16) {  // bf16 feature rows: the three plain modes (a prologue goes with f32 rows)
        switch (mode) {
        case 0: GNNX_STREAM(0, bf16_t); break;
        case 1: GNNX_STREAM(1, bf16_t); break;
        case 2: GNNX_STREAM(2, bf16_t); break;
        default: GNNX_STREAM(6, bf16_t); break;
        }
        return;
    }
    switch (mode) {
    case 0: GNNX_STREAM(0, float); break;
    case 1: GNNX_STREAM(1, float); break;
    case 2: GNNX_STREAM(2, float); break;
    case 3: GNNX_STREAM(3, float); break;
    case 4: GNNX_STREAM(4, float); break;
    case 5: GNNX_STREAM(5, float); break;
    default: GNNX_STREAM(6, float); break;
    }
#undef GNNX_STREAM
}

template <int G, int VEC, int U>
void launch_rows(int mode, dim3 grid, hipStream_t st, const SpmmArgs &a, int32_t n_item_blocks)
{
#define GNNX_ROWS(M, XT) hipLaunchKernelGGL((spmm_kernel<G, VEC, U, M, XT>), grid, dim3(256), 0, st, a, n_item_blocks)
    if (a.x_bf16) {
        switch (mode) {
        case 0: GNNX_ROWS(0, bf16_t); break;
        case 1: GNNX_ROWS(1, bf16_t); break;
        case 2: GNNX_ROWS(2, bf16_t); break;
        default: GNNX_ROWS(6, bf16_t); break;
        }
        return;
    }
    switch (mode) {
    case 0: GNNX_ROWS(0, float); break;
    case 1: GNNX_ROWS(1, float); break;
    case 2: GNNX_ROWS(2, float); break;
    case 3: GNNX_ROWS(3, float); break;
    case 4: GNNX_ROWS(4, float); break;
    case 5: GNNX_ROWS(5, float); break;
    default: GNNX_ROWS(6, float); break;
    }
#undef GNNX_ROWS
}

template <int G, int VEC, int U>
int launch_mode(const SpmmArgs &a_in, const gnnx_spmm_plan *plan, int pro, hipStream_t st)
{
    constexpr int GROUPS = 256 / G;
    const int feat_per_tile = G * VEC;
    SpmmArgs a = a_in;
    if (G < 2 * kPlanBlockRows) a.block_starts = nullptr;  // plan blocks hold up to kPlanBlockRows rows (< G needed)
    dim3 grid;
    int32_t n_item_blocks = (int32_t)ceil_div(a.n_items, GROUPS);
    grid.y = (uint32_t)ceil_div(a.n_feat, feat_per_tile);
    // kernel MODE: 0 forward, 1 colscale, 2 vals (the backward: norm streamed per entry), 6 vals + colscale, 3/4/5 forward with
    // a ReLU / BN / BN+ReLU prologue
    const int mode = a.vals != nullptr ? (a.colscale ? 6 : 2) : (a.colscale ? 1 : (pro ? 2 + pro : 0));
    bool stream = false;
    if constexpr (G >= 8) stream = use_stream_kernel(G);
#ifdef GNNX_EXPERIMENTS
    if constexpr (G == 64 && VEC == 4) {
        if (use_lds_variant() && a.n_feat == 256 && mode == 0 && !a.relu_out && !a.x_bf16) {  // measurement variant, forward mode only
            grid.x = (uint32_t)(a.n_items + (a.block_starts ? a.n_blocks : ceil_div(a.n_rows, StreamCfg<64>::R)));
            grid.y = 1;
            hipLaunchKernelGGL((spmm_lds_kernel<16>), grid, dim3(64), 0, st, a, (int32_t)a.n_items);
            GNNX_LAUNCH_CHECK();
            if (plan && plan->n_split_rows > 0) {
                dim3 cgrid((uint32_t)ceil_div(plan->n_split_rows, GROUPS), 1);
                hipLaunchKernelGGL((spmm_combine_kernel<G, VEC>), cgrid, dim3(256), 0, st, a, plan->d_rows, plan->n_split_rows);
                GNNX_LAUNCH_CHECK();
            }
            return GNNX_OK;
        }
    }
#endif
    if (stream) {
        if constexpr (G >= 8) {
            constexpr int R = StreamCfg<G>::R;
            // One wavefront per workgroup: a workgroup's CU slot is held until its slowest wavefront ends, and
            // blocks of a power-law graph differ a lot in non-zeros, so multi-wave workgroups strand slots.
            static const int tpb_env = [] { const char *v = experiment_env("GNNX_SPMM_TPB"); return v ? atoi(v) : 64; }();
            const int tpb = tpb_env == 256 ? 256 : 64;
            n_item_blocks = (int32_t)ceil_div(a.n_items, tpb / G);
            grid.x = (uint32_t)(n_item_blocks + (a.block_starts ? ceil_div(a.n_blocks, tpb / G)
                                                                : ceil_div(a.n_rows, (int64_t)(tpb / G) * R)));
            if (tpb == 256) launch_stream<G, VEC, U, 256>(mode, grid, st, a, n_item_blocks);
            else launch_stream<G, VEC, U, 64>(mode, grid, st, a, n_item_blocks);
        }
    } else {
        grid.x = (uint32_t)(n_item_blocks + ceil_div(a.n_rows, GROUPS));
        if (grid.x == 0) return GNNX_OK;
        launch_rows<G, VEC, U>(mode, grid, st, a, n_item_blocks);
    }
    GNNX_LAUNCH_CHECK();
    if (plan && plan->n_split_rows > 0) {
        dim3 cgrid((uint32_t)ceil_div(plan->n_split_rows, GROUPS), grid.y);
        hipLaunchKernelGGL((spmm_combine_kernel<G, VEC>), cgrid, dim3(256), 0, st, a, plan->d_rows, plan->n_split_rows);
        GNNX_LAUNCH_CHECK();
    }
    return GNNX_OK;
}

}  // namespace

GNNX_API int gnnx_spmm_plan_create(const int32_t *d_rowptr, int32_t n_rows, int32_t chunk, int32_t max_feat,
                                   gnnx_spmm_plan **plan_out, void *stream)
{
    GNNX_REQUIRE(d_rowptr && plan_out, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_REQUIRE(n_rows >= 0 && chunk > 0 && max_feat > 0, GNNX_ERR_INVALID_ARG, "n_rows/chunk/max_feat");
    hipStream_t st = as_stream(stream);
    auto *plan = new gnnx_spmm_plan();
    plan->n_rows = n_rows;
    plan->chunk = chunk;
    plan->max_feat = (max_feat + 3) & ~3;
    *plan_out = plan;
    GNNX_HIP_CHECK(hipMalloc(&plan->d_counters, 4 * sizeof(int32_t)));
    GNNX_HIP_CHECK(hipMemsetAsync(plan->d_counters, 0, 4 * sizeof(int32_t), st));
    if (n_rows == 0) return GNNX_OK;
    dim3 grid((uint32_t)ceil_div(n_rows, 256));
    hipLaunchKernelGGL(plan_count_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, chunk, plan->d_counters);
    GNNX_LAUNCH_CHECK();
    int32_t h[4];
    GNNX_HIP_CHECK(hipMemcpyAsync(h, plan->d_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    plan->n_split_rows = h[0];
    plan->n_chunks = h[1];
    if (plan->n_split_rows > 0) {
        GNNX_HIP_CHECK(hipMalloc(&plan->d_items, sizeof(int4) * (size_t)plan->n_chunks));
        GNNX_HIP_CHECK(hipMalloc(&plan->d_rows, sizeof(int4) * (size_t)plan->n_split_rows));
        GNNX_HIP_CHECK(hipMalloc(&plan->d_partial, sizeof(float) * (size_t)plan->n_chunks * plan->max_feat));
        hipLaunchKernelGGL(plan_fill_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, chunk, plan->d_counters,
                           plan->d_items, plan->d_rows);
        GNNX_LAUNCH_CHECK();
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
    }
    // non-zero-balanced row blocks
    {
        static const int env_bn = [] { const char *v = experiment_env("GNNX_SPMM_BLOCK_NNZ"); return v ? atoi(v) : 0; }();
        plan->block_nnz = env_bn > 0 ? env_bn : (chunk < 256 ? chunk : 256);
        int32_t *flag = nullptr, *pos = nullptr;
        void *tmp = nullptr;
        size_t tmp_bytes = 0;
        GNNX_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)n_rows,
                                               rocprim::plus<int32_t>()));
        GNNX_HIP_CHECK(hipMalloc(&flag, sizeof(int32_t) * (size_t)n_rows));
        GNNX_HIP_CHECK(hipMalloc(&pos, sizeof(int32_t) * (size_t)n_rows));
        GNNX_HIP_CHECK(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 4));
        hipLaunchKernelGGL(plan_block_flags_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, plan->block_nnz, flag);
        GNNX_LAUNCH_CHECK();
        GNNX_HIP_CHECK(rocprim::exclusive_scan(tmp, tmp_bytes, flag, pos, 0, (size_t)n_rows, rocprim::plus<int32_t>(), st));
        int32_t last[2];
        GNNX_HIP_CHECK(hipMemcpyAsync(&last[0], pos + (n_rows - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipMemcpyAsync(&last[1], flag + (n_rows - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        plan->n_blocks = last[0] + last[1];
        GNNX_HIP_CHECK(hipMalloc(&plan->d_block_starts, sizeof(int32_t) * ((size_t)plan->n_blocks + 1)));
        hipLaunchKernelGGL(plan_block_fill_kernel, grid, dim3(256), 0, st, flag, pos, n_rows, plan->n_blocks,
                           plan->d_block_starts);
        GNNX_LAUNCH_CHECK();
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFree(flag);
        (void)hipFree(pos);
        (void)hipFree(tmp);
    }
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_destroy(gnnx_spmm_plan *plan)
{
    if (!plan) return GNNX_OK;
    if (plan->d_items) (void)hipFree(plan->d_items);
    if (plan->d_rows) (void)hipFree(plan->d_rows);
    if (plan->d_partial) (void)hipFree(plan->d_partial);
    if (plan->d_counters) (void)hipFree(plan->d_counters);
    if (plan->d_block_starts) (void)hipFree(plan->d_block_starts);
    delete plan;
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_info(const gnnx_spmm_plan *plan, int64_t *n_split_rows, int64_t *n_chunks)
{
    GNNX_REQUIRE(plan, GNNX_ERR_INVALID_ARG, "plan is null");
    if (n_split_rows) *n_split_rows = plan->n_split_rows;
    if (n_chunks) *n_chunks = plan->n_chunks;
    return GNNX_OK;
}

namespace {
int spmm_impl(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr, const int32_t *d_colidx, const float *d_vals,
              const float *d_colscale, const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx, float beta,
              float *d_Y, int64_t ldy, const gnnx_spmm_fusion *fusion, const gnnx_spmm_plan *plan, void *stream, bool x_bf16 = false)
{
    GNNX_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n_rows == 0 || n_feat == 0) return GNNX_OK;
    GNNX_REQUIRE(d_rowptr && d_X && d_Y, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_REQUIRE(ldx >= n_feat && ldy >= n_feat, GNNX_ERR_SHAPE, "leading dimension smaller than n_feat");
    GNNX_REQUIRE(beta == 0.f || beta == 1.f, GNNX_ERR_UNSUPPORTED, "beta must be 0 or 1");
    GNNX_REQUIRE(d_X != d_Y, GNNX_ERR_INVALID_ARG, "X and Y alias");
    GNNX_REQUIRE(!x_bf16 || !fusion, GNNX_ERR_UNSUPPORTED, "bf16 feature rows take no fusion");
    if (plan) {
        GNNX_REQUIRE(plan->n_rows == n_rows, GNNX_ERR_SHAPE, "plan was built for %d rows, got %d", plan->n_rows, n_rows);
        GNNX_REQUIRE(plan->n_split_rows == 0 || n_feat <= plan->max_feat, GNNX_ERR_SHAPE,
                     "plan was built for at most %d features, got %d", plan->max_feat, n_feat);
    }
    SpmmArgs a{};
    int pro = 0;  // 0 none, 1 ReLU, 2 BatchNorm, 3 BatchNorm + ReLU
    if (fusion) {
        const bool bn = fusion->bn_mean != nullptr;
        GNNX_REQUIRE((fusion->bn_mean == nullptr) == (fusion->bn_var == nullptr), GNNX_ERR_INVALID_ARG, "bn_mean and bn_var go together");
        GNNX_REQUIRE(bn || (!fusion->bn_gamma && !fusion->bn_beta), GNNX_ERR_INVALID_ARG, "bn_gamma / bn_beta without bn_mean");
        pro = bn ? (fusion->relu_in ? 3 : 2) : (fusion->relu_in ? 1 : 0);
        GNNX_REQUIRE(pro == 0 || (!d_vals && !d_colscale), GNNX_ERR_UNSUPPORTED, "a prologue goes with the forward mode only");
        a.relu_out = fusion->relu_out != 0;
        a.pro_mean = fusion->bn_mean;
        a.pro_var = fusion->bn_var;
        a.pro_gamma = fusion->bn_gamma;
        a.pro_beta = fusion->bn_beta;
        a.pro_eps = fusion->bn_eps;
    }
    a.x_bf16 = x_bf16;
    a.n_rows = n_rows;
    a.n_feat = n_feat;
    a.rowptr = d_rowptr;
    a.colidx = d_colidx;
    a.vals = d_vals;
    a.colscale = d_colscale;
    a.rowscale = d_rowscale;
    a.bias = d_bias;
    a.X = d_X;
    a.ldx = ldx;
    a.Y = d_Y;
    a.ldy = ldy;
    a.beta = beta != 0.f;
    if (plan && plan->d_block_starts) {
        a.block_starts = plan->d_block_starts;
        a.n_blocks = plan->n_blocks;
    }
    if (plan && plan->n_split_rows > 0) {
        a.split_threshold = plan->chunk;
        a.items = plan->d_items;
        a.n_items = plan->n_chunks;
        a.partial = plan->d_partial;
        a.partial_ld = plan->max_feat;
    }
    hipStream_t st = as_stream(stream);
    auto aligned16 = [](const void *p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec4 = (n_feat % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (x_bf16 ? (reinterpret_cast<uintptr_t>(d_X) & 7u) == 0 : aligned16(d_X)) && aligned16(d_Y) &&
                      aligned16(d_bias) && aligned16(a.pro_mean) && aligned16(a.pro_var) && aligned16(a.pro_gamma) &&
                      aligned16(a.pro_beta);
    if (vec4) {
        int lanes = n_feat / 4;
        if (lanes > 32) return launch_mode<64, 4, 8>(a, plan, pro, st);
        if (lanes > 16) return launch_mode<32, 4, 8>(a, plan, pro, st);
        if (lanes > 8) return launch_mode<16, 4, 8>(a, plan, pro, st);
        if (lanes > 4) return launch_mode<8, 4, 8>(a, plan, pro, st);
        return launch_mode<4, 4, 8>(a, plan, pro, st);
    }
    if (n_feat > 32) return launch_mode<64, 1, 8>(a, plan, pro, st);
    if (n_feat > 16) return launch_mode<32, 1, 8>(a, plan, pro, st);
    if (n_feat > 8) return launch_mode<16, 1, 8>(a, plan, pro, st);
    if (n_feat > 4) return launch_mode<8, 1, 8>(a, plan, pro, st);
    return launch_mode<4, 1, 8>(a, plan, pro, st);
}
}  // namespace

GNNX_API int gnnx_spmm_csr_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                               const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                               const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx,
                               float beta, float *d_Y, int64_t ldy, const gnnx_spmm_plan *plan, void *stream)
{
    return spmm_impl(n_rows, n_cols, n_feat, d_rowptr, d_colidx, d_vals, d_colscale, d_rowscale, d_bias, d_X, ldx, beta, d_Y, ldy,
                     nullptr, plan, stream);
}

GNNX_API int gnnx_spmm_csr_fused_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                                     const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                                     const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx, float beta,
                                     float *d_Y, int64_t ldy, const gnnx_spmm_fusion *fusion, const gnnx_spmm_plan *plan,
                                     void *stream)
{
    return spmm_impl(n_rows, n_cols, n_feat, d_rowptr, d_colidx, d_vals, d_colscale, d_rowscale, d_bias, d_X, ldx, beta, d_Y, ldy,
                     fusion, plan, stream);
}

GNNX_API int gnnx_spmm_csr_bf16_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                                    const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                                    const float *d_rowscale, const float *d_bias, const uint16_t *d_X_bf16, int64_t ldx, float beta,
                                    float *d_Y, int64_t ldy, const gnnx_spmm_plan *plan, void *stream)
{
    return spmm_impl(n_rows, n_cols, n_feat, d_rowptr, d_colidx, d_vals, d_colscale, d_rowscale, d_bias,
                     reinterpret_cast<const float *>(d_X_bf16), ldx, beta, d_Y, ldy, nullptr, plan, stream, true);
}
