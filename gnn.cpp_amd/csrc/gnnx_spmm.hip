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
//   * power-law rows: an optional plan hands rows longer than `chunk` to spmm_hub_kernel -- one wavefront per (row,
//     64-feature slab), the neighbour rows' slices in flight in an LDS ring filled by LDS-DMA, ONE accumulator per feature
//     in the reference's order -- and cuts the remaining rows into non-zero-balanced blocks for the streaming kernel.
//     A planned aggregation is bit-identical to the unplanned one on every row.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "gnnx_common.h"

// Parity depends on separately rounded fp32 mul / add (the reference has no FMA): never contract.
#pragma clang fp contract(off)

using namespace gnnx;

struct gnnx_spmm_plan {
    int32_t n_rows = 0;
    int32_t chunk = 0;
    int32_t n_split_rows = 0;   // rows with degree > chunk (the hub rows)
    int64_t n_hub_nnz = 0;      // their non-zeros
    int64_t nnz = 0;            // all non-zeros (rowptr[n_rows])
    int32_t max_hub_degree = 0;
    // the longest rows of the list go to the producer / consumer kernel (spmm_hubpc_kernel): rows whose time is their own chain of
    // dependent adds, not their bytes.  How many is decided per call (big_rows_for: it depends on the feature width) unless a caller
    // fixed the threshold (gnnx_spmm_plan_set_big_row_threshold, >= 0)
    int32_t big_row_threshold = -1;
    std::vector<int32_t> h_hub_degrees;  // host copy of the sorted degrees, longest first
    std::vector<int64_t> h_hub_prefix;   // h_hub_prefix[k] = non-zeros of the k longest rows
    std::vector<int32_t> h_hub_rows;     // host copy of the hub rows' ids, longest first (gnnx_spmm_plan_hub_ids_structured)
    int32_t *d_hub_rows = nullptr;  // [n_split_rows] the hub rows, longest first: work list of the hub kernels
    unsigned long long *d_counters = nullptr;
    // error word of the producer / consumer kernel (a wait that timed out): pinned host memory the kernels write through its device
    // address, read by the host -- without any synchronisation -- at the plan's next call and at gnnx_spmm_plan_status()
    int32_t *h_err = nullptr, *d_err = nullptr;
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
    int32_t split_threshold; // rows with degree > this are left to the hub kernel (0 = none)
    const int32_t *hub_rows; // the plan's hub rows, longest first
    int32_t n_hub_rows;
    int32_t hub_beside;      // run the hub kernel on the side stream, beside the row kernel (its time is one row's add chain)
    int32_t n_big_rows;      // the first n_big_rows hub rows take the producer / consumer kernel (f32 rows of 16-byte pieces)
    int32_t pc_experiment;   // EXPERIMENTS build only (GNNX_PC_EXP): 1 = the consumer does not wait for the producers (timing only)
    int32_t exp_policy;      // EXPERIMENTS build only (GNNX_SPMM_POLICY): cache policy of the streaming kernel's one-touch traffic -- 1: colidx / vals
                             // loads nontemporal, 2: Y stores nontemporal, 4: Y stores sc1 (write-through, the line leaves L2),
                             // 8: gathers of rows >= exp_hot_k nontemporal (hot-first vertex labels: GNNX_SPMM_HOTK)
    int32_t exp_hot_k;
    int32_t *err_word;       // device address of the plan's error word (spmm_hubpc_kernel: a wait that timed out); never null with big rows
    // row blocks of the streaming kernel (plan): nullptr => fixed blocks of StreamCfg<G>::R rows
    const int32_t *block_starts;
    int32_t n_blocks;
    // fusion (gnnx_spmm_csr_fused_f32): ReLU on the stored row; BatchNorm / ReLU applied to every gathered row of X
    int32_t relu_out;
    int32_t x_bf16;          // X holds bf16 (gnnx_spmm_csr_bf16_f32)
    const float *pro_mean, *pro_var, *pro_gamma, *pro_beta;
    float pro_eps;
    // gnnx_spmm_csr_bn_sums_f32 (the backward aggregation of a fused BatchNorm + ReLU layer): besides dY = A^T . (vals (.) G), the
    // column sums dbeta = sum_i g_i and dgamma = sum_i g_i xhat_i of BatchNorm's backward (g = dY where relu(BN(h)) > 0) are
    // accumulated by the wavefront that stores the rows of dY: bn_h = the layer's pre-BatchNorm activations H, statistics and
    // affine parameters in the pro_* fields above, one partial row [2][F] per streaming wavefront / hub row in bn_partial.
    const float *bn_h;
    int64_t bn_ldh;
    float *bn_partial;
    int32_t bn_relu;
    int32_t blocks_per_wave;   // row blocks one streaming wavefront walks (1 unless bn sums: fewer, longer partial rows)
    int32_t n_stream_waves;    // partial rows [0, n_stream_waves) belong to the streaming kernel, the hub rows follow
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
// rsd[i] = RN_f64(1 / sd_i): the divisor of the normalisation is a per-column constant (div_by_const below)
template <int VEC> struct ProConst { typename Vec<VEC>::type mean, sd, gamma, beta; double rsd[VEC]; };

template <int MODE>
__device__ __forceinline__ float pro1(float v, float mean, float sd, double rsd, float gamma, float beta)
{
    if constexpr (pro_bn(MODE)) {
        v = div_by_const(__fsub_rn(v, mean), sd, rsd);
        v = __fmul_rn(v, gamma);
        v = __fadd_rn(v, beta);
    }
    if constexpr (pro_relu(MODE)) v = relu1(v);
    return v;
}
template <int MODE>
__device__ __forceinline__ float pro_apply(float v, const ProConst<1> &c) { return pro1<MODE>(v, c.mean, c.sd, c.rsd[0], c.gamma, c.beta); }
template <int MODE>
__device__ __forceinline__ float4 pro_apply(float4 v, const ProConst<4> &c)
{
    return make_float4(pro1<MODE>(v.x, c.mean.x, c.sd.x, c.rsd[0], c.gamma.x, c.beta.x), pro1<MODE>(v.y, c.mean.y, c.sd.y, c.rsd[1], c.gamma.y, c.beta.y),
                       pro1<MODE>(v.z, c.mean.z, c.sd.z, c.rsd[2], c.gamma.z, c.beta.z), pro1<MODE>(v.w, c.mean.w, c.sd.w, c.rsd[3], c.gamma.w, c.beta.w));
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
#pragma unroll
    for (int i = 0; i < VEC; i++) c.rsd[i] = 1.0;
    if constexpr (pro_bn(MODE)) {
        if (active) {
            c.mean = ld_vec(reinterpret_cast<const V *>(a.pro_mean + f0));
            c.sd = sd_of(ld_vec(reinterpret_cast<const V *>(a.pro_var + f0)), a.pro_eps);
            if constexpr (VEC == 4) {
                c.rsd[0] = 1.0 / (double)c.sd.x;
                c.rsd[1] = 1.0 / (double)c.sd.y;
                c.rsd[2] = 1.0 / (double)c.sd.z;
                c.rsd[3] = 1.0 / (double)c.sd.w;
            } else {
                c.rsd[0] = 1.0 / (double)c.sd;
            }
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
#ifdef GNNX_EXPERIMENTS
    if constexpr (VEC == 4) {
        if (a.exp_policy & 2) {
            __builtin_nontemporal_store(acc.x, &dst->x); __builtin_nontemporal_store(acc.y, &dst->y);
            __builtin_nontemporal_store(acc.z, &dst->z); __builtin_nontemporal_store(acc.w, &dst->w);
            return;
        }
        if (a.exp_policy & 4) {
            typedef float v4f_store __attribute__((ext_vector_type(4)));
            const v4f_store pk = {acc.x, acc.y, acc.z, acc.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(pk) : "memory");
            return;
        }
    }
#endif
    *dst = acc;
}

// grid.x = row blocks ; grid.y = feature tiles of G*VEC features.
template <int G, int VEC, int U, int MODE, class XT>
__global__ __launch_bounds__(256) void spmm_kernel(SpmmArgs a)
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

    int32_t row = blockIdx.x * GROUPS + grp;
    if (row >= a.n_rows) return;
    int32_t b = a.rowptr[row], e = a.rowptr[row + 1];
    if constexpr (G == 64) {
        b = __builtin_amdgcn_readfirstlane(b);
        e = __builtin_amdgcn_readfirstlane(e);
    }
    if (a.split_threshold > 0 && e - b > a.split_threshold) return;  // a hub row: spmm_hub_kernel owns it
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
//   * rows longer than the plan's threshold are skipped here (spmm_hub_kernel owns them): the
//     block is cut into segments of consecutive non-hub rows with a ballot mask.
// At G == 64 every control decision is wave-uniform (SALU + s_cbranch); at G == 32 the two half-waves
// stream independent blocks under the EXEC mask.
template <int G> struct StreamCfg { static constexpr int R = G / 2; };

// BatchNorm-backward column sums over rows [row_lo, row_hi) whose dY this wavefront has just stored (VEC 4: a lane owns features
// f0 .. f0+3).  Same per-element arithmetic as bn_bwd_vec_kernel<0> (gnnx_norm.hip): the ReLU mask is recomputed from h with the
// forward's separately rounded sub / div / mul / add, xhat = (h - mean) * rstd.  The rows of dY come back from L2 (they were written
// a moment ago; the stores are waited for first), the rows of H from HBM -- this replaces the separate sums pass over dY and H.
__device__ __forceinline__ void bn_sums_rows(const SpmmArgs &a, int32_t row_lo, int32_t row_hi, int32_t f0, float4 &s_beta, float4 &s_gamma)
{
    const float4 mean4 = *reinterpret_cast<const float4 *>(a.pro_mean + f0);
    const float4 var4 = *reinterpret_cast<const float4 *>(a.pro_var + f0);
    float4 gm4 = make_float4(1.f, 1.f, 1.f, 1.f), bt4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.pro_gamma) gm4 = *reinterpret_cast<const float4 *>(a.pro_gamma + f0);
    if (a.pro_beta) bt4 = *reinterpret_cast<const float4 *>(a.pro_beta + f0);
    const float mean[4] = {mean4.x, mean4.y, mean4.z, mean4.w}, var[4] = {var4.x, var4.y, var4.z, var4.w};
    const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w}, bt[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
    float sd[4], rstd[4], a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    double rsd[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        sd[c] = sqrtf(__fadd_rn(var[c], a.pro_eps));
        rsd[c] = 1.0 / (double)sd[c];
        rstd[c] = 1.0f / sqrtf(var[c] + a.pro_eps);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores of dY have reached L2
    for (int32_t rb = row_lo; rb < row_hi; rb += 4) {
        float4 d[4], h[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int32_t r = rb + u < row_hi ? rb + u : row_hi - 1;   // no predicated loads: the tail re-reads the last row
            d[u] = *reinterpret_cast<const float4 *>(a.Y + (int64_t)r * a.ldy + f0);
            h[u] = *reinterpret_cast<const float4 *>(a.bn_h + (int64_t)r * a.bn_ldh + f0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (rb + u < row_hi) {
                const float ds[4] = {d[u].x, d[u].y, d[u].z, d[u].w}, hs[4] = {h[u].x, h[u].y, h[u].z, h[u].w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float g = ds[c];
                    if (a.bn_relu) {
                        float v = div_by_const(__fsub_rn(hs[c], mean[c]), sd[c], rsd[c]);   // the forward's quotient, same bits
                        if (a.pro_gamma) v = __fmul_rn(v, gm[c]);
                        if (a.pro_beta) v = __fadd_rn(v, bt[c]);
                        if (!(v > 0.f)) g = 0.f;
                    }
                    const float xhat = (hs[c] - mean[c]) * rstd[c];
                    a0[c] += g;
                    a1[c] += g * xhat;
                }
            }
        }
    }
    s_beta.x += a0[0]; s_beta.y += a0[1]; s_beta.z += a0[2]; s_beta.w += a0[3];
    s_gamma.x += a1[0]; s_gamma.y += a1[1]; s_gamma.z += a1[2]; s_gamma.w += a1[3];
}

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
        if constexpr (has_val(MODE)) ch.val = a.vals[q];
#ifdef GNNX_EXPERIMENTS
        if (a.exp_policy & 1) {
            ch.c = __builtin_nontemporal_load(a.colidx + q);
            if constexpr (has_val(MODE)) ch.val = __builtin_nontemporal_load(a.vals + q);
        }
#endif
        if constexpr (has_sc(MODE)) ch.sc = a.colscale[ch.c];
        return ch;
    }

    struct Batch { V v[B]; float sc[B]; float val[B]; };

    __device__ __forceinline__ void issue(Batch &b, const Chunk &ch, int k0) const
    {
        int32_t c[B];
#pragma unroll
        for (int u = 0; u < B; u++) c[u] = bcast<G>(ch.c, k0 + u, gbase);
#ifdef GNNX_EXPERIMENTS
        if constexpr (VEC == 4 && sizeof(XT) == 4) {
            if (a.exp_policy & 8) {   // cold rows (labels >= exp_hot_k) with the streaming hint: they should not push the hot rows out of L2
#pragma unroll
                for (int u = 0; u < B; u++) {
                    const float *p = reinterpret_cast<const float *>(xf) + (int64_t)c[u] * a.ldx;
                    if (c[u] < a.exp_hot_k) {
                        b.v[u] = ld_x<VEC>(p);
                    } else {
                        typedef float v4f_ld __attribute__((ext_vector_type(4)));
                        const v4f_ld t = __builtin_nontemporal_load(reinterpret_cast<const v4f_ld *>(p));
                        b.v[u] = make_float4(t.x, t.y, t.z, t.w);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < B; u++) b.v[u] = ld_x<VEC>(xf + (int64_t)c[u] * a.ldx);
            }
        } else
#endif
        {
#pragma unroll
            for (int u = 0; u < B; u++) b.v[u] = ld_x<VEC>(xf + (int64_t)c[u] * a.ldx);
        }
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

template <int G, int VEC, int B, int MODE, int TPB, class XT, bool SUMS = false>
__global__ __launch_bounds__(TPB) void spmm_stream_kernel(SpmmArgs a)
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
    const int gbase = (tid & 63) - li;
    typename Vec<VEC>::type bias_v;
    zero(bias_v);
    if (a.bias && active) bias_v = ld_vec(reinterpret_cast<const typename Vec<VEC>::type *>(a.bias + f0));

    // a wavefront (a G-lane group) walks blocks_per_wave consecutive blocks: 1, or a few when it also carries the BatchNorm column
    // sums of the rows it stores (one partial row per wavefront instead of one per block)
    const int32_t wave_id = blockIdx.x * GROUPS + grp;
    const int32_t nbw = SUMS ? a.blocks_per_wave : 1;
    float4 s_beta = make_float4(0.f, 0.f, 0.f, 0.f), s_gamma = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int32_t bi = 0; bi < nbw; bi++) {
        const int32_t blk = wave_id * nbw + bi;
        int32_t r0, nr;
        if (a.block_starts) {  // non-zero-balanced blocks of the plan (at most kPlanBlockRows <= R rows each)
            if (blk >= a.n_blocks) break;
            r0 = a.block_starts[blk];
            nr = a.block_starts[blk + 1] - r0;
            if constexpr (G == 64) {
                r0 = __builtin_amdgcn_readfirstlane(r0);
                nr = __builtin_amdgcn_readfirstlane(nr);
            }
        } else {
            const int64_t r0l = (int64_t)blk * R;
            if (r0l >= a.n_rows) break;
            r0 = (int32_t)r0l;
            nr = a.n_rows - r0 < R ? a.n_rows - r0 : R;
        }
        const float rs_l = a.rowscale ? a.rowscale[r0 + (li < nr ? li : nr - 1)] : 1.f;
        Stream<G, VEC, B, MODE, XT> st{a, xf, li, gbase, f0, active, r0, a.rowptr[r0 + (li < nr ? li : nr)], rs_l, bias_v, pc};

        uint64_t hub = 0;  // bit l: local row l is a hub (left to spmm_hub_kernel)
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
            if constexpr (SUMS && VEC == 4)
                if (active) bn_sums_rows(a, r0 + sa, r0 + sb, f0, s_beta, s_gamma);
            sa = sb;
        }
    }
    if constexpr (SUMS && VEC == 4) {
        if (active) {
            float *prow = a.bn_partial + (int64_t)wave_id * 2 * a.n_feat;
            *reinterpret_cast<float4 *>(prow + f0) = s_beta;
            *reinterpret_cast<float4 *>(prow + a.n_feat + f0) = s_gamma;
        }
    }
}


// ---- hub rows in the reference's order ---------------------------------------------------------------------------------
// A row of 10^4..10^5 non-zeros summed by ONE accumulator per feature, top column first -- the reference's
// `(r_slice * l_slice).sum()` (functional.h:433-439) -- is a chain of dependent adds fed by a gather: with the 16 neighbour rows a
// streaming wavefront keeps in flight in registers it is latency-bound (tens of ms for the longest row of the RMAT 10M graph).
// Here nothing in flight lives in a register: a wavefront owns (hub row, 64-feature slab), lane l is feature slab*64 + l, and the
// slab's 256-byte slices of the neighbour rows land in an LDS ring by LDS-DMA (global_load_lds_dwordx4: 16 lanes per slice, four
// neighbours per wave-instruction, two whole 128-byte lines each), LAS sub-chunks of 16 neighbours ahead of the adds (LAS = 8:
// 128 neighbours = 32 KB in flight per wavefront, 4 wavefronts per CU).  The consumer side is one conflict-free ds_read_b32 and one
// add per neighbour, strictly in descending column order: bit-identical to the unsplit row whatever the degree.  Nothing is shared
// between wavefronts: no barrier, no atomics, no partial slab, no combine pass.
//   * indices (and per-entry values / gathered column scales) travel the same way: one 256-byte LDS-DMA per 64 neighbours (the
//     index chunk) into a small ring, 2 DI + 1 chunks ahead of the adds, so that by the time a DMA lane reads its neighbour's
//     column from LDS the chunk is older than every wait below;
//   * every wait is counted by hand: s_waitcnt vmcnt(LAS * IPS) after an issue leaves the LAS youngest sub-chunks in flight
//     (VMEM returns in order; the index DMAs in between only make the wait stricter), the tail drains with vmcnt(0);
//   * all LDS reads go through inline asm: behind a pending LDS-DMA hipcc would put s_waitcnt vmcnt(0) in front of every ds_read,
//     and a register load in flight across the loop's back edge gets copied (and waited for) by the register allocator.
// Work items are (row, slab) pairs, rows longest first (gnnx_spmm_plan_create sorts them).
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int kHubSlab = 64;   // features per work item
constexpr int kHubSub = 16;    // neighbours per sub-chunk: 4 KiB of LDS
constexpr int kHubChunk = 64;  // neighbours per index chunk
// no row shorter than this takes the producer / consumer kernel (spmm_hubpc_kernel): its start-up (flags, first index chunks, a CU
// of its own) is worth it only for a chain of thousands of adds
constexpr int kHubBigRowMin = 4096;

template <int OFF, class T>
__device__ __forceinline__ void hub_lds_read(T &dst, uint32_t addr)
{
    static_assert(sizeof(T) == 4 && OFF >= 0 && OFF < 65536, "one dword, 16-bit immediate offset");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
// one feature of a neighbour's slice as stored: an f32 word, or a bf16 halfword (zero-extended; widened after the wait)
template <int OFF, class XT>
__device__ __forceinline__ void hub_lds_read_x(float &dst, uint32_t addr)
{
    if constexpr (sizeof(XT) == 4) hub_lds_read<OFF>(dst, addr);
    else asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void hub_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// DMA shapes (bytes per lane / lanes per 64-feature slice / neighbours per wave-instruction):
//   f32 rows, 16-byte aligned (VEC 4): 16 / 16 / 4      f32 rows, any alignment (VEC 1): 4 / 64 / 1
//   bf16 rows, 8-byte aligned (VEC 4):  4 / 32 / 2      bf16 rows, any alignment (VEC 1): 2 / 64 / 1
// (SLAB: features per work item.  Narrower slabs -- 16 features, 64-byte slices, 384 neighbours in flight -- were tried for the
// longest rows and are slower: a row's time is its chain of dependent adds and the per-neighbour issue work, ~12 ns per neighbour
// whatever is in flight, so more wavefronts per row only repeat that work.)
template <int VEC, int MODE, int LAS, class XT, int SLAB = kHubSlab>
struct HubCfg {
    static constexpr int EB = (int)sizeof(XT);          // bytes per stored feature
    static constexpr int DS = VEC == 4 ? (EB == 4 ? 16 : 4) : EB;  // bytes per lane of one DMA instruction
    static constexpr int SB = SLAB * EB;                // bytes of a neighbour's slice (= its stride in the ring)
    static constexpr int EPI = 64 * DS / SB;            // neighbours per DMA wave-instruction
    static constexpr int LPE = 64 / EPI;                // lanes per neighbour
    static constexpr int FPL = DS / EB;                 // features per lane of a DMA instruction
    static constexpr int IPS = kHubSub / EPI;           // DMA instructions per sub-chunk
    static constexpr int IFLOATS = 64 * DS / 4;         // floats of LDS one DMA instruction fills
    static constexpr int SUB_FLOATS = kHubSub * SB / 4; // floats of a ring slot
    static constexpr int NS = LAS + 1;                  // ring slots: LAS in flight + the one being added
    static constexpr int SUBS = kHubChunk / kHubSub;    // sub-chunks per index chunk
    static constexpr int DI = (LAS + SUBS - 1) / SUBS;  // furthest index chunk (relative to the one being added) an issue reaches
    static constexpr int NC = 2 * DI + 1;               // index chunks fetched ahead: DI whole iterations older than their first use
    static constexpr int NI = NC + 1;                   // index ring slots (the chunk being added is still read for its values)
    static constexpr int RING = NS * SUB_FLOATS;                          // floats
    static constexpr int IR = RING;                                       // column indices
    static constexpr int VR = IR + NI * kHubChunk;                        // per-entry values (MODE 2 / 6)
    static constexpr int SR = VR + (has_val(MODE) ? NI * kHubChunk : 0);  // gathered column scales (MODE 1 / 6)
    static constexpr int LDS_FLOATS = SR + (has_sc(MODE) ? NI * kHubChunk : 0);
    static_assert(LAS * IPS <= 60, "vmcnt is a 6-bit counter");
    static_assert(SUBS * DI * IPS >= LAS * IPS, "an index chunk must be older than the counted wait when it is first read");
};

template <int VEC, int MODE, int LAS, class XT, bool SUMS = false, int SLAB = kHubSlab>
__global__ __launch_bounds__(256) void spmm_hub_kernel(SpmmArgs a, const int32_t *hub_rows, int32_t n_slabs, int32_t n_groups, int32_t ordinal0)
{
    using K = HubCfg<VEC, MODE, LAS, XT, SLAB>;
    constexpr int EPI = K::EPI, IPS = K::IPS, NS = K::NS, SUBS = K::SUBS, DI = K::DI, NC = K::NC, NI = K::NI;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    // a workgroup = the slabs of one row (up to 4 wavefronts, each with a ring of its own): nothing is shared and there is no
    // barrier, but wavefronts that start together and do the same work ask for the pieces of a neighbour row at about the same time
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float *lds = lds_all + wv * K::LDS_FLOATS;
    const int lane = threadIdx.x & 63;
    const int32_t row = __builtin_amdgcn_readfirstlane(hub_rows[blockIdx.x / n_groups]);
    const int32_t slab = (int32_t)(blockIdx.x % n_groups) * (int32_t)(blockDim.x >> 6) + wv;
    if (slab >= n_slabs) return;
    const int32_t f_slab = slab * SLAB;
    const int32_t lo = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
    const int32_t hi = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
    const int32_t total = hi - lo;
    const int32_t nsub = (total + kHubSub - 1) / kHubSub;
    const int32_t f = f_slab + (lane & (SLAB - 1));     // a narrow slab leaves lanes >= SLAB idle in the adds (they mirror lane & 15)
    const bool active = f < a.n_feat && lane < SLAB;
    // DMA source of this lane: neighbour sub_e of the instruction, FPL features at feature foff.  Lanes past the row's width
    // re-read the slab's first piece (never a byte outside the row); their LDS words are never stored.
    const int sub_e = lane / K::LPE;
    int32_t foff = f_slab + (lane % K::LPE) * K::FPL;
    if (foff + K::FPL > a.n_feat) foff = f_slab;
    const char *xsrc = reinterpret_cast<const char *>(reinterpret_cast<const XT *>(a.X) + foff);
    const uint32_t row_bytes = (uint32_t)(a.ldx * (int64_t)sizeof(XT));   // a neighbour row's address = one 32 x 32 -> 64-bit mad
    const ProConst<1> pc = pro_load<1, MODE>(a, active ? f : 0, active);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t *)lds;
    const uint32_t ring_lane = lds0 + (uint32_t)(lane & (SLAB - 1)) * (uint32_t)K::EB;   // this lane's feature of slice 0, slot 0
    const uint32_t col_lane = lds0 + (uint32_t)(K::IR + sub_e) * 4u;               // index slot 0, this lane's neighbour of instruction 0
    const uint32_t val_lane = lds0 + (uint32_t)(K::VR + (lane & 15)) * 4u;         // one value per lane 0..15 (read back by v_readlane)
    const uint32_t sc_lane = lds0 + (uint32_t)(K::SR + (lane & 15)) * 4u;

    auto idx_dma = [&](int32_t chunk, int islot) {  // index chunk -> ring slot: lane i brings neighbour chunk*64 + i (clamped into the row)
        int32_t q = hi - 1 - (chunk * kHubChunk + lane);
        q = q < lo ? lo : q;
        __builtin_amdgcn_global_load_lds(a.colidx + q, (lds_void_t *)(lds + K::IR + islot * kHubChunk), 4, 0, 0);
        if constexpr (has_val(MODE)) __builtin_amdgcn_global_load_lds(a.vals + q, (lds_void_t *)(lds + K::VR + islot * kHubChunk), 4, 0, 0);
    };
    auto sc_dma = [&](int islot) {  // colscale[col] of an index chunk that has landed
        int32_t c;
        hub_lds_read<0>(c, lds0 + (uint32_t)(K::IR + islot * kHubChunk + lane) * 4u);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c)::"memory");
        __builtin_amdgcn_global_load_lds(a.colscale + c, (lds_void_t *)(lds + K::SR + islot * kHubChunk), 4, 0, 0);
    };
    // a sub-chunk = neighbours k0 .. k0+15 of the index chunk in ring slot islot: the DMA lanes read their neighbour's column from
    // the index ring (idx_read: LDS reads only, the caller waits), then dma_issue puts the slices in flight
    auto idx_read = [&](int islot, int k0, int32_t(&c)[16]) {
        const uint32_t ca = col_lane + (uint32_t)(islot * kHubChunk + k0) * 4u;
#define GNNX_HUB_RC(i) if constexpr ((i) < IPS) hub_lds_read<(i) * EPI * 4>(c[i], ca)
        GNNX_HUB_RC(0); GNNX_HUB_RC(1); GNNX_HUB_RC(2); GNNX_HUB_RC(3); GNNX_HUB_RC(4); GNNX_HUB_RC(5); GNNX_HUB_RC(6); GNNX_HUB_RC(7);
        GNNX_HUB_RC(8); GNNX_HUB_RC(9); GNNX_HUB_RC(10); GNNX_HUB_RC(11); GNNX_HUB_RC(12); GNNX_HUB_RC(13); GNNX_HUB_RC(14); GNNX_HUB_RC(15);
#undef GNNX_HUB_RC
    };
    auto idx_wait = [&](int32_t(&c)[16]) {   // the reads above (and every older LDS operation) have returned
        static_assert(IPS == 4 || IPS == 8 || IPS == 16, "DMA instructions per sub-chunk");
        if constexpr (IPS == 4)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])::"memory");
        else if constexpr (IPS == 8)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]),
                           "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15])::"memory");
    };
    auto dma_issue = [&](int slot, const int32_t(&c)[16]) {
        float *dst = lds + slot * K::SUB_FLOATS;
#pragma unroll
        for (int i = 0; i < IPS; i++) {
            const char *srcp = xsrc + (uint64_t)(uint32_t)c[i] * row_bytes;
            lds_void_t *dstp = (lds_void_t *)(dst + i * K::IFLOATS);
            // (a source pointer of DEPENDENT type makes clang drop the whole instantiation without a diagnostic when the size is
            // 16: the kernel's symbol stays undefined -- hence the concrete types; tests/test_cabi_cpu.py runs `ldd -r` on the library)
            if constexpr (K::DS == 16) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(srcp), dstp, 16, 0, 0);
            else if constexpr (K::DS == 4) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(srcp), dstp, 4, 0, 0);
            else __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint16_t *>(srcp), dstp, 2, 0, 0);
        }
    };
    auto issue = [&](int slot, int islot, int k0) {   // prologue form: read, wait, issue
        int32_t c[16];
        idx_read(islot, k0, c);
        idx_wait(c);
        dma_issue(slot, c);
    };
    float acc = 0.f;
    // The prologue of a whole sub-chunk in straight-line code: 16 independent chains (sub, f64 multiply, convert, mul, add, max) that
    // the scheduler interleaves.  Applied element by element inside the add loop, each chain sat between exec-mask branches (the
    // subnormal guard of div_by_const, the j < cnt test) and ran at its full latency: the hub kernel's prologue modes were bound by
    // that, not by memory.  The guard is taken once per sub-chunk; slots past the row's end hold stale LDS words -- harmless, never added.
    auto pro_batch = [&](float(&v)[kHubSub]) {
        if constexpr (pro_bn(MODE)) {
            float d[kHubSub];
            bool bad = false;
#pragma unroll
            for (int j = 0; j < kHubSub; j++) {
                d[j] = __fsub_rn(v[j], pc.mean);
                v[j] = (float)((double)d[j] * pc.rsd[0]);
                bad |= !(fabsf(v[j]) >= 1.17549435e-38f) && d[j] != 0.f;
            }
            if (__builtin_expect(bad, 0)) {   // a subnormal quotient somewhere: the IEEE division for the sub-chunk (div_by_const)
#pragma unroll
                for (int j = 0; j < kHubSub; j++) v[j] = __fdiv_rn(d[j], pc.sd);
            }
#pragma unroll
            for (int j = 0; j < kHubSub; j++) v[j] = __fadd_rn(__fmul_rn(v[j], pc.gamma), pc.beta);
        }
        if constexpr (pro_relu(MODE)) {
#pragma unroll
            for (int j = 0; j < kHubSub; j++) v[j] = relu1(v[j]);
        }
    };
    auto add1 = [&](float x, float scv, float vv, int src) {
        if constexpr (has_sc(MODE)) x = mul_rn(x, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(scv), src)));
        if constexpr (has_val(MODE)) x = mul_rn(x, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), src)));
        acc = add_rn(acc, x);
    };
    // one step: sub-chunk t is added while sub-chunk t + LAS is put in flight.  The LDS reads of both -- the next issue's column
    // indices and this sub-chunk's 16 values -- go out together and share one wait (a second LDS round trip per step was a quarter
    // of a long row's time); the DMAs are issued in front of the adds.
    auto step = [&](int slot, int islot, int32_t cnt, int k0, bool more, int slot_next, int islot_next, int k0_next) {
        int32_t c[16];
        if (more) idx_read(islot_next, k0_next, c);
        const uint32_t ad = ring_lane + (uint32_t)slot * (uint32_t)(K::SUB_FLOATS * 4);
        float v[kHubSub], vv = 1.f, scv = 1.f;
        if constexpr (has_val(MODE)) hub_lds_read<0>(vv, val_lane + (uint32_t)(islot * kHubChunk + k0) * 4u);
        if constexpr (has_sc(MODE)) hub_lds_read<0>(scv, sc_lane + (uint32_t)(islot * kHubChunk + k0) * 4u);
        hub_lds_read_x<0 * K::SB, XT>(v[0], ad);
        hub_lds_read_x<1 * K::SB, XT>(v[1], ad);
        hub_lds_read_x<2 * K::SB, XT>(v[2], ad);
        hub_lds_read_x<3 * K::SB, XT>(v[3], ad);
        hub_lds_read_x<4 * K::SB, XT>(v[4], ad);
        hub_lds_read_x<5 * K::SB, XT>(v[5], ad);
        hub_lds_read_x<6 * K::SB, XT>(v[6], ad);
        hub_lds_read_x<7 * K::SB, XT>(v[7], ad);
        hub_lds_read_x<8 * K::SB, XT>(v[8], ad);
        hub_lds_read_x<9 * K::SB, XT>(v[9], ad);
        hub_lds_read_x<10 * K::SB, XT>(v[10], ad);
        hub_lds_read_x<11 * K::SB, XT>(v[11], ad);
        hub_lds_read_x<12 * K::SB, XT>(v[12], ad);
        hub_lds_read_x<13 * K::SB, XT>(v[13], ad);
        hub_lds_read_x<14 * K::SB, XT>(v[14], ad);
        hub_lds_read_x<15 * K::SB, XT>(v[15], ad);
        if (more) {
            idx_wait(c);   // lgkmcnt(0): the values below have returned as well (LDS returns in order)
            dma_issue(slot_next, c);
        }
        // every value read above is an operand of the wait (no use of it may be scheduled in front); vv / scv only in the modes that READ
        // them -- as an operand a constant 1.0 would be materialised in a register per step for nothing
#define GNNX_HUB_VWAIT(...)                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                             \
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),      \
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) __VA_ARGS__::"memory")
        if constexpr (has_val(MODE) && has_sc(MODE)) GNNX_HUB_VWAIT(, "+v"(vv), "+v"(scv));
        else if constexpr (has_val(MODE)) GNNX_HUB_VWAIT(, "+v"(vv));
        else if constexpr (has_sc(MODE)) GNNX_HUB_VWAIT(, "+v"(scv));
        else GNNX_HUB_VWAIT();
#undef GNNX_HUB_VWAIT
        if constexpr (sizeof(XT) == 2) {  // a bf16 is the top half of an f32: widening is exact
#pragma unroll
            for (int j = 0; j < kHubSub; j++) v[j] = __uint_as_float(__float_as_uint(v[j]) << 16);
        }
        if constexpr (has_pro(MODE)) pro_batch(v);
        if (cnt >= kHubSub) {
#pragma unroll
            for (int j = 0; j < kHubSub; j++) add1(v[j], scv, vv, j);
        } else {
#pragma unroll
            for (int j = 0; j < kHubSub; j++)
                if (j < cnt) add1(v[j], scv, vv, j);
        }
    };
    auto wrap = [](int x) { return x >= NI ? x - NI : x; };

    // prologue: the first NC index chunks, then (once they have landed) their column scales and the first LAS sub-chunks
#pragma unroll
    for (int k = 0; k < NC; k++) idx_dma(k, k);
    hub_wait_vm<0>();
    if constexpr (has_sc(MODE)) {
#pragma unroll
        for (int k = 0; k <= DI; k++) sc_dma(k);
    }
#pragma unroll
    for (int p = 0; p < LAS; p++)
        if (p < nsub) issue(p, p / SUBS, (p % SUBS) * kHubSub);
    int slot_i = LAS, slot_c = 0, islot_c = 0;  // LAS < NS
    const int32_t nchunk = (nsub + SUBS - 1) / SUBS;
    for (int32_t cc = 0; cc < nchunk; cc++) {
#pragma unroll
        for (int u = 0; u < SUBS; u++) {
            const int32_t t = cc * SUBS + u;
            if (t < nsub) {
                const bool more = t + LAS < nsub;
                // sub-chunk t has landed: at most the LAS - 1 younger ones are still in flight (the tail drains everything)
                if (more) hub_wait_vm<(LAS - 1) * IPS>();
                else hub_wait_vm<0>();
                step(slot_c, islot_c, total - t * kHubSub, u * kHubSub, more, slot_i, wrap(islot_c + (u + LAS) / SUBS),
                     ((u + LAS) % SUBS) * kHubSub);
                if (more) slot_i = slot_i + 1 == NS ? 0 : slot_i + 1;
                slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
            }
        }
        // index chunk cc + NC into the slot chunk cc - 1 has left; column scales of chunk cc + DI + 1, whose indices landed DI
        // iterations ago and whose adds are DI + 1 iterations away
        idx_dma(cc + NC, wrap(islot_c + NC));
        if constexpr (has_sc(MODE)) sc_dma(wrap(islot_c + DI + 1));
        islot_c = wrap(islot_c + 1);
    }
    hub_wait_vm<0>();  // index DMAs past the end of the row
    if (active) {  // the epilogue of epilogue_store<1>, same op order
        float v = acc;
        if (a.rowscale) v = mul_rn(v, a.rowscale[row]);
        if (a.bias) v = add_rn(v, a.bias[f]);
        float *dst = a.Y + (int64_t)row * a.ldy + f;
        if (a.beta) v = add_rn(*dst, v);
        if (a.relu_out) v = relu1(v);
        *dst = v;
        if constexpr (SUMS) {   // this row's term of BatchNorm's backward sums (bn_sums_rows' arithmetic), a partial row of its own
            const float h = a.bn_h[(int64_t)row * a.bn_ldh + f];
            const float mean = a.pro_mean[f], var = a.pro_var[f];
            float g = v;
            if (a.bn_relu) {
                float z = __fdiv_rn(__fsub_rn(h, mean), sqrtf(__fadd_rn(var, a.pro_eps)));
                if (a.pro_gamma) z = __fmul_rn(z, a.pro_gamma[f]);
                if (a.pro_beta) z = __fadd_rn(z, a.pro_beta[f]);
                if (!(z > 0.f)) g = 0.f;
            }
            const float xhat = (h - mean) * (1.0f / sqrtf(var + a.pro_eps));
            float *prow = a.bn_partial + ((int64_t)a.n_stream_waves + (int64_t)ordinal0 + (int64_t)(blockIdx.x / n_groups)) * 2 * a.n_feat;
            prow[f] = g;
            prow[a.n_feat + f] = g * xhat;
        }
    }
}

// ---- the LONGEST hub rows: producer / consumer form ---------------------------------------------------------------------
// In spmm_hub_kernel a (row, slab) is ONE wavefront that issues the DMAs, reads the ring and adds, with 32 KiB in flight: ~12 ns per
// neighbour.  On the whole graph that hides behind the bytes of the other hub rows; on one rank's shard of it (1/8 of the rows, the
// same longest row) or on a small graph the longest rows ARE the aggregation's time (a 250 k-entry row: 3.0 ms).  The sum itself
// cannot be split -- one accumulator per feature, the reference's order -- so everything else is taken out of its wavefront and the
// row is spread over more CUs:
//   * a workgroup = (row, SLAB-feature slab) = ONE consumer wavefront + hubpc::NP producer wavefronts, a CU of its own (the ring takes
//     128 KiB of its LDS);
//   * producers own the index chunks (64 neighbours) round-robin: index chunk -> LDS by DMA, column -> address, the slab's slices of
//     16 neighbours per ring slot by LDS-DMA (+ per index chunk the 64 per-entry values / gathered column scales into small
//     rings), up to LAS sub-chunks in flight each; when a sub-chunk has LANDED (the producer's own vmcnt) its count goes to an LDS flag;
//   * the consumer reads a landed slot with 8 two-address LDS reads (neighbours e, e + 1 of its feature in one instruction), the
//     next slot's reads in flight while this slot's 16 adds run -- strictly in descending column order, separately rounded: the same
//     bits as every other path -- and publishes the slots it has left (the producers' back-pressure).  ~6-7 cycles per neighbour.
// Flags are LDS words (ds_write / ds_read of one CU's LDS unit are processed in order; a producer writes its flag after the
// s_waitcnt that covers the DMA).  No barrier after the start, no wait without progress: a producer blocks only when nothing of
// its own is in flight and the ring is full, the consumer only on a sub-chunk that is not there yet.
namespace hubpc {
constexpr int NP = 3;                 // producer wavefronts
constexpr int kSpinCap = 1 << 22;     // polls of an LDS flag before a wait gives up (seconds; a wait lasts microseconds)
constexpr int kBigSlab = 16;          // features per workgroup
constexpr int kMaxLdsBytes = 160 * 1024;  // a CU's LDS: what the device reports per workgroup (163 840) -- the launcher checks the request against the runtime's answer (lds_opt_in)
// SLAB features per workgroup: 16 -- a row of F features is spread over F / 16 CUs.  While the row kernel streams beside it every
// CU keeps ~128 KiB in flight and the memory system serves them at about the same rate each (~30 GB/s per CU at 8 TB/s over 256
// CUs), so a row's rate is the number of CUs it sits on: with 64-feature slabs the 62 k-entry row of RMAT 1M / 10M took 0.51 ms
// (8 ns per neighbour: its two CUs' share of the bytes, not the add chain); 64-byte slices put it on 8.
template <int MODE, int SLAB> struct Cfg {
    static constexpr int SB = SLAB * 4;                  // bytes of a neighbour's slice
    static constexpr int EPI = 1024 / SB;                // neighbours per DMA wave-instruction
    static constexpr int LPE = 64 / EPI;                 // lanes per neighbour
    static constexpr int IPS = kHubSub / EPI;            // DMA instructions per sub-chunk of 16 neighbours
    static constexpr int SUBF = kHubSub * SLAB;          // floats per ring slot
    // ring slots: 128 KiB -- 64 KiB when per-entry values AND column scales ride along (the rare weighted Mode SYM; its two extra
    // small rings would bring the full-ring layout to 156 KiB).  Round 4 saw that 156 KiB layout "read wrong words" and capped every
    // layout at 152 000 bytes without a cause.  Round 5 looked: the device and the runtime grant the whole 160 KiB to a workgroup, and
    // plain LDS accesses and LDS-DMA of 4 and 16 bytes per lane hit every address up to 163 840 (bench_kernels/lds_probe.hip); the
    // very 156 KiB layout, rebuilt from today's source, passes every mode's test and the binary lint -- it was never an addressing
    // limit.  What is checked now: the dynamic-LDS request against device and runtime at the first launch (lds_opt_in), and the
    // built kernels against the one hazard that does produce "wrong words" here -- a register of an in-flight LDS read touched before
    // its wait (scripts/check_lds_asm_discipline.py).
    static constexpr bool kBoth = has_val(MODE) && has_sc(MODE);
    static constexpr int S = (kBoth && SLAB == 16 ? 16384 : 32768) / SUBF;
    static constexpr int LASC = SLAB == 64 ? (kBoth ? 1 : 2) : (kBoth ? 4 : 8);   // index chunks (4 ring slots each) in flight per producer
    static constexpr int NCX = LASC + 1;                 // index chunks fetched ahead per producer
    static constexpr int NI = NCX < 4 ? 4 : (NCX < 8 ? 8 : 16);   // index ring slots per producer (a power of two > NCX)
    static constexpr int XC = 1 + (has_val(MODE) ? 1 : 0) + (has_sc(MODE) ? 1 : 0);   // VMEM operations per index chunk besides the slices
    static constexpr int OPS = 4 * IPS + XC;             // VMEM operations per index chunk
    static constexpr int VR = S * SUBF;                                // [S][16] per-entry values (written 64 at a time, by index chunk)
    static constexpr int SR = VR + (has_val(MODE) ? S * kHubSub : 0);  // [S][16] gathered column scales
    static constexpr int IR = SR + (has_sc(MODE) ? S * kHubSub : 0);   // [NP][NI][64] column indices
    static constexpr int FL = IR + NP * NI * kHubChunk;                // landed[NP], consumed
    static constexpr int LDS_FLOATS = FL + 8;
    static_assert(SLAB == 64 || SLAB == 16, "slice = 256 or 64 bytes");
    static_assert(LDS_FLOATS * 4 <= kMaxLdsBytes, "one workgroup per CU: the layout must fit a CU's LDS");
    static_assert((S & (S - 1)) == 0 && S >= 4 * (NP * LASC + 2), "ring: a power of two, room for everything in flight plus the chunk being added");
    static_assert(NCX >= LASC + 1 && NI > NCX, "an index chunk has landed when it is read: a chunk issued behind its DMA has landed by then");
    static_assert(LASC * OPS <= 63, "vmcnt is a 6-bit counter");
    static_assert(LASC <= 32, "wait_vm_sub");
};
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int32_t flag_read(uint32_t addr)   // one LDS word, wave-uniform
{
    int32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void flag_write(uint32_t addr, int32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// s_waitcnt vmcnt(k * K): the k youngest sub-chunks (K slice DMAs each) stay in flight; k < LAS
template <int K, int LAS> __device__ __forceinline__ void wait_vm_sub(int k)
{
#define GNNX_PC_CASE(i) case i: if constexpr ((i) < LAS) hub_wait_vm<((i) < LAS ? (i) : 0) * K>(); break;
    switch (k) {
        GNNX_PC_CASE(0) GNNX_PC_CASE(1) GNNX_PC_CASE(2) GNNX_PC_CASE(3) GNNX_PC_CASE(4) GNNX_PC_CASE(5) GNNX_PC_CASE(6) GNNX_PC_CASE(7)
        GNNX_PC_CASE(8) GNNX_PC_CASE(9) GNNX_PC_CASE(10) GNNX_PC_CASE(11) GNNX_PC_CASE(12) GNNX_PC_CASE(13) GNNX_PC_CASE(14) GNNX_PC_CASE(15)
        GNNX_PC_CASE(16) GNNX_PC_CASE(17) GNNX_PC_CASE(18) GNNX_PC_CASE(19) GNNX_PC_CASE(20) GNNX_PC_CASE(21) GNNX_PC_CASE(22) GNNX_PC_CASE(23)
        GNNX_PC_CASE(24) GNNX_PC_CASE(25) GNNX_PC_CASE(26) GNNX_PC_CASE(27) GNNX_PC_CASE(28) GNNX_PC_CASE(29) GNNX_PC_CASE(30) GNNX_PC_CASE(31)
    default: hub_wait_vm<0>(); break;
    }
#undef GNNX_PC_CASE
}
// A wait of the producer / consumer kernel gave up: raise the plan's error word (pinned host memory: a system-scope store)
__device__ __forceinline__ void pc_fail(int32_t *err_word, int32_t who)
{
    if (err_word && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(err_word, who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// own sequence number u of producer p -> the row's sub-chunk: producers own the index chunks (4 sub-chunks) round-robin
__device__ __forceinline__ int32_t t_of(int32_t p, int32_t u) { return 4 * (p + (u >> 2) * NP) + (u & 3); }
}  // namespace hubpc

template <int MODE, bool SUMS, int SLAB>
__global__ __launch_bounds__(64 * (1 + hubpc::NP)) void spmm_hubpc_kernel(SpmmArgs a, const int32_t *hub_rows, int32_t n_slabs)
{
    using namespace hubpc;
    using K = Cfg<MODE, SLAB>;
    constexpr int S = K::S, IPS = K::IPS, NI = K::NI, NCX = K::NCX, SUBF = K::SUBF;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int32_t ordinal = (int32_t)(blockIdx.x / n_slabs);
    const int32_t row = __builtin_amdgcn_readfirstlane(hub_rows[ordinal]);
    int32_t sidx = (int32_t)(blockIdx.x % n_slabs);
    // workgroups b and b + 8 land on one XCD (observed round-robin dispatch; only speed depends on it): give them the two 64-byte
    // slabs of one 128-byte line
    if (SLAB == 16 && (n_slabs & 15) == 0) sidx = (sidx & ~15) | ((sidx & 7) << 1) | ((sidx >> 3) & 1);
    const int32_t f_slab = sidx * SLAB;
    const int32_t lo = __builtin_amdgcn_readfirstlane(a.rowptr[row]);
    const int32_t hi = __builtin_amdgcn_readfirstlane(a.rowptr[row + 1]);
    const int32_t total = hi - lo;
    const int32_t nsub = (total + kHubSub - 1) / kHubSub;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t *)lds;
    const uint32_t fl0 = lds0 + (uint32_t)K::FL * 4u;   // landed[p] at fl0 + 4 p, consumed at fl0 + 4 NP
    if (threadIdx.x < 8) reinterpret_cast<int32_t *>(lds + K::FL)[threadIdx.x] = 0;
    __syncthreads();

    if (wv > 0) {
        // ------------------------------------------------------------------ producer p
        // Everything at the granularity of an index chunk (64 neighbours = 4 ring slots): a lone wavefront pays ~4 cycles per instruction
        // and tens per taken branch, so the loop is long straight-line stretches -- one flag check, one LDS wait, the chunk's DMAs back
        // to back, one counted vmcnt wait, one publish per 64 neighbours.
#ifdef GNNX_EXPERIMENTS
        if (a.pc_experiment == 3 || a.pc_experiment == 4) return;   // 3: timing only, the consumer alone; 4: the consumer's wait must time out (test of the error word)
#endif
        const int32_t p = wv - 1;
        const int32_t nchunk = (nsub + 3) >> 2;
        const int32_t n_own = nchunk > p ? (nchunk - p + NP - 1) / NP : 0;   // own index chunks: own chunk j is the row's chunk p + j NP
        // DMA source of this lane: neighbour sub_e of the instruction, 4 features at foff (lanes past the row's width re-read the
        // slab's first piece; their LDS words are never stored)
        const int sub_e = lane / K::LPE;
        int32_t foff = f_slab + (lane % K::LPE) * 4;
        if (foff + 4 > a.n_feat) foff = f_slab;
        const char *xsrc = reinterpret_cast<const char *>(a.X + foff);
        const uint32_t row_bytes = (uint32_t)(a.ldx * (int64_t)sizeof(float));
        float *iring = lds + K::IR + p * NI * kHubChunk;
        const uint32_t col_lane = lds0 + (uint32_t)(K::IR + p * NI * kHubChunk + sub_e) * 4u;
        const uint32_t col64 = lds0 + (uint32_t)(K::IR + p * NI * kHubChunk + lane) * 4u;

        auto idx_dma = [&](int32_t j) {   // own index chunk j -> its ring slot: lane i brings neighbour chunk * 64 + i (clamped into the row)
            const int32_t c = p + j * NP;
            int32_t q = hi - 1 - (c * kHubChunk + lane);
            q = q < lo ? lo : q;
            __builtin_amdgcn_global_load_lds(a.colidx + q, (lds_void_t *)(iring + (j & (NI - 1)) * kHubChunk), 4, 0, 0);
        };
        // own chunk j: columns from the index ring; the chunk's 64 per-entry values / gathered column scales, then the slices of its
        // 64 neighbours into ring slots 4 c .. 4 c + 3 (mod S), then the index chunk NCX ahead: always K::OPS VMEM operations, in this
        // order (the row's last chunk too: neighbours past the row's end are clamped into it -- valid addresses, words nobody adds)
        auto issue_chunk = [&](int32_t j) {
            const int32_t c = p + j * NP;
            const int islot = j & (NI - 1), slot0 = (4 * c) & (S - 1);
            int32_t col[4 * IPS], c64 = 0;
            const uint32_t ca = col_lane + (uint32_t)(islot * kHubChunk) * 4u;
#define GNNX_PC_RC(i) hub_lds_read<(i) * K::EPI * 4>(col[i], ca)
            GNNX_PC_RC(0); GNNX_PC_RC(1); GNNX_PC_RC(2); GNNX_PC_RC(3);
            if constexpr (IPS == 4) {
                GNNX_PC_RC(4); GNNX_PC_RC(5); GNNX_PC_RC(6); GNNX_PC_RC(7); GNNX_PC_RC(8); GNNX_PC_RC(9); GNNX_PC_RC(10); GNNX_PC_RC(11);
                GNNX_PC_RC(12); GNNX_PC_RC(13); GNNX_PC_RC(14); GNNX_PC_RC(15);
            }
#undef GNNX_PC_RC
            if constexpr (has_sc(MODE)) hub_lds_read<0>(c64, col64 + (uint32_t)(islot * kHubChunk) * 4u);
            if constexpr (IPS == 4)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(col[0]), "+v"(col[1]), "+v"(col[2]), "+v"(col[3]), "+v"(col[4]), "+v"(col[5]), "+v"(col[6]), "+v"(col[7]), "+v"(col[8]),
                               "+v"(col[9]), "+v"(col[10]), "+v"(col[11]), "+v"(col[12]), "+v"(col[13]), "+v"(col[14]), "+v"(col[15]), "+v"(c64)::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(col[0]), "+v"(col[1]), "+v"(col[2]), "+v"(col[3]), "+v"(c64)::"memory");
            if constexpr (has_val(MODE)) {
                int32_t q = hi - 1 - (c * kHubChunk + lane);
                q = q < lo ? lo : q;
                __builtin_amdgcn_global_load_lds(a.vals + q, (lds_void_t *)(lds + K::VR + slot0 * kHubSub), 4, 0, 0);
            }
            if constexpr (has_sc(MODE)) __builtin_amdgcn_global_load_lds(a.colscale + c64, (lds_void_t *)(lds + K::SR + slot0 * kHubSub), 4, 0, 0);
            float *dst = lds + slot0 * SUBF;
#pragma unroll
            for (int i = 0; i < 4 * IPS; i++) {
                const char *srcp = xsrc + (uint64_t)(uint32_t)col[i] * row_bytes;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float *>(srcp), (lds_void_t *)(dst + i * 256), 16, 0, 0);
            }
            idx_dma(j + NCX);
        };
        // A landed chunk is FINISHED by its producer, in place: the BatchNorm / ReLU prologue, the gathered column scale and the
        // per-entry value are applied to this lane's 16-byte piece of every neighbour's slice -- the same separately rounded operations
        // in the same order as every other path (pro_apply, then x sc, then x val) -- so that the consumer's loop is "read, add" in
        // every MODE (its instruction count is the row's time; the producers have the issue slots to spare).
        const ProConst<4> pc4 = pro_load<4, MODE>(a, foff, true);
        auto finish_chunk = [&](int32_t j) {
            const int32_t slot0 = (4 * (p + j * NP)) & (S - 1);
#pragma unroll
            for (int i = 0; i < 4 * IPS; i++) {
                const uint32_t xa = lds0 + (uint32_t)(slot0 * SUBF + i * 256) * 4u + (uint32_t)lane * 16u;
                const uint32_t e = (uint32_t)(slot0 * kHubSub + i * K::EPI + sub_e) * 4u;   // this lane's neighbour in the small rings
                v4f x;
                float vv = 1.f, sv = 1.f;
                asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(xa) : "memory");
                if constexpr (has_val(MODE)) hub_lds_read<0>(vv, lds0 + (uint32_t)K::VR * 4u + e);
                if constexpr (has_sc(MODE)) hub_lds_read<0>(sv, lds0 + (uint32_t)K::SR * 4u + e);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(vv), "+v"(sv)::"memory");
                float4 y = make_float4(x.x, x.y, x.z, x.w);
                if constexpr (has_pro(MODE)) y = pro_apply<MODE>(y, pc4);
                if constexpr (has_sc(MODE)) y = mul_rn(y, sv);
                if constexpr (has_val(MODE)) y = mul_rn(y, vv);
                x = (v4f){y.x, y.y, y.z, y.w};
                asm volatile("ds_write_b128 %0, %1" ::"v"(xa), "v"(x) : "memory");
            }
        };

#pragma unroll
        for (int j = 0; j < NCX; j++) idx_dma(j);
        hub_wait_vm<0>();
        int32_t issued = 0, landed = 0, consumed = 0, spins = 0;   // own chunks
        while (landed < n_own) {
            // put in flight whatever the window (LASC chunks) and the ring allow: slots 4 c .. 4 c + 3 (mod S) are free once the consumer
            // has left sub-chunk 4 c + 3 - S
            while (issued < n_own && issued - landed < K::LASC) {
                const int32_t need = 4 * (p + issued * NP) + 3 - S + 1;
                if (need > consumed) {
                    consumed = flag_read(fl0 + 4u * NP);
                    if (need > consumed) break;
                }
                issue_chunk(issued++);
            }
            if (issued == landed) {   // ring full and nothing of ours in flight: the consumer is behind
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinCap) {   // (never reached: every wait here ends when the consumer moves.)  A bounded exit all the
                    pc_fail(a.err_word, 2);  // same -- and a LOUD one: the plan's error word, GNNX_ERR_HIP at the plan's next call
                    break;
                }
                continue;
            }
            spins = 0;
            wait_vm_sub<K::OPS, K::LASC>(issued - landed - 1);   // own chunk `landed` is in LDS: exactly K::OPS operations per younger chunk
            if constexpr (has_pro(MODE) || has_val(MODE) || has_sc(MODE)) finish_chunk(landed);
            landed++;
            flag_write(fl0 + 4u * (uint32_t)p, 4 * landed);   // (behind finish_chunk's LDS writes: one wavefront's LDS operations are processed in order)
        }
        hub_wait_vm<0>();   // index DMAs past the end of the row
        return;
    }

    // ---------------------------------------------------------------------- consumer
    // A lone wavefront issues one instruction every ~4 cycles, whatever its kind: the consumer's time is its instruction COUNT.  So
    // everything but "read, add" is somewhere else -- the per-entry value, the column scale and the BatchNorm / ReLU prologue are
    // applied to a landed slot by its producer (in place, the same separately rounded operations in the same order), the bookkeeping
    // (flags, slot addresses) is per index chunk of 64 neighbours -- and the loop is the same for every MODE: per neighbour half a
    // two-address LDS read and one add.
    const int32_t f = f_slab + (lane & (SLAB - 1));
    const bool active = lane < SLAB && f < a.n_feat;
    const uint32_t ring_lane = lds0 + (uint32_t)(lane & (SLAB - 1)) * 4u;
#ifdef GNNX_EXPERIMENTS
    if (a.pc_experiment == 2) {   // timing only: the producers alone (no back-pressure)
        flag_write(fl0 + 4u * NP, 0x7fffffff);
        return;
    }
#endif
    struct Set {
        v2f d[8];
    };
    constexpr uint32_t SLOT_BYTES = (uint32_t)SUBF * 4u;
    // neighbours e, e + 1 of this lane's feature in one instruction: slices are 256 bytes apart (read2st64) or 64 (read2, dword offsets)
#define GNNX_PC_R2(s, ad, i)                                                                                                           \
    do {                                                                                                                               \
        if constexpr (SLAB == 64)                                                                                                      \
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"((s).d[i]) : "v"(ad), "i"(2 * (i)), "i"(2 * (i) + 1) : "memory"); \
        else                                                                                                                           \
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"((s).d[i]) : "v"(ad), "i"(32 * (i)), "i"(32 * (i) + 16) : "memory"); \
    } while (0)
    float acc = 0.f;
    // one ring slot -> 16 registers.  The accumulator is an in-out operand of the block's first read -- nothing is done to it: it
    // only keeps the adds of the sub-chunk in front of this block IN FRONT of it.  Left to itself the compiler issues a chunk's four
    // blocks of reads first and its 64 adds last, and three of the four waits then stall for most of an LDS round trip (4.8 ns per
    // neighbour for the consumer alone); with 16 adds between the blocks the round trips are covered.  (Pinning the adds
    // themselves with volatile asm cost a register copy per add -- a half of a 64-bit asm operand -- and was slower: 6.1 ns.)
    auto issue_reads = [&](Set &s, uint32_t ad) {
        if constexpr (SLAB == 64)
            asm volatile("ds_read2st64_b32 %0, %2 offset0:0 offset1:1" : "=v"(s.d[0]), "+v"(acc) : "v"(ad) : "memory");
        else
            asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:16" : "=v"(s.d[0]), "+v"(acc) : "v"(ad) : "memory");
        GNNX_PC_R2(s, ad, 1); GNNX_PC_R2(s, ad, 2); GNNX_PC_R2(s, ad, 3);
        GNNX_PC_R2(s, ad, 4); GNNX_PC_R2(s, ad, 5); GNNX_PC_R2(s, ad, 6); GNNX_PC_R2(s, ad, 7);
    };
#undef GNNX_PC_R2
    // this set's reads have returned; N: the LDS operations issued after them that may stay in flight.  The registers are in-out
    // operands so that no use of them is scheduled in front of the wait
    auto wait_set = [&](Set &s, auto keep) {
        constexpr int N = decltype(keep)::value;
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(s.d[0]), "+v"(s.d[1]), "+v"(s.d[2]), "+v"(s.d[3]), "+v"(s.d[4]), "+v"(s.d[5]), "+v"(s.d[6]), "+v"(s.d[7])
                     : "i"(N)
                     : "memory");
    };
    using Keep8 = std::integral_constant<int, 8>;
    using None = std::integral_constant<int, 0>;
    // acc = RN(acc + x), sixteen times
    auto adds = [&](const Set &s) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc = add_rn(acc, s.d[i].x);
            acc = add_rn(acc, s.d[i].y);
        }
    };
    auto adds_n = [&](const Set &s, int32_t cnt) {   // the row's last sub-chunk: cnt of 16
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (2 * i < cnt) acc = add_rn(acc, s.d[i].x);
            if (2 * i + 1 < cnt) acc = add_rn(acc, s.d[i].y);
        }
    };
    // index chunk c belongs to producer c % NP, its sub-chunks are that producer's own 4 (c / NP) ..: (p, own0) walk along with c
    // producer p has landed (and finished) `need` own sub-chunks.  A wait that gives up (never reached: the producers wait for nothing
    // but this wavefront) sets `dead`: from then on no wait is made, the loop runs out on whatever the ring holds, and the row is NOT
    // stored -- the consumer raises the plan's error word instead and the plan's next call returns GNNX_ERR_HIP.  (`dead` is a sticky
    // scalar rather than an early return: an exit inside the loop changed the register allocation of the in-flight read sets, and
    // the compiler copied them before their waits -- scripts/check_lds_asm_discipline.py checks the built kernel for exactly that.)
    int32_t dead = 0;
    auto wait_chunk = [&](int32_t p, int32_t need, int32_t seen) {
        if (seen >= need || dead) return;
#ifdef GNNX_EXPERIMENTS
        if (a.pc_experiment == 1 || a.pc_experiment == 3) return;
#endif
        int32_t spins = 0;
        while (flag_read(fl0 + 4u * (uint32_t)p) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinCap) {
                dead = 1;
                break;
            }
        }
    };
    const int32_t nchunk = (nsub + 3) >> 2;
    int32_t p = 0, own0 = 0;   // of the chunk being added
    Set A, B;
    wait_chunk(0, nchunk > 1 ? 4 : nsub, 0);
    issue_reads(A, ring_lane);
    int32_t c = 0;
    for (; c + 1 < nchunk; c++) {
        // precondition: chunk c is complete in LDS, the reads of its first sub-chunk are in flight into A
        const int32_t pn = p + 1 == NP ? 0 : p + 1, own0n = p + 1 == NP ? own0 + 4 : own0;   // the next chunk's producer / first own sub-chunk
        const int32_t need_n = own0n + (c + 2 < nchunk ? 4 : nsub - 4 * (c + 1));            // (the row's last chunk may be short)
        int32_t fl;
        asm volatile("ds_read_b32 %0, %1" : "=v"(fl) : "v"(fl0 + 4u * (uint32_t)pn) : "memory");   // early: it has returned long before it is looked at
        const uint32_t ad = ring_lane + (uint32_t)((4 * c) & (S - 1)) * SLOT_BYTES;
        issue_reads(B, ad + SLOT_BYTES);
        wait_set(A, Keep8{});
        adds(A);
        issue_reads(A, ad + 2 * SLOT_BYTES);
        wait_set(B, Keep8{});
        adds(B);
        issue_reads(B, ad + 3 * SLOT_BYTES);
        wait_set(A, Keep8{});
        adds(A);
        // `fl` is the output of an asm LDS read that no compiler-visible wait covers: it has landed behind the wait_set above (LDS
        // operations return in order), so its first use is tied BEHIND that wait -- volatile asm statements keep their order, a plain
        // v_readfirstlane of an asm output does not, and hoisted above the wait it reads the register before the load has written it
        asm volatile("" : "+v"(fl)::"memory");
        wait_chunk(pn, need_n, __builtin_amdgcn_readfirstlane(fl));
        issue_reads(A, ring_lane + (uint32_t)((4 * c + 4) & (S - 1)) * SLOT_BYTES);
        wait_set(B, Keep8{});
        flag_write(fl0 + 4u * NP, 4 * c + 4);   // the four slots of chunk c are free (their words are in registers)
        adds(B);
        p = pn;
        own0 = own0n;
    }
    {   // the row's last index chunk: 1 .. 4 sub-chunks, the last one possibly short; the reads of its first are in flight into A
        const uint32_t ad = ring_lane + (uint32_t)((4 * c) & (S - 1)) * SLOT_BYTES;
        const int32_t r = nsub - 4 * c;
        const int32_t cnt_last = total - (nsub - 1) * kHubSub;
        wait_set(A, None{});
        for (int32_t k = 0; k < r; k++) {
            if (k + 1 < r) {
                adds(A);
                issue_reads(A, ad + (uint32_t)(k + 1) * SLOT_BYTES);
                wait_set(A, None{});
            } else {
                adds_n(A, cnt_last);
            }
        }
    }
    if (dead) {   // a wait gave up: the sum is not the row's -- say so, store nothing
        pc_fail(a.err_word, 1);
        return;
    }
    if (active) {  // the epilogue of epilogue_store<1>, same op order
        float v = acc;
        if (a.rowscale) v = mul_rn(v, a.rowscale[row]);
        if (a.bias) v = add_rn(v, a.bias[f]);
        float *dst = a.Y + (int64_t)row * a.ldy + f;
        if (a.beta) v = add_rn(*dst, v);
        if (a.relu_out) v = relu1(v);
        *dst = v;
        if constexpr (SUMS) {   // this row's term of BatchNorm's backward sums (bn_sums_rows' arithmetic), a partial row of its own
            const float h = a.bn_h[(int64_t)row * a.bn_ldh + f];
            const float mean = a.pro_mean[f], var = a.pro_var[f];
            float g = v;
            if (a.bn_relu) {
                float z = __fdiv_rn(__fsub_rn(h, mean), sqrtf(__fadd_rn(var, a.pro_eps)));
                if (a.pro_gamma) z = __fmul_rn(z, a.pro_gamma[f]);
                if (a.pro_beta) z = __fadd_rn(z, a.pro_beta[f]);
                if (!(z > 0.f)) g = 0.f;
            }
            const float xhat = (h - mean) * (1.0f / sqrtf(var + a.pro_eps));
            float *prow = a.bn_partial + ((int64_t)a.n_stream_waves + (int64_t)ordinal) * 2 * a.n_feat;
            prow[f] = g;
            prow[a.n_feat + f] = g * xhat;
        }
    }
}

// ---- plan construction -------------------------------------------------------------------------
// hub rows: count, then list as {row, degree} (positions by atomics; the host sorts the list, so its order does not matter)
__global__ void plan_count_kernel(const int32_t *rowptr, int32_t n_rows, int32_t chunk, unsigned long long *counters)
{
    int32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    int32_t deg = rowptr[row + 1] - rowptr[row];
    if (deg > chunk) {
        atomicAdd(&counters[0], 1ull);
        atomicAdd(&counters[1], (unsigned long long)deg);
    }
}

__global__ void plan_fill_kernel(const int32_t *rowptr, int32_t n_rows, int32_t chunk, unsigned long long *counters, int2 *rows)
{
    int32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    int32_t deg = rowptr[row + 1] - rowptr[row];
    if (deg <= chunk) return;
    rows[atomicAdd(&counters[2], 1ull)] = make_int2(row, deg);
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

template <int VEC, int MODE, class XT, bool SUMS, int SLAB>
int launch_hub_rows(hipStream_t st, const SpmmArgs &a, const int32_t *rows, int32_t n_rows_hub, int32_t ordinal0)
{
    // look-ahead in sub-chunks of 16 neighbours: 8 where a sub-chunk is 4 DMA instructions, less where it is 8 or 16 (vmcnt counts
    // at most 63 operations).
    // (Prologue modes: with the sub-chunk's prologue in straight-line code the ring length no longer matters -- 4.90 ms on the hub
    // rows of the bench graph with LAS 3 or 8, against 5.7 / 6.2 ms when the prologue ran element by element between branches, and
    // 3.4 ms without a prologue.)
    constexpr int LAS = (VEC == 4 && sizeof(XT) == 4) ? 8 : (VEC == 4 ? 6 : 3);
    using K = HubCfg<VEC, MODE, LAS, XT, SLAB>;
    if (n_rows_hub <= 0) return GNNX_OK;
    static std::atomic<uint64_t> done{0};
    constexpr size_t lds_wave = sizeof(float) * K::LDS_FLOATS;
    constexpr int max_waves = 160 * 1024 / lds_wave < 4 ? (int)(160 * 1024 / lds_wave) : 4;
    const int32_t n_slabs = (int32_t)ceil_div(a.n_feat, SLAB);
    const int waves = n_slabs < max_waves ? n_slabs : max_waves;
    const int32_t n_groups = (int32_t)ceil_div(n_slabs, waves);
    {   // dynamic-LDS opt-in, once per kernel and device, checked against what the runtime grants
        const int rc_lds = lds_opt_in(&spmm_hub_kernel<VEC, MODE, LAS, XT, SUMS, SLAB>, lds_wave * max_waves, done, "spmm_hub_kernel");
        if (rc_lds != GNNX_OK) return rc_lds;
    }
    const dim3 grid((uint32_t)((int64_t)n_rows_hub * n_groups));
    const size_t lds_wg = lds_wave * waves;
    hipLaunchKernelGGL((spmm_hub_kernel<VEC, MODE, LAS, XT, SUMS, SLAB>), grid, dim3(64 * waves), lds_wg, st, a, rows, n_slabs, n_groups, ordinal0);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

template <int VEC, int MODE, class XT, bool SUMS = false>
int launch_hub_kernel(hipStream_t st, const SpmmArgs &a)
{
    // the rows behind the first n_big_rows (those belong to spmm_hubpc_kernel: launch_hubpc)
    return launch_hub_rows<VEC, MODE, XT, SUMS, kHubSlab>(st, a, a.hub_rows + a.n_big_rows, a.n_hub_rows - a.n_big_rows, a.n_big_rows);
}

template <int VEC>
int launch_hub(int mode, hipStream_t st, const SpmmArgs &a)
{
    if constexpr (VEC == 4)
        if (a.bn_partial) return launch_hub_kernel<VEC, 2, float, true>(st, a);   // backward aggregation + BatchNorm sums (mode 2, f32)
    if (a.x_bf16) {  // bf16 feature rows: the three plain modes (a prologue goes with f32 rows)
        switch (mode) {
        case 0: return launch_hub_kernel<VEC, 0, bf16_t>(st, a);
        case 1: return launch_hub_kernel<VEC, 1, bf16_t>(st, a);
        case 2: return launch_hub_kernel<VEC, 2, bf16_t>(st, a);
        default: return launch_hub_kernel<VEC, 6, bf16_t>(st, a);
        }
    }
    switch (mode) {
    case 0: return launch_hub_kernel<VEC, 0, float>(st, a);
    case 1: return launch_hub_kernel<VEC, 1, float>(st, a);
    case 2: return launch_hub_kernel<VEC, 2, float>(st, a);
    case 3: return launch_hub_kernel<VEC, 3, float>(st, a);
    case 4: return launch_hub_kernel<VEC, 4, float>(st, a);
    case 5: return launch_hub_kernel<VEC, 5, float>(st, a);
    default: return launch_hub_kernel<VEC, 6, float>(st, a);
    }
}

template <int MODE, bool SUMS, int SLAB>
int launch_hubpc_slab(hipStream_t st, const SpmmArgs &a)
{
    using L = hubpc::Cfg<MODE, SLAB>;
    static std::atomic<uint64_t> done{0};
    constexpr size_t lds_bytes = sizeof(float) * L::LDS_FLOATS;
    static_assert(lds_bytes <= 160 * 1024, "one workgroup per CU");
    {   // dynamic-LDS opt-in, once per kernel and device, checked against what the runtime grants
        const int rc_lds = lds_opt_in(&spmm_hubpc_kernel<MODE, SUMS, SLAB>, lds_bytes, done, "spmm_hubpc_kernel");
        if (rc_lds != GNNX_OK) return rc_lds;
    }
    const int32_t n_slabs = (int32_t)ceil_div(a.n_feat, SLAB);
    const dim3 grid((uint32_t)((int64_t)a.n_big_rows * n_slabs));
    hipLaunchKernelGGL((spmm_hubpc_kernel<MODE, SUMS, SLAB>), grid, dim3(64 * (1 + hubpc::NP)), lds_bytes, st, a, a.hub_rows, n_slabs);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

template <int MODE, bool SUMS>
int launch_hubpc_kernel(hipStream_t st, const SpmmArgs &a)
{
#ifdef GNNX_EXPERIMENTS
    static const int slab_env = [] { const char *e = experiment_env("GNNX_PC_SLAB"); return e ? atoi(e) : 0; }();   // A/B: 64-feature slabs
    if (slab_env == 64) return launch_hubpc_slab<MODE, SUMS, 64>(st, a);
#endif
    return launch_hubpc_slab<MODE, SUMS, hubpc::kBigSlab>(st, a);
}


// the first n_big_rows hub rows (f32 rows of 16-byte pieces) on the producer / consumer kernel
int launch_hubpc(int mode, hipStream_t st, const SpmmArgs &a)
{
    if (a.bn_partial) return launch_hubpc_kernel<2, true>(st, a);
    switch (mode) {
    case 0: return launch_hubpc_kernel<0, false>(st, a);
    case 1: return launch_hubpc_kernel<1, false>(st, a);
    case 2: return launch_hubpc_kernel<2, false>(st, a);
    case 3: return launch_hubpc_kernel<3, false>(st, a);
    case 4: return launch_hubpc_kernel<4, false>(st, a);
    case 5: return launch_hubpc_kernel<5, false>(st, a);
    default: return launch_hubpc_kernel<6, false>(st, a);
    }
}

template <int G, int VEC, int U, int TPB>
void launch_stream(int mode, dim3 grid, hipStream_t st, const SpmmArgs &a)
{
#define GNNX_STREAM(M, XT) hipLaunchKernelGGL((spmm_stream_kernel<G, VEC, U, M, TPB, XT>), grid, dim3(TPB), 0, st, a)
    if (a.x_bf16) {  // bf16 feature rows: the three plain modes (a prologue goes with f32 rows)
        switch (mode) {
        case 0: GNNX_STREAM(0, bf16_t); break;
        case 1: GNNX_STREAM(1, bf16_t); break;
        case 2: GNNX_STREAM(2, bf16_t); break;
        default: GNNX_STREAM(6, bf16_t); break;
        }
        return;
    }
    if constexpr (VEC == 4 && G >= 32) {
        if (a.bn_partial) {   // backward aggregation + BatchNorm sums (mode 2, f32 rows)
            hipLaunchKernelGGL((spmm_stream_kernel<G, VEC, U, 2, TPB, float, true>), grid, dim3(TPB), 0, st, a);
            return;
        }
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
void launch_rows(int mode, dim3 grid, hipStream_t st, const SpmmArgs &a)
{
#define GNNX_ROWS(M, XT) hipLaunchKernelGGL((spmm_kernel<G, VEC, U, M, XT>), grid, dim3(256), 0, st, a)
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

// Side streams of the aggregation (the hub kernels beside the row kernel): one pair per host thread and device, created on first
// use, with the events of the fork / join.  `stream` carries spmm_hub_kernel when it runs beside the row kernel, `stream2` the
// producer / consumer kernel of the longest rows (always beside: the two hub kernels overlap, each fills the CUs the other leaves).
// Per thread because a rank = one host thread = one stream (gnnx.h): two threads never share the events.  Released when the thread
// ends (a rank thread of an in-process group that exits leaves nothing behind).
struct SideStream {
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, join2 = nullptr;
    void destroy()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        if (stream2) (void)hipStreamSynchronize(stream2);
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (join2) (void)hipEventDestroy(join2);
        if (stream) (void)hipStreamDestroy(stream);
        if (stream2) (void)hipStreamDestroy(stream2);
        *this = SideStream{};
    }
};
constexpr int kSideMaxDev = 64;
struct SideStreamTable {
    SideStream t[kSideMaxDev];
    ~SideStreamTable()
    {
        for (SideStream &s : t) s.destroy();
    }
};
// the side streams live on the device that owns the CALLER's stream (the current device for the null stream)
SideStream *side_stream(hipStream_t caller)
{
    static thread_local SideStreamTable table;
    int dev = 0;
    if (caller == nullptr || hipStreamGetDevice(caller, &dev) != hipSuccess) {
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    }
    if (dev < 0 || dev >= kSideMaxDev) return nullptr;
    SideStream &s = table.t[dev];
    if (!s.stream) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return nullptr;
        if (cur != dev && hipSetDevice(dev) != hipSuccess) return nullptr;
        // Priorities, for the QUEUES they come with: HIP maps streams of one priority onto a small pool of hardware queues, and two
        // streams that share a queue are not independent -- a kernel packet that follows another kernel of ITS stream carries the
        // barrier bit and waits for every packet in front of it in the queue, the other stream's too.  With the producer / consumer
        // kernel's stream on the caller's queue the row kernel started only when that kernel had finished (one rank of eight: 1.97
        // instead of 1.79 ms per aggregation on the runs that drew that mapping; kernel timeline in DESIGN_HISTORY.md).  The three
        // priority classes have queue pools of their own: high for the chain-bound kernel of the longest rows (it should never wait
        // for a CU), low for the hub kernel that fills what the row kernel leaves.
        int prio_low = 0, prio_high = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_low = prio_high = 0;
        const bool ok = hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio_low) == hipSuccess &&
                        hipStreamCreateWithPriority(&s.stream2, hipStreamNonBlocking, prio_high) == hipSuccess &&
                        hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess &&
                        hipEventCreateWithFlags(&s.join2, hipEventDisableTiming) == hipSuccess;
        if (cur != dev) (void)hipSetDevice(cur);
        if (!ok) {
            s.destroy();
            return nullptr;
        }
    }
    return &s;
}

template <int G, int VEC, int U>
int launch_mode(const SpmmArgs &a_in, const gnnx_spmm_plan *plan, int pro, hipStream_t st)
{
    constexpr int GROUPS = 256 / G;
    const int feat_per_tile = G * VEC;
    SpmmArgs a = a_in;
    if (G < 2 * kPlanBlockRows) a.block_starts = nullptr;  // plan blocks hold up to kPlanBlockRows rows (< G needed)
    dim3 grid;
    grid.y = (uint32_t)ceil_div(a.n_feat, feat_per_tile);
    // kernel MODE: 0 forward, 1 colscale, 2 vals (the backward: norm streamed per entry), 6 vals + colscale, 3/4/5 forward with
    // a ReLU / BN / BN+ReLU prologue
    const int mode = a.vals != nullptr ? (a.colscale ? 6 : 2) : (a.colscale ? 1 : (pro ? 2 + pro : 0));
    bool stream = false;
    if constexpr (G >= 8) stream = use_stream_kernel(G);
    // The hub kernel (LDS rings, few registers, 4 wavefronts per CU) and the row kernel (registers, no LDS) touch disjoint rows and
    // fit on a CU together.  When the hub kernel is latency-bound (spmm_impl: hub_beside) it goes to a side stream -- forked from
    // and joined back into the caller's stream with events -- and hides under the row kernel (RMAT 1M / 10M, F = 128: a 0.49 ms hub
    // kernel beside a 0.45 ms streaming kernel); when it is bandwidth-bound the two would only share the same bytes per second.
    SideStream *side = nullptr;
    struct JoinGuard {   // every exit path orders the caller's stream behind whatever this call put on the side streams
        SideStream *s = nullptr;
        hipStream_t to = nullptr;
        bool used1 = false, used2 = false;
        ~JoinGuard()
        {
            if (!s) return;
            if (used1) {
                if (hipEventRecord(s->join, s->stream) == hipSuccess) (void)hipStreamWaitEvent(to, s->join, 0);
                else (void)hipStreamSynchronize(s->stream);
            }
            if (used2) {
                if (hipEventRecord(s->join2, s->stream2) == hipSuccess) (void)hipStreamWaitEvent(to, s->join2, 0);
                else (void)hipStreamSynchronize(s->stream2);
            }
        }
    } joiner;
    static const int side_env = [] { const char *e = experiment_env("GNNX_SPMM_SIDE"); return e ? atoi(e) : -1; }();   // A/B: 0 never, 1 always
    const bool beside = side_env < 0 ? a.hub_beside != 0 : side_env != 0;
    const int32_t n_rest = a.n_hub_rows - a.n_big_rows;   // hub rows of spmm_hub_kernel
    if (a.n_big_rows > 0 || (n_rest > 0 && beside)) {
        side = side_stream(st);
        GNNX_REQUIRE(side, GNNX_ERR_HIP, "could not create the aggregation's side streams");
        GNNX_HIP_CHECK(hipEventRecord(side->fork, st));
        joiner.s = side;
        joiner.to = st;
    }
    if (a.n_big_rows > 0) {
        // the longest rows -- each a chain of dependent adds, a CU of its own per (row, slab) -- always beside everything else, on the
        // high-priority side stream (side_stream(): a hardware queue of its own)
        GNNX_HIP_CHECK(hipStreamWaitEvent(side->stream2, side->fork, 0));
        joiner.used2 = true;   // from here on every exit path joins the stream back into the caller's (an error return included)
        const int rc = launch_hubpc(mode, side->stream2, a);
        if (rc != GNNX_OK) return rc;
    }
    if (n_rest > 0) {
        if (beside) {
            GNNX_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
            joiner.used1 = true;
        }
        const int rc = launch_hub<VEC>(mode, beside ? side->stream : st, a);
        if (rc != GNNX_OK) return rc;
    }
    if (stream) {
        if constexpr (G >= 8) {
            constexpr int R = StreamCfg<G>::R;
            // One wavefront per workgroup: a workgroup's CU slot is held until its slowest wavefront ends, and
            // blocks of a power-law graph differ a lot in non-zeros, so multi-wave workgroups strand slots.
            static const int tpb_env = [] { const char *v = experiment_env("GNNX_SPMM_TPB"); return v ? atoi(v) : 64; }();
            const int tpb = tpb_env == 256 ? 256 : 64;
            const int64_t n_blk = a.block_starts ? a.n_blocks : ceil_div(a.n_rows, R);
            const int64_t n_waves = ceil_div(n_blk, a.blocks_per_wave > 0 ? a.blocks_per_wave : 1);
            grid.x = (uint32_t)ceil_div(n_waves, tpb / G);
            if (tpb == 256) launch_stream<G, VEC, U, 256>(mode, grid, st, a);
            else launch_stream<G, VEC, U, 64>(mode, grid, st, a);
        }
    } else {
        grid.x = (uint32_t)ceil_div(a.n_rows, GROUPS);
        if (grid.x == 0) return GNNX_OK;
        launch_rows<G, VEC, U>(mode, grid, st, a);
    }
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;   // (the JoinGuard orders whatever follows on the caller's stream behind the hub kernel)
}

}  // namespace

namespace {
int plan_build(gnnx_spmm_plan *plan, const int32_t *d_rowptr, int32_t n_rows, int32_t chunk, hipStream_t st)
{
    GNNX_HIP_CHECK(hipMalloc(&plan->d_counters, 4 * sizeof(unsigned long long)));
    GNNX_HIP_CHECK(hipMemsetAsync(plan->d_counters, 0, 4 * sizeof(unsigned long long), st));
    GNNX_HIP_CHECK(hipHostMalloc((void **)&plan->h_err, sizeof(int32_t), hipHostMallocMapped));
    *plan->h_err = 0;
    GNNX_HIP_CHECK(hipHostGetDevicePointer((void **)&plan->d_err, plan->h_err, 0));
    if (n_rows == 0) return GNNX_OK;
    dim3 grid((uint32_t)ceil_div(n_rows, 256));
    hipLaunchKernelGGL(plan_count_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, chunk, plan->d_counters);
    GNNX_LAUNCH_CHECK();
    unsigned long long h[2];
    int32_t h_nnz = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(h, plan->d_counters, sizeof(h), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_nnz, d_rowptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    plan->nnz = h_nnz;
    plan->n_split_rows = (int32_t)h[0];
    plan->n_hub_nnz = (int64_t)h[1];
    if (plan->n_split_rows > 0) {
        // work list of spmm_hub_kernel: the hub rows, longest first (ties by row id: the list does not depend on the order the
        // atomics handed out positions)
        DeviceFreeSync list;   // every temporary below is released on every exit path
        GNNX_HIP_CHECK(hipMalloc(&list.p, sizeof(int2) * (size_t)plan->n_split_rows));
        int2 *d_list = static_cast<int2 *>(list.p);
        hipLaunchKernelGGL(plan_fill_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, chunk, plan->d_counters, d_list);
        GNNX_LAUNCH_CHECK();
        std::vector<int2> h_rows((size_t)plan->n_split_rows);
        GNNX_HIP_CHECK(hipMemcpyAsync(h_rows.data(), d_list, sizeof(int2) * h_rows.size(), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        std::sort(h_rows.begin(), h_rows.end(), [](const int2 &x, const int2 &y) { return x.y != y.y ? x.y > y.y : x.x < y.x; });
        std::vector<int32_t> h_hub(h_rows.size());
        for (size_t i = 0; i < h_rows.size(); i++) h_hub[i] = h_rows[i].x;
        plan->max_hub_degree = h_rows.empty() ? 0 : h_rows[0].y;
        plan->h_hub_rows = h_hub;
        plan->h_hub_degrees.resize(h_rows.size());
        plan->h_hub_prefix.assign(h_rows.size() + 1, 0);
        for (size_t i = 0; i < h_rows.size(); i++) {
            plan->h_hub_degrees[i] = h_rows[i].y;
            plan->h_hub_prefix[i + 1] = plan->h_hub_prefix[i] + h_rows[i].y;
        }
        GNNX_HIP_CHECK(hipMalloc(&plan->d_hub_rows, sizeof(int32_t) * h_hub.size()));
        GNNX_HIP_CHECK(hipMemcpyAsync(plan->d_hub_rows, h_hub.data(), sizeof(int32_t) * h_hub.size(), hipMemcpyHostToDevice, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
    }
    // non-zero-balanced row blocks
    {
        static const int env_bn = [] { const char *v = experiment_env("GNNX_SPMM_BLOCK_NNZ"); return v ? atoi(v) : 0; }();
        plan->block_nnz = env_bn > 0 ? env_bn : (chunk < 256 ? chunk : 256);
        DeviceFreeSync flag_g, pos_g, tmp_g;
        size_t tmp_bytes = 0;
        GNNX_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp_bytes, (int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)n_rows,
                                               rocprim::plus<int32_t>()));
        GNNX_HIP_CHECK(hipMalloc(&flag_g.p, sizeof(int32_t) * (size_t)n_rows));
        GNNX_HIP_CHECK(hipMalloc(&pos_g.p, sizeof(int32_t) * (size_t)n_rows));
        GNNX_HIP_CHECK(hipMalloc(&tmp_g.p, tmp_bytes > 0 ? tmp_bytes : 4));
        int32_t *flag = static_cast<int32_t *>(flag_g.p), *pos = static_cast<int32_t *>(pos_g.p);
        hipLaunchKernelGGL(plan_block_flags_kernel, grid, dim3(256), 0, st, d_rowptr, n_rows, plan->block_nnz, flag);
        GNNX_LAUNCH_CHECK();
        GNNX_HIP_CHECK(rocprim::exclusive_scan(tmp_g.p, tmp_bytes, flag, pos, 0, (size_t)n_rows, rocprim::plus<int32_t>(), st));
        int32_t last[2];
        GNNX_HIP_CHECK(hipMemcpyAsync(&last[0], pos + (n_rows - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipMemcpyAsync(&last[1], flag + (n_rows - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GNNX_HIP_CHECK(hipStreamSynchronize(st));
        plan->n_blocks = last[0] + last[1];
        GNNX_HIP_CHECK(hipMalloc(&plan->d_block_starts, sizeof(int32_t) * ((size_t)plan->n_blocks + 1)));
        hipLaunchKernelGGL(plan_block_fill_kernel, grid, dim3(256), 0, st, flag, pos, n_rows, plan->n_blocks,
                           plan->d_block_starts);
        GNNX_LAUNCH_CHECK();
        GNNX_HIP_CHECK(hipStreamSynchronize(st));   // the temporaries are freed (hipFree) when this scope ends: the kernels are done
    }
    return GNNX_OK;
}
}  // namespace

GNNX_API int gnnx_spmm_plan_create(const int32_t *d_rowptr, int32_t n_rows, int32_t chunk, int32_t max_feat,
                                   gnnx_spmm_plan **plan_out, void *stream)
{
    GNNX_REQUIRE(d_rowptr && plan_out, GNNX_ERR_INVALID_ARG, "null pointer");
    *plan_out = nullptr;
    GNNX_REQUIRE(n_rows >= 0 && chunk > 0 && max_feat > 0, GNNX_ERR_INVALID_ARG, "n_rows/chunk/max_feat");
    (void)max_feat;  // a plan holds no per-feature storage any more (kept in the signature: callers size nothing by it)
    auto *plan = new gnnx_spmm_plan();
    plan->n_rows = n_rows;
    plan->chunk = chunk;
    const int rc = plan_build(plan, d_rowptr, n_rows, chunk, as_stream(stream));
    if (rc != GNNX_OK) {   // nothing half-built reaches the caller
        (void)hipStreamSynchronize(as_stream(stream));
        (void)gnnx_spmm_plan_destroy(plan);
        return rc;
    }
    *plan_out = plan;
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_hub_ids_structured(const gnnx_spmm_plan *plan, int *structured)
{
    GNNX_REQUIRE(plan && structured, GNNX_ERR_INVALID_ARG, "null pointer");
    // The hub rows of this CSR are the rows the TRANSPOSED aggregation gathers most (a vertex with many out-edges is read by many
    // rows of A^T).  Synthetic power-law generators (R-MAT) draw the bits of an id independently, so the hubs are the ids with few
    // one-bits: with a power-of-two row pitch of the gathered matrix the address bits that pick the memory channel / cache slice are
    // mostly zero for the hottest rows and the caller should use gnnx_gather_row_stride.  Measured here as the non-zero-weighted mean
    // popcount of the hub ids against half the id width (what ids spread at random show).  Hubs sorted to the front as one dense
    // block of consecutive ids have few one-bits too but consecutive addresses, which spread by themselves: not flagged.
    // (The residues of the ids modulo 128 do NOT tell: the multiplicative relabelling keeps them -- 128 divides 10^7 -- and cures the
    // pile-up all the same; the bits that matter sit higher.)
    *structured = 0;
    if (plan->h_hub_rows.size() < 64 || plan->n_rows < 1024) return GNNX_OK;
    double bits = 0.0, weight = 0.0;
    int32_t lo = plan->h_hub_rows[0], hi = plan->h_hub_rows[0];
    for (size_t i = 0; i < plan->h_hub_rows.size(); i++) {
        const double w = (double)plan->h_hub_degrees[i];
        bits += w * (double)__builtin_popcount((uint32_t)plan->h_hub_rows[i]);
        weight += w;
        lo = plan->h_hub_rows[i] < lo ? plan->h_hub_rows[i] : lo;
        hi = plan->h_hub_rows[i] > hi ? plan->h_hub_rows[i] : hi;
    }
    int width = 0;
    while ((1ll << width) < (long long)plan->n_rows) width++;
    const bool dense_block = (int64_t)hi - (int64_t)lo < 4 * (int64_t)plan->h_hub_rows.size();
    *structured = !dense_block && bits / weight < 0.7 * 0.5 * (double)width;
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_set_big_row_threshold(gnnx_spmm_plan *plan, int32_t threshold)
{
    GNNX_REQUIRE(plan, GNNX_ERR_INVALID_ARG, "plan is null");
    plan->big_row_threshold = threshold < 0 ? -1 : threshold;
    return GNNX_OK;
}

// GNNX_OK, or GNNX_ERR_HIP when a launch of this plan's producer / consumer kernel that has COMPLETED gave up a wait (its rows were
// not written).  Reads one word of pinned host memory: no synchronisation; call it behind a stream synchronisation to cover the
// launches before it.  Every planned aggregation call makes the same check on entry.
GNNX_API int gnnx_spmm_plan_status(const gnnx_spmm_plan *plan)
{
    GNNX_REQUIRE(plan, GNNX_ERR_INVALID_ARG, "plan is null");
    if (plan->h_err) {
        const int32_t e = *(volatile const int32_t *)plan->h_err;
        GNNX_REQUIRE(e == 0, GNNX_ERR_HIP,
                     "spmm_hubpc_kernel: a %s wait timed out in an earlier launch of this plan; the rows it owned were not written",
                     (e & 1) ? "consumer" : "producer");
    }
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_destroy(gnnx_spmm_plan *plan)
{
    if (!plan) return GNNX_OK;
    if (plan->d_hub_rows) (void)hipFree(plan->d_hub_rows);
    if (plan->d_counters) (void)hipFree(plan->d_counters);
    if (plan->h_err) (void)hipHostFree(plan->h_err);
    if (plan->d_block_starts) (void)hipFree(plan->d_block_starts);
    delete plan;
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_plan_info(const gnnx_spmm_plan *plan, int64_t *n_hub_rows, int64_t *n_hub_nnz)
{
    GNNX_REQUIRE(plan, GNNX_ERR_INVALID_ARG, "plan is null");
    if (n_hub_rows) *n_hub_rows = plan->n_split_rows;
    if (n_hub_nnz) *n_hub_nnz = plan->n_hub_nnz;
    return GNNX_OK;
}

namespace {
// BatchNorm-backward sums riding in the backward aggregation (gnnx_spmm_csr_bn_sums_f32)
struct BnSums {
    const float *h;
    int64_t ldh;
    const float *mean, *var, *gamma, *beta;
    float eps;
    int relu;
    float *partial;
    int32_t blocks_per_wave, n_stream_waves;
};
constexpr int kBnSumsBlocksPerWave = 4;

// partial rows the streaming kernel writes for a graph / width (= the G-lane groups it launches); 0: shape not covered
int64_t bn_sums_stream_waves(int32_t n_rows, int32_t n_feat, const gnnx_spmm_plan *plan)
{
    if (n_feat % 4 || n_feat <= 64) return 0;           // needs the VEC 4 streaming kernel (G >= 32)
    const int G = n_feat / 4 > 32 ? 64 : 32, groups = 64 / G, R = G / 2;
    const int64_t n_blk = plan && plan->d_block_starts ? plan->n_blocks : ceil_div(n_rows, R);
    return ceil_div(ceil_div(n_blk, kBnSumsBlocksPerWave), groups) * groups;
}

// Hub kernel beside the row kernel (side stream) or in front of it (same stream)?  Beside, when its time is the dependent-add chain
// of its LONGEST row rather than its bytes: the two measured constants of the hub kernel, in one place.
//   chain: kHubNsPerNeighbour per non-zero of the longest row (one accumulator per feature, the reference's order);
//   bytes: hub non-zeros x row bytes at kHubBytesPerNs.
// RMAT 1M / 10M, F = 128: beside (0.92 -> 0.79 ms); RMAT 10M / 100M, F = 256 (28 GB of hub rows): in front (13.59 vs 13.76 ms beside).
constexpr double kHubNsPerNeighbour = 17.0;   // spmm_hub_kernel, one row alone: 4.29 ms for 250 k neighbours (scripts/exp_hub_row.py; the pc kernel: 4.5)   // profiles/r03: 0.75 ms for the 62 k-entry longest row of RMAT 1M / 10M
constexpr double kHubBytesPerNs = 7000.0;     // ~7 TB/s on the hub rows of the headline graph
constexpr double kHubBesideBytes = 8.0e9;     // hub rows of fewer bytes run beside the row kernel (28 GB on the whole headline graph: in front)
bool hub_is_chain_bound(int32_t max_hub_degree, int64_t n_hub_nnz, int32_t n_feat, int bytes_per_feature)
{
    const double chain_ns = (double)max_hub_degree * kHubNsPerNeighbour;
    const double bytes = (double)n_hub_nnz * (double)n_feat * bytes_per_feature;
    // ... or when its bytes are few: a small hub kernel does not fill the memory system by itself (one rank of 8 of the 10 M / 100 M
    // graph: 2.5 GB in 0.45 ms = 5.5 TB/s, in FRONT of a 1.33 ms streaming kernel -- kernel trace, profiles/r04_shard_rank0_trace.txt);
    // beside the streaming kernel the two share what the memory system delivers
    return chain_ns > bytes / kHubBytesPerNs || bytes < kHubBesideBytes;
}

// How many of the plan's hub rows (longest first) take the producer / consumer kernel for an aggregation of `n_feat` f32 features:
// those whose chain in spmm_hub_kernel (kHubNsPerNeighbour per non-zero) would be longer than HALF of what all the hub rows' bytes
// take at kHubBytesPerNs -- on the whole 10 M / 100 M graph a handful of rows beyond ~100 k non-zeros (everything else hides behind
// 28 GB of hub rows), on one rank's eighth of it the rows beyond ~15 k (a rank keeps the longest rows whole but only an eighth of the
// bytes), on RMAT 1M / 10M everything beyond kHubBigRowMin.  (Measured against a rule that also let the OTHER rows' bytes hide a
// chain -- fewer rows here, more on spmm_hub_kernel: one rank of 8, aggregation 1.78 -> 2.15 ms; RMAT 1M / 10M 0.59 -> 0.65.)  A
// threshold fixed by the caller (tests) wins.
int32_t big_rows_for(const gnnx_spmm_plan *plan, int32_t n_feat)
{
    const auto &deg = plan->h_hub_degrees;
    int32_t thr = plan->big_row_threshold;
    if (thr < 0) {
        const double bytes_ns = (double)plan->n_hub_nnz * (double)n_feat * 4.0 / kHubBytesPerNs;
        const double t = 0.5 * bytes_ns / kHubNsPerNeighbour;
        thr = t > 2.0e9 ? 2000000000 : (int32_t)t;
        if (thr < kHubBigRowMin) thr = kHubBigRowMin;
    }
    return (int32_t)(std::partition_point(deg.begin(), deg.end(), [thr](int32_t d) { return d > thr; }) - deg.begin());
}

int spmm_impl(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr, const int32_t *d_colidx, const float *d_vals,
              const float *d_colscale, const float *d_rowscale, const float *d_bias, const float *d_X, int64_t ldx, float beta,
              float *d_Y, int64_t ldy, const gnnx_spmm_fusion *fusion, const gnnx_spmm_plan *plan, void *stream, bool x_bf16 = false,
              const BnSums *bs = nullptr)
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
    a.blocks_per_wave = 1;
    if (bs) {
        a.bn_h = bs->h;
        a.bn_ldh = bs->ldh;
        a.bn_partial = bs->partial;
        a.bn_relu = bs->relu;
        a.blocks_per_wave = bs->blocks_per_wave;
        a.n_stream_waves = bs->n_stream_waves;
        a.pro_mean = bs->mean;
        a.pro_var = bs->var;
        a.pro_gamma = bs->gamma;
        a.pro_beta = bs->beta;
        a.pro_eps = bs->eps;
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
#ifdef GNNX_EXPERIMENTS
    static const int policy_env = [] { const char *e = experiment_env("GNNX_SPMM_POLICY"); return e ? atoi(e) : 0; }();
    a.exp_policy = policy_env;
    static const int hotk_env = [] { const char *e = experiment_env("GNNX_SPMM_HOTK"); return e ? atoi(e) : 0; }();
    a.exp_hot_k = hotk_env;
#endif
    if (plan) {
        const int rc = gnnx_spmm_plan_status(plan);   // an earlier launch of this plan gave up a wait: say so, never carry on silently
        if (rc != GNNX_OK) return rc;
        a.err_word = plan->d_err;
    }
    if (plan && plan->n_split_rows > 0) {
        a.split_threshold = plan->chunk;
        a.hub_rows = plan->d_hub_rows;
        a.n_hub_rows = plan->n_split_rows;
        // Hub kernel beside the row kernel (side stream) or in front of it (same stream)?  Beside, when its time is the dependent-add
        // chain of its longest row (~12 ns per neighbour) rather than its bytes (hub non-zeros x row bytes at ~7 TB/s):
        // RMAT 1M / 10M, F = 128: 0.92 -> 0.79 ms; RMAT 10M / 100M, F = 256 (28 GB of hub rows): in front, 13.59 vs 13.76 ms beside.
        a.hub_beside = hub_is_chain_bound(plan->max_hub_degree, plan->n_hub_nnz, n_feat, x_bf16 ? 2 : 4);
    }
    hipStream_t st = as_stream(stream);
    auto aligned16 = [](const void *p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec4 = (n_feat % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (x_bf16 ? (reinterpret_cast<uintptr_t>(d_X) & 7u) == 0 : aligned16(d_X)) && aligned16(d_Y) &&
                      aligned16(d_bias) && aligned16(a.pro_mean) && aligned16(a.pro_var) && aligned16(a.pro_gamma) &&
                      aligned16(a.pro_beta);
    // the producer / consumer kernel reads f32 rows in 16-byte pieces; other rows: spmm_hub_kernel for every hub row
    if (a.n_hub_rows > 0 && !x_bf16 && vec4) {
#ifdef GNNX_EXPERIMENTS
        static const int pc_exp = [] { const char *e = experiment_env("GNNX_PC_EXP"); return e ? atoi(e) : 0; }();
        a.pc_experiment = pc_exp;
#endif
        a.n_big_rows = big_rows_for(plan, n_feat);
#ifdef GNNX_EXPERIMENTS
        static const int pc_off = [] { const char *e = experiment_env("GNNX_PC_OFF"); return e ? atoi(e) : 0; }();
        if (pc_off) a.n_big_rows = 0;
#endif
        // With the chain-bound rows on the producer / consumer kernel (beside everything, on a queue of its own) what is left for
        // spmm_hub_kernel goes IN FRONT of the row kernel: three kernels at once share one fabric, and the one that loses is the chain
        // kernel, which then ends last (one rank of eight 8.65 -> 8.43 ms per step, RMAT 1M / 10M F = 128 2.41 -> 2.34, products-shaped
        // 11.87 -> 11.74: GNNX_SPMM_SIDE A/B on one box each, round 5).  Round 4's rule put it beside whenever its bytes were few.
        if (a.n_big_rows > 0) a.hub_beside = 0;
    }
    if (x_bf16 && !vec4) {
        // bf16 rows that are not 8-byte pieces (n_feat % 4 != 0 or an unaligned X): LDS-DMA moves one DWORD per lane whatever the
        // load size, so a 2-byte-per-lane ring layout does not exist -- the hub rows stay with the row / streaming kernel (same bits)
        a.split_threshold = 0;
        a.hub_rows = nullptr;
        a.n_hub_rows = 0;
    }
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

GNNX_API int gnnx_spmm_csr_bn_sums_workspace(int32_t n_rows, int32_t n_feat, const gnnx_spmm_plan *plan, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_rows >= 0 && n_feat >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    const int64_t sw = bn_sums_stream_waves(n_rows, n_feat, plan);
    const int64_t n_partial = sw + (plan ? plan->n_split_rows : 0);
    size_t cs = 0;
    int rc = gnnx_colsum_workspace(n_partial > 0 ? n_partial : 1, 2 * n_feat, &cs);
    if (rc) return rc;
    *bytes = sizeof(float) * ((size_t)n_partial * 2 * (size_t)n_feat + 2 * (size_t)n_feat) + cs + 256;
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_csr_bn_sums_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr, const int32_t *d_colidx,
                                       const float *d_vals, const float *d_G, int64_t ldg, float *d_dY, int64_t ldy, const float *d_H,
                                       int64_t ldh, const float *d_mean, const float *d_var, float eps, const float *d_gamma,
                                       const float *d_beta, int relu, float *d_dgamma, float *d_dbeta, void *d_workspace,
                                       size_t workspace_bytes, const gnnx_spmm_plan *plan, void *stream)
{
    GNNX_REQUIRE(n_rows > 0 && n_feat > 0, GNNX_ERR_INVALID_ARG, "empty problem");
    GNNX_REQUIRE(d_vals && d_G && d_dY && d_H && d_mean && d_var && d_dgamma && d_dbeta, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_REQUIRE(ldh >= n_feat, GNNX_ERR_SHAPE, "leading dimension smaller than n_feat");
    auto aligned16 = [](const void *p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int64_t sw = bn_sums_stream_waves(n_rows, n_feat, plan);
    const bool ok = sw > 0 && ldg % 4 == 0 && ldy % 4 == 0 && ldh % 4 == 0 && aligned16(d_G) && aligned16(d_dY) && aligned16(d_H) &&
                    aligned16(d_mean) && aligned16(d_var) && aligned16(d_gamma) && aligned16(d_beta);
    GNNX_REQUIRE(ok, GNNX_ERR_UNSUPPORTED, "shape not covered by the fused sums (n_feat %% 4 == 0, n_feat > 64, 16-byte aligned rows): use "
                                           "gnnx_spmm_csr_f32 + gnnx_bn_relu_bwd_sums_f32");
    size_t need = 0;
    int rc = gnnx_spmm_csr_bn_sums_workspace(n_rows, n_feat, plan, &need);
    if (rc) return rc;
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    const int64_t n_partial = sw + (plan ? plan->n_split_rows : 0);
    float *partial = static_cast<float *>(d_workspace);
    float *sums = partial + (size_t)n_partial * 2 * n_feat;
    char *cs_ws = reinterpret_cast<char *>(sums + 2 * n_feat);
    cs_ws += (256 - (reinterpret_cast<uintptr_t>(cs_ws) & 255)) & 255;
    const size_t cs_bytes = workspace_bytes - (size_t)(cs_ws - static_cast<char *>(d_workspace));
    BnSums bs{d_H, ldh, d_mean, d_var, d_gamma, d_beta, eps, relu, partial, kBnSumsBlocksPerWave, (int32_t)sw};
    rc = spmm_impl(n_rows, n_cols, n_feat, d_rowptr, d_colidx, d_vals, nullptr, nullptr, nullptr, d_G, ldg, 0.0f, d_dY, ldy, nullptr, plan,
                   stream, false, &bs);
    if (rc) return rc;
    // partial rows -> [2F] in a fixed order (two-stage deterministic tree), then to the caller's two [F] vectors
    rc = gnnx_colsum_f32(partial, 2 * (int64_t)n_feat, n_partial, 2 * n_feat, 0.0f, sums, cs_ws, cs_bytes, stream);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    GNNX_HIP_CHECK(hipMemcpyAsync(d_dbeta, sums, sizeof(float) * (size_t)n_feat, hipMemcpyDeviceToDevice, st));
    GNNX_HIP_CHECK(hipMemcpyAsync(d_dgamma, sums + n_feat, sizeof(float) * (size_t)n_feat, hipMemcpyDeviceToDevice, st));
    return GNNX_OK;
}

GNNX_API int gnnx_spmm_csr_bf16_f32(int32_t n_rows, int32_t n_cols, int32_t n_feat, const int32_t *d_rowptr,
                                    const int32_t *d_colidx, const float *d_vals, const float *d_colscale,
                                    const float *d_rowscale, const float *d_bias, const uint16_t *d_X_bf16, int64_t ldx, float beta,
                                    float *d_Y, int64_t ldy, const gnnx_spmm_plan *plan, void *stream)
{
    return spmm_impl(n_rows, n_cols, n_feat, d_rowptr, d_colidx, d_vals, d_colscale, d_rowscale, d_bias,
                     reinterpret_cast<const float *>(d_X_bf16), ldx, beta, d_Y, ldy, nullptr, plan, stream, true);
}
