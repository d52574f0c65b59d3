// Training-step pieces around the hot path: softmax cross-entropy on the logits of the
// last GCN layer and the SGD parameter update, so a 2/3-layer GCN runs as a whole training step on the device.
//   loss   (reference nn.cpp:442-453, forward only -- its backward throws):  l_i = -log( exp(x_i[t_i]) / (sum_c exp(x_ic) + 1e-20) ),
//          loss = (sum_i l_i) / N.   No max-subtraction, like the reference (logits of a GCN layer are O(1..100)).
//   dlogits (textbook; the reference has none that works):  (softmax(x_i) - onehot(t_i)) / N
//   SGD    (textbook; the reference's step() reads an empty velocity vector, nn.cpp:414):  p -= lr * (g + wd * p)
#include "gnnx_common.h"

#pragma clang fp contract(off)

using namespace gnnx;

namespace {

// one wavefront per row: lanes stride the classes; per-row loss to a buffer, gradient written in place
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float *X, int64_t ldx, const int32_t *target, int64_t n_rows,
                                                          int32_t n_cls, float inv_n, float *row_loss, float *dX, int64_t ldd,
                                                          int32_t *bad)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float *x = X + row * ldx;
    float sum = 0.f;
    for (int32_t c = lane; c < n_cls; c += 64) sum += expf(x[c]);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const int32_t t = target[row];
    if (t < 0 || t >= n_cls) {
        if (lane == 0) atomicOr(bad, 1);
        return;
    }
    const float denom = sum + 1e-20f;
    if (lane == 0 && row_loss) row_loss[row] = -logf(expf(x[t]) / denom);
    if (dX) {
        float *d = dX + row * ldd;
        const float rden = 1.0f / denom;
        for (int32_t c = lane; c < n_cls; c += 64) d[c] = (expf(x[c]) * rden - (c == t ? 1.f : 0.f)) * inv_n;
    }
}

// Vector form for class counts that are a multiple of 4 and at most 1024 (row stride a multiple of 4, 16-byte aligned): one
// wavefront per row, a lane holds classes 256 k + 4 lane .. + 3 (one 16-byte load per k: a 256-class row is ONE 1-KiB
// wave-instruction), exp evaluated once per element and kept in registers, two rows in flight per wavefront.  Waves walk the rows
// with a grid stride; with `colsum_partial` a wavefront also keeps the column sums of the gradient rows it wrote -- the last
// layer's bias gradient, which otherwise is one more 4 N C-byte pass -- and stores them as row `wave id` of the partials (summed
// by ce_colsum_reduce in a fixed order: deterministic).  Same expressions per element as the generic kernel below
// (expf(x) * (1 / (sum + 1e-20)), then - onehot, then * 1/N); only the ORDER of the row sum differs (tolerance-level, as documented).
template <int KV>
__global__ __launch_bounds__(256) void softmax_ce_vec_kernel(const float *X, int64_t ldx, const int32_t *target, int64_t n_rows,
                                                              int32_t n_cls, float inv_n, float *row_loss, float *dX, int64_t ldd,
                                                              int32_t *bad, float *colsum_partial)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float4 cs[KV];
#pragma unroll
    for (int k = 0; k < KV; k++) cs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r0 = wave; r0 < n_rows; r0 += 2 * n_waves) {
        float4 e[2][KV];
        int32_t t[2];
        bool have[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int64_t row = r0 + u * n_waves;
            have[u] = row < n_rows;
            t[u] = have[u] ? target[row] : 0;
#pragma unroll
            for (int k = 0; k < KV; k++) {
                const int32_t c = 256 * k + 4 * lane;
                e[u][k] = (have[u] && c < n_cls) ? *reinterpret_cast<const float4 *>(X + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (!have[u]) continue;   // wave-uniform
            const int64_t row = r0 + u * n_waves;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KV; k++) {
                if (256 * k + 4 * lane < n_cls) {
                    e[u][k].x = expf(e[u][k].x);
                    e[u][k].y = expf(e[u][k].y);
                    e[u][k].z = expf(e[u][k].z);
                    e[u][k].w = expf(e[u][k].w);
                    sum += e[u][k].x;
                    sum += e[u][k].y;
                    sum += e[u][k].z;
                    sum += e[u][k].w;
                }
            }
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
            if (t[u] < 0 || t[u] >= n_cls) {
                if (lane == 0) atomicOr(bad, 1);
                continue;
            }
            const float denom = sum + 1e-20f;
            if (lane == 0 && row_loss) row_loss[row] = -logf(expf(X[row * ldx + t[u]]) / denom);
            if (dX) {
                const float rden = 1.0f / denom;   // the gradient has no reference arithmetic to follow (its backward throws): one
                                                   // division per row, a multiplication per element
#pragma unroll
                for (int k = 0; k < KV; k++) {
                    const int32_t c = 256 * k + 4 * lane;
                    if (c < n_cls) {
                        float4 d;
                        d.x = (e[u][k].x * rden - (c + 0 == t[u] ? 1.f : 0.f)) * inv_n;
                        d.y = (e[u][k].y * rden - (c + 1 == t[u] ? 1.f : 0.f)) * inv_n;
                        d.z = (e[u][k].z * rden - (c + 2 == t[u] ? 1.f : 0.f)) * inv_n;
                        d.w = (e[u][k].w * rden - (c + 3 == t[u] ? 1.f : 0.f)) * inv_n;
                        *reinterpret_cast<float4 *>(dX + row * ldd + c) = d;
                        cs[k].x += d.x;
                        cs[k].y += d.y;
                        cs[k].z += d.z;
                        cs[k].w += d.w;
                    }
                }
            }
        }
    }
    if (colsum_partial) {
#pragma unroll
        for (int k = 0; k < KV; k++) {
            const int32_t c = 256 * k + 4 * lane;
            if (c < n_cls) *reinterpret_cast<float4 *>(colsum_partial + wave * n_cls + c) = cs[k];
        }
    }
}

// out[c] = sum over the partial rows, 4 interleaved parts per column combined in part order (fixed order: deterministic)
__global__ __launch_bounds__(256) void ce_colsum_reduce(const float *partial, int32_t n_part, int32_t n_cls, float *out)
{
    __shared__ float red[256];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int32_t f = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (f < n_cls) {
        int32_t b = part;
        for (; b + 12 < n_part; b += 16) {
            a0 += partial[(int64_t)b * n_cls + f];
            a1 += partial[(int64_t)(b + 4) * n_cls + f];
            a2 += partial[(int64_t)(b + 8) * n_cls + f];
            a3 += partial[(int64_t)(b + 12) * n_cls + f];
        }
        for (; b < n_part; b += 4) a0 += partial[(int64_t)b * n_cls + f];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && f < n_cls) out[f] = ((red[c] + red[64 + c]) + red[128 + c]) + red[192 + c];
}

// generic class counts: a lane walks the classes with stride 64; the column sums (when asked for) by one thread per class
// over the finished gradient (a second pass; only for shapes off the vector form)
__global__ __launch_bounds__(256) void ce_colsum_generic(const float *dX, int64_t ldd, int64_t n_rows, int32_t n_cls, float *partial,
                                                          int64_t rows_per_block)
{
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cls) return;
    float acc = 0.f;
    for (int64_t r = r0; r < r1; r++) acc += dX[r * ldd + c];
    partial[(int64_t)blockIdx.y * n_cls + c] = acc;
}

// deterministic two-stage mean of the per-row losses
__global__ __launch_bounds__(256) void sum_stage1(const float *v, int64_t n, float *partial)
{
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += v[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void sum_stage2(const float *partial, int n_blocks, float scale, float *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int b = 0; b < n_blocks; b++) acc += partial[b];
        *out = acc * scale;
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float *p, const float *g, int64_t n, float lr, float wd)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i];
        if (wd != 0.f) gi = gi + wd * p[i];
        p[i] = p[i] - lr * gi;
    }
}

constexpr int kSumBlocks = 256;
constexpr int kCeMaxGroups = 2048;   // vector form: at most this many workgroups of 4 wavefronts (8 per CU)
constexpr int kCeGenericParts = 256;

inline int ce_groups(int64_t n_rows)
{
    int64_t gsz = ceil_div(n_rows, 8);   // 4 wavefronts, 2 rows in flight each
    return (int)(gsz < 1 ? 1 : (gsz > kCeMaxGroups ? kCeMaxGroups : gsz));
}

}  // namespace

GNNX_API int gnnx_softmax_ce_colsum_workspace(int64_t n_rows, int32_t n_classes, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_rows >= 0 && n_classes >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    const size_t parts = (size_t)(4 * ce_groups(n_rows) > kCeGenericParts ? 4 * ce_groups(n_rows) : kCeGenericParts);
    *bytes = sizeof(float) * ((size_t)n_rows + kSumBlocks + 64 + parts * (size_t)n_classes) + 256;
    return GNNX_OK;
}

GNNX_API int gnnx_softmax_ce_workspace(int64_t n_rows, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_rows >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(float) * ((size_t)n_rows + kSumBlocks) + 256;
    return GNNX_OK;
}

// A shard's share of the loss over n_total rows: the mean's divisor is n_total (>= n_rows), so d_loss is THIS rank's term of the
// mean (summed over the ranks by the caller) and dlogits / the column sums carry 1 / n_total -- the same expression per element as
// the unsharded call with n_total rows.
GNNX_API int gnnx_softmax_ce_partial_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                                         int64_t n_total, float *d_loss, float *d_dlogits, int64_t ldd, float *d_colsum,
                                         void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(n_rows > 0 && n_classes > 0, GNNX_ERR_INVALID_ARG, "empty batch");
    GNNX_REQUIRE(n_total >= n_rows, GNNX_ERR_INVALID_ARG, "n_total < n_rows");
    GNNX_REQUIRE(d_logits && d_target && ldx >= n_classes, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_classes");
    GNNX_REQUIRE(!d_dlogits || ldd >= n_classes, GNNX_ERR_INVALID_ARG, "ldd < n_classes");
    GNNX_REQUIRE(!d_colsum || d_dlogits, GNNX_ERR_INVALID_ARG, "column sums are those of dlogits: dlogits is null");
    size_t need = 0;
    if (d_colsum) gnnx_softmax_ce_colsum_workspace(n_rows, n_classes, &need);
    else gnnx_softmax_ce_workspace(n_rows, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float *row_loss = static_cast<float *>(d_workspace);
    float *partial = row_loss + n_rows;
    int32_t *bad = reinterpret_cast<int32_t *>(partial + kSumBlocks);
    // column-sum partials: 16-byte aligned, behind the flag
    float *cpart = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(bad + 4) + 255u) & ~(uintptr_t)255u);
    GNNX_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
    const float inv_n = 1.0f / (float)n_total;
    const bool vec = n_classes % 4 == 0 && n_classes <= 1024 && ldx % 4 == 0 && (!d_dlogits || ldd % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(d_logits) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_dlogits) & 15u) == 0;
    if (vec) {
        const int groups = ce_groups(n_rows);
        float *cp = d_colsum ? cpart : nullptr;
        float *rl = d_loss ? row_loss : nullptr;
        const int kv = (n_classes + 255) / 256;
#define GNNX_CE_LAUNCH(KV)                                                                                                       \
    hipLaunchKernelGGL(softmax_ce_vec_kernel<KV>, dim3((uint32_t)groups), dim3(256), 0, st, d_logits, ldx, d_target, n_rows, n_classes, \
                       inv_n, rl, d_dlogits, ldd, bad, cp)
        if (kv == 1) GNNX_CE_LAUNCH(1);
        else if (kv == 2) GNNX_CE_LAUNCH(2);
        else if (kv == 3) GNNX_CE_LAUNCH(3);
        else GNNX_CE_LAUNCH(4);
#undef GNNX_CE_LAUNCH
        GNNX_LAUNCH_CHECK();
        if (d_colsum) {
            hipLaunchKernelGGL(ce_colsum_reduce, dim3((uint32_t)ceil_div(n_classes, 64)), dim3(256), 0, st, cpart, 4 * groups, n_classes,
                               d_colsum);
            GNNX_LAUNCH_CHECK();
        }
    } else {
        hipLaunchKernelGGL(softmax_ce_kernel, dim3((uint32_t)ceil_div(n_rows, 4)), dim3(256), 0, st, d_logits, ldx, d_target, n_rows,
                           n_classes, inv_n, d_loss ? row_loss : nullptr, d_dlogits, ldd, bad);
        GNNX_LAUNCH_CHECK();
        if (d_colsum) {
            const int parts = (int)(n_rows < kCeGenericParts ? n_rows : kCeGenericParts);
            const int64_t rpb = ceil_div(n_rows, parts);
            hipLaunchKernelGGL(ce_colsum_generic, dim3((uint32_t)ceil_div(n_classes, 256), (uint32_t)ceil_div(n_rows, rpb)), dim3(256), 0, st,
                               d_dlogits, ldd, n_rows, n_classes, cpart, rpb);
            GNNX_LAUNCH_CHECK();
            hipLaunchKernelGGL(ce_colsum_reduce, dim3((uint32_t)ceil_div(n_classes, 64)), dim3(256), 0, st, cpart,
                               (int32_t)ceil_div(n_rows, rpb), n_classes, d_colsum);
            GNNX_LAUNCH_CHECK();
        }
    }
    if (d_loss) {
        hipLaunchKernelGGL(sum_stage1, dim3(kSumBlocks), dim3(256), 0, st, row_loss, n_rows, partial);
        GNNX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(64), 0, st, partial, kSumBlocks, 1.0f / (float)n_total, d_loss);
        GNNX_LAUNCH_CHECK();
    }
    int32_t h_bad = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE, "target class out of range");
    return GNNX_OK;
}

GNNX_API int gnnx_softmax_ce_colsum_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                                        float *d_loss, float *d_dlogits, int64_t ldd, float *d_colsum, void *d_workspace,
                                        size_t workspace_bytes, void *stream)
{
    return gnnx_softmax_ce_partial_f32(d_logits, ldx, d_target, n_rows, n_classes, n_rows, d_loss, d_dlogits, ldd, d_colsum, d_workspace,
                                       workspace_bytes, stream);
}

GNNX_API int gnnx_softmax_ce_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                                 float *d_loss, float *d_dlogits, int64_t ldd, void *d_workspace, size_t workspace_bytes, void *stream)
{
    return gnnx_softmax_ce_colsum_f32(d_logits, ldx, d_target, n_rows, n_classes, d_loss, d_dlogits, ldd, nullptr, d_workspace,
                                      workspace_bytes, stream);
}

GNNX_API int gnnx_sgd_step_f32(float *d_param, const float *d_grad, int64_t n, float lr, float weight_decay, void *stream)
{
    GNNX_REQUIRE(n >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_param && d_grad, GNNX_ERR_INVALID_ARG, "null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sgd_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), d_param, d_grad, n, lr, weight_decay);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}
