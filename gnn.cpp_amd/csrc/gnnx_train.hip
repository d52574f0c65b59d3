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
        for (int32_t c = lane; c < n_cls; c += 64) d[c] = (expf(x[c]) / denom - (c == t ? 1.f : 0.f)) * inv_n;
    }
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

}  // namespace

GNNX_API int gnnx_softmax_ce_workspace(int64_t n_rows, size_t *bytes)
{
    GNNX_REQUIRE(bytes && n_rows >= 0, GNNX_ERR_INVALID_ARG, "bad arguments");
    *bytes = sizeof(float) * ((size_t)n_rows + kSumBlocks) + 256;
    return GNNX_OK;
}

GNNX_API int gnnx_softmax_ce_f32(const float *d_logits, int64_t ldx, const int32_t *d_target, int64_t n_rows, int32_t n_classes,
                                 float *d_loss, float *d_dlogits, int64_t ldd, void *d_workspace, size_t workspace_bytes, void *stream)
{
    GNNX_REQUIRE(n_rows > 0 && n_classes > 0, GNNX_ERR_INVALID_ARG, "empty batch");
    GNNX_REQUIRE(d_logits && d_target && ldx >= n_classes, GNNX_ERR_INVALID_ARG, "null pointer or ld < n_classes");
    GNNX_REQUIRE(!d_dlogits || ldd >= n_classes, GNNX_ERR_INVALID_ARG, "ldd < n_classes");
    size_t need = 0;
    gnnx_softmax_ce_workspace(n_rows, &need);
    GNNX_REQUIRE(d_workspace && workspace_bytes >= need, GNNX_ERR_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    float *row_loss = static_cast<float *>(d_workspace);
    float *partial = row_loss + n_rows;
    int32_t *bad = reinterpret_cast<int32_t *>(partial + kSumBlocks);
    GNNX_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(softmax_ce_kernel, dim3((uint32_t)ceil_div(n_rows, 4)), dim3(256), 0, st, d_logits, ldx, d_target, n_rows,
                       n_classes, 1.0f / (float)n_rows, d_loss ? row_loss : nullptr, d_dlogits, ldd, bad);
    GNNX_LAUNCH_CHECK();
    if (d_loss) {
        hipLaunchKernelGGL(sum_stage1, dim3(kSumBlocks), dim3(256), 0, st, row_loss, n_rows, partial);
        GNNX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(64), 0, st, partial, kSumBlocks, 1.0f / (float)n_rows, d_loss);
        GNNX_LAUNCH_CHECK();
    }
    int32_t h_bad = 0;
    GNNX_HIP_CHECK(hipMemcpyAsync(&h_bad, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GNNX_HIP_CHECK(hipStreamSynchronize(st));
    GNNX_REQUIRE(!h_bad, GNNX_ERR_INDEX_RANGE, "target class out of range");
    return GNNX_OK;
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
