// Synthetic inputs on the device: counter-based SplitMix64, bit-identical to gnn.cpp_amd/synth.py
// (graphs are generated from seeds on the GPU box, no files are shipped).
#include <cmath>

#include "gnnx_common.h"

using namespace gnnx;

namespace {

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + kGolden;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void rmat_kernel(uint64_t key, int32_t n_nodes, int64_t n_edges, int64_t first_edge,
                                                    int scale, uint32_t ta, uint32_t tb, uint32_t tc, int32_t *src,
                                                    int32_t *dst)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_edges; i += (int64_t)gridDim.x * 256) {
        uint64_t e = (uint64_t)(first_edge + i);
        uint64_t s = 0, d = 0;
        for (int l = 0; l < scale; l++) {
            uint32_t r = (uint32_t)(splitmix64(key ^ ((e * 64 + (uint64_t)l) * kGolden)) >> 32);
            uint64_t sbit = r >= tb;
            uint64_t dbit = ((r >= ta) & (r < tb)) | (r >= tc);
            s = (s << 1) | sbit;
            d = (d << 1) | dbit;
        }
        src[i] = (int32_t)(s % (uint64_t)n_nodes);
        dst[i] = (int32_t)(d % (uint64_t)n_nodes);
    }
}

__global__ __launch_bounds__(256) void uniform_kernel(uint64_t key, int64_t n, float scale, float *out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t r = splitmix64(key ^ ((uint64_t)i * kGolden)) >> 40;  // 24 bits
        float v = __fsub_rn(__fmul_rn((float)r, 1.1920928955078125e-07f), 1.0f);
        out[i] = scale == 1.0f ? v : __fmul_rn(v, scale);
    }
}

}  // namespace

GNNX_API int gnnx_rmat_edges(uint64_t seed, int32_t n_nodes, int64_t n_edges, int64_t first_edge, double a, double b,
                             double c, int32_t *d_src, int32_t *d_dst, void *stream)
{
    GNNX_REQUIRE(n_nodes > 0 && n_edges >= 0 && first_edge >= 0, GNNX_ERR_INVALID_ARG, "bad sizes");
    GNNX_REQUIRE(a > 0 && b >= 0 && c >= 0 && a + b + c <= 1.0, GNNX_ERR_INVALID_ARG, "bad quadrant probabilities");
    if (n_edges == 0) return GNNX_OK;
    GNNX_REQUIRE(d_src && d_dst, GNNX_ERR_INVALID_ARG, "null pointer");
    int scale = 1;
    while ((1ll << scale) < (long long)n_nodes) scale++;
    uint32_t ta = (uint32_t)(uint64_t)(a * 4294967296.0);
    uint32_t tb = (uint32_t)(uint64_t)((a + b) * 4294967296.0);
    uint32_t tc = (uint32_t)(uint64_t)((a + b + c) * 4294967296.0);
    int64_t blocks = ceil_div(n_edges, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(rmat_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), splitmix64(seed), n_nodes,
                       n_edges, first_edge, scale, ta, tb, tc, d_src, d_dst);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}

GNNX_API int gnnx_uniform_pm1_f32(uint64_t seed, int64_t n, float scale, float *d_out, void *stream)
{
    GNNX_REQUIRE(n >= 0, GNNX_ERR_INVALID_ARG, "negative size");
    if (n == 0) return GNNX_OK;
    GNNX_REQUIRE(d_out, GNNX_ERR_INVALID_ARG, "null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(uniform_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), splitmix64(seed), n, scale,
                       d_out);
    GNNX_LAUNCH_CHECK();
    return GNNX_OK;
}
