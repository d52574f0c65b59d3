// MEASUREMENT KERNELS for bench.py -- not part of the product library, not in include/gnnx.h.
// Two in-run ceilings of the memory system, so that bench.py's roofline line is a fraction of something measured on the same box
// in the same process (MI355X_MICROARCH.md gives 6.29 TB/s for the copy and 7.4-8.6 TB/s for gathers of whole rows from a table
// resident in the 256 MiB Infinity Cache):
//   ceil_copy_f4      float4 grid-stride copy: bytes read + bytes written per second = what HBM sustains for streams;
//   ceil_gather_rows  every output row is the sum of `deg` whole rows of a table picked by an index array (uniformly random rows):
//                     the access pattern of the aggregation (graph.cpp:208 as a CSR gather) with every structural difficulty
//                     removed -- constant degree, no row pointers, no epilogue, table as small as the caller makes it.
// Plain C ABI, device pointers, caller's stream; compiled for gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>

#define CEIL_API extern "C" __attribute__((visibility("default")))

__global__ __launch_bounds__(256) void copy_f4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// One wavefront per block of output rows; lane l owns floats 4l..4l+3 of a row (row_floats == 256: one 1-KiB row per
// wave-instruction).  B rows in flight per batch, two batches (the next output row's gathers are issued before the current row's are
// added), indices by one coalesced load per output row and v_readlane.
template <int B>
__global__ __launch_bounds__(64) void gather_rows_kernel(const float *__restrict__ table, const int32_t *__restrict__ idx, int64_t n_out,
                                                          int rows_per_wave, float *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wave;
    const float *tl = table + lane * 4;
    float4 cur[B], nxt[B];
    auto issue = [&](float4(&v)[B], int64_t r) {
        const int32_t mine = idx[r * B + (lane < B ? lane : 0)];
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int32_t c = __builtin_amdgcn_readlane(mine, u);
            v[u] = *reinterpret_cast<const float4 *>(tl + (int64_t)c * 256);
        }
    };
    auto reduce_store = [&](const float4(&v)[B], int64_t r) {
        float4 s = v[0];
#pragma unroll
        for (int u = 1; u < B; u++) {
            s.x += v[u].x;
            s.y += v[u].y;
            s.z += v[u].z;
            s.w += v[u].w;
        }
        *reinterpret_cast<float4 *>(out + r * 256 + lane * 4) = s;
    };
    if (r0 >= n_out) return;
    const int64_t r1 = r0 + rows_per_wave < n_out ? r0 + rows_per_wave : n_out;
    issue(cur, r0);
    for (int64_t r = r0; r < r1; r += 2) {
        if (r + 1 < r1) issue(nxt, r + 1);
        reduce_store(cur, r);
        if (r + 1 >= r1) break;
        if (r + 2 < r1) issue(cur, r + 2);
        reduce_store(nxt, r + 1);
    }
}

CEIL_API int ceil_copy_f4(const void *src, void *dst, size_t bytes, void *stream)
{
    if (!src || !dst || bytes % 16) return 1;
    hipLaunchKernelGGL(copy_f4_kernel, dim3(256 * 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// table: [table_rows, 256] f32; idx: [n_out * deg] int32 in [0, table_rows); out: [n_out, 256] f32.  deg is 8 or 16.
CEIL_API int ceil_gather_rows(const float *table, const int32_t *idx, int64_t n_out, int deg, float *out, void *stream)
{
    if (!table || !idx || !out || n_out <= 0 || (deg != 8 && deg != 16)) return 1;
    const int rows_per_wave = 16;
    const dim3 grid((uint32_t)((n_out + rows_per_wave - 1) / rows_per_wave));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (deg == 16) hipLaunchKernelGGL((gather_rows_kernel<16>), grid, dim3(64), 0, st, table, idx, n_out, rows_per_wave, out);
    else hipLaunchKernelGGL((gather_rows_kernel<8>), grid, dim3(64), 0, st, table, idx, n_out, rows_per_wave, out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
