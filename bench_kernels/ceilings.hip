// MEASUREMENT KERNELS for bench.py -- not part of the product library, not in include/gnnx.h.
// Two in-run ceilings of the memory system, so that bench.py's roofline line is a fraction of something measured on the same box
// in the same process (MI355X_MICROARCH.md gives 6.29 TB/s for the copy and 7.4-8.6 TB/s for gathers of whole rows from a table
// resident in the 256 MiB Infinity Cache):
//   ceil_copy_f4      float4 copy: bytes read + bytes written per second = what HBM sustains for streams;
//   ceil_gather_rows  every output row is the sum of `deg` whole 1-KiB rows of a table picked by an index array (uniformly random
//                     rows): the access pattern of the aggregation (graph.cpp:208 as a CSR gather) with every structural
//                     difficulty removed -- constant degree, no row pointers, no epilogue, table as small as the caller makes it.
// `variant` selects the kernel shape (swept in round 3; bench.py uses the fastest, variant 0).
// Plain C ABI, device pointers, caller's stream; compiled for gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#define CEIL_API extern "C" __attribute__((visibility("default")))
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- copy -------------------------------------------------------------------------------------------------------------------
// A workgroup owns a contiguous tile of U * 4 KiB: every thread puts U loads in flight (consecutive threads = consecutive 16-byte
// pieces: each wave-instruction is 1 KiB), then stores them.
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tile_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    f32x4 v[U];
    if (base + 256 * (U - 1) < n) {
#pragma unroll
        for (int k = 0; k < U; k++) {
            if constexpr (NT) v[k] = __builtin_nontemporal_load(src + base + 256 * k);
            else v[k] = src[base + 256 * k];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            if constexpr (NT) __builtin_nontemporal_store(v[k], dst + base + 256 * k);
            else dst[base + 256 * k] = v[k];
        }
    } else {
        for (int k = 0; k < U; k++)
            if (base + 256 * k < n) dst[base + 256 * k] = src[base + 256 * k];
    }
}

__global__ __launch_bounds__(256) void copy_stride_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// ---- gather, register-staged ---------------------------------------------------------------------------------------------------
// One wavefront per block of RPW output rows; lane l owns floats 4l..4l+3 of a row (one 1-KiB row per wave-instruction).  All the
// block's indices are fetched up front (RPW * B of them, B per output row), B rows in flight per batch, two batches (the next output
// row's gathers are issued before the current row's are added).
template <int B, int RPW>
__global__ __launch_bounds__(64) void gather_rows_kernel(const float *__restrict__ table, const int32_t *__restrict__ idx, int64_t n_out,
                                                          float *__restrict__ out)
{
    constexpr int NI = RPW * B / 64;  // index registers per lane
    static_assert(RPW * B % 64 == 0, "whole index registers");
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * RPW;
    if (r0 + RPW > n_out) return;  // the caller sizes n_out as a multiple of RPW
    const float *tl = table + lane * 4;
    int32_t ix[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) ix[k] = idx[r0 * B + k * 64 + lane];
    float4 buf[2][B];
#pragma unroll
    for (int r = 0; r <= RPW; r++) {
        if (r < RPW) {
#pragma unroll
            for (int u = 0; u < B; u++) {
                const int e = r * B + u;
                const int32_t c = __builtin_amdgcn_readlane(ix[e / 64], e % 64);
                buf[r & 1][u] = *reinterpret_cast<const float4 *>(tl + (int64_t)c * 256);
            }
        }
        if (r > 0) {
            float4 s = buf[(r - 1) & 1][0];
#pragma unroll
            for (int u = 1; u < B; u++) {
                s.x += buf[(r - 1) & 1][u].x;
                s.y += buf[(r - 1) & 1][u].y;
                s.z += buf[(r - 1) & 1][u].z;
                s.w += buf[(r - 1) & 1][u].w;
            }
            *reinterpret_cast<float4 *>(out + (r0 + r - 1) * 256 + lane * 4) = s;
        }
    }
}

// ---- gather, LDS-DMA ring ----------------------------------------------------------------------------------------------------
// One wavefront streams RPW output rows; the table rows land in a ring of NS slots of B rows (1 KiB each) by LDS-DMA
// (global_load_lds_dwordx4: one row per wave-instruction), LA slots ahead of the adds; nothing in flight costs a register.
typedef __attribute__((address_space(3))) void lds_void_t;
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4 &dst, uint32_t addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int LA, int WPB>
__global__ __launch_bounds__(64 * WPB) void gather_rows_dma_kernel(const float *__restrict__ table, const int32_t *__restrict__ idx, int64_t n_out,
                                                                  int rows_per_wave, float *__restrict__ out)
{
    constexpr int B = 8, NS = LA + 1;
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float *ring = lds_all + wv * (NS * B * 256);
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * WPB + wv) * rows_per_wave;
    if (r0 + rows_per_wave > n_out) return;
    const float *tl = table + lane * 4;
    const uint32_t ring_lane = (uint32_t)(uintptr_t)(lds_void_t *)ring + (uint32_t)lane * 16u;
    // indices: one coalesced load per 8 output rows (64 table rows), the next group's fetched a group ahead
    int32_t ix_cur = idx[r0 * B + lane];
    int32_t ix_nxt = idx[(r0 + 8 < n_out ? r0 + 8 : r0) * B + lane];
    int slot_i = 0, slot_c = 0;
    const int n_rows = rows_per_wave;  // multiple of 8
    // prologue
    for (int p = 0; p < LA; p++) {
#pragma unroll
        for (int u = 0; u < B; u++) {
            const int32_t c = __shfl(p < 8 ? ix_cur : ix_nxt, (p % 8) * B + u, 64);
            __builtin_amdgcn_global_load_lds(tl + (int64_t)c * 256, (lds_void_t *)(ring + (slot_i * B + u) * 256), 16, 0, 0);
        }
        slot_i = slot_i + 1 == NS ? 0 : slot_i + 1;
    }
    for (int r = 0; r < n_rows; r++) {
        const int ri = r + LA;  // row to issue
        if (ri < n_rows) {
            const int32_t src = ((ri / 8) == (r / 8)) ? ix_cur : ix_nxt;
#pragma unroll
            for (int u = 0; u < B; u++) {
                const int32_t c = __shfl(src, (ri % 8) * B + u, 64);
                __builtin_amdgcn_global_load_lds(tl + (int64_t)c * 256, (lds_void_t *)(ring + (slot_i * B + u) * 256), 16, 0, 0);
            }
            slot_i = slot_i + 1 == NS ? 0 : slot_i + 1;
            wait_vm<LA * B>();
        } else {
            wait_vm<0>();
        }
        const uint32_t ad = ring_lane + (uint32_t)slot_c * (B * 1024);
        f32x4 v[B];
        lds_read_b128<0 * 1024>(v[0], ad);
        lds_read_b128<1 * 1024>(v[1], ad);
        lds_read_b128<2 * 1024>(v[2], ad);
        lds_read_b128<3 * 1024>(v[3], ad);
        lds_read_b128<4 * 1024>(v[4], ad);
        lds_read_b128<5 * 1024>(v[5], ad);
        lds_read_b128<6 * 1024>(v[6], ad);
        lds_read_b128<7 * 1024>(v[7], ad);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
        f32x4 s = v[0];
#pragma unroll
        for (int u = 1; u < B; u++) s += v[u];
        *reinterpret_cast<f32x4 *>(out + (r0 + r) * 256 + lane * 4) = s;
        slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
        if (r % 8 == 7) {  // next group of 8 output rows: rotate the index registers (LA <= 8: ix_nxt is the furthest needed)
            ix_cur = ix_nxt;
            const int64_t rn = r0 + r + 1 + 8;
            ix_nxt = idx[(rn < n_out ? rn : r0) * B + lane];
        }
    }
}

CEIL_API int ceil_copy_f4(const void *src, void *dst, size_t bytes, int variant, void *stream)
{
    if (!src || !dst || bytes % 16) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src);
    f32x4 *d = reinterpret_cast<f32x4 *>(dst);
    const size_t n = bytes / 16;
    auto blocks = [&](int u) { return dim3((uint32_t)((n + 256 * (size_t)u - 1) / (256 * (size_t)u))); };
    switch (variant) {  // measured (MI355X, 4 GiB): U = 2: 5.69 TB/s, 4: 5.60, 8: 5.35, 16: 5.17, nontemporal 5.13, grid-stride 4.72
    case 0: hipLaunchKernelGGL((copy_tile_kernel<2, false>), blocks(2), dim3(256), 0, st, s, d, n); break;
    case 1: hipLaunchKernelGGL((copy_tile_kernel<4, false>), blocks(4), dim3(256), 0, st, s, d, n); break;
    case 2: hipLaunchKernelGGL((copy_tile_kernel<1, false>), blocks(1), dim3(256), 0, st, s, d, n); break;
    case 3: hipLaunchKernelGGL((copy_tile_kernel<2, true>), blocks(2), dim3(256), 0, st, s, d, n); break;
    case 4: hipLaunchKernelGGL(copy_stride_kernel, dim3(256 * 16), dim3(256), 0, st, s, d, n); break;
    case 5: hipLaunchKernelGGL((copy_tile_kernel<8, false>), blocks(8), dim3(256), 0, st, s, d, n); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// table: [table_rows, 256] f32; idx: [n_out * 8] int32 in [0, table_rows); out: [n_out, 256] f32; n_out a multiple of 256.
CEIL_API int ceil_gather_rows(const float *table, const int32_t *idx, int64_t n_out, int variant, float *out, void *stream)
{
    if (!table || !idx || !out || n_out <= 0 || n_out % 256) return 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto dma = [&](auto la_tag, int rpw) -> int {  // LDS-DMA ring, LA output rows (8 table rows each) ahead, 4 wavefronts per workgroup
        constexpr int LA = decltype(la_tag)::value, WPB = 4;
        const size_t lds = sizeof(float) * (LA + 1) * 8 * 256 * WPB;
        static bool opted = false;
        if (!opted) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_rows_dma_kernel<LA, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) return 3;
            opted = true;
        }
        hipLaunchKernelGGL((gather_rows_dma_kernel<LA, WPB>), dim3((uint32_t)(n_out / (rpw * WPB))), dim3(64 * WPB), lds, st, table, idx, n_out, rpw, out);
        return 0;
    };
    // measured (MI355X, 160 MB table): LDS-DMA ring 8.03 TB/s (38 MB table: 8.58), register-staged 7.35 (7.87)
    switch (variant) {
    case 0: { int rc = dma(std::integral_constant<int, 4>{}, 64); if (rc) return rc; break; }   // 32 KB in flight per wavefront
    case 1: { int rc = dma(std::integral_constant<int, 2>{}, 64); if (rc) return rc; break; }   // 16 KB
    case 2: hipLaunchKernelGGL((gather_rows_kernel<8, 16>), dim3((uint32_t)(n_out / 16)), dim3(64), 0, st, table, idx, n_out, out); break;
    case 3: hipLaunchKernelGGL((gather_rows_kernel<8, 32>), dim3((uint32_t)(n_out / 32)), dim3(64), 0, st, table, idx, n_out, out); break;
    case 4: { int rc = dma(std::integral_constant<int, 6>{}, 64); if (rc) return rc; break; }   // 48 KB: 3 wavefronts per CU
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
