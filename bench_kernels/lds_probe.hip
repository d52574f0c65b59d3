// What does the runtime grant, and what does the hardware address, when a workgroup asks for (nearly) all of a CU's LDS?
// Round 4 found that a 156 KB layout of spmm_hubpc_kernel "read wrong words" from the LDS-DMA targets that sat highest, and capped
// its layouts at 152 000 bytes without an explanation.  This probe establishes the facts (one line of JSON per requested size):
//   * what hipDeviceAttributeMaxSharedMemoryPerBlock / hipFuncSetAttribute / hipFuncGetAttributes report,
//   * whether the launch is accepted,
//   * the lowest LDS byte address at which (a) plain ds_write / ds_read, (b) LDS-DMA with 4 bytes per lane (the index rings),
//     (c) LDS-DMA with 16 bytes per lane (the slice rings) stop returning what was written.
// Usage: lds_probe            (run by scripts through gpurun; not part of the library or of the tests)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));              \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

// res[0] = lowest bad byte address of the plain test (or -1), res[1] = of the 4-byte DMA, res[2] = of the 16-byte DMA,
// res[3..5] = number of bad words of each
__global__ __launch_bounds__(256) void probe_kernel(const uint32_t *pattern, int32_t lds_bytes, int32_t *res)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t words = lds_bytes / 4;
    // (a) plain stores and loads
    for (int32_t w = tid; w < words; w += 256) lds[w] = 0x5a000000u ^ (uint32_t)w;
    __syncthreads();
    for (int32_t w = tid; w < words; w += 256)
        if (lds[w] != (0x5a000000u ^ (uint32_t)w)) {
            atomicMin(&res[0], w * 4);
            atomicAdd(&res[3], 1);
        }
    __syncthreads();
    // (b) LDS-DMA, one dword per lane: 256 bytes per wave-instruction, every 256-byte piece of [0, lds_bytes)
    for (int32_t w = tid; w < words; w += 256) lds[w] = 0;
    __syncthreads();
    const int32_t pieces4 = lds_bytes / 256;
    for (int32_t p = wave; p < pieces4; p += 4)
        __builtin_amdgcn_global_load_lds(pattern + p * 64 + lane, (lds_void_t *)(lds + p * 64), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int32_t w = tid; w < pieces4 * 64; w += 256)
        if (lds[w] != pattern[w]) {
            atomicMin(&res[1], w * 4);
            atomicAdd(&res[4], 1);
        }
    __syncthreads();
    // (c) LDS-DMA, 16 bytes per lane: 1 KiB per wave-instruction
    for (int32_t w = tid; w < words; w += 256) lds[w] = 0;
    __syncthreads();
    const int32_t pieces16 = lds_bytes / 1024;
    for (int32_t p = wave; p < pieces16; p += 4)
        __builtin_amdgcn_global_load_lds(pattern + p * 256 + lane * 4, (lds_void_t *)(lds + p * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int32_t w = tid; w < pieces16 * 256; w += 256)
        if (lds[w] != pattern[w]) {
            atomicMin(&res[2], w * 4);
            atomicAdd(&res[5], 1);
        }
}

int main()
{
    int dev = 0, max_lds = 0, max_lds_cu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    CK(hipDeviceGetAttribute(&max_lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("{\"device\": \"%s\", \"maxSharedMemoryPerBlock\": %d, \"maxSharedMemoryPerMultiprocessor\": %d, \"prop.sharedMemPerBlock\": %zu}\n",
           prop.gcnArchName, max_lds, max_lds_cu, (size_t)prop.sharedMemPerBlock);
    const int32_t max_words = 160 * 1024 / 4;
    std::vector<uint32_t> h((size_t)max_words);
    for (int32_t i = 0; i < max_words; i++) h[(size_t)i] = 0xc0de0000u + (uint32_t)i * 2654435761u;
    uint32_t *d_pat = nullptr;
    int32_t *d_res = nullptr;
    CK(hipMalloc((void **)&d_pat, sizeof(uint32_t) * (size_t)max_words));
    CK(hipMalloc((void **)&d_res, 6 * sizeof(int32_t)));
    CK(hipMemcpy(d_pat, h.data(), sizeof(uint32_t) * (size_t)max_words, hipMemcpyHostToDevice));
    const int32_t sizes[] = {64 * 1024, 128 * 1024, 148 * 1024, 151584, 152576, 154 * 1024, 156 * 1024, 158 * 1024, 160 * 1024, 160 * 1024 + 1024};
    for (int32_t bytes : sizes) {
        const hipError_t es = hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        hipFuncAttributes attr;
        const hipError_t eg = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&probe_kernel));
        int32_t init[6] = {1 << 30, 1 << 30, 1 << 30, 0, 0, 0};
        CK(hipMemcpy(d_res, init, sizeof(init), hipMemcpyHostToDevice));
        (void)hipGetLastError();
        hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(256), (size_t)bytes, 0, d_pat, bytes, d_res);
        const hipError_t el = hipGetLastError();
        const hipError_t ey = hipDeviceSynchronize();
        int32_t r[6] = {-2, -2, -2, -2, -2, -2};
        if (el == hipSuccess && ey == hipSuccess) CK(hipMemcpy(r, d_res, sizeof(r), hipMemcpyDeviceToHost));
        auto lo = [](int32_t v) { return v == (1 << 30) ? -1 : v; };
        printf("{\"request\": %d, \"set_attribute\": \"%s\", \"get_attributes\": \"%s\", \"maxDynamicSharedSizeBytes\": %d, \"sharedSizeBytes\": %zu, "
               "\"launch\": \"%s\", \"sync\": \"%s\", \"first_bad_plain\": %d, \"first_bad_dma4\": %d, \"first_bad_dma16\": %d, "
               "\"bad_plain\": %d, \"bad_dma4\": %d, \"bad_dma16\": %d}\n",
               bytes, hipGetErrorName(es), hipGetErrorName(eg), eg == hipSuccess ? attr.maxDynamicSharedSizeBytes : -1,
               eg == hipSuccess ? (size_t)attr.sharedSizeBytes : (size_t)0, hipGetErrorName(el), hipGetErrorName(ey), lo(r[0]), lo(r[1]), lo(r[2]),
               r[3], r[4], r[5]);
        if (ey != hipSuccess) break;   // a failed launch leaves the context unusable: stop here
    }
    return 0;
}
