// Internal helpers shared by the HIP translation units of libgnnx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "gnnx.h"

#define GNNX_API extern "C" __attribute__((visibility("default")))

namespace gnnx {

constexpr int kWave = 64;       // CDNA4 wavefront width
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

int set_error(int status, const char *fmt, ...);

// Device copy of table[k] = the HOST libm's powf((float)k, -0.5f), k = 0 .. *len_out - 1 >= need_len - 1, on the current device:
// process-wide, built once under a mutex, immutable and never freed (gnnx_graph.hip) -- usable from any thread on any stream.
int libm_pow_m05_table(size_t need_len, const float **d_table_out, size_t *len_out);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Scope guard for a device temporary (hipMalloc): released on EVERY exit path of the enclosing scope (the GNNX_HIP_CHECK /
// GNNX_REQUIRE early returns included).  Temporaries are only made by build-time calls (CSR, plans, norm), which synchronise their
// stream before the guard runs; hipFree itself synchronises the device before the memory can be handed out again.
struct DeviceFreeSync {
    void *p = nullptr;
    ~DeviceFreeSync() { if (p) (void)hipFree(p); }
};

// Measurement switches (kernel variants, forced tile shapes, ablation flags, the LDS-staged SpMM) exist only in an EXPERIMENTS
// build (`make EXPERIMENTS=1`, scripts/exp_*.py); the shipped library has one code path per shape and reads no environment.
#ifdef GNNX_EXPERIMENTS
inline const char *experiment_env(const char *name) { return std::getenv(name); }
#else
inline const char *experiment_env(const char *) { return nullptr; }
#endif

}  // namespace gnnx

namespace gnnx {
// RN_f32(d / sd) -- the reference's `(x - mean) / (var + eps)->pow(0.5)` element (nn.cpp:301-316 -> functional.h Div: one IEEE
// division) -- from ONE f64 multiply by the precomputed RN_f64(1 / sd) instead of the ~10-instruction f32 division sequence: the
// prologue is bound by the vector ALU, and the division was most of it.  Correctly rounded, not approximately:
//   * with 24-bit operands the quotient d / sd is never half-way between two floats (that would need 2^25 | B for a 24-bit B), and
//     if it is not a float itself it lies at least 2^-49 (relative) away from every float and every half-way point:
//     |A / B - N 2^-25| = |A 2^25 - N B| / (B 2^25) >= 1 / (B 2^25) > 2^-49 for integers A, B < 2^24;
//   * (double)d * rsd carries two f64 roundings: relative error <= 2^-52 -- an eighth of that distance -- so rounding it to f32
//     (v_cvt_f32_f64, round to nearest even) lands on the float the exact quotient rounds to; the overflow threshold is a half-way
//     point of the same form;
//   * the argument needs a NORMAL quotient: a subnormal (or flushed) result takes the IEEE division itself (never seen on
//     normalised activations; a wave-uniform-in-practice branch).  NaN / inf / zero operands give what the division gives.
// tests/test_gpu_parity.py::test_division_by_column_constant_is_ieee_exact sweeps it against numpy's division.
__device__ __forceinline__ float div_by_const(float d, float sd, double rsd)
{
    float q = (float)((double)d * rsd);
    if (__builtin_expect(!(fabsf(q) >= 1.17549435e-38f) && d != 0.f, 0)) q = __fdiv_rn(d, sd);
    return q;
}

}  // namespace gnnx

#define GNNX_HIP_CHECK(expr)                                                                              \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return gnnx::set_error(GNNX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                   __FILE__, __LINE__);                                                   \
    } while (0)

#define GNNX_REQUIRE(cond, status, ...)                                  \
    do {                                                                 \
        if (!(cond)) return gnnx::set_error((status), __VA_ARGS__);      \
    } while (0)

#define GNNX_LAUNCH_CHECK() GNNX_HIP_CHECK(hipGetLastError())

namespace gnnx {
// Dynamic-LDS opt-in of a kernel, once per kernel AND device (bit d of the caller's mask; thread-safe), CHECKED: the request is held
// against what the device reports per workgroup (hipDeviceAttributeMaxSharedMemoryPerBlock) and, after the opt-in, against what
// the runtime reports for the function (maxDynamicSharedSizeBytes + its static sharedSizeBytes) -- a layout the hardware or the
// runtime would clamp is refused with GNNX_ERR_UNSUPPORTED instead of reading and writing LDS words that are not there.
template <class K>
inline int lds_opt_in(K kernel, size_t lds, std::atomic<uint64_t> &done, const char *what)
{
    int dev = 0;
    GNNX_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 64 || !(done.load(std::memory_order_acquire) & (1ull << dev))) {
        int max_lds = 0;
        GNNX_HIP_CHECK(hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        if ((size_t)max_lds < lds)
            return set_error(GNNX_ERR_UNSUPPORTED, "%s: %zu bytes of workgroup LDS requested, device %d offers %d", what, lds, dev, max_lds);
        GNNX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipFuncAttributes attr;
        GNNX_HIP_CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)));
        if ((size_t)attr.maxDynamicSharedSizeBytes < lds || attr.sharedSizeBytes + lds > (size_t)max_lds)
            return set_error(GNNX_ERR_UNSUPPORTED, "%s: %zu bytes of dynamic LDS requested; the runtime grants %d dynamic beside %zu static of %d",
                             what, lds, attr.maxDynamicSharedSizeBytes, (size_t)attr.sharedSizeBytes, max_lds);
        if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    }
    return GNNX_OK;
}
}  // namespace gnnx
