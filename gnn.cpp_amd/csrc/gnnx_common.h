// Internal helpers shared by the HIP translation units of libgnnx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

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

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Measurement switches (kernel variants, forced tile shapes, ablation flags, the LDS-staged SpMM) exist only in an EXPERIMENTS
// build (`make EXPERIMENTS=1`, scripts/exp_*.py); the shipped library has one code path per shape and reads no environment.
#ifdef GNNX_EXPERIMENTS
inline const char *experiment_env(const char *name) { return std::getenv(name); }
#else
inline const char *experiment_env(const char *) { return nullptr; }
#endif

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
