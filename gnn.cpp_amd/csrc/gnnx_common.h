// Internal helpers shared by the HIP translation units of libgnnx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "gnnx.h"

#define GNNX_API extern "C" __attribute__((visibility("default")))

namespace gnnx {

constexpr int kWave = 64;       // CDNA4 wavefront width
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

int set_error(int status, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

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
