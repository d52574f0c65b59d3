// Runtime plumbing of the C-ABI: status strings, memory, streams, events (thin HIP wrappers).
#include <cstring>

#include "gnnx_common.h"

namespace gnnx {
static thread_local char g_err[512] = "";

int set_error(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}
}  // namespace gnnx

using namespace gnnx;

GNNX_API int gnnx_version(void) { return GNNX_VERSION; }

GNNX_API const char *gnnx_status_string(int status)
{
    switch (status) {
    case GNNX_OK: return "ok";
    case GNNX_ERR_INVALID_ARG: return "invalid argument";
    case GNNX_ERR_SHAPE: return "operand shapes do not agree";
    case GNNX_ERR_INDEX_RANGE: return "edge index out of range";
    case GNNX_ERR_WORKSPACE: return "workspace too small";
    case GNNX_ERR_HIP: return "HIP runtime error";
    case GNNX_ERR_NO_DEVICE: return "no gfx950 device";
    case GNNX_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

GNNX_API const char *gnnx_last_error(void) { return g_err; }

GNNX_API int gnnx_device_count(int *count)
{
    GNNX_REQUIRE(count, GNNX_ERR_INVALID_ARG, "count is null");
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return set_error(GNNX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    return GNNX_OK;
}

GNNX_API int gnnx_set_device(int device)
{
    GNNX_HIP_CHECK(hipSetDevice(device));
    return GNNX_OK;
}

GNNX_API int gnnx_device_name(int device, char *buf, size_t buflen)
{
    GNNX_REQUIRE(buf && buflen > 0, GNNX_ERR_INVALID_ARG, "buf is null");
    hipDeviceProp_t prop;
    GNNX_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return GNNX_OK;
}

GNNX_API int gnnx_malloc(void **d_ptr, size_t bytes)
{
    GNNX_REQUIRE(d_ptr, GNNX_ERR_INVALID_ARG, "d_ptr is null");
    *d_ptr = nullptr;
    if (bytes == 0) return GNNX_OK;
    GNNX_HIP_CHECK(hipMalloc(d_ptr, bytes));
    return GNNX_OK;
}

GNNX_API int gnnx_free(void *d_ptr)
{
    if (d_ptr) GNNX_HIP_CHECK(hipFree(d_ptr));
    return GNNX_OK;
}

GNNX_API int gnnx_memset(void *d_ptr, int value, size_t bytes, void *stream)
{
    if (bytes == 0) return GNNX_OK;
    GNNX_REQUIRE(d_ptr, GNNX_ERR_INVALID_ARG, "d_ptr is null");
    GNNX_HIP_CHECK(hipMemsetAsync(d_ptr, value, bytes, as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    if (bytes == 0) return GNNX_OK;
    GNNX_REQUIRE(d_dst && h_src, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    GNNX_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));  // pageable source: make it safe to reuse
    return GNNX_OK;
}

GNNX_API int gnnx_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (bytes == 0) return GNNX_OK;
    GNNX_REQUIRE(h_dst && d_src, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    GNNX_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (bytes == 0) return GNNX_OK;
    GNNX_REQUIRE(d_dst && d_src, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_memcpy2d_d2d(void *d_dst, size_t dst_pitch_bytes, const void *d_src, size_t src_pitch_bytes, size_t width_bytes, size_t rows,
                               void *stream)
{
    if (width_bytes == 0 || rows == 0) return GNNX_OK;
    GNNX_REQUIRE(d_dst && d_src && dst_pitch_bytes >= width_bytes && src_pitch_bytes >= width_bytes, GNNX_ERR_INVALID_ARG, "null pointer or pitch < width");
    GNNX_HIP_CHECK(hipMemcpy2DAsync(d_dst, dst_pitch_bytes, d_src, src_pitch_bytes, width_bytes, rows, hipMemcpyDeviceToDevice, as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_stream_create(void **stream)
{
    GNNX_REQUIRE(stream, GNNX_ERR_INVALID_ARG, "stream is null");
    hipStream_t s;
    GNNX_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return GNNX_OK;
}

GNNX_API int gnnx_stream_destroy(void *stream)
{
    if (stream) GNNX_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_stream_sync(void *stream)
{
    GNNX_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_device_sync(void)
{
    GNNX_HIP_CHECK(hipDeviceSynchronize());
    return GNNX_OK;
}

GNNX_API int gnnx_event_create(void **event)
{
    GNNX_REQUIRE(event, GNNX_ERR_INVALID_ARG, "event is null");
    hipEvent_t e;
    GNNX_HIP_CHECK(hipEventCreate(&e));
    *event = e;
    return GNNX_OK;
}

GNNX_API int gnnx_event_destroy(void *event)
{
    if (event) GNNX_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return GNNX_OK;
}

GNNX_API int gnnx_event_record(void *event, void *stream)
{
    GNNX_REQUIRE(event, GNNX_ERR_INVALID_ARG, "event is null");
    GNNX_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return GNNX_OK;
}

GNNX_API int gnnx_event_sync(void *event)
{
    GNNX_REQUIRE(event, GNNX_ERR_INVALID_ARG, "event is null");
    GNNX_HIP_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return GNNX_OK;
}

GNNX_API int gnnx_event_elapsed_ms(void *start, void *stop, float *ms)
{
    GNNX_REQUIRE(start && stop && ms, GNNX_ERR_INVALID_ARG, "null pointer");
    GNNX_HIP_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return GNNX_OK;
}

GNNX_API int gnnx_stream_wait_event(void *stream, void *event)
{
    GNNX_REQUIRE(event, GNNX_ERR_INVALID_ARG, "event is null");
    GNNX_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return GNNX_OK;
}
