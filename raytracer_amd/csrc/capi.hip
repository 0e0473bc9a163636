// Error string, build info and the small memory/stream helpers of the C ABI.
#include "kk_host.h"

#include <atomic>
#include <stdarg.h>
#include <string.h>

namespace kk {
static thread_local char g_error[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
int device_cu_count() {
    constexpr int MAX_DEVICES = 64;
    static std::atomic<int> cached[MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES)
        return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
} // namespace kk

extern "C" const char *kk_last_error(void) { return kk::g_error; }
extern "C" const char *kk_build_info(void) { return "gfx950;0.1"; }

extern "C" int kk_set_device(int device) {
    KK_HIP_TRY(hipSetDevice(device));
    return KK_OK;
}
extern "C" int kk_malloc(void **d_ptr, size_t bytes) {
    KK_REQUIRE(d_ptr != nullptr, "null pointer");
    KK_HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return KK_OK;
}
extern "C" int kk_free(void *d_ptr) {
    if (d_ptr)
        KK_HIP_TRY(hipFree(d_ptr));
    return KK_OK;
}
extern "C" int kk_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
    KK_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice,
                              static_cast<hipStream_t>(stream)));
    KK_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return KK_OK;
}
extern "C" int kk_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    KK_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost,
                              static_cast<hipStream_t>(stream)));
    KK_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return KK_OK;
}
extern "C" int kk_memset(void *d_dst, int value, size_t bytes, void *stream) {
    KK_HIP_TRY(hipMemsetAsync(d_dst, value, bytes, static_cast<hipStream_t>(stream)));
    return KK_OK;
}
extern "C" int kk_stream_synchronize(void *stream) {
    KK_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return KK_OK;
}
extern "C" int kk_stream_create(void **out_stream) {
    KK_REQUIRE(out_stream != nullptr, "null pointer");
    hipStream_t s = nullptr;
    KK_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out_stream = s;
    return KK_OK;
}
extern "C" int kk_stream_destroy(void *stream) {
    if (stream)
        KK_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return KK_OK;
}
