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

// ---- index check for meshes that already live on the device ------------------------------------------
namespace kk {
__global__ __launch_bounds__(256) void max_index_kernel(const uint32_t *__restrict__ idx, size_t count,
                                                        uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < count; i += gridDim.x * 256ull)
        m = idx[i] > m ? idx[i] : m;
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0)
        atomicMax(out, m);
}
} // namespace kk

extern "C" int kk_check_indices(const uint32_t *d_indices, size_t count, uint32_t limit, uint32_t *d_word,
                                void *stream_) {
    KK_REQUIRE(d_indices != nullptr && d_word != nullptr, "null pointer");
    if (count == 0)
        return KK_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KK_HIP_TRY(hipMemsetAsync(d_word, 0, sizeof(uint32_t), stream));
    const size_t want = (count + 255) / 256;
    const uint32_t grid = static_cast<uint32_t>(want < 2048 ? want : 2048);
    kk::max_index_kernel<<<grid, 256, 0, stream>>>(d_indices, count, d_word);
    KK_LAUNCH_CHECK();
    uint32_t top = 0;
    KK_HIP_TRY(hipMemcpyAsync(&top, d_word, sizeof(top), hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    if (top >= limit) {
        kk::set_error("an index of %u where the array it points into has %u elements", top, limit);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}
