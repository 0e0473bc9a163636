// Host-side helpers shared by the C-ABI translation units: error string, HIP call checking,
// launch-size arithmetic, workspace carving.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/kouek_hip.h"

namespace kk {

void set_error(const char *fmt, ...);

#define KK_HIP_TRY(call)                                                                           \
    do {                                                                                           \
        hipError_t kk_e_ = (call);                                                                 \
        if (kk_e_ != hipSuccess) {                                                                 \
            kk::set_error("%s failed at %s:%d: %s", #call, __FILE__, __LINE__,                     \
                          hipGetErrorString(kk_e_));                                               \
            return KK_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

#define KK_LAUNCH_CHECK()                                                                          \
    do {                                                                                           \
        hipError_t kk_e_ = hipGetLastError();                                                      \
        if (kk_e_ != hipSuccess) {                                                                 \
            kk::set_error("kernel launch failed at %s:%d: %s", __FILE__, __LINE__,                 \
                          hipGetErrorString(kk_e_));                                               \
            return KK_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

#define KK_REQUIRE(cond, msg)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            kk::set_error("%s (%s) at %s:%d", msg, #cond, __FILE__, __LINE__);                     \
            return KK_ERR_INVALID;                                                                 \
        }                                                                                          \
    } while (0)

// Compute units of the current device (asked once per device); 256 when the query fails.
int device_cu_count();

inline uint32_t div_up(uint64_t a, uint64_t b) { return static_cast<uint32_t>((a + b - 1) / b); }
inline size_t align_up(size_t a, size_t b = 256) { return (a + b - 1) / b * b; }

// Carves consecutive 256-byte aligned pieces out of one allocation.
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes);
        return o;
    }
};

} // namespace kk
