// Shared host-side helpers for libssd_hip.so (gfx950 only; no portability shims).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/ssd_hip.h"

namespace ssd {

void set_error(const char* fmt, ...);

#define SSD_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ssd::set_error(__VA_ARGS__);    \
            return SSD_E_INVALID;           \
        }                                   \
    } while (0)

#define SSD_UNSUPPORTED_IF(cond, ...)       \
    do {                                    \
        if (cond) {                         \
            ssd::set_error(__VA_ARGS__);    \
            return SSD_E_UNSUPPORTED;       \
        }                                   \
    } while (0)

#define SSD_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) {                                                   \
            ssd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                           __FILE__, __LINE__);                                    \
            return SSD_E_HIP;                                                      \
        }                                                                          \
    } while (0)

#define SSD_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            ssd::set_error("kernel launch failed: %s (%s:%d)",                    \
                           hipGetErrorString(e__), __FILE__, __LINE__);           \
            return SSD_E_HIP;                                                     \
        }                                                                         \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace ssd
