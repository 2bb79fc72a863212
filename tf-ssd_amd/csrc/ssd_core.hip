// Library-level entry points: version, thread-local error text, device init.
#include "common.h"

#include <cstring>

namespace ssd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ssd

extern "C" {

const char* ssd_version(void) { return "ssd_hip 0.3 (gfx950)"; }
#include "build/build_id.h"
const char* ssd_build_id(void) { return SSD_BUILD_ID; }

const char* ssd_last_error(void) { return ssd::g_err; }

void* ssd_stream_create(int high_priority) {
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;     // (least, greatest): numerically lower = higher priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : 0);      // 0 = the default priority
    if (e != hipSuccess) {
        ssd::set_error("ssd_stream_create: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return nullptr;
    }
    return (void*)s;
}
int ssd_stream_destroy(void* stream) {
    if (stream) SSD_HIP(hipStreamDestroy((hipStream_t)stream));
    return SSD_OK;
}

int ssd_init(int device) {
    int n = 0;
    SSD_HIP(hipGetDeviceCount(&n));
    SSD_CHECK_ARG(device >= 0 && device < n, "ssd_init: device %d out of range (%d visible)", device, n);
    SSD_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    SSD_HIP(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        ssd::set_error("ssd_init: device %d is %s; this library is built for gfx950 only", device,
                       p.gcnArchName);
        return SSD_E_UNSUPPORTED;
    }
    return SSD_OK;
}

}  // extern "C"
