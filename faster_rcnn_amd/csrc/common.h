// Shared host-side helpers for libfrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/frcnn_hip.h"

namespace frcnn {

void set_error(const char* fmt, ...);

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FRCNN_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return FRCNN_OK;
}

struct AnchorTable {          // passed by value as a kernel argument
    int32_t h[FRCNN_MAX_ANCHORS];
    int32_t w[FRCNN_MAX_ANCHORS];
};

inline int load_anchor_table(const int32_t* hw_h, int A, AnchorTable* t) {
    if (!hw_h || A <= 0 || A > FRCNN_MAX_ANCHORS) return fail(FRCNN_E_ARG, "anchor table: A=%d out of range", A);
    for (int a = 0; a < A; ++a) { t->h[a] = hw_h[2 * a]; t->w[a] = hw_h[2 * a + 1]; }
    return FRCNN_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) takes effect on the CURRENT device only, and the C ABI promises plain-C hosts
// (tests/tools/c_abi_host.c) nothing about one process per GPU or one thread per process: every launcher that needs more than
// the default 64 KB of LDS keeps one bit per device ordinal (a `static bool` raised the limit on the first device
// a process used and launched without it on the second).
inline int raise_lds_once(std::atomic<uint64_t>& seen, const void* fn, size_t lds, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(FRCNN_E_HIP, "%s: no current HIP device", what);
    const uint64_t bit = 1ull << (dev & 63);
    if (seen.load(std::memory_order_acquire) & bit) return FRCNN_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(FRCNN_E_HIP, "%s: cannot raise dynamic LDS to %zu bytes on device %d", what, lds, dev);
    seen.fetch_or(bit, std::memory_order_release);
    return FRCNN_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace frcnn
