// Shared host-side helpers for libfrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace frcnn
