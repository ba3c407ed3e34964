// csrc/common.h -- shared helpers for the gfx950 kernels behind include/eap_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eap_hip.h"

namespace eap {

void set_error(const char *msg);

static inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        set_error(buf);
        return (int)e;
    }
    return 0;
}

static inline int bad_arg(const char *what) {
    set_error(what);
    return (int)hipErrorInvalidValue;
}

static inline int hip_fail(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    set_error(buf);
    return (int)e;
}

static inline hipStream_t S(eap_stream_t s) { return (hipStream_t)s; }

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace eap
