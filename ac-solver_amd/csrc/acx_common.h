// acx_common.h -- host-side helpers shared by the C-ABI translation units of libacx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/acx.h"

namespace acx {

// thread-local message behind acx_last_error()
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define ACX_HIP_TRY(expr)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) return ::acx::fail(ACX_E_NODEVICE, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// true when at least one HIP device is visible; sets the error message otherwise
bool have_device();

// grow-only per-thread device scratch for the host-buffer convenience entry points
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    ~Scratch();
};
Scratch& scratch(int slot);

template <typename T> static inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

}  // namespace acx
