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

// Register padding of the kernels ("the top of a wave's register allocation").  Twice now a kernel that
// keeps live values in the LAST vector register it declares (k_expand with 24, k_bfs_compact with 32) returned corrupted
// packed words on MI355X, differently on every run, while the byte-identical code with a few more registers declared is
// bit-exact.  Every kernel therefore names a register at least 8 above its own need as clobbered in an empty asm statement:
// no instruction, no occupancy (the pads stay inside the next occupancy step), and tests/test_abi_cpu.py compiles the
// library with and without the pads (-DACX_NO_VGPR_PAD) to check the margin of every kernel.
#ifdef ACX_NO_VGPR_PAD
#define ACX_VGPR_PAD(reg) ((void)0)
#else
#define ACX_VGPR_PAD(reg) asm volatile("" ::: reg)
#endif
// by key width: W = uint64_t kernels and the wider unsigned __int128 ones
#define ACX_VGPR_PAD_W(W, reg64, reg128) \
    do {                                 \
        if (sizeof(W) == 8) {            \
            ACX_VGPR_PAD(reg64);         \
        } else {                         \
            ACX_VGPR_PAD(reg128);        \
        }                                \
    } while (0)

}  // namespace acx
