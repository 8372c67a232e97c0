// Library-level entry points of libdeepemia_hip.so: version, arch, last-error text.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void demia_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* demia_last_error(void) { return g_err; }
extern "C" int demia_abi_version(void) { return 4; }

// flagged NON-parity mode (--precision f16): P32 producers write a ZERO low plane and the conv kernels issue the h x h
// MFMA only -- single-plane fp16 operands with f32 accumulation (what autocast gives the reference, inference.py:1390-1395)
int g_demia_single_plane = 0;
extern "C" int demia_p32_single_plane(int on) {
    const int prev = g_demia_single_plane;
    if (on >= 0) g_demia_single_plane = on ? 1 : 0;
    return prev;
}
extern "C" const char* demia_build_arch(void) { return "gfx950"; }
