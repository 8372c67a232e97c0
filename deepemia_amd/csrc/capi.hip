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
extern "C" int demia_abi_version(void) { return 6; }

extern "C" const char* demia_build_arch(void) { return "gfx950"; }
#ifndef DEMIA_DEV
#define DEMIA_DEV 0
#endif
extern "C" const char* demia_build_flavor(void) { return DEMIA_DEV ? "dev" : "product"; }
