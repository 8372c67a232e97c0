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

// A HIP stream whose kernels may only run on the compute units whose bit is set in `mask` (bit i of word i / 32; 256 CUs = 8
// words on MI355X).  The image loop gives its NETWORK stream such a stream with a few CUs per XCD left out: the short,
// latency-bound post-processing kernels of the images in flight then find free compute units while a forward's convolution
// grids -- one 128-KiB-LDS workgroup per CU -- hold all the others.
extern "C" int demia_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
    DEMIA_REQUIRE(mask && stream && words > 0, "args");
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) {
        demia_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
        return DEMIA_ELAUNCH;
    }
    *stream = (void*)s;
    return DEMIA_OK;
}
extern "C" int demia_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return DEMIA_ELAUNCH;
    return DEMIA_OK;
}
