// Shared helpers for the gfx950 kernels of libdeepemia_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/deepemia_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

extern "C" void demia_set_error(const char* fmt, ...);

#define DEMIA_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            demia_set_error("%s: %s", name, hipGetErrorString(e__));               \
            return DEMIA_ELAUNCH;                                                  \
        }                                                                          \
    } while (0)

#define DEMIA_REQUIRE(cond, msg)                                                   \
    do {                                                                           \
        if (!(cond)) {                                                             \
            demia_set_error("%s: requirement failed: %s", __func__, msg);          \
            return DEMIA_EINVAL;                                                   \
        }                                                                          \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of a 1-D grid: blocks that share an XCD (bid % 8) get a
// contiguous run of logical ids, so neighbouring tiles hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
