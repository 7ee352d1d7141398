// Shared definitions for libams_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ams.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern thread_local int g_ams_last_hip_error;

static inline ams_status ams_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_ams_last_hip_error = (int)e;
        return AMS_E_LAUNCH_FAILED;
    }
    return AMS_OK;
}

#define AMS_REQUIRE(cond)            \
    do {                             \
        if (!(cond)) return AMS_E_INVALID_ARG; \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
