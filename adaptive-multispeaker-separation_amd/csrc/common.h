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

// csrc/gemm_ps.hip, used by csrc/gemm.hip (ams_front_maxpool_fwd): the stride-1 conv + max-pool partial on pre-split images
namespace ams_detail {
size_t conv_maxpool_ps_bytes(int Bt, int L, int W, int N);
bool conv_maxpool_ps_applies(int Bt, int L, int W, int N);
ams_status conv_maxpool_ps(const float* x, const float* f, float* pmax, int32_t* pidx, int Bt, int L, int W, int N, int pl,
                           const float* amax_x, const float* amax_f, void* img, hipStream_t st);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// value of lane (l + S) for the lanes l that have one (S = 32: lanes 0..31, S = 16: lanes 0..15 of each 32, S < 16: within 16-lane rows)
template <int S>
__device__ __forceinline__ float lane_above(float v) {
    const unsigned b = __builtin_bit_cast(unsigned, v);
    if constexpr (S == 32 || S == 16) {
        // "swap the upper half (S = 32) / the odd 16-lane rows (S = 16) of the first register with the lower half / even rows of the
        // second": afterwards lane l of `lo` holds what lane l + S of `hi` held.  Written as asm: with the builtin hipcc (ROCm 7.2)
        // picked the other result register in some contexts (tools/lane_probe*.hip).
        unsigned hi = b, lo = 0u;
        if constexpr (S == 32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hi), "+v"(lo));
        else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hi), "+v"(lo));
        return __builtin_bit_cast(float, lo);
    }
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, b, 0x100 | S, 0xf, 0xf, true));      // row_shl:S
}

// Sum of the 64 lanes, valid in LANE 0 ONLY, as the halving tree l += l + s (s = 32 .. 1) on the VALU: 6 instructions instead of the
// 6 ds_bpermute round trips of the butterfly above.  For per-block reductions of many values (one result lane is all they need).
__device__ __forceinline__ float wave_sum_lane0(float v) {
    v += lane_above<32>(v);
    v += lane_above<16>(v);
    v += lane_above<8>(v);
    v += lane_above<4>(v);
    v += lane_above<2>(v);
    v += lane_above<1>(v);
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
