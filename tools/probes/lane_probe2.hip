#include <hip/hip_runtime.h>
#include <cstdio>
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
    else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, b, 0x100 | S, 0xf, 0xf, true));
}
__global__ void k(const float* in, float* o, int NV) {
    const int l = threadIdx.x;
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = in[i * 64 + l];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = acc[i], w = acc[i];
        v = __fadd_rn(v, lane_above<32>(v));
        v = __fadd_rn(v, lane_above<16>(v));
        v = __fadd_rn(v, lane_above<8>(v));
        v = __fadd_rn(v, lane_above<4>(v));
        v = __fadd_rn(v, lane_above<2>(v));
        v = __fadd_rn(v, lane_above<1>(v));
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) w = __fadd_rn(w, __shfl_down(w, s, 64));
        if (l == 0) { o[2 * i] = v; o[2 * i + 1] = w; }
    }
}
int main() {
    float h[8 * 64]; for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f - 0.3f;
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8);
    float r[16]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("%d  dpp %.9g  shfl %.9g  %s\n", i, r[2 * i], r[2 * i + 1], r[2 * i] == r[2 * i + 1] ? "==" : "DIFF");
    return 0;
}
