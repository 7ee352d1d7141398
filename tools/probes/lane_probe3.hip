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
__global__ void k(const float* in, float* o) {
    const int l = threadIdx.x;
    float v = in[l];
    o[0 * 64 + l] = lane_above<32>(v);
    v = __fadd_rn(v, lane_above<32>(v)); o[1 * 64 + l] = v;
    o[2 * 64 + l] = lane_above<16>(v);
    v = __fadd_rn(v, lane_above<16>(v)); o[3 * 64 + l] = v;
    v = __fadd_rn(v, lane_above<8>(v)); o[4 * 64 + l] = v;
    v = __fadd_rn(v, lane_above<4>(v)); o[5 * 64 + l] = v;
    v = __fadd_rn(v, lane_above<2>(v)); o[6 * 64 + l] = v;
    v = __fadd_rn(v, lane_above<1>(v)); o[7 * 64 + l] = v;
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = (float)(i + 1);
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * 64 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    float r[8 * 64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    const char* n[8] = {"above32", "sum32", "above16", "sum16", "sum8", "sum4", "sum2", "sum1"};
    for (int q = 0; q < 8; ++q) { printf("%-8s", n[q]); for (int l = 0; l < 34; ++l) printf(" %4.0f", r[q * 64 + l]); printf("\n"); }
    printf("expected total %d\n", 64 * 65 / 2);
    return 0;
}
