// What do v_permlane32_swap / v_permlane16_swap / row_shl DPP deliver to each lane?  (gfx950; hipcc tools/probes/lane_probe.hip -o /tmp/lp && /tmp/lp)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x + 100, z = 0;
    o[0 * 64 + threadIdx.x] = __builtin_amdgcn_permlane32_swap(l, z, false, false)[0];
    o[1 * 64 + threadIdx.x] = __builtin_amdgcn_permlane32_swap(l, z, false, false)[1];
    o[2 * 64 + threadIdx.x] = __builtin_amdgcn_permlane32_swap(z, l, false, false)[0];
    o[3 * 64 + threadIdx.x] = __builtin_amdgcn_permlane32_swap(z, l, false, false)[1];
    o[4 * 64 + threadIdx.x] = __builtin_amdgcn_permlane16_swap(l, z, false, false)[0];
    o[5 * 64 + threadIdx.x] = __builtin_amdgcn_permlane16_swap(l, z, false, false)[1];
    o[6 * 64 + threadIdx.x] = __builtin_amdgcn_permlane16_swap(z, l, false, false)[0];
    o[7 * 64 + threadIdx.x] = __builtin_amdgcn_permlane16_swap(z, l, false, false)[1];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 8 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[8 * 64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* n[8] = {"p32(l,0)[0]", "p32(l,0)[1]", "p32(0,l)[0]", "p32(0,l)[1]", "p16(l,0)[0]", "p16(l,0)[1]", "p16(0,l)[0]", "p16(0,l)[1]"};
    for (int r = 0; r < 8; ++r) { printf("%-12s", n[r]); for (int l = 0; l < 64; ++l) printf(" %3u", h[r * 64 + l]); printf("\n"); }
    return 0;
}
