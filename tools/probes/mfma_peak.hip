// Sustained fp32 MFMA ceiling of the box the bench runs on: v_mfma_f32_32x32x2_f32 only, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probes/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 1e-9f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    for (int wg_per_cu : {1, 2, 4}) {
        const int iters = 20000, grid = 256 * wg_per_cu;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 100, 1.0f, 1e-9f);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, 1e-9f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * (2.0 * 32 * 32 * 2);
        printf("%d workgroup(s) of 4 waves per CU: %.1f ms, %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms / 1e9);
    }
    return 0;
}
