// Which CU does bit b of a hipExtStreamCreateWithCUMask mask select, and does a mask survive hipGraph capture + replay?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe tools/probes/cu_mask_probe.hip && /tmp/cu_mask_probe
// Output: one line per mask bit (xcc, se, sh, cu as the wave itself reads them from HW_REG_XCC_ID / HW_REG_HW_ID), then the number
// of distinct CUs a 2048-workgroup launch touched (a) directly on a masked stream, (b) replayed from a graph captured on that stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void whoami(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    // keep the workgroup resident for a while so that a big launch spreads over every CU it is allowed on
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {}
}

static unsigned key(unsigned xcc, unsigned hw) { return (xcc << 16) | (hw & 0xff00); }     // xcc, se, sh, cu

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    unsigned* d;
    CK(hipMalloc(&d, 2 * 4096 * sizeof(unsigned)));
    std::vector<unsigned> h(2 * 4096);
    const int words = (ncu + 31) / 32;
    for (int b = 0; b < ncu; ++b) {
        std::vector<uint32_t> mask(words, 0u);
        mask[b / 32] = 1u << (b % 32);
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
        hipLaunchKernelGGL(whoami, dim3(1), dim3(64), 0, s, d, 0);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 8, hipMemcpyDeviceToHost));
        printf("bit %3d -> xcc %u se %u sh %u cu %2u\n", b, h[0], (h[1] >> 13) & 7, (h[1] >> 12) & 1, (h[1] >> 8) & 0xf);
        CK(hipStreamDestroy(s));
    }
    // a mask of the first 64 bits: direct launch vs graph replay
    std::vector<uint32_t> mask(words, 0u);
    mask[0] = 0xffffffffu; mask[1] = 0xffffffffu;
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
    auto distinct = [&](const char* what) {
        hipMemcpy(h.data(), d, 2 * 2048 * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::set<unsigned> cus;
        int per_xcc[16] = {0};
        for (int i = 0; i < 2048; ++i) { cus.insert(key(h[2 * i], h[2 * i + 1])); }
        for (unsigned k : cus) per_xcc[k >> 16]++;
        printf("%s: %zu distinct CUs; per xcc:", what, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n");
    };
    hipLaunchKernelGGL(whoami, dim3(2048), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    distinct("masked stream, direct launch (64 bits set)");
    hipLaunchKernelGGL(whoami, dim3(2048), dim3(256), 0, 0, d, 20000);
    CK(hipDeviceSynchronize());
    distinct("null stream (no mask)");
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(whoami, dim3(2048), dim3(256), 0, s, d, 20000);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemset(d, 0xff, 2 * 2048 * sizeof(unsigned)));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    distinct("graph captured on the masked stream, replayed on the masked stream");
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    CK(hipGraphLaunch(ge, plain));
    CK(hipStreamSynchronize(plain));
    distinct("same graph replayed on an unmasked stream");
    // block id -> xcc under the mask (the ring kernels rely on id % 8)
    hipLaunchKernelGGL(whoami, dim3(64), dim3(256), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    hipMemcpy(h.data(), d, 2 * 64 * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("block -> xcc under the mask:");
    for (int i = 0; i < 32; ++i) printf(" %u", h[2 * i]);
    printf("\n");
    return 0;
}
