// What ds_read_b64_tr_b16 returns: LDS holds halves whose bit pattern is their own index; lane l supplies byte address a(l) (three address
// patterns); the four returned 16-bit values per lane are printed.  hipcc --offload-arch=gfx950 tools/probes/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                  // consecutive 8-byte chunks
    else if (mode == 1) addr = (l & 15) * 128 + (l >> 4) * 8;     // 16 rows of 128 bytes, chunk (l >> 4) of the row
    else addr = (l >> 2) * 128 + (l & 3) * 8;                     // row l >> 2 (128-byte pitch), chunk l & 3
    addr += (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (values are half indices = byte address / 2)\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
