// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 element e at byte 2e (value = e).  Every lane supplies its own
// 8-byte-aligned address; the result (4 x u16 per lane) is printed so that the lane <-> (source lane, element) permutation can be read off.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(const int* __restrict__ addr, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)lds;          // LDS byte address (low 32 bits of the generic pointer's offset)
    unsigned a = base + (unsigned)addr[threadIdx.x];
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (uint16_t)(r.x & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint16_t)(r.x >> 16);
    out[threadIdx.x * 4 + 2] = (uint16_t)(r.y & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint16_t)(r.y >> 16);
}

int main() {
    int h_addr[64];
    uint16_t h_out[256];
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int variant = 0; variant < 3; ++variant) {
        for (int l = 0; l < 64; ++l) {
            if (variant == 0) h_addr[l] = l * 8;                                  // lane-linear: lane l -> elements 4l .. 4l+3
            else if (variant == 1) h_addr[l] = (l & 15) * 8 + (l >> 4) * 1024;    // 16-lane groups 1 KB apart
            else h_addr[l] = ((l & 3) * 8) + ((l >> 2) & 3) * 256 + (l >> 4) * 2048;   // rows of 4 lanes, row stride 256 B
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("variant %d (element index = byte address / 2)\n", variant);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d addr %5d -> %5u %5u %5u %5u\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
