// STFT / iSTFT glue around the MFMA DFT products (reference models/network.py:480-502, 584-607).
//
// tf.contrib.signal.stft(frame_length=W, frame_step=hop, fft_length=W) on [R,L] is computed as ONE fp32 MFMA GEMM
//   frames[R*T, W] (A_FRAMES loader, pad 0)  x  D[W, 2F],  D[n, f] = hann[n] cos(2 pi f n / W),  D[n, F+f] = -hann[n] sin(..)
// (the window is folded into the DFT matrix), which leaves [Re | Im] side by side.  The kernels here turn that
// into magnitude + unit phasor, and re-attach the mixture phase for the inverse transform, which is again a GEMM
// against the inverse-DFT matrix (inverse window folded in) followed by ams_overlap_add.
// W = 512: 2*W*2F = 1.05 MFLOP per frame -- far cheaper on the matrix cores than a radix kernel's launch/sync.
#include "common.h"

namespace {

// ri [rows, 2F] -> mag [rows, F], phasor [rows, 2F] = (cos theta | sin theta), theta = angle (angle(0) = 0)
__global__ void cplx_mag_phase_kernel(const float* __restrict__ ri, float* __restrict__ mag, float* __restrict__ ph, long rows,
                                      int F, long ld) {
    const long total = rows * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F;
        const int f = (int)(i - r * F);
        const float re = ri[r * ld + f], im = ri[r * ld + F + f];
        const float m = sqrtf(re * re + im * im);
        mag[i] = m;
        if (ph) {
            ph[r * 2 * F + f] = m > 0.f ? re / m : 1.f;
            ph[r * 2 * F + F + f] = m > 0.f ? im / m : 0.f;
        }
    }
}

// z [R*T, 2F] = (sep * cos | sep * sin) with the mixture's phasor tiled over the S speakers of each utterance:
// sep row index r = (b*S + s)*T + t uses phasor row b*T + t          (network.py:589-596)
__global__ void cplx_apply_kernel(const float* __restrict__ sep, const float* __restrict__ ph, float* __restrict__ z, long rows,
                                  int F, int S, int T) {
    const long total = rows * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F;
        const int f = (int)(i - r * F);
        const long bs = r / T, t = r - bs * T;
        const long pr = (bs / S) * T + t;
        const float v = sep[i];
        z[r * 2 * F + f] = v * ph[pr * 2 * F + f];
        z[r * 2 * F + F + f] = v * ph[pr * 2 * F + F + f];
    }
}
__global__ void cplx_apply_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ ph, float* __restrict__ dsep,
                                      long rows, int F, int S, int T) {
    const long total = rows * F;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / F;
        const int f = (int)(i - r * F);
        const long bs = r / T, t = r - bs * T;
        const long pr = (bs / S) * T + t;
        dsep[i] = dz[r * 2 * F + f] * ph[pr * 2 * F + f] + dz[r * 2 * F + F + f] * ph[pr * 2 * F + F + f];
    }
}

inline int blocks_for(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace

extern "C" {

// ld_ri: floats between two rows of ri (>= 2F; the DFT product may have padded its output rows to a multiple of 4 floats)
ams_status ams_cplx_mag_phase(const float* ri, float* mag, float* phasor, long rows, int F, long ld_ri, void* stream) {
    AMS_REQUIRE(ri && mag && rows > 0 && F > 0);
    hipLaunchKernelGGL(cplx_mag_phase_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, ri, mag, phasor, rows, F, ld_ri);
    return ams_check_launch();
}

ams_status ams_cplx_apply_fwd(const float* sep, const float* phasor, float* z, long rows, int F, int S, int T, void* stream) {
    AMS_REQUIRE(sep && phasor && z && rows > 0 && F > 0 && S > 0 && T > 0);
    hipLaunchKernelGGL(cplx_apply_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, sep, phasor, z, rows, F, S, T);
    return ams_check_launch();
}

ams_status ams_cplx_apply_bwd(const float* dz, const float* phasor, float* dsep, long rows, int F, int S, int T, void* stream) {
    AMS_REQUIRE(dz && phasor && dsep && rows > 0 && F > 0 && S > 0 && T > 0);
    hipLaunchKernelGGL(cplx_apply_bwd_kernel, dim3(blocks_for(rows * F)), dim3(256), 0, (hipStream_t)stream, dz, phasor, dsep, rows, F, S, T);
    return ams_check_launch();
}

}  // extern "C"
