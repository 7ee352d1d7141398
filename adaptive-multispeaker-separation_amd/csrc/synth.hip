// Synthesis-side and waveform-loss kernels (HBM-bound streaming; deterministic, no atomics).
//
//  * overlap_add:   frames [R,T,W] -> out [R,L], out[r,l] = sum_t frames[r,t,l+pl-t*hop]   (gather form)
//                   second half of tf.nn.conv2d_transpose (reference models/adapt.py:241-243) and of
//                   tf.contrib.signal.inverse_stft's overlap_and_add (models/network.py:598-602); the first
//                   half (z . f2^T, or the irfft as a DFT product) is an MFMA GEMM in gemm.hip.
//  * pair_stats:    per utterance the S x S table of <t_s, a_s'> plus |a_s'|^2, |t_s|^2, <t_s, mix>, |mix|^2 --
//                   everything the SDR / L2 / permutation-invariant costs need (adapt.py:321-372, 404-431,
//                   network.py:196-221, 662-724) in ONE pass over the waveforms instead of a P-fold tiling.
//  * apply_masks:   separated = X_input * masks, transposed to (b,s) rows (network.py:577-581).
//  * overlap_metric: adapt.py:141-160.
#include "common.h"

namespace {

__global__ void overlap_add_kernel(const float* __restrict__ fr, float* __restrict__ out, int R, int T, int W, int L, int hop,
                                   int pl) {
    const long total = (long)R * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / L), l = (int)(i - (long)r * L);
        const int q = l + pl;                       // t*hop <= q < t*hop + W
        int t_hi = q / hop;
        if (t_hi > T - 1) t_hi = T - 1;
        int t_lo = (q - W + hop) / hop;             // ceil((q-W+1)/hop)
        if (q - W + 1 <= 0) t_lo = 0;
        float s = 0.f;
        for (int t = t_lo; t <= t_hi; ++t) s += fr[((long)r * T + t) * W + (q - t * hop)];
        out[i] = s;
    }
}

// stats layout per utterance b (floats), NS = 2 S^2 + 3 S + 1:
//   D[S*S] (D[s*S+s'] = <t_s, a_s'>) | Q[S*S] (|t_s - a_s'|^2) | Na[S] | Nt[S] | Tm[S] (<t_s, mix>) | Nm[1]
__global__ __launch_bounds__(256) void pair_stats_kernel(const float* __restrict__ tgt, const float* __restrict__ est,
                                                         const float* __restrict__ mix, float* __restrict__ part, int S, long L,
                                                         int nchunk, long chunk) {
    constexpr int MAXS = 4;
    constexpr int MAXNS = 2 * MAXS * MAXS + 3 * MAXS + 1;
    __shared__ float sm[4][MAXNS];
    const int b = blockIdx.y, c = blockIdx.x;
    const int SS = S * S, NS = 2 * SS + 3 * S + 1;
    float acc[MAXNS];
#pragma unroll
    for (int i = 0; i < MAXNS; ++i) acc[i] = 0.f;
    const long l0 = (long)c * chunk, l1 = min(L, l0 + chunk);
    for (long l = l0 + threadIdx.x; l < l1; l += 256) {
        float t[MAXS], a[MAXS];
        const float m = mix ? mix[(long)b * L + l] : 0.f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            t[s] = s < S ? tgt[((long)b * S + s) * L + l] : 0.f;
            a[s] = s < S ? est[((long)b * S + s) * L + l] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            if (s < S) {
#pragma unroll
                for (int s2 = 0; s2 < MAXS; ++s2) {
                    if (s2 < S) {
                        const float d = t[s] - a[s2];
                        acc[s * MAXS + s2] += t[s] * a[s2];
                        acc[MAXS * MAXS + s * MAXS + s2] += d * d;
                    }
                }
                acc[2 * MAXS * MAXS + s] += a[s] * a[s];
                acc[2 * MAXS * MAXS + MAXS + s] += t[s] * t[s];
                acc[2 * MAXS * MAXS + 2 * MAXS + s] += t[s] * m;
            }
        }
        acc[2 * MAXS * MAXS + 3 * MAXS] += m * m;
    }
    // compact [MAXS-strided] accumulators into the S-strided output layout
    for (int k = 0; k < NS; ++k) {
        int src;
        if (k < SS) src = (k / S) * MAXS + (k % S);
        else if (k < 2 * SS) src = MAXS * MAXS + ((k - SS) / S) * MAXS + ((k - SS) % S);
        else if (k < 2 * SS + S) src = 2 * MAXS * MAXS + (k - 2 * SS);
        else if (k < 2 * SS + 2 * S) src = 2 * MAXS * MAXS + MAXS + (k - 2 * SS - S);
        else if (k < 2 * SS + 3 * S) src = 2 * MAXS * MAXS + 2 * MAXS + (k - 2 * SS - 2 * S);
        else src = 2 * MAXS * MAXS + 3 * MAXS;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < MAXNS; ++i) v = (i == src) ? acc[i] : v;      // static indexing keeps acc in registers
        v = wave_sum(v);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS)
        part[((long)b * nchunk + c) * NS + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void pair_stats_final_kernel(const float* __restrict__ part, float* __restrict__ stats, int NS, int nchunk, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * NS) return;
    const int b = i / NS, k = i - b * NS;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[((long)b * nchunk + c) * NS + k];
    stats[i] = s;
}

// d est[b,s',l] = sum_s ( gD[b,s,s'] * t[b,s,l] - 2 gQ[b,s,s'] (t[b,s,l] - est[b,s',l]) ) + 2 gNa[b,s'] est[b,s',l]
__global__ void pair_stats_bwd_kernel(const float* __restrict__ tgt, const float* __restrict__ est, const float* __restrict__ g,
                                      float* __restrict__ dest, int B, int S, long L) {
    const int SS = S * S, NS = 2 * SS + 3 * S + 1;
    const long total = (long)B * S * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bs = i / L, l = i - bs * L;
        const int b = (int)(bs / S), s2 = (int)(bs - (long)b * S);
        const float* gb = g + (long)b * NS;
        const float a = est[i];
        float v = 2.0f * gb[2 * SS + s2] * a;
        for (int s = 0; s < S; ++s) {
            const float t = tgt[((long)b * S + s) * L + l];
            v += gb[s * S + s2] * t - 2.0f * gb[SS + s * S + s2] * (t - a);
        }
        dest[i] = v;
    }
}

// X_input [B,TF], masks [B,TF,S] -> sep rows (b,s): [B*S, TF]
__global__ void apply_masks_kernel(const float* __restrict__ X, const float* __restrict__ masks, float* __restrict__ sep, int B,
                                   int S, long TF) {
    const long total = (long)B * S * TF;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bs = i / TF, p = i - bs * TF;
        const long b = bs / S;
        const int s = (int)(bs - b * S);
        sep[i] = X[b * TF + p] * masks[(b * TF + p) * S + s];
    }
}
// dmasks[b,p,s] = dsep[(b,s),p] * X[b,p]
__global__ void apply_masks_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dsep, float* __restrict__ dmasks,
                                       int B, int S, long TF) {
    const long total = (long)B * TF * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bp = i / S;
        const int s = (int)(i - bp * S);
        const long b = bp / TF, p = bp - b * TF;
        dmasks[i] = dsep[(b * S + s) * TF + p] * X[bp];
    }
}

// overlap = mean_b mean_pairs mean_bins ( 1 - | |a|-|b| | / (max(|a|,|b|) + 1e-8) )   on the non-mix rows of the front output
__global__ __launch_bounds__(256) void overlap_partial_kernel(const float* __restrict__ y, float* __restrict__ part, int B, int S,
                                                              long TN) {
    __shared__ float sm[4];
    const long total = (long)B * TN;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / TN, p = i - b * TN;
        const float* base = y + ((long)B + b * S) * TN + p;      // rows B.. are (b,s)
        for (int s = 0; s < S; ++s)
            for (int s2 = s + 1; s2 < S; ++s2) {
                const float a = fabsf(base[(long)s * TN]), c = fabsf(base[(long)s2 * TN]);
                acc += 1.0f - fabsf(a - c) / (fmaxf(a, c) + 1e-8f);
            }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void overlap_final_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (sm[0] + sm[1] + sm[2] + sm[3]) * scale;
}
// dy for all rows (mixture rows get 0).  upstream = device scalar d loss/d overlap.
__global__ void overlap_bwd_kernel(const float* __restrict__ y, const float* __restrict__ upstream, float* __restrict__ dy, int B,
                                   int S, long TN, float scale) {
    const long total = (long)B * (S + 1) * TN;
    const float up = upstream[0] * scale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / TN, p = i - row * TN;
        if (row < B) { dy[i] = 0.f; continue; }
        const long b = (row - B) / S;
        const int s = (int)((row - B) - b * S);
        const float ys = y[i], a = fabsf(ys);
        const float* base = y + ((long)B + b * S) * TN + p;
        float g = 0.f;
        for (int s2 = 0; s2 < S; ++s2) {
            if (s2 == s) continue;
            const float c = fabsf(base[(long)s2 * TN]);
            // m = 1 - |a-c| / (max(a,c)+eps);  d m / d a
            const float mx = fmaxf(a, c) + 1e-8f, diff = a - c;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            float d = -sgn / mx;
            if (a >= c) d += fabsf(diff) / (mx * mx);            // max picks a (tf.reduce_max: first on ties -> split evenly is TF's rule; a==c gives diff 0 anyway)
            g += d;
        }
        const float sa = ys > 0.f ? 1.f : (ys < 0.f ? -1.f : 0.f);
        dy[i] = up * g * sa;
    }
}

}  // namespace

extern "C" {

ams_status ams_overlap_add(const float* frames, float* out, int R, int T, int W, int L, int hop, int pad_left, void* stream) {
    AMS_REQUIRE(frames && out && R > 0 && T > 0 && W > 0 && L > 0 && hop > 0 && pad_left >= 0);
    long n = (long)R * L;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(overlap_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, frames, out, R, T, W, L, hop, pad_left);
    return ams_check_launch();
}

size_t ams_pair_stats_workspace_bytes(int B, int S, long L) {
    const int NS = 2 * S * S + 3 * S + 1;
    const int nchunk = ceil_div(L, 4096);
    return (size_t)B * nchunk * NS * sizeof(float);
}

ams_status ams_pair_stats_fwd(const float* target, const float* est, const float* mix, float* stats, int B, int S, long L, void* ws,
                              size_t ws_bytes, void* stream) {
    AMS_REQUIRE(target && est && stats && ws && B > 0 && S > 0 && S <= 4 && L > 0);
    const int NS = 2 * S * S + 3 * S + 1;
    const int nchunk = ceil_div(L, 4096);
    if (ws_bytes < (size_t)B * nchunk * NS * sizeof(float)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pair_stats_kernel, dim3(nchunk, B), dim3(256), 0, st, target, est, mix, (float*)ws, S, L, nchunk, (long)4096);
    hipLaunchKernelGGL(pair_stats_final_kernel, dim3(ceil_div(B * NS, 256)), dim3(256), 0, st, (const float*)ws, stats, NS, nchunk, B);
    return ams_check_launch();
}

ams_status ams_pair_stats_bwd(const float* target, const float* est, const float* gstats, float* dest, int B, int S, long L,
                              void* stream) {
    AMS_REQUIRE(target && est && gstats && dest && B > 0 && S > 0 && S <= 4 && L > 0);
    long n = (long)B * S * L;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pair_stats_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, target, est, gstats, dest, B, S, L);
    return ams_check_launch();
}

ams_status ams_apply_masks_fwd(const float* X, const float* masks, float* sep, int B, int S, long TF, void* stream) {
    AMS_REQUIRE(X && masks && sep && B > 0 && S > 0 && TF > 0);
    long n = (long)B * S * TF;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(apply_masks_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, masks, sep, B, S, TF);
    return ams_check_launch();
}

ams_status ams_apply_masks_bwd(const float* X, const float* dsep, float* dmasks, int B, int S, long TF, void* stream) {
    AMS_REQUIRE(X && dsep && dmasks && B > 0 && S > 0 && TF > 0);
    long n = (long)B * S * TF;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(apply_masks_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, X, dsep, dmasks, B, S, TF);
    return ams_check_launch();
}

// y [B(1+S), TN] front output; out[0] = overlap metric.  ws >= 1024 floats.
ams_status ams_overlap_metric_fwd(const float* y, float* out, int B, int S, long TN, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(y && out && ws && B > 0 && S > 1 && TN > 0 && ws_bytes >= 1024 * sizeof(float));
    long n = (long)B * TN;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    const int npairs = S * (S - 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(overlap_partial_kernel, dim3(blocks), dim3(256), 0, st, y, (float*)ws, B, S, TN);
    hipLaunchKernelGGL(overlap_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, blocks, 1.0f / ((float)B * npairs * TN));
    return ams_check_launch();
}

ams_status ams_overlap_metric_bwd(const float* y, const float* upstream, float* dy, int B, int S, long TN, void* stream) {
    AMS_REQUIRE(y && upstream && dy && B > 0 && S > 1 && TN > 0);
    long n = (long)B * (S + 1) * TN;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    const int npairs = S * (S - 1) / 2;
    hipLaunchKernelGGL(overlap_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, upstream, dy, B, S, TN,
                       1.0f / ((float)B * npairs * TN));
    return ams_check_launch();
}

}  // extern "C"
