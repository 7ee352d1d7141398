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


// ---- costs from the pair table: one workgroup, fixed-order reductions (the [B,S,S] glue of adapt.py:321-372, network.py:662-724) ----
// mode 0  PRETRAIN  out[0] = mean_b sum_s Q[b,s,s]                    out[1] = mean_{b,s} Nt Na / (D[b,s,s]^2 + 1e-12)
// mode 1  PIT_L2    out[0] = mean_b min_p red_s (Q[b,s,p(s)] * cl)     out[1] = 0          red_s = sum (cs = 1) or mean (cs = 1/S)
// mode 2  ADAPT     out[0] = mean_b min_p sum_s Q[b,s,p(s)] / L        out[1] = mean_i sum_s min_j Nt[i,s] Na[j,s] / (D2[s,i,j]^2 + 1e-12)
//                   (the cross-batch minimum is the reference's broadcast of [B,1,S,L] against [B,S,L], adapt.py:361-365)
// perms [P,S] int32 in lexicographic order; ties keep the first minimum (tf.reduce_min / torch.min sub-gradient convention).
constexpr int PC_PRETRAIN = 0, PC_ADAPT = 2;     // (1 = PIT_L2: the generic branch)

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    const float r = sm[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void pair_combine_fwd_kernel(const float* __restrict__ st, const float* __restrict__ D2,
                                                               const int* __restrict__ perms, float* __restrict__ out,
                                                               int* __restrict__ pbest, int* __restrict__ jbest, int B, int S, int P,
                                                               int mode, float cl, float cs) {
    __shared__ float sm[256];
    const int SS = S * S, NS = 2 * SS + 3 * S + 1, tid = threadIdx.x;
    float a0 = 0.f, a1 = 0.f;
    if (mode == PC_PRETRAIN) {
        for (int b = tid; b < B; b += 256) {
            const float* r = st + (long)b * NS;
            float q = 0.f;
            for (int s = 0; s < S; ++s) {
                q += r[SS + s * S + s];
                const float d = r[s * S + s];
                a1 += r[2 * SS + S + s] * r[2 * SS + s] / (d * d + 1e-12f);
            }
            a0 += q;
        }
        const float l2 = block_sum_256(a0, sm), sdr = block_sum_256(a1, sm);
        if (tid == 0) { out[0] = l2 / (float)B; out[1] = sdr / ((float)B * (float)S); }
        return;
    }
    for (int b = tid; b < B; b += 256) {
        const float* q = st + (long)b * NS + SS;
        float best = 0.f;
        int bp = 0;
        for (int p = 0; p < P; ++p) {
            float c = 0.f;
            for (int s = 0; s < S; ++s) c += q[s * S + perms[p * S + s]] * cl;
            c *= cs;
            if (p == 0 || c < best) { best = c; bp = p; }
        }
        pbest[b] = bp;
        a0 += best;
    }
    if (mode == PC_ADAPT) {
        for (int e = tid; e < B * S; e += 256) {
            const int i = e / S, s = e - i * S;
            const float nt = st[(long)i * NS + 2 * SS + S + s];
            const float* drow = D2 + ((long)s * B + i) * B;
            float best = 0.f;
            int bj = 0;
            for (int j = 0; j < B; ++j) {
                const float d = drow[j];
                const float v = nt * st[(long)j * NS + 2 * SS + s] / (d * d + 1e-12f);
                if (j == 0 || v < best) { best = v; bj = j; }
            }
            jbest[e] = bj;
            a1 += best;
        }
    }
    const float l2 = block_sum_256(a0, sm), sdr = block_sum_256(a1, sm);
    if (tid == 0) { out[0] = l2 / (float)B; out[1] = sdr / (float)B; }
}

// gstats [B,NS] and gD2 [S,B,B] are fully written (zeros where nothing flows)
__global__ __launch_bounds__(256) void pair_combine_bwd_kernel(const float* __restrict__ st, const float* __restrict__ D2,
                                                               const int* __restrict__ perms, const float* __restrict__ g,
                                                               const int* __restrict__ pbest, const int* __restrict__ jbest,
                                                               float* __restrict__ gst, float* __restrict__ gD2, int B, int S, int P,
                                                               int mode, float cl, float cs) {
    const int SS = S * S, NS = 2 * SS + 3 * S + 1, tid = threadIdx.x;
    const float g0 = g[0], g1 = g[1];
    for (long i = tid; i < (long)B * NS; i += 256) gst[i] = 0.f;
    if (mode == PC_ADAPT)
        for (long i = tid; i < (long)S * B * B; i += 256) gD2[i] = 0.f;
    __syncthreads();
    if (mode == PC_PRETRAIN) {
        for (int e = tid; e < B * S; e += 256) {
            const int b = e / S, s = e - b * S;
            const float* r = st + (long)b * NS;
            float* o = gst + (long)b * NS;
            const float d = r[s * S + s], nt = r[2 * SS + S + s], na = r[2 * SS + s];
            const float den = d * d + 1e-12f, w = g1 / ((float)B * (float)S);
            o[SS + s * S + s] = g0 / (float)B;
            o[2 * SS + s] = w * nt / den;                                   // d/d Na
            o[s * S + s] = -w * nt * na * 2.0f * d / (den * den);           // d/d D
        }
        return;
    }
    for (int e = tid; e < B * S; e += 256) {
        const int b = e / S, s = e - b * S;
        gst[(long)b * NS + SS + s * S + perms[pbest[b] * S + s]] = g0 * cl * cs / (float)B;
    }
    if (mode == PC_ADAPT) {
        // d/d Na[j,s]: every (i, s) whose minimum sits at j contributes -- gathered per (j, s) in i order (deterministic)
        for (int e = tid; e < B * S; e += 256) {
            const int j = e / S, s = e - j * S;
            float acc = 0.f;
            for (int i = 0; i < B; ++i) {
                if (jbest[i * S + s] != j) continue;
                const float d = D2[((long)s * B + i) * B + j];
                acc += st[(long)i * NS + 2 * SS + S + s] / (d * d + 1e-12f);
            }
            gst[(long)j * NS + 2 * SS + s] = acc * g1 / (float)B;
        }
        for (int e = tid; e < B * S; e += 256) {
            const int i = e / S, s = e - i * S, j = jbest[e];
            const float d = D2[((long)s * B + i) * B + j];
            const float den = d * d + 1e-12f;
            const float nt = st[(long)i * NS + 2 * SS + S + s], na = st[(long)j * NS + 2 * SS + s];
            gD2[((long)s * B + i) * B + j] = -(g1 / (float)B) * nt * na * 2.0f * d / (den * den);
        }
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

// Costs from the pair table in one launch (see pair_combine_fwd_kernel): mode 0 pre-training (l2, sdr), 1 PIT squared error
// (out[0]; cl multiplies every entry -- scale, or scale / L -- cs = 1 or 1/S), 2 Adapt.cost non-pretraining branch (l2 / L, the
// cross-batch SDR term; D2 [S,B,B], cl = 1/L).  perms [P,S] int32 (lexicographic); pbest [B] / jbest [B,S] int32 are written by
// the forward pass and read by the backward pass.  gD2 may be NULL unless mode == 2.
ams_status ams_pair_combine_fwd(const float* stats, const float* D2, const int* perms, float* out, int* pbest, int* jbest, int B, int S,
                                int P, int mode, float cl, float cs, void* stream) {
    AMS_REQUIRE(stats && out && B > 0 && S > 0 && S <= 4 && mode >= 0 && mode <= 2);
    AMS_REQUIRE(mode == PC_PRETRAIN || (perms && pbest && P > 0));
    AMS_REQUIRE(mode != PC_ADAPT || (D2 && jbest));
    hipLaunchKernelGGL(pair_combine_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, D2, perms, out, pbest, jbest, B, S, P,
                       mode, cl, cs);
    return ams_check_launch();
}

ams_status ams_pair_combine_bwd(const float* stats, const float* D2, const int* perms, const float* gout, const int* pbest,
                                const int* jbest, float* gstats, float* gD2, int B, int S, int P, int mode, float cl, float cs,
                                void* stream) {
    AMS_REQUIRE(stats && gout && gstats && B > 0 && S > 0 && S <= 4 && mode >= 0 && mode <= 2);
    AMS_REQUIRE(mode == PC_PRETRAIN || (perms && pbest && P > 0));
    AMS_REQUIRE(mode != PC_ADAPT || (D2 && jbest && gD2));
    hipLaunchKernelGGL(pair_combine_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, D2, perms, gout, pbest, jbest, gstats,
                       gD2, B, S, P, mode, cl, cs);
    return ams_check_launch();
}

}  // extern "C"
