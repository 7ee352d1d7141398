// Per-utterance input conditioning of the separator (reference models/network.py:409-454,504-521) and the magnitude /
// silence weightings of masks and k-means (network.py:381-396, Kmeans_2.py:76-80).  One workgroup per utterance row;
// every statistic is a fixed-order block reduction (deterministic).  These flags are off in the shipped launchers, the
// kernels exist so that turning them on does not put framework arithmetic on the path.
#include "common.h"

namespace {

enum { PRE_NONE = 0, PRE_ABS = 1, PRE_SQRT = 2, PRE_LOG10 = 3 };
enum { NORM_NONE = 0, NORM_01 = 1, NORM_MEANSTD = 2, NORM_SILENT = 3 };
enum { W_NONE = 0, W_LINEAR = 1, W_SQRT = 2, W_SQUARE = 3 };

__device__ __forceinline__ float pre_apply(float x, int pre) {
    switch (pre) {
        case PRE_ABS: return fabsf(x);
        case PRE_SQRT: return sqrtf(x);
        case PRE_LOG10: return log10f(x + 1e-12f);
        default: return x;
    }
}

struct Red { float mn, mx, sum; };

__device__ __forceinline__ float blk_reduce(float v, float* sm, int op) {       // op 0 sum, 1 min, 2 max; result to all threads
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_down(v, o);
        v = op == 0 ? v + t : (op == 1 ? fminf(v, t) : fmaxf(v, t));
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = op == 0 ? r + sm[i] : (op == 1 ? fminf(r, sm[i]) : fmaxf(r, sm[i]));
    return r;
}

// out = norm(pre(x)) per row.  NORM_01: (z - min)/(max - min); NORM_MEANSTD: (z - mean)/sqrt(population var);
// NORM_SILENT: z * [max - z < thr]  (network.py:440-443, thr = silence_mask_db / 20)
__global__ __launch_bounds__(256) void row_transform_kernel(const float* __restrict__ x, float* __restrict__ out, long n, int pre,
                                                            int norm, float thr) {
    __shared__ float sm[4];
    const float* xr = x + (long)blockIdx.x * n;
    float* orow = out + (long)blockIdx.x * n;
    float a = 0.f, b = 1.f, mxv = 0.f;
    if (norm == NORM_01 || norm == NORM_SILENT) {
        float mn = 3.0e38f, mx = -3.0e38f;
        for (long i = threadIdx.x; i < n; i += 256) { const float z = pre_apply(xr[i], pre); mn = fminf(mn, z); mx = fmaxf(mx, z); }
        mn = blk_reduce(mn, sm, 1);
        mx = blk_reduce(mx, sm, 2);
        a = mn; b = mx - mn; mxv = mx;
    } else if (norm == NORM_MEANSTD) {
        float s = 0.f;
        for (long i = threadIdx.x; i < n; i += 256) s += pre_apply(xr[i], pre);
        const float mean = blk_reduce(s, sm, 0) / (float)n;
        float v = 0.f;
        for (long i = threadIdx.x; i < n; i += 256) { const float d = pre_apply(xr[i], pre) - mean; v += d * d; }
        const float var = blk_reduce(v, sm, 0) / (float)n;
        a = mean; b = sqrtf(var);
    }
    for (long i = threadIdx.x; i < n; i += 256) {
        const float z = pre_apply(xr[i], pre);
        float r;
        if (norm == NORM_SILENT) r = (mxv - z < thr) ? z : 0.f;
        else if (norm == NORM_NONE) r = z;
        else r = (z - a) / b;
        orow[i] = r;
    }
}

// y[b,p,s] *= f(|X[b,p]| / max_p |X[b,:]|)  and/or  [log10(max/|X|) < sil_thr]   (network.py:381-396)
__global__ __launch_bounds__(256) void weight_masks_kernel(const float* __restrict__ X, float* __restrict__ y, long TF, int S, int mode,
                                                           int use_sil, float sil_thr) {
    __shared__ float sm[4];
    const float* xr = X + (long)blockIdx.x * TF;
    float* yr = y + (long)blockIdx.x * TF * S;
    float mx = 0.f;
    for (long i = threadIdx.x; i < TF; i += 256) mx = fmaxf(mx, fabsf(xr[i]));
    mx = blk_reduce(mx, sm, 2);
    for (long i = threadIdx.x; i < TF; i += 256) {
        const float ax = fabsf(xr[i]);
        const float r = ax / mx;
        float w = 1.f;
        if (mode == W_LINEAR) w = r;
        else if (mode == W_SQRT) w = sqrtf(r);
        else if (mode == W_SQUARE) w = r * r;
        if (use_sil) w *= (log10f(mx / ax) < sil_thr) ? 1.f : 0.f;
        for (int s = 0; s < S; ++s) yr[i * S + s] *= w;
    }
}

// w[b,l] = [log10(max_l lat[b,:] / lat[b,l]) < thr]   (Kmeans_2.py:76-80)
__global__ __launch_bounds__(256) void silence_weights_kernel(const float* __restrict__ lat, float* __restrict__ w, long n, float thr) {
    __shared__ float sm[4];
    const float* r = lat + (long)blockIdx.x * n;
    float mx = -3.0e38f;
    for (long i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, r[i]);
    mx = blk_reduce(mx, sm, 2);
    for (long i = threadIdx.x; i < n; i += 256) w[(long)blockIdx.x * n + i] = (log10f(mx / r[i]) < thr) ? 1.f : 0.f;
}

// Oracle separator of the pre-training objective (adapt.py:173-196).  y rows: B mixtures then (b,s) sources, TN values each.
//   mode 0 'mask'    : out[b,s] = mix * (nm[b,s] / mix)          (NaN where mix == 0, quirk C-4, kept)
//   mode 1 'perfect' : out[b,s] = mix - (sum_s' nm[b,s'] - nm[b,s])
__global__ __launch_bounds__(256) void pretrain_sep_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int S,
                                                               long TN, int mode) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < TN; p += (long)gridDim.x * 256) {
        const float mix = y[(long)b * TN + p];
        float tot = 0.f;
        if (mode == 1)
            for (int s = 0; s < S; ++s) tot += y[((long)B + (long)b * S + s) * TN + p];
        for (int s = 0; s < S; ++s) {
            const float nm = y[((long)B + (long)b * S + s) * TN + p];
            out[((long)b * S + s) * TN + p] = mode == 0 ? mix * (nm / mix) : mix - (tot - nm);
        }
    }
}
// dy (all rows, fully written).  mask: d/d nm = 1, d/d mix = 0 (the two autodiff terms cancel);  perfect: d/d mix = sum_s d,
// d/d nm[s] = -(sum_s' d - d[s]).
__global__ __launch_bounds__(256) void pretrain_sep_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dy, int B, int S,
                                                               long TN, int mode) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < TN; p += (long)gridDim.x * 256) {
        float tot = 0.f;
        for (int s = 0; s < S; ++s) tot += dout[((long)b * S + s) * TN + p];
        dy[(long)b * TN + p] = mode == 0 ? 0.f : tot;
        for (int s = 0; s < S; ++s) {
            const float d = dout[((long)b * S + s) * TN + p];
            dy[((long)B + (long)b * S + s) * TN + p] = mode == 0 ? d : -(tot - d);
        }
    }
}


// ---- default-on terms of the pre-training objective (reference models/adapt.py:127-132, 310-316, 377-384; CLI defaults
// utils/trainer.py:151-161: beta = 1e-2, regularization = 1e-4) ------------------------------------------------------------------
//
// p_hat[m] = sum_b |y[b, m]|  (adapt.py:130-131), rows split over grid.y, partials added in row-slab order (deterministic).
__global__ __launch_bounds__(256) void abs_colsum_part_kernel(const float* __restrict__ y, float* __restrict__ part, int Bt, long M,
                                                              int rows_per) {
    const long m = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (m >= M) return;
    const int r0 = blockIdx.y * rows_per, r1 = min(Bt, r0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m + 3 < M && (M & 3) == 0) {
        for (int r = r0; r < r1; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(y + (long)r * M + m);
            s.x += fabsf(v.x); s.y += fabsf(v.y); s.z += fabsf(v.z); s.w += fabsf(v.w);
        }
        *reinterpret_cast<float4*>(part + (long)blockIdx.y * M + m) = s;
    } else {
        for (int j = 0; j < 4 && m + j < M; ++j) {
            float t = 0.f;
            for (int r = r0; r < r1; ++r) t += fabsf(y[(long)r * M + m + j]);
            part[(long)blockIdx.y * M + m + j] = t;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, long M, int nslab) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += part[(long)k * M + m];
    out[m] = s;
}

// kl_div(p, p_hat) = logfunc(p, p_hat) + logfunc(1 - p, 1 - p_hat), logfunc(a, b) = a * log(clip(a) / clip(b)), clip to [1e-10, 1]
// (utils/ops.py:46-54).  tf.clip_by_value passes the gradient where 1e-10 <= b <= 1.
__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 1e-10f), 1.0f); }
__device__ __forceinline__ float kl_term(float p, float ph) {
    const float q = 1.0f - p, qh = 1.0f - ph;
    return p * logf(clip01(p) / clip01(ph)) + q * logf(clip01(q) / clip01(qh));
}
__device__ __forceinline__ float kl_dph(float p, float ph) {       // d kl_term / d p_hat
    const float q = 1.0f - p, qh = 1.0f - ph;
    float g = 0.f;
    if (ph >= 1e-10f && ph <= 1.0f) g -= p / ph;
    if (qh >= 1e-10f && qh <= 1.0f) g += q / qh;
    return g;
}
__global__ __launch_bounds__(256) void kl_part_kernel(const float* __restrict__ p_hat, float* __restrict__ part, long M, float p) {
    __shared__ float sm[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M; i += (long)gridDim.x * 256) s += kl_term(p, p_hat[i]);
    s = blk_reduce(s, sm, 0);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void small_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = blk_reduce(s, sm, 0);
    if (threadIdx.x == 0) out[0] = s * scale;
}
// dy[b, m] (+)= up * gscale * sign(y[b, m]) * d kl / d p_hat[m]     (tf.abs' = sign, 0 at 0)
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ y, const float* __restrict__ p_hat,
                                                     const float* __restrict__ up, float gscale, float* __restrict__ dy, int Bt, long M,
                                                     float p, int accumulate) {
    const float g0 = up[0] * gscale;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        const float g = g0 * kl_dph(p, p_hat[m]);
        for (int r = blockIdx.y; r < Bt; r += gridDim.y) {
            const float v = y[(long)r * M + m];
            const float d = v > 0.f ? g : (v < 0.f ? -g : 0.f);
            float* o = dy + (long)r * M + m;
            *o = accumulate ? *o + d : d;
        }
    }
}

// non-negativity term: mean_b sum_{t,n} min(y, 0)^2   (adapt.py:314-316)
__global__ __launch_bounds__(256) void negsq_part_kernel(const float* __restrict__ y, float* __restrict__ part, long n) {
    __shared__ float sm[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float v = fminf(y[i], 0.f); s += v * v; }
    s = blk_reduce(s, sm, 0);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// dx (+)= up * scale * f(x):  mode 0  f = 2 x  (sum of squares),  mode 1  f = 2 min(x, 0)  (negative energy)
__global__ __launch_bounds__(256) void sq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ up, float scale,
                                                     float* __restrict__ dx, long n, int mode, int accumulate) {
    const float g = 2.0f * up[0] * scale;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = mode ? fminf(x[i], 0.f) : x[i];
        dx[i] = accumulate ? dx[i] + g * v : g * v;
    }
}

}  // namespace

extern "C" {

// p_hat[M] = sum over the Bt rows of |y|  (adapt.py:130-131).  ws: ams_abs_colsum_workspace_bytes(Bt, M).
size_t ams_abs_colsum_workspace_bytes(int Bt, long M) {
    if (Bt <= 0 || M <= 0) return 0;
    int nslab = (Bt + 11) / 12;
    if (nslab > 32) nslab = 32;
    return (size_t)nslab * M * sizeof(float);
}
ams_status ams_abs_colsum_fwd(const float* y, float* p_hat, int Bt, long M, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(y && p_hat && ws && Bt > 0 && M > 0);
    if (ws_bytes < ams_abs_colsum_workspace_bytes(Bt, M)) return AMS_E_WORKSPACE_TOO_SMALL;
    int nslab = (Bt + 11) / 12;
    if (nslab > 32) nslab = 32;
    const int rows_per = (Bt + nslab - 1) / nslab;
    nslab = (Bt + rows_per - 1) / rows_per;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(abs_colsum_part_kernel, dim3((unsigned)((M + 1023) / 1024), nslab), dim3(256), 0, st, y, (float*)ws, Bt, M, rows_per);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, (const float*)ws, p_hat, M, nslab);
    return ams_check_launch();
}
// out[0] = sum_m kl_div(p, p_hat[m])  (utils/ops.py:46-54).  ws: 1024 floats.
ams_status ams_kl_sparsity_fwd(const float* p_hat, float* out, long M, float p, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(p_hat && out && ws && M > 0);
    if (ws_bytes < 1024 * sizeof(float)) return AMS_E_WORKSPACE_TOO_SMALL;
    int blocks = (int)((M + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(kl_part_kernel, dim3(blocks), dim3(256), 0, st, p_hat, (float*)ws, M, p);
    hipLaunchKernelGGL(small_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, blocks, 1.0f);
    return ams_check_launch();
}
// dy[Bt, M] (+)= upstream[0] * gscale * sign(y) * d kl / d p_hat   (the |.| column sum and the KL in one pass)
ams_status ams_kl_sparsity_bwd(const float* y, const float* p_hat, const float* upstream, float gscale, float* dy, int Bt, long M,
                               float p, int accumulate, void* stream) {
    AMS_REQUIRE(y && p_hat && upstream && dy && Bt > 0 && M > 0);
    int bx = (int)((M + 255) / 256);
    if (bx > 1024) bx = 1024;
    int by = Bt < 8 ? Bt : 8;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, y, p_hat, upstream, gscale, dy, Bt, M, p, accumulate);
    return ams_check_launch();
}
// out[0] = (1 / Bt) * sum min(y, 0)^2 over all Bt * M entries  (adapt.py:314-316).  ws: 1024 floats.
ams_status ams_negative_energy_fwd(const float* y, float* out, int Bt, long M, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(y && out && ws && Bt > 0 && M > 0);
    if (ws_bytes < 1024 * sizeof(float)) return AMS_E_WORKSPACE_TOO_SMALL;
    const long n = (long)Bt * M;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(negsq_part_kernel, dim3(blocks), dim3(256), 0, st, y, (float*)ws, n);
    hipLaunchKernelGGL(small_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, blocks, 1.0f / (float)Bt);
    return ams_check_launch();
}
// dx[n] (+)= upstream[0] * scale * 2 x  (mode 0: gradient of sum x^2)  /  * 2 min(x, 0)  (mode 1: of sum min(x, 0)^2)
ams_status ams_sumsq_bwd(const float* x, const float* upstream, float scale, float* dx, long n, int mode, int accumulate, void* stream) {
    AMS_REQUIRE(x && upstream && dx && n > 0 && (mode == 0 || mode == 1));
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sq_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, upstream, scale, dx, n, mode, accumulate);
    return ams_check_launch();
}

// y [B(1+S), TN] -> out [B*S, TN];  mode 0 mask, 1 perfect
ams_status ams_pretrain_separator_fwd(const float* y, float* out, int B, int S, long TN, int mode, void* stream) {
    AMS_REQUIRE(y && out && B > 0 && S > 0 && TN > 0 && (mode == 0 || mode == 1));
    int bx = (int)((TN + 255) / 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(pretrain_sep_fwd_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, y, out, B, S, TN, mode);
    return ams_check_launch();
}
ams_status ams_pretrain_separator_bwd(const float* dout, float* dy, int B, int S, long TN, int mode, void* stream) {
    AMS_REQUIRE(dout && dy && B > 0 && S > 0 && TN > 0 && (mode == 0 || mode == 1));
    int bx = (int)((TN + 255) / 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(pretrain_sep_bwd_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, dout, dy, B, S, TN, mode);
    return ams_check_launch();
}

// pre: 0 none, 1 abs, 2 sqrt, 3 log10(x + 1e-12);  norm: 0 none, 1 (z-min)/(max-min), 2 (z-mean)/sqrt(var), 3 z*[max-z < thr]
ams_status ams_row_transform(const float* x, float* out, int rows, long n, int pre, int norm, float thr, void* stream) {
    AMS_REQUIRE(x && out && rows > 0 && n > 0 && pre >= 0 && pre <= 3 && norm >= 0 && norm <= 3);
    hipLaunchKernelGGL(row_transform_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, out, n, pre, norm, thr);
    return ams_check_launch();
}

// in place on y [B,TF,S]; mode: 0 none, 1 linear, 2 sqrt, 3 square; sil_thr used when use_silence != 0
ams_status ams_weight_masks(const float* X, float* y, int B, long TF, int S, int mode, int use_silence, float sil_thr, void* stream) {
    AMS_REQUIRE(X && y && B > 0 && TF > 0 && S > 0 && mode >= 0 && mode <= 3);
    hipLaunchKernelGGL(weight_masks_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, X, y, TF, S, mode, use_silence, sil_thr);
    return ams_check_launch();
}

ams_status ams_silence_weights(const float* lat, float* w, int rows, long n, float thr, void* stream) {
    AMS_REQUIRE(lat && w && rows > 0 && n > 0);
    hipLaunchKernelGGL(silence_weights_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, lat, w, n, thr);
    return ams_check_launch();
}

}  // extern "C"
