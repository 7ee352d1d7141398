// Output stage of the enhance layer (reference models/network.py:640-660) and the speaker-vector gather of the L41
// loss (models/L41.py:60-68).  Both are small streaming maps; they exist so that the default recipes run no framework
// arithmetic between the hand-written kernels.
//
//   enhance:  u [B,S,TF] (network output, rows (b,s))  ->  act over the SPEAKER axis (softmax | tanh | identity),
//             times the mixture representation X [B,TF]  ->  cost_in [B,TF,S]  and  separated [B,S,TF].
//   L41:      vs[b,s,:] = l2_normalize(table)[I[b,s], :]   (table [NSPK,E]);  backward scatters through the normalise
//             Jacobian with ONE workgroup per table row scanning the B*S indices in order (deterministic, no atomics).
#include "common.h"

namespace {

enum { NL_NONE = 0, NL_SOFTMAX = 1, NL_TANH = 2 };
constexpr int MAXS = 8;

__global__ __launch_bounds__(256) void enhance_out_fwd_kernel(const float* __restrict__ u, const float* __restrict__ X,
                                                              float* __restrict__ cost_in, float* __restrict__ sep, int S,
                                                              long TF, int nonlin) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < TF; p += (long)gridDim.x * blockDim.x) {
        float v[MAXS];
        float mx = -3.0e38f;
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < S) { v[s] = u[((long)b * S + s) * TF + p]; mx = fmaxf(mx, v[s]); }
        if (nonlin == NL_SOFTMAX) {
            float den = 0.f;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) { v[s] = expf(v[s] - mx); den += v[s]; }
            const float r = 1.0f / den;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) v[s] *= r;
        } else if (nonlin == NL_TANH) {
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) v[s] = tanhf(v[s]);
        }
        const float x = X[(long)b * TF + p];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < S) {
                const float y = v[s] * x;
                cost_in[((long)b * TF + p) * S + s] = y;
                if (sep) sep[((long)b * S + s) * TF + p] = y;
            }
    }
}

// du = d act / d u applied to dy = (d_cost_in[b,p,s] + d_sep[b,s,p]) * X[b,p]
__global__ __launch_bounds__(256) void enhance_out_bwd_kernel(const float* __restrict__ u, const float* __restrict__ X,
                                                              const float* __restrict__ d_cost_in, const float* __restrict__ d_sep,
                                                              float* __restrict__ du, int S, long TF, int nonlin) {
    const int b = blockIdx.y;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < TF; p += (long)gridDim.x * blockDim.x) {
        float v[MAXS], dy[MAXS];
        float mx = -3.0e38f;
        const float x = X[(long)b * TF + p];
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < S) {
                v[s] = u[((long)b * S + s) * TF + p];
                mx = fmaxf(mx, v[s]);
                float g = 0.f;
                if (d_cost_in) g += d_cost_in[((long)b * TF + p) * S + s];
                if (d_sep) g += d_sep[((long)b * S + s) * TF + p];
                dy[s] = g * x;
            }
        if (nonlin == NL_SOFTMAX) {
            float den = 0.f, dot = 0.f;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) { v[s] = expf(v[s] - mx); den += v[s]; }
            const float r = 1.0f / den;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) { v[s] *= r; dot += dy[s] * v[s]; }
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) dy[s] = v[s] * (dy[s] - dot);
        } else if (nonlin == NL_TANH) {
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (s < S) { const float t = tanhf(v[s]); dy[s] *= (1.0f - t * t); }
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < S) du[((long)b * S + s) * TF + p] = dy[s];
    }
}

// one 64-lane wave per gathered row
__global__ __launch_bounds__(64) void l41_speaker_fwd_kernel(const float* __restrict__ table, const int* __restrict__ I,
                                                             float* __restrict__ vs, int E, int nspk, int normalize) {
    const int r = blockIdx.x, lane = threadIdx.x;
    int k = I[r];
    k = k < 0 ? 0 : (k >= nspk ? nspk - 1 : k);
    const float* src = table + (long)k * E;
    float ss = 0.f;
    for (int e = lane; e < E; e += 64) { const float t = src[e]; ss += t * t; }
    ss = wave_sum(ss);
    const float iv = normalize ? 1.0f / sqrtf(fmaxf(ss, 1e-12f)) : 1.0f;       // tf.nn.l2_normalize epsilon
    for (int e = lane; e < E; e += 64) vs[(long)r * E + e] = src[e] * iv;
}

// d table[k,:] = sum over rows r with I[r] == k (ascending r) of J_k^T d_vs[r,:],  J = (I - v v^T) / |t_k|  (or I)
__global__ __launch_bounds__(64) void l41_speaker_bwd_kernel(const float* __restrict__ table, const int* __restrict__ I,
                                                             const float* __restrict__ d_vs, float* __restrict__ d_table, int R,
                                                             int E, int normalize) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const float* src = table + (long)k * E;
    float ss = 0.f;
    for (int e = lane; e < E; e += 64) { const float t = src[e]; ss += t * t; }
    ss = wave_sum(ss);
    const bool active = ss >= 1e-12f;
    const float iv = normalize ? 1.0f / sqrtf(fmaxf(ss, 1e-12f)) : 1.0f;
    // per-lane accumulators for up to 4 strided entries of the row (E <= 256)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
        if (I[r] != k) continue;                                  // uniform across the wave
        const float* g = d_vs + (long)r * E;
        float dot = 0.f;
        if (normalize && active)
            for (int e = lane; e < E; e += 64) dot += g[e] * src[e] * iv;
        dot = wave_sum(dot);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = lane + 64 * j;
            if (e < E) acc[j] += normalize ? (active ? (g[e] - src[e] * iv * dot) * iv : g[e] * iv) : g[e];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        if (e < E) d_table[(long)k * E + e] = acc[j];
    }
}

}  // namespace

extern "C" {

// u [B,S,TF], X [B,TF] -> cost_in [B,TF,S], separated [B,S,TF] (may be NULL).  nonlin: 0 none, 1 softmax over S, 2 tanh.
ams_status ams_enhance_output_fwd(const float* u, const float* X, float* cost_in, float* separated, int B, int S, long TF,
                                  int nonlin, void* stream) {
    AMS_REQUIRE(u && X && cost_in && B > 0 && S > 0 && S <= MAXS && TF > 0 && nonlin >= 0 && nonlin <= 2);
    int bx = (int)((TF + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(enhance_out_fwd_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, u, X, cost_in, separated, S, TF, nonlin);
    return ams_check_launch();
}

// du [B,S,TF] from d_cost_in [B,TF,S] and/or d_separated [B,S,TF] (either may be NULL, not both).
ams_status ams_enhance_output_bwd(const float* u, const float* X, const float* d_cost_in, const float* d_separated, float* du, int B,
                                  int S, long TF, int nonlin, void* stream) {
    AMS_REQUIRE(u && X && du && (d_cost_in || d_separated) && B > 0 && S > 0 && S <= MAXS && TF > 0 && nonlin >= 0 && nonlin <= 2);
    int bx = (int)((TF + 255) / 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(enhance_out_bwd_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, u, X, d_cost_in, d_separated, du, S, TF,
                       nonlin);
    return ams_check_launch();
}

// vs [R,E] = (l2-normalised if normalize) rows I[r] of table [nspk,E]
ams_status ams_l41_speaker_fwd(const float* table, const int* I, float* vs, int R, int E, int nspk, int normalize, void* stream) {
    AMS_REQUIRE(table && I && vs && R > 0 && E > 0 && nspk > 0);
    hipLaunchKernelGGL(l41_speaker_fwd_kernel, dim3(R), dim3(64), 0, (hipStream_t)stream, table, I, vs, E, nspk, normalize);
    return ams_check_launch();
}

// d_table [nspk,E] (fully written) from d_vs [R,E]
ams_status ams_l41_speaker_bwd(const float* table, const int* I, const float* d_vs, float* d_table, int R, int E, int nspk,
                               int normalize, void* stream) {
    AMS_REQUIRE(table && I && d_vs && d_table && R > 0 && E > 0 && E <= 256 && nspk > 0);
    hipLaunchKernelGGL(l41_speaker_bwd_kernel, dim3(nspk), dim3(64), 0, (hipStream_t)stream, table, I, d_vs, d_table, R, E, normalize);
    return ams_check_launch();
}

}  // extern "C"
