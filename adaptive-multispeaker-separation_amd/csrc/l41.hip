// Lab41 source-contrastive loss (reference models/L41.py:150-178, sampling=None):
//   cost = mean_{t,f} mean_b mean_s  -log(sigmoid(y[b,t,f,s] * <Vspk[b,s,:], emb[b,t,f,:]>))
// The reference broadcasts a [B,T,F,S,E] temporary (420 MB at the benchmark shape); here emb is streamed once per
// pass, thread-per-point with the S speaker vectors in LDS, points transposed through LDS for coalesced traffic.
// HBM-bound: algorithmic bytes = TF*(E+S)*4 per utterance forward, + TF*E*4 written backward.
//
// Negative sampling (`--sampling K`, models/L41.py:69-147,165-166): K further (already gathered / normalised) speaker vectors
// per bin enter with label -1 and weight ns_rate / K:  cost[b,t,f] += ns_rate * mean_k -log(sigmoid(-<neg_k, emb>)).  The caller
// passes negs [B, NSEL, K, E]: NSEL = 1 -- one set per utterance ('random', :117-139) -- or NSEL = S -- the set of the bin's
// dominant speaker argmax_s y ('k-nearest', :91-116).  Same kernel, same pass over emb.
#include "common.h"

namespace {

constexpr int MAXS = 4;
constexpr int MAXK = 16;          // negatives per set
constexpr int MAXN = 32;          // NSEL * K

__device__ __forceinline__ float softplus_neg(float z) {      // -log(sigmoid(z)) = log(1 + exp(-z)), stable
    return z > 0.f ? log1pf(expf(-z)) : -z + log1pf(expf(z));
}

struct NegArgs {
    const float* negs;            // [B, NSEL, K, E]; nullptr = no negative sampling
    float* dneg_part;             // [B, nblk, NSEL*K*E] (backward)
    int NSEL, K;
    float wn;                     // ns_rate * S / K: weight of one negative term relative to one speaker term
};

// FROM_U (round 4): emb is the network output BEFORE tf.nn.l2_normalize (models/L41.py:43 Normalize(3), utils/ops.py:323): the point
// is normalised in registers on the way in, and the backward applies the normalise Jacobian to the gradient before it writes it -- no
// l2-normalise pass before the loss (read + write of the embedding tensor) and none after it (two reads + one write): 0.9 ms of a 9.5 ms
// cfg5 step.  VEC: 16-byte global accesses and LDS rows (E % 4 == 0, 16-byte aligned tensors): the 4-byte form ran the backward at
// 2.3 TB/s.
template <int E_, bool BWD, bool NEG, bool FROM_U, bool VEC>
__global__ __launch_bounds__(256) void l41_kernel(const float* __restrict__ emb, const float* __restrict__ y,
                                                  const float* __restrict__ vs, const float* __restrict__ upstream,
                                                  float* __restrict__ part, float* __restrict__ demb, float* __restrict__ dvs_part,
                                                  float* __restrict__ amax_part, long TF, int S, int nblk, float scale, NegArgs na) {
    constexpr int LD = VEC ? E_ + 4 : E_ + 1;
    constexpr int V4 = E_ / 4;
    __shared__ __attribute__((aligned(16))) float tile[256 * LD];
    __shared__ float svs[MAXS * E_];
    __shared__ float red[BWD ? ((256 / E_) * MAXS * E_ > 4 * (MAXS * E_ + 1) ? (256 / E_) * MAXS * E_ : 4 * (MAXS * E_ + 1)) : 4];     // forward: 4 wave sums; backward: the parts' partial dVs sums
    __shared__ __attribute__((aligned(16))) float sdzs[BWD ? 256 * MAXS : 1];   // d cost / d <Vs_s, emb> of every point (backward: the dVs sums read them)
    __shared__ float sneg[NEG ? MAXN * E_ : 1];
    __shared__ float sdz[NEG && BWD ? 256 * MAXK : 1];          // d cost / d <neg_k, emb> of every point
    __shared__ int ssel[NEG && BWD ? 256 : 1];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * 256;
    const int npts = (int)min((long)256, TF - p0);
    const float* eb = emb + ((long)b * TF + p0) * E_;
    if constexpr (VEC) {
        const float4* src = reinterpret_cast<const float4*>(eb);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
            const float4 v4 = src[min(i, npts * V4 - 1)];                    // unconditional, clamped (rows past npts are never read back)
            *reinterpret_cast<float4*>(&tile[row * LD + c4 * 4]) = v4;
        }
    } else {
        for (int i = tid; i < npts * E_; i += 256) tile[(i / E_) * LD + (i % E_)] = eb[i];
    }
    for (int i = tid; i < S * E_; i += 256) svs[i] = vs[(long)b * S * E_ + i];
    if (NEG)
        for (int i = tid; i < na.NSEL * na.K * E_; i += 256) sneg[i] = na.negs[(long)b * na.NSEL * na.K * E_ + i];
    __syncthreads();
    float cost = 0.f;
    float dz[MAXS] = {0.f, 0.f, 0.f, 0.f};
    float v[E_];
    int sel = 0;
    float inv_u = 1.0f;
    if (tid < npts) {
#pragma unroll
        for (int e = 0; e < E_; ++e) v[e] = tile[tid * LD + e];
        if constexpr (FROM_U) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < E_; ++e) ss += v[e] * v[e];
            inv_u = 1.0f / sqrtf(fmaxf(ss, 1e-12f));                          // tf.nn.l2_normalize epsilon (utils/ops.py:323)
#pragma unroll
            for (int e = 0; e < E_; ++e) v[e] *= inv_u;
            if (BWD) {                                                       // the dVs sums (and the negatives' gradient) read the points from LDS
#pragma unroll
                for (int e = 0; e < E_; ++e) tile[tid * LD + e] = v[e];
            }
        }
        const float* yp = y + ((long)b * TF + p0 + tid) * S;
        const float up = BWD ? upstream[0] * scale : 0.f;
        for (int s = 0; s < S; ++s) {
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < E_; ++e) dot += v[e] * svs[s * E_ + e];
            const float ys = yp[s], z = ys * dot;
            cost += softplus_neg(z);
            if (BWD) dz[s] = -ys * up / (1.0f + expf(z));          // d/d dot of -log sigmoid(y dot) = -y sigmoid(-z)
        }
        if (NEG) {
            if (na.NSEL > 1) {                                     // dominant speaker: first maximum of y (tf.argmax, L41.py:75)
                float best = yp[0];
                for (int s = 1; s < S; ++s) { const float ys = yp[s]; if (ys > best) { best = ys; sel = s; } }
            }
            for (int k = 0; k < na.K; ++k) {
                const float* nv = &sneg[(sel * na.K + k) * E_];
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < E_; ++e) dot += v[e] * nv[e];
                cost += na.wn * softplus_neg(-dot);
                if (BWD) sdz[tid * MAXK + k] = na.wn * up / (1.0f + expf(-dot));       // d/d dot of -log sigmoid(-dot) = sigmoid(dot)
            }
            if (BWD) ssel[tid] = sel;
        }
    }
    if (!BWD) {
        cost = wave_sum(cost);
        if ((tid & 63) == 0) red[tid >> 6] = cost;
        __syncthreads();
        if (tid == 0) part[(long)b * nblk + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        return;
    }
    // backward: demb = sum_s dz_s * Vs_s ; dVs_s += dz_s * emb (block partial)
    __syncthreads();
    if (NEG) {
        // d negs of this block: output (set j, negative k, e) = sum over the block's points that used set j, in point order
        // (fixed order -> deterministic); `tile` still holds emb here
        const int nout = na.NSEL * na.K * E_;
        for (int o = tid; o < nout; o += 256) {
            const int e = o % E_, jk = o / E_, k = jk % na.K, j = jk / na.K;
            float acc = 0.f;
            for (int p = 0; p < npts; ++p)
                if (ssel[p] == j) acc += sdz[p * MAXK + k] * tile[p * LD + e];
            na.dneg_part[((long)b * nblk + blockIdx.x) * nout + o] = acc;
        }
        __syncthreads();
    }
    float dv[E_];
    float amx = 0.f, dvs_tot = 0.f;
    if (tid < npts) {
#pragma unroll
        for (int e = 0; e < E_; ++e) {
            float d = 0.f;
            for (int s = 0; s < S; ++s) d += dz[s] * svs[s * E_ + e];
            if (NEG)
                for (int k = 0; k < na.K; ++k) d += sdz[tid * MAXK + k] * sneg[(sel * na.K + k) * E_ + e];
            dv[e] = d;
        }
        if constexpr (FROM_U) {                                                  // du = (dv - v <v, dv>) / |u|
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < E_; ++e) dot += v[e] * dv[e];
#pragma unroll
            for (int e = 0; e < E_; ++e) dv[e] = (dv[e] - v[e] * dot) * inv_u;
        }
#pragma unroll
        for (int e = 0; e < E_; ++e) amx = fmaxf(amx, fabsf(dv[e]));
    }
    // dVs_s[e] of the block = sum over its points of dz_s * v_e.  Round 5 ran S * E wave-wide halving trees per block (120 at S = 3:
    // ~10 instructions each with their nops -- the longest phase of the kernel, 0.34 of HBM at cfg5); now the points' dz lie in LDS beside
    // the points themselves and thread (o = (s, e), half) adds its half of the block's points in point order: 128 steps of two LDS reads
    // and one FMA, fixed order (deterministic).
#pragma unroll
    for (int s = 0; s < MAXS; ++s) sdzs[tid * MAXS + s] = (tid < npts && s < S) ? dz[s] : 0.f;
    __syncthreads();
    {
        // thread (feature e, part): the part's points one after another, ONE read of the point's feature and one 16-byte read of its
        // dz serve all S sums (S * E threads with one sum each read 3 x as much); the parts meet below in part order
        constexpr int NPART = 256 / E_;                                          // 6 parts at E = 40 (240 threads busy)
        const int eo = tid % E_, part = tid / E_;
        if (part < NPART) {
            const int per = (256 + NPART - 1) / NPART;
            const int pa = part * per, pb = min(npts, pa + per);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int p = pa; p < pb; ++p) {
                const float vv = tile[p * LD + eo];
                const float4 dz4 = *reinterpret_cast<const float4*>(&sdzs[p * MAXS]);
                a0 += dz4.x * vv; a1 += dz4.y * vv; a2 += dz4.z * vv; a3 += dz4.w * vv;
            }
            float* const r = &red[part * (MAXS * E_)];
            r[eo] = a0; r[E_ + eo] = a1; r[2 * E_ + eo] = a2; r[3 * E_ + eo] = a3;
        }
        __syncthreads();
        if (tid < S * E_) {                                                      // (S * E <= 160: one output per thread, kept in a register)
#pragma unroll
            for (int q = 0; q < NPART; ++q) dvs_tot += red[q * (MAXS * E_) + tid];
        }
    }
    __syncthreads();                                                             // the points have been read: the tile takes the gradients
    if (tid < npts) {
#pragma unroll
        for (int e = 0; e < E_; ++e) tile[tid * LD + e] = dv[e];
    }
    if (amax_part != nullptr) {                                                  // max |d emb| of the block (the launch's final kernel folds them)
        amx = wave_max(amx);
        if ((tid & 63) == 0) sdzs[tid >> 6] = amx;                               // (the points' dz are dead by now)
    }
    __syncthreads();
    float* db = demb + ((long)b * TF + p0) * E_;
    if constexpr (VEC) {
        float4* dst = reinterpret_cast<float4*>(db);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
            if (i < npts * V4) dst[i] = *reinterpret_cast<const float4*>(&tile[row * LD + c4 * 4]);
        }
    } else {
        for (int i = tid; i < npts * E_; i += 256) db[i] = tile[(i / E_) * LD + (i % E_)];
    }
    if (tid < S * E_) dvs_part[((long)b * nblk + blockIdx.x) * (S * E_) + tid] = dvs_tot;
    if (amax_part != nullptr && tid == 0)
        amax_part[(long)b * nblk + blockIdx.x] = fmaxf(fmaxf(sdzs[0], sdzs[1]), fmaxf(sdzs[2], sdzs[3]));
}

__global__ __launch_bounds__(1024) void l41_cost_final_kernel(const float* __restrict__ part, float* __restrict__ out, long n, float scale) {
    __shared__ float sm[16];                            // one workgroup of 16 waves: 20480 partials at cfg5 took 256 threads 27 us
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
        out[0] = t * scale;
    }
}

// (block 0 also folds the blocks' max |d emb| into amax_out[0]: the operand bound of the two products that read d emb -- no pass over it)
__global__ void l41_dvs_final_kernel(const float* __restrict__ part, float* __restrict__ dvs, int nblk, int SE, int B,
                                     const float* __restrict__ amax_part, float* __restrict__ amax_out) {
    if (amax_out != nullptr && blockIdx.x == 0) {
        __shared__ float sm[4];
        float m = 0.f;
        for (long i = threadIdx.x; i < (long)B * nblk; i += blockDim.x) m = fmaxf(m, amax_part[i]);
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) amax_out[0] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    }
    // one WAVE per output (b, k): its lanes take the blocks c = lane, lane + 64, ... and meet in a fixed tree -- a thread per output
    // walked 160 blocks one load at a time (94 us at cfg5: 2.4 M floats, 15 K threads)
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= B * SE) return;
    const int b = i / SE, k = i - b * SE;
    float s = 0.f;
    for (int c = lane; c < nblk; c += 64) s += part[((long)b * nblk + c) * SE + k];
    s = wave_sum(s);
    if (lane == 0) dvs[i] = s;
}

}  // namespace

extern "C" {

size_t ams_l41_workspace_bytes(int B, long TF, int E, int S) {
    const int nblk = ceil_div(TF, 256);
    return sizeof(float) * (size_t)B * nblk * ((S * E > 1 ? S * E : 1) + 1);          // + one maximum per block (backward: amax_out)
}

#define AMS_L41_CASE(EE, BWD, NEG, U, V, ...) hipLaunchKernelGGL((l41_kernel<EE, BWD, NEG, U, V>), grid, dim3(256), 0, st, __VA_ARGS__)
#define AMS_L41_E(EE, BWD, NEG, ...)                                                                          \
    if (EE % 4 == 0 && vec) { if (from_u) AMS_L41_CASE(EE, BWD, NEG, true, (EE % 4 == 0), __VA_ARGS__); else AMS_L41_CASE(EE, BWD, NEG, false, (EE % 4 == 0), __VA_ARGS__); } \
    else { if (from_u) AMS_L41_CASE(EE, BWD, NEG, true, false, __VA_ARGS__); else AMS_L41_CASE(EE, BWD, NEG, false, false, __VA_ARGS__); }
#define AMS_L41_DISPATCH(BWD, NEG, ...)                                                                       \
    switch (E) {                                                                                              \
        case 40: AMS_L41_E(40, BWD, NEG, __VA_ARGS__) break;                                                   \
        case 32: AMS_L41_E(32, BWD, NEG, __VA_ARGS__) break;                                                   \
        case 20: AMS_L41_E(20, BWD, NEG, __VA_ARGS__) break;                                                   \
        case 16: AMS_L41_E(16, BWD, NEG, __VA_ARGS__) break;                                                   \
        case 8: AMS_L41_E(8, BWD, NEG, __VA_ARGS__) break;                                                     \
        case 4: AMS_L41_E(4, BWD, NEG, __VA_ARGS__) break;                                                     \
        case 3: AMS_L41_E(3, BWD, NEG, __VA_ARGS__) break;                                                     \
        default: return AMS_E_INVALID_ARG;                                                                    \
    }
inline bool l41_vec(const void* a, const void* b) { return (((uintptr_t)a | (uintptr_t)b) & 15) == 0; }

// emb [B,TF,E], y [B,TF,S] (+1/-1), vspk [B,S,E] (already gathered / normalised) -> cost[0]
// emb_is_u (all four entry points): emb is the network output BEFORE tf.nn.l2_normalize over E (models/L41.py:43): it is normalised
// inside the pass, and the backward returns the gradient w.r.t. that un-normalised tensor.
ams_status ams_l41_loss_fwd(const float* emb, const float* y, const float* vspk, float* cost, int B, long TF, int E, int S, int emb_is_u,
                            void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(emb && y && vspk && cost && ws && B > 0 && TF > 0 && S > 0 && S <= MAXS);
    if (ws_bytes < ams_l41_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ceil_div(TF, 256);
    dim3 grid(nblk, B);
    const float scale = 1.0f / ((float)B * (float)TF * S);
    const bool from_u = emb_is_u != 0, vec = l41_vec(emb, emb);
    AMS_L41_DISPATCH(false, false, emb, y, vspk, (const float*)nullptr, (float*)ws, (float*)nullptr, (float*)nullptr, (float*)nullptr, TF, S, nblk, scale, NegArgs{})
    hipLaunchKernelGGL(l41_cost_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)ws, cost, (long)B * nblk, scale);
    return ams_check_launch();
}

// demb [B,TF,E], dvspk [B,S,E]; upstream = device scalar d loss / d cost
// amax_out (optional, both backward entry points): one float that receives max |demb| of this launch
ams_status ams_l41_loss_bwd(const float* emb, const float* y, const float* vspk, const float* upstream, float* demb, float* dvspk,
                            float* amax_out, int B, long TF, int E, int S, int emb_is_u, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(emb && y && vspk && upstream && demb && dvspk && ws && B > 0 && TF > 0 && S > 0 && S <= MAXS);
    if (ws_bytes < ams_l41_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ceil_div(TF, 256);
    dim3 grid(nblk, B);
    const float scale = 1.0f / ((float)B * (float)TF * S);
    const bool from_u = emb_is_u != 0, vec = l41_vec(emb, demb);
    float* const amax_part = amax_out ? (float*)ws + (size_t)B * nblk * S * E : nullptr;
    AMS_L41_DISPATCH(true, false, emb, y, vspk, upstream, (float*)nullptr, demb, (float*)ws, amax_part, TF, S, nblk, scale, NegArgs{})
    hipLaunchKernelGGL(l41_dvs_final_kernel, dim3(ceil_div(B * S * E, 4)), dim3(256), 0, st, (const float*)ws, dvspk, nblk, S * E, B,
                       (const float*)amax_part, amax_out);
    return ams_check_launch();
}

// ---- the same loss with negative sampling (models/L41.py:69-147,165-166).  negs [B,NSEL,K,E]: NSEL = 1 or S, NSEL * K <= 32, K <= 16.
size_t ams_l41_ns_workspace_bytes(int B, long TF, int E, int S, int NSEL, int K) {
    const int nblk = ceil_div(TF, 256);
    return sizeof(float) * (size_t)B * nblk * ((size_t)S * E + (size_t)NSEL * K * E + 1);
}

ams_status ams_l41_loss_ns_fwd(const float* emb, const float* y, const float* vspk, const float* negs, float* cost, int B, long TF, int E,
                               int S, int NSEL, int K, float ns_rate, int emb_is_u, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(emb && y && vspk && negs && cost && ws && B > 0 && TF > 0 && S > 0 && S <= MAXS);
    AMS_REQUIRE((NSEL == 1 || NSEL == S) && K > 0 && K <= MAXK && NSEL * K <= MAXN);
    if (ws_bytes < ams_l41_ns_workspace_bytes(B, TF, E, S, NSEL, K)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ceil_div(TF, 256);
    dim3 grid(nblk, B);
    const float scale = 1.0f / ((float)B * (float)TF * S);
    NegArgs na{negs, nullptr, NSEL, K, ns_rate * (float)S / (float)K};
    const bool from_u = emb_is_u != 0, vec = l41_vec(emb, emb);
    AMS_L41_DISPATCH(false, true, emb, y, vspk, (const float*)nullptr, (float*)ws, (float*)nullptr, (float*)nullptr, (float*)nullptr, TF, S, nblk, scale, na)
    hipLaunchKernelGGL(l41_cost_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)ws, cost, (long)B * nblk, scale);
    return ams_check_launch();
}

// demb [B,TF,E], dvspk [B,S,E], dnegs [B,NSEL,K,E]
ams_status ams_l41_loss_ns_bwd(const float* emb, const float* y, const float* vspk, const float* negs, const float* upstream, float* demb,
                               float* dvspk, float* dnegs, float* amax_out, int B, long TF, int E, int S, int NSEL, int K, float ns_rate,
                               int emb_is_u, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(emb && y && vspk && negs && upstream && demb && dvspk && dnegs && ws && B > 0 && TF > 0 && S > 0 && S <= MAXS);
    AMS_REQUIRE((NSEL == 1 || NSEL == S) && K > 0 && K <= MAXK && NSEL * K <= MAXN);
    if (ws_bytes < ams_l41_ns_workspace_bytes(B, TF, E, S, NSEL, K)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = ceil_div(TF, 256);
    dim3 grid(nblk, B);
    const float scale = 1.0f / ((float)B * (float)TF * S);
    float* dvs_part = (float*)ws;
    float* dneg_part = dvs_part + (size_t)B * nblk * S * E;
    NegArgs na{negs, dneg_part, NSEL, K, ns_rate * (float)S / (float)K};
    const bool from_u = emb_is_u != 0, vec = l41_vec(emb, demb);
    float* const amax_part = amax_out ? dneg_part + (size_t)B * nblk * NSEL * K * E : nullptr;
    AMS_L41_DISPATCH(true, true, emb, y, vspk, upstream, (float*)nullptr, demb, dvs_part, amax_part, TF, S, nblk, scale, na)
    hipLaunchKernelGGL(l41_dvs_final_kernel, dim3(ceil_div(B * S * E, 4)), dim3(256), 0, st, (const float*)dvs_part, dvspk, nblk, S * E, B,
                       (const float*)amax_part, amax_out);
    const int NE = NSEL * K * E;
    hipLaunchKernelGGL(l41_dvs_final_kernel, dim3(ceil_div(B * NE, 4)), dim3(256), 0, st, (const float*)dneg_part, dnegs, nblk, NE, B,
                       (const float*)nullptr, (float*)nullptr);
    return ams_check_launch();
}

}  // extern "C"
