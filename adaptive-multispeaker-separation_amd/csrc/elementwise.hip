// HBM-bound streaming kernels: filter construction, mask creation, l2-normalise, column sums,
// optimizers, global norm.  All are single-pass, float4 where the layout allows, grid-stride with
// <= 2048 workgroups (8 per CU) as the CDNA4 guide recommends for memory-bound work.
#include "common.h"

thread_local int g_ams_last_hip_error = 0;

#ifndef AMS_ABSMAX_BLOCKS
#define AMS_ABSMAX_BLOCKS 256
#endif
namespace {

inline int stream_blocks(long n, int per_block = 256) {
    long b = (n + per_block - 1) / per_block;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// f[k,n] = |w[k]| * bases[k,n]   (reference models/adapt.py:106, :234)
__global__ void front_filter_fwd_kernel(const float* __restrict__ w, const float* __restrict__ bases, float* __restrict__ f,
                                        int W, int N) {
    const long total = (long)W * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        f[i] = fabsf(w[i / N]) * bases[i];
}

// dbases = |w| * df ; dw[k] = sign(w[k]) * sum_n bases[k,n]*df[k,n]   (one wave per tap k)
__global__ void front_filter_bwd_kernel(const float* __restrict__ w, const float* __restrict__ bases,
                                        const float* __restrict__ df, float* __restrict__ dw, float* __restrict__ dbases,
                                        int W, int N) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= W) return;
    const float wk = w[k], aw = fabsf(wk);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float d = df[(long)k * N + n];
        s += bases[(long)k * N + n] * d;
        dbases[(long)k * N + n] = aw * d;
    }
    s = wave_sum(s);
    if (lane == 0) dw[k] = (wk > 0.f ? 1.f : (wk < 0.f ? -1.f : 0.f)) * s;
}

// Plugged-separator mask creation (reference models/network.py:369-378; STFT variant :499-502).
// rep_nm rows are (b,s) row-major, each [TF]; Y[b,i,s] = a if s == argmax_s' |rep[b,s',i]| else b_.
__global__ void make_masks_kernel(const float* __restrict__ rep_nm, float* __restrict__ Y, int32_t* __restrict__ am,
                                  int B, int S, long TF, float a, float b_, int take_abs) {
    const long total = (long)B * TF;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / TF, p = i - b * TF;
        float best = 0.f;
        int bi = 0;
        for (int s = 0; s < S; ++s) {
            float v = rep_nm[((long)b * S + s) * TF + p];
            if (take_abs) v = fabsf(v);
            if (s == 0 || v > best) { best = v; bi = s; }       // first index wins ties (tf.argmax)
        }
        for (int s = 0; s < S; ++s) Y[i * S + s] = (s == bi) ? a : b_;
        if (am) am[i] = bi;
    }
}

// tf.nn.l2_normalize over groups of E contiguous floats: v = u * rsqrt(max(sum u^2, 1e-12)).
// Quarter-wave (16 lanes) per group keeps 64 B..256 B contiguous per request for E = 40.
__global__ void l2norm_fwd_kernel(const float* __restrict__ u, float* __restrict__ v, float* __restrict__ inv, long rows, int E) {
    const int sub = threadIdx.x & 15;
    const long gid0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long gstride = ((long)gridDim.x * blockDim.x) >> 4;
    for (long r = gid0; r < rows; r += gstride) {
        const float* p = u + r * E;
        float ss = 0.f;
        for (int e = sub; e < E; e += 16) { const float x = p[e]; ss += x * x; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 16);
        const float iv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        for (int e = sub; e < E; e += 16) v[r * E + e] = p[e] * iv;
        if (inv && sub == 0) inv[r] = iv;
    }
}

// du = (dv - v <v,dv>) * inv   (clamp active -> du = dv * inv)
__global__ void l2norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ inv, const float* __restrict__ dv,
                                  float* __restrict__ du, long rows, int E) {
    const int sub = threadIdx.x & 15;
    const long gid0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long gstride = ((long)gridDim.x * blockDim.x) >> 4;
    for (long r = gid0; r < rows; r += gstride) {
        float dot = 0.f;
        for (int e = sub; e < E; e += 16) dot += v[r * E + e] * dv[r * E + e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 16);
        const float iv = inv[r];
        const bool active = iv < 0.999999e6f;
        for (int e = sub; e < E; e += 16) {
            const float d = dv[r * E + e];
            du[r * E + e] = active ? (d - v[r * E + e] * dot) * iv : d * iv;
        }
    }
}

// 16-byte, LDS-staged forms of the two kernels above for E % 4 == 0: a workgroup moves a slab of 256 rows (256 * E floats,
// contiguous in memory) with coalesced 16-byte accesses, and thread t works on row t out of LDS (row pitch E + 4 floats: 16-byte
// aligned rows, conflict-free 16-byte reads).  The quarter-wave kernels issue 4-byte accesses in 64-byte runs and reach
// 3.2-3.5 TB/s on the 210 MB embedding tensor; these are HBM-speed passes.
// TWICE: v = normalise(normalise(u)) in one pass -- what a k-means that normalises its input (Kmeans_2.py:56) computes on the output of
// the embedding network's own Normalize layer (dpcl.py:32).  The once-normalised row is rounded to f32 in between exactly as the
// stored tensor of the two-pass form is, so v, inv and inv2 carry the bits of two ams_l2norm_fwd calls.
// TWICE = 2: the second normalisation is the k-means' own (ams_kmeans_normalize, csrc/kmeans.hip: squares added left to right, separate
// multiply and add) -- the bits of ams_l2norm_fwd followed by ams_kmeans_normalize, which the bit-exact hard k-means consumes.
template <int E_, int TWICE = 0>
__global__ __launch_bounds__(256) void l2norm_fwd_slab_kernel(const float* __restrict__ u, float* __restrict__ v, float* __restrict__ inv,
                                                              long rows, float* __restrict__ inv2 = nullptr) {
    constexpr int V4 = E_ / 4, LD = E_ + 4;
    __shared__ __attribute__((aligned(16))) float tile[256 * LD];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * 256;
    const int nr = (int)min((long)256, rows - r0);
    const float4* src = reinterpret_cast<const float4*>(u + r0 * E_);
    float4 pre[V4];
#pragma unroll
    for (int k = 0; k < V4; ++k) pre[k] = src[min(tid + 256 * k, nr * V4 - 1)];           // unconditional, clamped
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
        *reinterpret_cast<float4*>(&tile[row * LD + c4 * 4]) = pre[k];
    }
    __syncthreads();
    if (tid < nr) {
        float4* p = reinterpret_cast<float4*>(&tile[tid * LD]);
        float4 x[V4];
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < V4; ++k) { x[k] = p[k]; ss += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w; }
        const float iv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
        for (int k = 0; k < V4; ++k) x[k] = make_float4(x[k].x * iv, x[k].y * iv, x[k].z * iv, x[k].w * iv);
        if (TWICE == 1) {
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < V4; ++k) s2 += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w;
            const float iv2 = 1.0f / sqrtf(fmaxf(s2, 1e-12f));
#pragma unroll
            for (int k = 0; k < V4; ++k) x[k] = make_float4(x[k].x * iv2, x[k].y * iv2, x[k].z * iv2, x[k].w * iv2);
            inv2[r0 + tid] = iv2;
        } else if (TWICE == 2) {
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < V4; ++k) {
                s2 = __fadd_rn(s2, __fmul_rn(x[k].x, x[k].x)); s2 = __fadd_rn(s2, __fmul_rn(x[k].y, x[k].y));
                s2 = __fadd_rn(s2, __fmul_rn(x[k].z, x[k].z)); s2 = __fadd_rn(s2, __fmul_rn(x[k].w, x[k].w));
            }
            const float iv2 = 1.0f / sqrtf(fmaxf(s2, 1e-12f));
#pragma unroll
            for (int k = 0; k < V4; ++k) x[k] = make_float4(__fmul_rn(x[k].x, iv2), __fmul_rn(x[k].y, iv2), __fmul_rn(x[k].z, iv2), __fmul_rn(x[k].w, iv2));
        }
#pragma unroll
        for (int k = 0; k < V4; ++k) p[k] = x[k];
        if (inv) inv[r0 + tid] = iv;
    }
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(v + r0 * E_);
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
        if (i < nr * V4) dst[i] = *reinterpret_cast<const float4*>(&tile[row * LD + c4 * 4]);
    }
}

template <int E_>
__global__ __launch_bounds__(256) void l2norm_bwd_slab_kernel(const float* __restrict__ v, const float* __restrict__ inv,
                                                              const float* __restrict__ dv, float* __restrict__ du, long rows) {
    constexpr int V4 = E_ / 4, LD = E_ + 4;
    __shared__ __attribute__((aligned(16))) float tile[256 * LD];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * 256;
    const int nr = (int)min((long)256, rows - r0);
    const float4* sv = reinterpret_cast<const float4*>(v + r0 * E_);
    const float4* sd = reinterpret_cast<const float4*>(dv + r0 * E_);
    float4 pv[V4], pd[V4];
#pragma unroll
    for (int k = 0; k < V4; ++k) pv[k] = sv[min(tid + 256 * k, nr * V4 - 1)];
#pragma unroll
    for (int k = 0; k < V4; ++k) pd[k] = sd[min(tid + 256 * k, nr * V4 - 1)];
    const float iv = inv[r0 + min(tid, nr - 1)];
    // v through LDS into this thread's row registers, then dv through the same buffer
    float4 xv[V4], xd[V4];
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
        *reinterpret_cast<float4*>(&tile[row * LD + c4 * 4]) = pv[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V4; ++k) xv[k] = *reinterpret_cast<const float4*>(&tile[tid * LD + k * 4]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
        *reinterpret_cast<float4*>(&tile[row * LD + c4 * 4]) = pd[k];
    }
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        xd[k] = *reinterpret_cast<const float4*>(&tile[tid * LD + k * 4]);
        dot += xv[k].x * xd[k].x + xv[k].y * xd[k].y + xv[k].z * xd[k].z + xv[k].w * xd[k].w;
    }
    const bool active = iv < 0.999999e6f;
    const float dd = active ? dot : 0.f;                      // clamp active -> du = dv * inv
#pragma unroll
    for (int k = 0; k < V4; ++k)
        *reinterpret_cast<float4*>(&tile[tid * LD + k * 4]) =
            make_float4((xd[k].x - xv[k].x * dd) * iv, (xd[k].y - xv[k].y * dd) * iv, (xd[k].z - xv[k].z * dd) * iv, (xd[k].w - xv[k].w * dd) * iv);
    __syncthreads();
    float4* dst = reinterpret_cast<float4*>(du + r0 * E_);
#pragma unroll
    for (int k = 0; k < V4; ++k) {
        const int i = tid + 256 * k, row = i / V4, c4 = i - row * V4;
        if (i < nr * V4) dst[i] = *reinterpret_cast<const float4*>(&tile[row * LD + c4 * 4]);
    }
}

// Column sums of a [rows, cols] matrix (bias gradients).  Two-stage, deterministic.
// Stage 1: a block owns 64 columns x `rows_per_block` rows; its 4 waves take interleaved rows (256-byte coalesced
// segments), then meet in LDS.  Stage 2 adds the few per-block partials in fixed order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long rows, int cols,
                                                             long ld, int rows_per_block) {
    __shared__ float sm[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < cols)
        for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + c];
    sm[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) part[(long)blockIdx.y * cols + c] = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}
// 16-byte variant (cols % 4 == 0, ld % 4 == 0, 16-byte aligned base): a block owns 256 columns, each wave reads whole
// 1-KB row segments, 8 independent loads per thread in flight.
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* __restrict__ x, float* __restrict__ part, long rows,
                                                                 int cols, long ld, int rows_per_block) {
    __shared__ float4 sm[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cl * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c < cols) {
        const float* base = x + c;
        long r = r0 + rl;
#pragma unroll 4
        for (; r + 4 < r1; r += 8) {
            const float4 a = *reinterpret_cast<const float4*>(base + r * ld);
            const float4 b = *reinterpret_cast<const float4*>(base + (r + 4) * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (r < r1) {
            const float4 a = *reinterpret_cast<const float4*>(base + r * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    sm[rl][cl] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float4 a = sm[0][cl], b = sm[1][cl], d = sm[2][cl], e = sm[3][cl];
        *reinterpret_cast<float4*>(part + (long)blockIdx.y * cols + c) =
            make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int cols, int nparts, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(long)p * cols + c];
    out[c] = accumulate ? out[c] + s : s;
}

// max |p| of what an optimizer kernel has just written, for the NEXT step's fp16x3 products (ops.param_amax), folded into `bound` by
// the launch itself: per thread in the update loop; per block one atomicMax into slot blockIdx & 63 (2048 blocks finishing together
// on ONE address are served one after another, ~10 ns each; 64 addresses are not); the block that completes its slot's group counts
// the group in, and the block that completes the last group folds the 64 slots into bound[0] and clears slots and tickets.
// slots: [0, 64) maxima, [64, 128) group tickets, [128] final ticket -- zero before the first launch, left zero.
constexpr int AMS_BOUND_SLOTS = 64;
__device__ __forceinline__ void block_amax_finish(unsigned m, unsigned* __restrict__ slots, float* __restrict__ bound) {
    __shared__ unsigned sm_amax[4];
    __shared__ int sm_last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) sm_amax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = blockIdx.x & (AMS_BOUND_SLOTS - 1);
        const unsigned old = atomicMax(slots + s, max(max(sm_amax[0], sm_amax[1]), max(sm_amax[2], sm_amax[3])));
        asm volatile("" :: "v"(old));                   // returned: the maximum is in place before this block is counted in
        const unsigned in_group = (gridDim.x - s + AMS_BOUND_SLOTS - 1) / AMS_BOUND_SLOTS;
        int last = 0;
        if (atomicAdd(slots + AMS_BOUND_SLOTS + s, 1u) == in_group - 1u) {
            const unsigned groups = min(gridDim.x, (unsigned)AMS_BOUND_SLOTS);
            last = atomicAdd(slots + 2 * AMS_BOUND_SLOTS, 1u) == groups - 1u;
        }
        sm_last = last;
    }
    __syncthreads();
    if (sm_last && threadIdx.x < AMS_BOUND_SLOTS) {
        unsigned t = __hip_atomic_load(slots + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slots + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slots + AMS_BOUND_SLOTS + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o));
        if (threadIdx.x == 0) {
            __hip_atomic_store(slots + 2 * AMS_BOUND_SLOTS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t != 0u) bound[0] = __uint_as_float(t);
        }
    }
}

// ---- optimizers (reference utils/ops.py:686-703; TF RMSProp/Momentum) ----
__global__ void amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               float* __restrict__ vh, long n, float lr_t, float b1, float b2, float eps, float gscale,
                               const unsigned* __restrict__ skip, unsigned* __restrict__ amax_slots, float* __restrict__ bound, const float* __restrict__ gscale_dev) {
    // skip: a recurrence launch of this step gave up a bounded wait (csrc/lstm_ring.hip, sticky error word): its gradients are
    // garbage -- leave parameters and slots untouched so that the caller can repeat the step on the per-step kernels
    if (skip && *skip != 0u) return;
    if (gscale_dev) gscale *= gscale_dev[0];         // tf.clip_by_global_norm factor, computed on the device (ams_clip_scale)
    unsigned am = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float vhi = fmaxf(vi, vh[i]);
        m[i] = mi; v[i] = vi; vh[i] = vhi;
        const float pn = p[i] - lr_t * mi / (sqrtf(vhi) + eps);
        p[i] = pn;
        am = max(am, __float_as_uint(pn) & 0x7fffffffu);
    }
    if (amax_slots) block_amax_finish(am, amax_slots, bound);
}
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms, long n, float lr,
                               float decay, float eps, float gscale, const unsigned* __restrict__ skip, unsigned* __restrict__ amax_slots, float* __restrict__ bound,
                               const float* __restrict__ gscale_dev) {
    if (skip && *skip != 0u) return;
    if (gscale_dev) gscale *= gscale_dev[0];
    unsigned am = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float s = decay * ms[i] + (1.0f - decay) * gi * gi;
        ms[i] = s;
        const float pn = p[i] - lr * gi / sqrtf(s + eps);
        p[i] = pn;
        am = max(am, __float_as_uint(pn) & 0x7fffffffu);
    }
    if (amax_slots) block_amax_finish(am, amax_slots, bound);
}
__global__ void momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ acc, long n, float lr,
                                float mom, float gscale, const unsigned* __restrict__ skip, unsigned* __restrict__ amax_slots, float* __restrict__ bound,
                                const float* __restrict__ gscale_dev) {
    if (skip && *skip != 0u) return;
    if (gscale_dev) gscale *= gscale_dev[0];
    unsigned am = 0u;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = mom * acc[i] + g[i] * gscale;
        acc[i] = a;
        const float pn = p[i] - lr * a;
        p[i] = pn;
        am = max(am, __float_as_uint(pn) & 0x7fffffffu);
    }
    if (amax_slots) block_amax_finish(am, amax_slots, bound);
}

// sum of squares -> per-block partials -> single value (deterministic two-stage)
__global__ void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ part, long n) {
    __shared__ float sm[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i] * x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void sum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int n) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sm[0] + sm[1] + sm[2] + sm[3];
}

// max |x| as float bits (non-negative floats order like unsigned integers; a NaN anywhere wins and comes out as NaN): operand bound
// of the fp16x3 products (csrc/gemm.hip).  out[0] must be zero on entry (the entry point clears it).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    __shared__ unsigned sm[4];
    unsigned m = 0;
    const long n4 = n / 4;
    const bool al = ((uintptr_t)x & 15) == 0;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    auto fold = [&](const uint4& v) {
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    };
    if (al) {
        const uint4* x4 = reinterpret_cast<const uint4*>(x);
        for (; i + 7 * stride < n4; i += 8 * stride) {          // eight independent 16-byte loads in flight per thread
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = x4[i + k * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) fold(v[k]);
        }
        for (; i < n4; i += stride) fold(x4[i]);
        for (long j = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) m = max(m, __float_as_uint(x[j]) & 0x7fffffffu);
    } else {
        for (; i < n; i += stride) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

// One batch into the static buffers of a captured step, and max |x| of it on the way: dst[0..n) = src[0..n) (16-byte loads), the second,
// small pair (the speaker indices) by block 0, per-block maxima as write-through stores, and the block that arrives last folds them
// (the last-arriver form of csrc/dpcl.hip: workgroup-scope release before the ticket, agent-scope loads of the partials).
// scratch: [0] ticket (zero on entry, left zero), [1 .. 1 + gridDim.x) partials.
__global__ __launch_bounds__(256) void stage_inputs_kernel(const float* __restrict__ src, float* __restrict__ dst, long n,
                                                           const unsigned char* __restrict__ src2, unsigned char* __restrict__ dst2, long n2,
                                                           float* __restrict__ amax_out, unsigned* __restrict__ scratch) {
    __shared__ unsigned sm[4];
    __shared__ int last_sh;
    unsigned m = 0u;
    const long n4 = n / 4, stride = (long)gridDim.x * 256;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    auto fold = [&](const uint4& v) {
        m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    };
    for (; i + 7 * stride < n4; i += 8 * stride) {          // eight independent 16-byte loads in flight per thread
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s4[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) { d4[i + k * stride] = v[k]; fold(v[k]); }
    }
    for (; i < n4; i += stride) { const uint4 v = s4[i]; d4[i] = v; fold(v); }
    for (long j = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) { dst[j] = src[j]; m = max(m, __float_as_uint(src[j]) & 0x7fffffffu); }
    if (blockIdx.x == 0)
        for (long j = threadIdx.x; j < n2; j += 256) dst2[j] = src2[j];
    if (!amax_out) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + 1 + blockIdx.x, max(max(sm[0], sm[1]), max(sm[2], sm[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last_sh = (atomicAdd(scratch, 1u) == gridDim.x - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!last_sh) return;
    unsigned t = 0u;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t = max(t, __hip_atomic_load(scratch + 1 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        amax_out[0] = __uint_as_float(max(max(sm[0], sm[1]), max(sm[2], sm[3])));
        __hip_atomic_store(scratch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// zero two regions in one launch (16-byte stores; both 16-byte aligned, sizes multiples of 16)
__global__ __launch_bounds__(256) void zero2_kernel(uint4* __restrict__ a, long na, uint4* __restrict__ b, long nb) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = z; else b[i - na] = z;
    }
}

// out[0] = clip / max(sqrt(sumsq[0]) * pre_scale, clip): tf.clip_by_global_norm's factor for gradients that the optimizer kernel will
// first multiply by pre_scale (the 1 / world of the data-parallel mean) -- on the device, so that a clipped step has no host round trip
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float pre_scale, float clip, float* __restrict__ out) {
    const float gn = sqrtf(sumsq[0]) * pre_scale;
    out[0] = clip / fmaxf(gn, clip);
}

// fp16x3 range audit of one operand (ops.f16_audit): out[0] += number of non-zero entries with |x| < bound * 2^-17 (the entries the
// arithmetic keeps fewer than 22 bits of), out[1] += their energy, out[2] += the operand's energy.  Float atomics: an audit, not a result.
__global__ __launch_bounds__(256) void range_share_kernel(const float* __restrict__ x, long rows, long cols, long ld, const float* __restrict__ bound,
                                                          float* __restrict__ out) {
    __shared__ float sm[3][4];
    const float thr = bound[0] * 1.52587890625e-05f * 0.5f;      // 2^-17
    float cnt = 0.f, es = 0.f, et = 0.f;
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols, c = i - r * cols;
        const float v = x[r * ld + c], a = fabsf(v);
        const float e = v * v;
        et += e;
        if (a < thr && a > 0.f) { cnt += 1.f; es += e; }
    }
    cnt = wave_sum(cnt); es = wave_sum(es); et = wave_sum(et);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = cnt; sm[1][threadIdx.x >> 6] = es; sm[2][threadIdx.x >> 6] = et; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(out + threadIdx.x, sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// one device-clock stamp (constant 100 MHz counter): brackets a launch INSIDE a captured hipGraph, where HIP events cannot be read back
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int slot) { buf[slot] = wall_clock64(); }

}  // namespace

extern "C" {

ams_status ams_absmax_f32(const float* x, long n, float* out, void* stream) {
    AMS_REQUIRE(x && out && n > 0);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return ams_check_launch();
    long blocks = (n / 4 + 255) / 256 / 8;          // >= 8 float4 per thread; at most two workgroups per CU: their final atomics all
    if (blocks < 1) blocks = 1;                     // arrive at one address at about the same time and are served one after another
    if (blocks > AMS_ABSMAX_BLOCKS) blocks = AMS_ABSMAX_BLOCKS;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, (unsigned*)out);
    return ams_check_launch();
}

ams_status ams_range_share(const float* x, long rows, long cols, long ld, const float* bound, float* out3, void* stream) {
    AMS_REQUIRE(x && bound && out3 && rows > 0 && cols > 0 && ld >= cols);
    long blocks = (rows * cols + 255) / 256 / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(range_share_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, bound, out3);
    return ams_check_launch();
}

ams_status ams_clip_scale(const float* sumsq, float pre_scale, float clip, float* out, void* stream) {
    AMS_REQUIRE(sumsq && out && clip > 0.f);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, pre_scale, clip, out);
    return ams_check_launch();
}

ams_status ams_zero2(void* a, size_t a_bytes, void* b, size_t b_bytes, void* stream) {
    AMS_REQUIRE((a || a_bytes == 0) && (b || b_bytes == 0) && a_bytes % 16 == 0 && b_bytes % 16 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0);
    const long n = (long)(a_bytes + b_bytes) / 16;
    if (n == 0) return AMS_OK;
    long blocks = (n + 255) / 256 / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)a, (long)a_bytes / 16, (uint4*)b, (long)b_bytes / 16);
    return ams_check_launch();
}

size_t ams_stage_inputs_scratch_bytes(void) { return (1 + 1024) * sizeof(unsigned); }
ams_status ams_stage_inputs(const float* src, float* dst, long n, const void* src2, void* dst2, long n2_bytes, float* amax_out,
                            void* scratch, void* stream) {
    AMS_REQUIRE(src && dst && n > 0 && (n2_bytes == 0 || (src2 && dst2)) && (!amax_out || scratch));
    AMS_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    long blocks = (n / 4 + 255) / 256 / 4;            // every block ends with a ticket on ONE word (~10 ns each): 256, not 1024 (20 us instead of 15)
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(stage_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n, (const unsigned char*)src2,
                       (unsigned char*)dst2, n2_bytes, amax_out, (unsigned*)scratch);
    return ams_check_launch();
}

ams_status ams_stamp(void* buf, int slot, void* stream) {
    AMS_REQUIRE(buf && slot >= 0);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)buf, slot);
    return ams_check_launch();
}
/* ticks per second of the ams_stamp counter */
long ams_stamp_rate(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) return 100000000L;
    return (long)khz * 1000L;
}

int ams_last_error(void) { return g_ams_last_hip_error; }
int ams_abi_version(void) { return AMS_ABI_VERSION; }

ams_status ams_front_filter_fwd(const float* w, const float* bases, float* f, int W, int N, void* stream) {
    AMS_REQUIRE(w && bases && f && W > 0 && N > 0);
    hipLaunchKernelGGL(front_filter_fwd_kernel, dim3(stream_blocks((long)W * N)), dim3(256), 0, (hipStream_t)stream, w, bases, f, W, N);
    return ams_check_launch();
}

ams_status ams_front_filter_bwd(const float* w, const float* bases, const float* df, float* dw, float* dbases, int W, int N,
                                void* stream) {
    AMS_REQUIRE(w && bases && df && dw && dbases && W > 0 && N > 0);
    hipLaunchKernelGGL(front_filter_bwd_kernel, dim3(ceil_div(W, 4)), dim3(256), 0, (hipStream_t)stream, w, bases, df, dw, dbases, W, N);
    return ams_check_launch();
}

ams_status ams_make_masks(const float* rep_non_mix, float* Y, int32_t* argmax, int B, int S, long TF, float a, float b,
                          int take_abs, void* stream) {
    AMS_REQUIRE(rep_non_mix && Y && B > 0 && S > 0 && TF > 0);
    hipLaunchKernelGGL(make_masks_kernel, dim3(stream_blocks((long)B * TF)), dim3(256), 0, (hipStream_t)stream, rep_non_mix, Y,
                       argmax, B, S, TF, a, b, take_abs);
    return ams_check_launch();
}

ams_status ams_l2norm_fwd(const float* u, float* v, float* inv, long rows, int E, void* stream) {
    AMS_REQUIRE(u && v && rows > 0 && E > 0);
    const bool vec = (((uintptr_t)u | (uintptr_t)v) & 15) == 0 && rows < (1L << 31) * 256;
    const dim3 sgrid((unsigned)ceil_div(rows, 256));
    hipStream_t st = (hipStream_t)stream;
    if (vec && E == 40) hipLaunchKernelGGL(l2norm_fwd_slab_kernel<40>, sgrid, dim3(256), 0, st, u, v, inv, rows);
    else if (vec && E == 32) hipLaunchKernelGGL(l2norm_fwd_slab_kernel<32>, sgrid, dim3(256), 0, st, u, v, inv, rows);
    else if (vec && E == 20) hipLaunchKernelGGL(l2norm_fwd_slab_kernel<20>, sgrid, dim3(256), 0, st, u, v, inv, rows);
    else if (vec && E == 8) hipLaunchKernelGGL(l2norm_fwd_slab_kernel<8>, sgrid, dim3(256), 0, st, u, v, inv, rows);
    else hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(stream_blocks(rows * 16)), dim3(256), 0, st, u, v, inv, rows, E);
    return ams_check_launch();
}

// xn = normalise(normalise(u)), inv = 1/|u|, inv2 = 1/|normalise(u)|: ONE pass (16-byte addressable rows of E = 40, 32, 20 or 8 floats;
// AMS_E_INVALID_ARG otherwise -- the caller then makes two ams_l2norm_fwd calls, which give the same bits).
ams_status ams_l2norm2_fwd(const float* u, float* xn, float* inv, float* inv2, long rows, int E, void* stream) {
    AMS_REQUIRE(u && xn && inv && inv2 && rows > 0 && E > 0);
    const bool vec = (((uintptr_t)u | (uintptr_t)xn) & 15) == 0 && rows < (1L << 31) * 256;
    if (!vec) return AMS_E_INVALID_ARG;
    const dim3 sgrid((unsigned)ceil_div(rows, 256));
    hipStream_t st = (hipStream_t)stream;
    if (E == 40) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<40, 1>), sgrid, dim3(256), 0, st, u, xn, inv, rows, inv2);
    else if (E == 32) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<32, 1>), sgrid, dim3(256), 0, st, u, xn, inv, rows, inv2);
    else if (E == 20) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<20, 1>), sgrid, dim3(256), 0, st, u, xn, inv, rows, inv2);
    else if (E == 8) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<8, 1>), sgrid, dim3(256), 0, st, u, xn, inv, rows, inv2);
    else return AMS_E_INVALID_ARG;
    return ams_check_launch();
}

// xn = kmeans_normalize(l2norm(u)) in ONE pass: the embedding network's Normalize layer (models/dpcl.py:32) followed by the k-means' own
// normalisation (models/Kmeans_2.py:40-41), with the bits of ams_l2norm_fwd followed by ams_kmeans_normalize.  E = 40, 32, 20, 8 and
// 16-byte addressable rows; AMS_E_INVALID_ARG otherwise (the caller then makes the two calls).
ams_status ams_l2norm_kmeans_normalize(const float* u, float* xn, long rows, int E, void* stream) {
    AMS_REQUIRE(u && xn && rows > 0 && E > 0);
    const bool vec = (((uintptr_t)u | (uintptr_t)xn) & 15) == 0 && rows < (1L << 31) * 256;
    if (!vec) return AMS_E_INVALID_ARG;
    const dim3 sgrid((unsigned)ceil_div(rows, 256));
    hipStream_t st = (hipStream_t)stream;
    if (E == 40) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<40, 2>), sgrid, dim3(256), 0, st, u, xn, (float*)nullptr, rows, (float*)nullptr);
    else if (E == 32) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<32, 2>), sgrid, dim3(256), 0, st, u, xn, (float*)nullptr, rows, (float*)nullptr);
    else if (E == 20) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<20, 2>), sgrid, dim3(256), 0, st, u, xn, (float*)nullptr, rows, (float*)nullptr);
    else if (E == 8) hipLaunchKernelGGL((l2norm_fwd_slab_kernel<8, 2>), sgrid, dim3(256), 0, st, u, xn, (float*)nullptr, rows, (float*)nullptr);
    else return AMS_E_INVALID_ARG;
    return ams_check_launch();
}

ams_status ams_l2norm_bwd(const float* v, const float* inv, const float* dv, float* du, long rows, int E, void* stream) {
    AMS_REQUIRE(v && inv && dv && du && rows > 0 && E > 0);
    const bool vec = (((uintptr_t)v | (uintptr_t)dv | (uintptr_t)du) & 15) == 0 && rows < (1L << 31) * 256;
    const dim3 sgrid((unsigned)ceil_div(rows, 256));
    hipStream_t st = (hipStream_t)stream;
    if (vec && E == 40) hipLaunchKernelGGL(l2norm_bwd_slab_kernel<40>, sgrid, dim3(256), 0, st, v, inv, dv, du, rows);
    else if (vec && E == 32) hipLaunchKernelGGL(l2norm_bwd_slab_kernel<32>, sgrid, dim3(256), 0, st, v, inv, dv, du, rows);
    else if (vec && E == 20) hipLaunchKernelGGL(l2norm_bwd_slab_kernel<20>, sgrid, dim3(256), 0, st, v, inv, dv, du, rows);
    else if (vec && E == 8) hipLaunchKernelGGL(l2norm_bwd_slab_kernel<8>, sgrid, dim3(256), 0, st, v, inv, dv, du, rows);
    else hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(stream_blocks(rows * 16)), dim3(256), 0, st, v, inv, dv, du, rows, E);
    return ams_check_launch();
}

constexpr int COLSUM_RPB = 128;
size_t ams_colsum_workspace_bytes(long rows, int cols) {
    const int nparts = ceil_div(rows, COLSUM_RPB);
    return (size_t)nparts * cols * sizeof(float);
}

ams_status ams_colsum(const float* x, float* out, long rows, int cols, long ld, int accumulate, void* ws, size_t ws_bytes,
                      void* stream) {
    AMS_REQUIRE(x && out && rows > 0 && cols > 0 && ws);
    const int rpb = COLSUM_RPB;
    const int nparts = ceil_div(rows, rpb);
    if ((size_t)nparts * cols * sizeof(float) > ws_bytes) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 15) == 0)
        hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3(ceil_div(cols, 256), nparts), dim3(256), 0, st, x, (float*)ws, rows, cols, ld, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3(ceil_div(cols, 64), nparts), dim3(256), 0, st, x, (float*)ws, rows, cols, ld, rpb);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 256)), dim3(256), 0, st, (const float*)ws, out, cols, nparts, accumulate);
    return ams_check_launch();
}

ams_status ams_opt_amsgrad(float* p, const float* g, float* m, float* v, float* vhat, long n, float lr_t, float beta1,
                           float beta2, float eps, float grad_scale, const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream) {
    AMS_REQUIRE(p && g && m && v && vhat && n > 0);
    hipLaunchKernelGGL(amsgrad_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vhat, n, lr_t, beta1,
                       beta2, eps, grad_scale, (const unsigned*)skip_if_set, bound_out ? (unsigned*)amax_slots : nullptr, bound_out, grad_scale_dev);
    return ams_check_launch();
}

ams_status ams_opt_rmsprop(float* p, const float* g, float* ms, long n, float lr, float decay, float eps, float grad_scale,
                           const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream) {
    AMS_REQUIRE(p && g && ms && n > 0);
    hipLaunchKernelGGL(rmsprop_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, ms, n, lr, decay, eps, grad_scale,
                       (const unsigned*)skip_if_set, bound_out ? (unsigned*)amax_slots : nullptr, bound_out, grad_scale_dev);
    return ams_check_launch();
}

ams_status ams_opt_momentum(float* p, const float* g, float* accum, long n, float lr, float momentum, float grad_scale,
                            const void* skip_if_set, void* amax_slots, float* bound_out, const float* grad_scale_dev, void* stream) {
    AMS_REQUIRE(p && g && accum && n > 0);
    hipLaunchKernelGGL(momentum_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, accum, n, lr, momentum, grad_scale,
                       (const unsigned*)skip_if_set, bound_out ? (unsigned*)amax_slots : nullptr, bound_out, grad_scale_dev);
    return ams_check_launch();
}

// out[0] = sum x^2  (ws: >= 1024 floats)
ams_status ams_sumsq(const float* x, float* out, long n, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(x && out && n > 0 && ws);
    int blocks = stream_blocks(n);
    if (blocks > 1024) blocks = 1024;
    if ((size_t)blocks * sizeof(float) > ws_bytes) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, x, (float*)ws, n);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, blocks);
    return ams_check_launch();
}

}  // extern "C"
