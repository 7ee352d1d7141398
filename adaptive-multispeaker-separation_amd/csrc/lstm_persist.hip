// Persistent BLSTM recurrence for gfx950: ONE launch per layer and pass instead of one per time step.
//
// Why: the per-step kernels of lstm.hip cost ~6-8 us each although their MFMA work is ~1-2 us -- the rest is launch,
// drain and first-touch latency, paid 480 times per training step (reference utils/ops.py:358-383: TF pays a whole
// while_loop iteration of ~10 small kernels instead).  Here the (direction, 16-row batch tile) GROUPS of the step
// kernels become independent rings of n_ut workgroups that stay resident for all T steps and hand h_t (forward) /
// da_t (backward) to each other in-launch.
//
// Hand-off = the R2 "granule" form of the CDNA4 guide (cdna_hip_programming.md G16): every exchanged float travels
// as ONE naturally aligned 8-byte {epoch tag, value} word written by ONE relaxed agent-scope atomic store (lowers to
// a write-through `sc1` store) and read by relaxed agent-scope atomic loads (`sc1`, bypass the CU's L1); the data is
// the flag, so there is no fence and no separate counter.  Epoch = time step + 1; slots are double-buffered by step
// parity (a writer can be at most one step ahead of any reader of its group); the granule buffer is
// zeroed by a memset node before every launch (epoch 0 = never written), never by a per-launch salt (frozen under
// hipGraph replay).  Placement-independent for correctness; block -> group = blockIdx % n_groups only AIMS to keep a
// ring on one XCD (blocks are observed to go to XCD b % 8) so the exchange stays in that XCD's L2.
// Every spin is bounded: on timeout the kernel raises an error word, all rings stop waiting, and the (wrong) launch
// ends -- the host reads the word (ams_blstm_persist_error).
//
// Residency: n_groups * n_ut workgroups of 256 threads must all be resident (<= 256 CUs, one per CU); the host
// wrapper falls back to the per-step kernels otherwise.  Weights (packed MFMA fragments) are loaded into registers
// ONCE and reused for all T steps; cell state / running dc live in registers.
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int TU = 16, TB = 16;
constexpr unsigned SPIN_LIMIT = 1u << 21;

struct PArgs {
    float* G; float* out; float* cst; const float* pk; u64* gran; unsigned* err;
    const float* dout;
    int B, T, H, n_ut, n_g, n_groups, KP;
};

__device__ __forceinline__ u64 gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool spin_check(unsigned& spins, unsigned* err) {
    // returns true when the wait must be abandoned (timeout here, or another ring already gave up)
    ++spins;
    if ((spins & 1023u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (spins >= SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void lstm_persist_fwd_kernel(PArgs a) {
    constexpr int CH = 5;                                   // k-groups per wave (n_g <= 20)
    __shared__ __attribute__((aligned(16))) float red[4][4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = blockIdx.x % a.n_groups, ut = blockIdx.x / a.n_groups;
    const int dir = group & 1, bt = group >> 1;
    const int H = a.H, T = a.T, KP = a.KP;
    const int b_row = bt * TB + (lane & 15);
    const bool row_ok = b_row < a.B;
    const int kq = (lane >> 4) * 4;

    // recurrent weights: packed fragments, loaded once for all T steps
    float4 bv[CH][4];
    {
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (4 * 64 * 4) + lane * 4;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int g = wave + 4 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bv[i][q] = (g < a.n_g) ? *reinterpret_cast<const float4*>(pk + ((long)g * 4 + q) * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    const bool live = (b < a.B && u < H);
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    u64* gp = a.gran + (size_t)group * 2 * TB * KP;
    float c_state = 0.f;

    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        float* grow = a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H);
        float zq[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = grow[q * H + u];
        }
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const u64* src = gp + (size_t)((s - 1) & 1) * TB * KP + (size_t)(lane & 15) * KP;
            const unsigned epoch = (unsigned)s;             // h_{s-1} carries epoch s
            float4 av[CH];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int k = (wave + 4 * i) * 16 + kq;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = 0.f;
                        if (row_ok && k + j < H) {
                            const u64 x = gran_load(src + k + j);
                            ok &= ((unsigned)(x >> 32) == epoch);
                            v[j] = __uint_as_float((unsigned)x);
                        }
                    }
                    av[i] = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (__all(ok)) break;
                if (spin_check(spins, a.err)) break;
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i][q].x, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i][q].y, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i][q].z, acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i][q].w, acc[q], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&red[wave][q][lane][0]) = acc[q];
        __syncthreads();
        if (live) {
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = zq[q];
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[w][q][src_lane][src_reg];
                pre[q] = v;
            }
            const float ig = 1.0f / (1.0f + expf(-pre[0]));
            const float gg = tanhf(pre[1]);
            const float fg = 1.0f / (1.0f + expf(-(pre[2] + 1.0f)));
            const float og = 1.0f / (1.0f + expf(-pre[3]));
            const float c = c_state * fg + ig * gg;
            const float h = tanhf(c) * og;
            c_state = c;
            gran_store(gp + (size_t)(s & 1) * TB * KP + (size_t)bl * KP + u, (unsigned)(s + 1), h);   // hand h_s to the ring first
            grow[0 * H + u] = ig;
            grow[1 * H + u] = gg;
            grow[2 * H + u] = fg;
            grow[3 * H + u] = og;
            a.cst[(((long)b * T + t) * 2 + dir) * H + u] = c;
            a.out[((long)b * T + t) * (2 * H) + dir * H + u] = h;
        }
        __syncthreads();                                    // red[] is rewritten next step
    }
}

// ----------------------------------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void lstm_persist_bwd_kernel(PArgs a) {
    constexpr int CH = 20;                                  // k-groups per wave (n_g <= 80, i.e. 4H <= 1280)
    __shared__ __attribute__((aligned(16))) float red[4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = blockIdx.x % a.n_groups, ut = blockIdx.x / a.n_groups;
    const int dir = group & 1, bt = group >> 1;
    const int H = a.H, T = a.T, KP = a.KP, H4 = 4 * a.H;
    const int b_row = bt * TB + (lane & 15);
    const bool row_ok = b_row < a.B;
    const int kq = (lane >> 4) * 4;

    float4 bv[CH];
    {
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (64 * 4) + lane * 4;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int g = wave + 4 * i;
            bv[i] = (g < a.n_g) ? *reinterpret_cast<const float4*>(pk + (long)g * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    unsigned need = 0;                                      // k-groups this lane must receive each step
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (wave + 4 * i < a.n_g && row_ok && (wave + 4 * i) * 16 + kq < H4) need |= (1u << i);

    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    const bool live = (b < a.B && u < H);
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    u64* gp = a.gran + (size_t)group * 2 * TB * KP;
    float dc_state = 0.f;

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);
        const int tp = dir ? t + 1 : t - 1;
        const bool has_prev = (tp >= 0 && tp < T);
        float* grow = a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H);
        float dh = 0.f, ig = 0.f, gg = 0.f, fg = 0.f, og = 0.f, c = 0.f, c_prev = 0.f;
        if (live) {
            dh = a.dout[((long)b * T + t) * (2 * H) + dir * H + u];
            ig = grow[0 * H + u]; gg = grow[1 * H + u]; fg = grow[2 * H + u]; og = grow[3 * H + u];
            c = a.cst[(((long)b * T + t) * 2 + dir) * H + u];
            if (has_prev) c_prev = a.cst[(((long)b * T + tp) * 2 + dir) * H + u];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const u64* src = gp + (size_t)((s - 1) & 1) * TB * KP + (size_t)(lane & 15) * KP;
            const unsigned epoch = (unsigned)s;
            float4 av[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) av[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned got = 0, spins = 0;
            for (;;) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if ((need & ~got) & (1u << i)) {
                        const int k = (wave + 4 * i) * 16 + kq;       // 4H is a multiple of 4: the float4 never straddles the end
                        const u64 x0 = gran_load(src + k), x1 = gran_load(src + k + 1);
                        const u64 x2 = gran_load(src + k + 2), x3 = gran_load(src + k + 3);
                        if ((unsigned)(x0 >> 32) == epoch && (unsigned)(x1 >> 32) == epoch && (unsigned)(x2 >> 32) == epoch &&
                            (unsigned)(x3 >> 32) == epoch) {
                            av[i] = make_float4(__uint_as_float((unsigned)x0), __uint_as_float((unsigned)x1),
                                                __uint_as_float((unsigned)x2), __uint_as_float((unsigned)x3));
                            got |= (1u << i);
                        }
                    }
                }
                if (__all(got == need)) break;
                if (spin_check(spins, a.err)) break;
            }
#pragma unroll
            for (int i = 0; i < CH; i += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].x, bv[i + 1].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].y, bv[i + 1].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].z, bv[i + 1].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].w, bv[i + 1].w, acc1, 0, 0, 0);
            }
        }
        acc0 += acc1;
        *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc0;
        __syncthreads();
        if (live) {
#pragma unroll
            for (int w = 0; w < 4; ++w) dh += red[w][src_lane][src_reg];
            const float tc = tanhf(c);
            const float d_o = dh * tc;
            const float dcv = dc_state + dh * og * (1.0f - tc * tc);
            const float da0 = dcv * gg * ig * (1.0f - ig);
            const float da1 = dcv * ig * (1.0f - gg * gg);
            const float da2 = dcv * c_prev * fg * (1.0f - fg);
            const float da3 = d_o * og * (1.0f - og);
            dc_state = dcv * fg;
            u64* dst = gp + (size_t)(s & 1) * TB * KP + (size_t)bl * KP + u;
            gran_store(dst + 0 * H, (unsigned)(s + 1), da0);
            gran_store(dst + 1 * H, (unsigned)(s + 1), da1);
            gran_store(dst + 2 * H, (unsigned)(s + 1), da2);
            gran_store(dst + 3 * H, (unsigned)(s + 1), da3);
            grow[0 * H + u] = da0;
            grow[1 * H + u] = da1;
            grow[2 * H + u] = da2;
            grow[3 * H + u] = da3;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

// 0 when the persistent path cannot be used for this shape (caller falls back to the per-step kernels).
size_t ams_blstm_persist_sync_bytes(int B, int H, int backward) {
    const int n_ut = ceil_div(H, TU), n_groups = 2 * ceil_div(B, TB);
    if (n_ut * n_groups > 256) return 0;                    // every workgroup must be resident
    if (H > 320 || H % 4 != 0) return 0;                    // register-resident weight fragments: n_g <= 20 / 80
    const int KP = backward ? ceil_div(4 * H, 16) * 16 : ceil_div(H, 16) * 16;
    return 256 + (size_t)n_groups * 2 * TB * KP * sizeof(u64);
}

// Same contract as ams_blstm_recurrent_fwd/bwd, plus `sync` (ams_blstm_persist_sync_bytes bytes; word 0 = error flag).
ams_status ams_blstm_persist_fwd(float* G, float* out, float* cst, const float* Uf, const float* Ub, long ldu, float* pack, void* sync,
                                 size_t sync_bytes, int B, int T, int H, void* stream) {
    AMS_REQUIRE(G && out && cst && Uf && Ub && pack && sync && B > 0 && T > 0 && H > 0);
    const size_t need = ams_blstm_persist_sync_bytes(B, H, 0);
    AMS_REQUIRE(need != 0);
    if (sync_bytes < need) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    ams_status s0 = ams_blstm_pack(Uf, Ub, ldu, pack, H, 0, stream);
    if (s0 != AMS_OK) return s0;
    if (hipMemsetAsync(sync, 0, need, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;
    PArgs a{};
    a.G = G; a.out = out; a.cst = cst; a.pk = pack; a.err = (unsigned*)sync; a.gran = (u64*)((char*)sync + 256);
    a.B = B; a.T = T; a.H = H; a.n_ut = ceil_div(H, TU); a.n_g = ceil_div(H, 16); a.n_groups = 2 * ceil_div(B, TB);
    a.KP = a.n_g * 16;
    hipLaunchKernelGGL(lstm_persist_fwd_kernel, dim3(a.n_groups * a.n_ut), dim3(256), 0, st, a);
    return ams_check_launch();
}

ams_status ams_blstm_persist_bwd(float* G, const float* cst, const float* dout, const float* Uf, const float* Ub, long ldu, float* pack,
                                 void* sync, size_t sync_bytes, int B, int T, int H, void* stream) {
    AMS_REQUIRE(G && cst && dout && Uf && Ub && pack && sync && B > 0 && T > 0 && H > 0);
    const size_t need = ams_blstm_persist_sync_bytes(B, H, 1);
    AMS_REQUIRE(need != 0);
    if (sync_bytes < need) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    ams_status s0 = ams_blstm_pack(Uf, Ub, ldu, pack, H, 1, stream);
    if (s0 != AMS_OK) return s0;
    if (hipMemsetAsync(sync, 0, need, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;
    PArgs a{};
    a.G = G; a.cst = const_cast<float*>(cst); a.dout = dout; a.pk = pack; a.err = (unsigned*)sync; a.gran = (u64*)((char*)sync + 256);
    a.B = B; a.T = T; a.H = H; a.n_ut = ceil_div(H, TU); a.n_g = ceil_div(4 * H, 16); a.n_groups = 2 * ceil_div(B, TB);
    a.KP = a.n_g * 16;
    hipLaunchKernelGGL(lstm_persist_bwd_kernel, dim3(a.n_groups * a.n_ut), dim3(256), 0, st, a);
    return ams_check_launch();
}

}  // extern "C"
