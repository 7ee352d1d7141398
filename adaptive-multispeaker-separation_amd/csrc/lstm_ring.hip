// BLSTM recurrence as ONE launch per layer and pass: chain-per-XCD rings (reference utils/ops.py:358-383, TF-1.4 BasicLSTMCell).
//
// The per-step kernels of lstm.hip pay a kernel boundary plus an L2-cold re-fetch of the packed recurrent weights (12.7 / 13.5 MB
// per launch, profiles/r01_c_hbm_traffic.txt) 480 times per training step: 5.8 us (forward) / 6.4-7.1 us (backward) per step for
// ~1 us of MFMA work.  Here a CHAIN = (direction, 16-row batch tile) is a ring of NW = ceil(H / 12) workgroups that stay resident
// for all T steps, keep their slice of the recurrent matrix in registers, and exchange the recurrent state in-launch.
//
// Placement.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8; observed, not promised), so chain c takes ids = c mod 8:
// B = 64 gives exactly 8 chains, one per XCD, and every hand-off of a chain goes through ONE L2.  That is what makes the exchange
// cheap (tools/xcd_exchange_bench.hip, 25 workgroups per chain, all 8 chains live):
//     plain 16-byte stores + L1-bypassing (sc1) loads through the shared L2 ....... 1.5 us per step, one hop ~500 cycles
//     write-through (sc1) stores + sc1 loads, placement independent ................. 3.8 us per step
//     plain stores + agent release fence + flag (the generic recipe) ................. 4.9 us per step
// The first form is correct ONLY when producer and consumer share an L2, so placement is VERIFIED, not assumed: at launch every
// workgroup publishes its HW_REG_XCC_ID with an agent-scope atomic, reads its chain's ids back, and the chain uses plain stores
// only if all of them agree; otherwise (or with force_safe) it falls back to write-through stores -- slower, never wrong.
//
// Forward hand-off: the data is the flag.  h_t travels as 16-byte granules {h[3k], h[3k+1], h[3k+2], tag = step + 1}; a consumer
// re-requests its granules (ALL of them, unconditionally -- per-granule "have it" tests make hipcc wait per load) until every
// tag matches.  Three values per granule cost nothing: a v_mfma_f32_16x16x4 k-group is four lanes' worth of k anyway, so the
// k-order of the register-resident weights is simply permuted to (producer, granule, component).  Slots are double-buffered by
// step parity (a writer is at most one step ahead of any reader of its chain) and zeroed by a memset node before each launch.
//
// Backward hand-off: reduce-scatter.  Workgroup w owns 12 units = 48 gate columns of da_t; dh_{t-1} = da_t . U^T needs ALL
// 4H columns, so each workgroup multiplies its own [16 x 48] block with its [48 x H] slice of U^T (registers) and publishes the
// [16 x H] PARTIAL sums as 16x12 tiles addressed to the workgroup that owns those units; a consumer adds NW partials in a fixed
// order (deterministic).  19 KB read + 19 KB written per workgroup and step instead of gathering 77 KB of da.  Tiles are
// published with plain (or write-through) stores, s_waitcnt vmcnt(0), barrier, one flag per producer.
//
// Every wait is bounded: on timeout an error word is set, every ring stops waiting, the (wrong) launch ends, and the host reads
// the word (ams_blstm_ring_error).  Residency: n_chains * NW <= 512 workgroups of 256 threads (<= 2 per CU).
#include "common.h"
#include <stdlib.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int TB = 16;          // batch rows per chain
constexpr int UW = 12;          // hidden units per workgroup = 4 granules of 3
constexpr int MAXR = 7;         // producers per wave (K split over 4 waves): NW <= 28, i.e. H <= 336
constexpr int MAXT = 6;         // backward: 16-unit output tiles per wave: ceil(NW * 12 / 16) <= 24
constexpr int IDS_STRIDE = 32;  // per-chain slots in the id / flag tables (NW <= 28)
constexpr unsigned SPIN_LIMIT = 1u << 20;

struct RingArgs {
    float* G; float* out; float* cst; const float* dout;
    const float* Uf; const float* Ub; long ldu;
    unsigned* err;              // word 0 of the sync buffer
    unsigned* ids;              // [n_chains][IDS_STRIDE]   XCC id + 1 of every workgroup (placement agreement)
    unsigned* flags;            // [n_chains][IDS_STRIDE]   backward: last published step + 1
    float* xbuf;                // forward: granules [n_chains][2][TB][NW*4] float4; backward: partial tiles [n_chains][2][NW][NW][UW*TB]
    int B, T, H, NW, n_chains, force_safe;
};

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// aux 16 = sc1: the load bypasses this CU's L1 and is served by the L2 (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ float4 ld16_l2(rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}
__device__ __forceinline__ float ld4_l2(rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 16));
}
__device__ __forceinline__ void st16(rsrc_t rs, float* base, unsigned byte_off, float4 v, bool fast) {
    if (fast) *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;       // stays in this XCD's L2
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rs, byte_off, 0, 16);   // write-through
}

// returns true when the wait must be abandoned (timeout here, or another ring already gave up)
__device__ __forceinline__ bool spin_check(unsigned& spins, unsigned* err) {
    ++spins;
    if ((spins & 255u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (spins >= SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
    }
    return false;
}

// block -> (chain, member); chain c = 8 * cgrp + x lives on ids = x mod 8.  false: surplus block of a partly filled chain group.
__device__ __forceinline__ bool ring_coords(const RingArgs& a, int& chain, int& w) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    chain = (j / a.NW) * 8 + x;
    w = j % a.NW;
    return chain < a.n_chains;
}

// Placement agreement: true when every workgroup of this chain reports the same XCC id (plain-store hand-off is then valid).
__device__ __forceinline__ bool chain_shares_l2(const RingArgs& a, int chain, int w, int* lds_flag, bool& abort) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        __hip_atomic_store(a.ids + chain * IDS_STRIDE + w, (id & 0xfu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 64) {
        unsigned v = 1, spins = 0;
        bool dead = false;
        for (;;) {
            v = (tid < a.NW) ? __hip_atomic_load(a.ids + chain * IDS_STRIDE + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            if (__all(v != 0u)) break;
            if (spin_check(spins, a.err)) { dead = true; break; }
        }
        const unsigned first = __shfl(v, 0, 64);
        const bool same = __all(tid >= a.NW || v == first);
        if (tid == 0) *lds_flag = (same ? 1 : 0) | (dead ? 2 : 0);
    }
    __syncthreads();
    const int f = *lds_flag;
    abort = (f & 2) != 0;
    return (f & 1) != 0 && !a.force_safe;
}

// ------------------------------------------------------------------------------------------------------ forward
// NR = ceil(NW / 4) producer rounds per wave, a COMPILE-TIME bound: the MFMA chain is then straight-line code (rounds past a wave's
// last producer multiply clamped granules with zero weights); a run-time `wave + 4 i < NW` test per round costs a branch and
// accumulator copies per MFMA.
template <int NR>
__global__ __launch_bounds__(256) void lstm_ring_fwd_kernel(RingArgs a) {
    __shared__ __attribute__((aligned(16))) float red[2][4][3][64][4];     // [step parity][wave][column tile][lane][reg]
    __shared__ int lds_flag;
    int chain, w;
    if (!ring_coords(a, chain, w)) return;
    __builtin_amdgcn_s_setprio(3);
    // `wave` must be KNOWN uniform: with tid >> 6 hipcc treats every `wave + 4 i < NW` test as divergent and wraps each MFMA in an
    // exec-mask branch with accumulator copies around it (measured on the backward kernel: 12.9 us per step instead of 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = chain & 1, bt = chain >> 1;
    const int H = a.H, T = a.T, NW = a.NW, NG = NW * 4;
    const float* U = dir ? a.Ub : a.Uf;
    bool abort = false;
    const bool fast = chain_shares_l2(a, chain, w, &lds_flag, abort);

    // recurrent weights of this workgroup's 48 gate columns (column c = gate * 12 + local unit), k in (producer, granule, component)
    // order, as v_mfma_f32_16x16x4 B fragments: lane (n = lane & 15, q = lane >> 4) holds k = 12 r + 3 q + j for column tile t
    const int n16 = lane & 15, q = lane >> 4;
    float bw[NR][3][3];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = 12 * r + 3 * q + j;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int c = t * 16 + n16, gate = c / UW, unit = w * UW + c % UW;
                bw[i][j][t] = (r < NW && k < H && unit < H) ? U[(long)k * a.ldu + gate * H + unit] : 0.f;
            }
        }
    }

    // epilogue element of this thread: (row, local unit); 16 threads per row, 12 of them live
    const int row = tid >> 4, ul = tid & 15;
    const int b = bt * TB + row, u = w * UW + ul;
    const bool live = (ul < UW && b < a.B && u < H);
    // accumulator element (row, c): tile = c >> 4, lane (row >> 2) * 16 + (c & 15), reg row & 3
    int src_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = g * UW + (ul < UW ? ul : 0);
        src_off[g] = ((c >> 4) * 64 + (row >> 2) * 16 + (c & 15)) * 4 + (row & 3);
    }
    float* xb = a.xbuf + (size_t)chain * 2 * TB * NG * 4;
    const rsrc_t rs = make_rsrc(xb, (unsigned)((size_t)2 * TB * NG * 16));
    const int rowl = lane & 15;
    float c_state = 0.f;

    auto gaddr = [&](int t) { return a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H) + (live ? u : 0); };
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* g0 = gaddr(dir ? T - 1 : 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) zq[g] = g0[g * H];
    }

    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        const int par = s & 1;
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (s > 0) {
            float4 hv[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned spins = 0;
            const unsigned want = (unsigned)s;                  // h_{s-1} carries tag s
            const unsigned rbase = (unsigned)(((par ^ 1) * TB + rowl) * NG) * 16u;
            if (!abort) {
                for (;;) {
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        const int r = min(wave + 4 * i, NW - 1);
                        hv[i] = ld16_l2(rs, rbase + (unsigned)(r * 4 + q) * 16u);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // all requests in flight before the first tag is looked at
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < NR; ++i) ok &= (__float_as_uint(hv[i].w) == want);
                    if (__all(ok)) break;
                    if (spin_check(spins, a.err)) { abort = true; break; }
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i].x, bw[i][0][t3], acc[t3], 0, 0, 0);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i].y, bw[i][1][t3], acc[t3], 0, 0, 0);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) acc[t3] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i].z, bw[i][2][t3], acc[t3], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) *reinterpret_cast<f32x4*>(&red[par][wave][t3][lane][0]) = acc[t3];
        __syncthreads();

        float pre[4];
        const float* rp = &red[par][0][0][0][0];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = zq[g];
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) v += rp[wv * (3 * 64 * 4) + src_off[g]];
            pre[g] = v;
        }
        const float ig = 1.0f / (1.0f + expf(-pre[0]));
        const float gg = tanhf(pre[1]);
        const float fg = 1.0f / (1.0f + expf(-(pre[2] + 1.0f)));     // forget_bias = 1.0
        const float og = 1.0f / (1.0f + expf(-pre[3]));
        const float c = c_state * fg + ig * gg;
        const float h = live ? tanhf(c) * og : 0.f;                   // dead rows / units publish zeros
        c_state = c;
        // hand h_s to the ring first: lanes ul = 0, 3, 6, 9 assemble {h[ul], h[ul+1], h[ul+2], tag}
        const float h1 = __shfl_down(h, 1, 64), h2 = __shfl_down(h, 2, 64);
        if (ul < UW && ul % 3 == 0)
            st16(rs, xb, (unsigned)(((par * TB + row) * NG) + w * 4 + ul / 3) * 16u, make_float4(h, h1, h2, __uint_as_float((unsigned)(s + 1))), fast);
        if (live) {
            float* gr = gaddr(t);
            gr[0 * H] = ig;
            gr[1 * H] = gg;
            gr[2 * H] = fg;
            gr[3 * H] = og;
            a.cst[(((long)b * T + t) * 2 + dir) * H + u] = c;
            a.out[((long)b * T + t) * (2 * H) + dir * H + u] = h;
            if (s + 1 < T) {                                           // next step's pre-activations: in flight during the wait
                const float* gn = gaddr(dir ? t - 1 : t + 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) zq[g] = gn[g * H];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------- backward
// NI = ceil(NT / 4) output-tile rounds per wave (compile-time, as NR above); NP = 4 * ceil(NW / 4) partial tiles summed per element.
template <int NI, int NP>
__global__ __launch_bounds__(256) void lstm_ring_bwd_kernel(RingArgs a) {
    __shared__ __attribute__((aligned(16))) float lds_a[TB][4 * UW + 1];    // da of this workgroup: [row][gate * 12 + local unit]
    __shared__ int lds_flag;
    int chain, w;
    if (!ring_coords(a, chain, w)) return;
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = chain & 1, bt = chain >> 1;
    const int H = a.H, T = a.T, NW = a.NW;
    const int NT = (NW * UW + 15) / 16;                         // 16-unit output tiles covering the chain's NW * 12 units
    const float* U = dir ? a.Ub : a.Uf;
    bool abort = false;
    const bool fast = chain_shares_l2(a, chain, w, &lds_flag, abort);

    // U^T slice as B fragments: k = own gate column (gate * 12 + local unit), n = unit of the output tile
    const int n16 = lane & 15, q = lane >> 4;
    float bw[NI][UW];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int tl = wave + 4 * i, unit_row = tl * 16 + n16;
#pragma unroll
        for (int kk = 0; kk < UW; ++kk) {
            const int k = 4 * kk + q, gate = k / UW, ucol = w * UW + k % UW;
            bw[i][kk] = (tl < NT && unit_row < H && ucol < H) ? U[(long)unit_row * a.ldu + gate * H + ucol] : 0.f;
        }
    }

    // element of this thread: (local unit n, row r) -- the order the partial tiles are stored in
    const int nl = tid >> 4, r = tid & 15;
    const int b = bt * TB + r, u = w * UW + nl;
    const bool live = (nl < UW && b < a.B && u < H);
    const size_t tile_f = (size_t)UW * TB;                      // floats per partial tile
    float* pb = a.xbuf + (size_t)chain * 2 * NW * NW * tile_f;
    const rsrc_t rs = make_rsrc(pb, (unsigned)((size_t)2 * NW * NW * tile_f * 4));
    unsigned* fl = a.flags + chain * IDS_STRIDE;
    const rsrc_t rf = make_rsrc(fl, IDS_STRIDE * 4);
    float dc_state = 0.f;

    auto gaddr = [&](int t) { return a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H) + (live ? u : 0); };
    float dh = 0.f, ig = 0.f, gg = 0.f, fg = 0.f, og = 0.f, c = 0.f, c_prev = 0.f;
    auto fetch = [&](int t) {
        const int tp = dir ? t + 1 : t - 1;
        dh = a.dout[((long)b * T + t) * (2 * H) + dir * H + u];
        const float* gr = gaddr(t);
        ig = gr[0 * H]; gg = gr[1 * H]; fg = gr[2 * H]; og = gr[3 * H];
        c = a.cst[(((long)b * T + t) * 2 + dir) * H + u];
        c_prev = (tp >= 0 && tp < T) ? a.cst[(((long)b * T + tp) * 2 + dir) * H + u] : 0.f;
    };
    if (live) fetch(dir ? 0 : T - 1);

    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);
        const int par = s & 1;
        if (s > 0 && !abort) {
            unsigned spins = 0;
            for (;;) {                                          // every wave watches its chain's flags itself: no extra barrier
                const unsigned v = (lane < NW) ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rf, lane * 4, 0, 16) : 0xffffffffu;
                if (__all(v >= (unsigned)s)) break;
                if (spin_check(spins, a.err)) { abort = true; break; }
            }
        }
        float part[NP];
        if (s > 0) {
            const unsigned base = (unsigned)((((size_t)(par ^ 1) * NW + w) * NW) * tile_f + (nl < UW ? nl : 0) * TB + r) * 4u;
#pragma unroll
            for (int p = 0; p < NP; ++p) part[p] = ld4_l2(rs, base + (unsigned)(min(p, NW - 1) * tile_f) * 4u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (p < NW) dh += part[p];                       // scalar (uniform) test; fixed order: deterministic
        }
        const float tc = tanhf(c);
        const float d_o = dh * tc;
        const float dcv = dc_state + dh * og * (1.0f - tc * tc);
        const float da0 = live ? dcv * gg * ig * (1.0f - ig) : 0.f;
        const float da1 = live ? dcv * ig * (1.0f - gg * gg) : 0.f;
        const float da2 = live ? dcv * c_prev * fg * (1.0f - fg) : 0.f;
        const float da3 = live ? d_o * og * (1.0f - og) : 0.f;
        dc_state = dcv * fg;
        if (nl < UW) {
            lds_a[r][0 * UW + nl] = da0;
            lds_a[r][1 * UW + nl] = da1;
            lds_a[r][2 * UW + nl] = da2;
            lds_a[r][3 * UW + nl] = da3;
        }
        __syncthreads();
        if (s + 1 < T) {
            // partial dh_{next} = da_own [16 x 48] . U^T slice [48 x NT*16]
            f32x4 acc[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* arow = &lds_a[lane & 15][q];
#pragma unroll
            for (int kk = 0; kk < UW; ++kk) {
                const float av = arow[4 * kk];
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[i][kk], acc[i], 0, 0, 0);
            }
            // tile column n16 of output tile tl = unit tl*16 + n16 -> consumer unit / 12, slot unit % 12; rows 4q..4q+3 contiguous
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int unit = (wave + 4 * i) * 16 + n16;
                if (unit < NW * UW) {
                    const int cw = unit / UW, slot = unit % UW;
                    const unsigned off = (unsigned)((((size_t)par * NW + cw) * NW + w) * tile_f + slot * TB + 4 * q) * 4u;
                    st16(rs, pb, off, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]), fast);
                }
            }
        }
        if (live) {
            float* gr = gaddr(t);
            gr[0 * H] = da0;
            gr[1 * H] = da1;
            gr[2 * H] = da2;
            gr[3 * H] = da3;
        }
        if (s + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every tile of this wave has reached the L2 (or memory, write-through)
            __syncthreads();                                    // ... of every wave; also: lds_a may be rewritten from here on
            if (tid == 0) {
                if (fast) fl[w] = (unsigned)(s + 1);
                else __hip_atomic_store(fl + w, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (live) fetch(dir ? t + 1 : t - 1);               // next step's operands: in flight during the wait
        }
    }
}

inline bool ring_shape(int B, int H, int& NW, int& n_chains) {
    NW = ceil_div(H, UW);
    n_chains = 2 * ceil_div(B, TB);
    return NW <= 4 * MAXR && ceil_div(NW * UW, 16) <= 4 * MAXT && (long)n_chains * NW <= 512;
}

struct RingLayout { size_t ids, flags, x, total, head; };

inline RingLayout ring_layout(int NW, int n_chains, int backward) {
    RingLayout L;
    L.ids = 256;
    L.flags = L.ids + (size_t)n_chains * IDS_STRIDE * 4;
    L.head = L.flags + (size_t)n_chains * IDS_STRIDE * 4;      // [0, head): zeroed before every launch
    L.x = (L.head + 255) & ~(size_t)255;
    const size_t xbytes = backward ? (size_t)n_chains * 2 * NW * NW * UW * TB * 4 : (size_t)n_chains * 2 * TB * NW * 4 * 16;
    L.total = L.x + xbytes;
    if (!backward) L.head = L.total;                            // forward: the granule tags must start at 0 as well
    return L;
}

inline int ring_force_safe() {
    static const int v = getenv("AMS_LSTM_RING_SAFE") ? atoi(getenv("AMS_LSTM_RING_SAFE")) : 0;     // read once; testing aid
    return v;
}

}  // namespace

extern "C" {

// 0 when the ring path cannot be used for this shape (caller falls back to the per-step kernels of lstm.hip).
size_t ams_blstm_ring_sync_bytes(int B, int H, int backward) {
    int NW, n_chains;
    if (B <= 0 || H <= 0 || !ring_shape(B, H, NW, n_chains)) return 0;
    return ring_layout(NW, n_chains, backward).total;
}

// Same contract as ams_blstm_recurrent_fwd (G: pre-activations in, activated gates out; out; cst), plus `sync`
// (ams_blstm_ring_sync_bytes(B, H, 0) bytes, word 0 = timeout flag).  safe != 0 forces the placement-independent hand-off.
ams_status ams_blstm_ring_fwd(float* G, float* out, float* cst, const float* Uf, const float* Ub, long ldu, void* sync, size_t sync_bytes,
                              int B, int T, int H, int safe, void* stream) {
    AMS_REQUIRE(G && out && cst && Uf && Ub && sync && B > 0 && T > 0 && H > 0);
    int NW, n_chains;
    AMS_REQUIRE(ring_shape(B, H, NW, n_chains));
    const RingLayout L = ring_layout(NW, n_chains, 0);
    if (sync_bytes < L.total) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, L.head, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;
    RingArgs a{};
    a.G = G; a.out = out; a.cst = cst; a.Uf = Uf; a.Ub = Ub; a.ldu = ldu;
    a.err = (unsigned*)sync; a.ids = (unsigned*)((char*)sync + L.ids); a.flags = (unsigned*)((char*)sync + L.flags);
    a.xbuf = (float*)((char*)sync + L.x);
    a.B = B; a.T = T; a.H = H; a.NW = NW; a.n_chains = n_chains; a.force_safe = (safe || ring_force_safe()) ? 1 : 0;
    const dim3 grid(8 * NW * ceil_div(n_chains, 8));
    switch (ceil_div(NW, 4)) {
        case 1: hipLaunchKernelGGL(lstm_ring_fwd_kernel<1>, grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(lstm_ring_fwd_kernel<2>, grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(lstm_ring_fwd_kernel<3>, grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(lstm_ring_fwd_kernel<4>, grid, dim3(256), 0, st, a); break;
        case 5: hipLaunchKernelGGL(lstm_ring_fwd_kernel<5>, grid, dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL(lstm_ring_fwd_kernel<6>, grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(lstm_ring_fwd_kernel<7>, grid, dim3(256), 0, st, a); break;
    }
    return ams_check_launch();
}

// Same contract as ams_blstm_recurrent_bwd (on return G holds da), without the dc workspace (the running dc lives in registers).
ams_status ams_blstm_ring_bwd(float* G, const float* cst, const float* dout, const float* Uf, const float* Ub, long ldu, void* sync,
                              size_t sync_bytes, int B, int T, int H, int safe, void* stream) {
    AMS_REQUIRE(G && cst && dout && Uf && Ub && sync && B > 0 && T > 0 && H > 0);
    int NW, n_chains;
    AMS_REQUIRE(ring_shape(B, H, NW, n_chains));
    const RingLayout L = ring_layout(NW, n_chains, 1);
    if (sync_bytes < L.total) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sync, 0, L.head, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;
    RingArgs a{};
    a.G = G; a.cst = const_cast<float*>(cst); a.dout = dout; a.Uf = Uf; a.Ub = Ub; a.ldu = ldu;
    a.err = (unsigned*)sync; a.ids = (unsigned*)((char*)sync + L.ids); a.flags = (unsigned*)((char*)sync + L.flags);
    a.xbuf = (float*)((char*)sync + L.x);
    a.B = B; a.T = T; a.H = H; a.NW = NW; a.n_chains = n_chains; a.force_safe = (safe || ring_force_safe()) ? 1 : 0;
    const dim3 grid(8 * NW * ceil_div(n_chains, 8));
    // NW = 1..28 -> NT = ceil(12 NW / 16) = 1..21 -> NI = ceil(NT / 4) = 1..6; partials summed: NP = 4 * ceil(NW / 4) >= NW
    const int NI = ceil_div(ceil_div(NW * UW, 16), 4);
    if (NW <= 4)       hipLaunchKernelGGL((lstm_ring_bwd_kernel<1, 4>), grid, dim3(256), 0, st, a);       // NT <= 3
    else if (NI <= 2)  hipLaunchKernelGGL((lstm_ring_bwd_kernel<2, 12>), grid, dim3(256), 0, st, a);      // NW <= 10
    else if (NI <= 3)  hipLaunchKernelGGL((lstm_ring_bwd_kernel<3, 16>), grid, dim3(256), 0, st, a);      // NW <= 16
    else if (NI <= 4)  hipLaunchKernelGGL((lstm_ring_bwd_kernel<4, 24>), grid, dim3(256), 0, st, a);      // NW <= 21
    else if (NI <= 5)  hipLaunchKernelGGL((lstm_ring_bwd_kernel<5, 28>), grid, dim3(256), 0, st, a);      // NW <= 26
    else               hipLaunchKernelGGL((lstm_ring_bwd_kernel<6, 28>), grid, dim3(256), 0, st, a);      // NW <= 28
    return ams_check_launch();
}

}  // extern "C"
