// BLSTM recurrence as ONE launch per layer and pass: chain-per-XCD rings (reference utils/ops.py:358-383, TF-1.4 BasicLSTMCell).
//
// The per-step kernels of lstm.hip pay a kernel boundary plus an L2-cold re-fetch of the packed recurrent weights (12.7 / 13.5 MB
// per launch, profiles/r01_c_hbm_traffic.txt) 480 times per training step: 5.8 us (forward) / 6.4-7.1 us (backward) per step for
// ~1 us of MFMA work.  Here a CHAIN = (direction, 16-row batch tile) is a ring of NW = ceil(H / 12) workgroups that stay resident
// for all T steps, keep their slice of the recurrent matrix in registers, and exchange the recurrent state in-launch.
//
// Placement.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8; observed, not promised), so chain c takes ids = c mod 8:
// B = 64 gives exactly 8 chains, one per XCD, and every hand-off of a chain goes through ONE L2.  That is what makes the exchange
// cheap (tools/probes/xcd_exchange_bench.hip, 25 workgroups per chain, all 8 chains live):
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

// tuning aids (A/B builds through AMS_HIP_LIB): where the backward kernel requests the next step's operands
//   0 = right behind this step's waits (ahead of the MFMA chain)   1 = behind the flag store (in flight during the next wait)
#ifndef AMS_RING_FETCH_LATE
#define AMS_RING_FETCH_LATE 0
#endif
#ifndef AMS_RING_FWD_FETCH_LATE
#define AMS_RING_FWD_FETCH_LATE 0
#endif
// s_sleep units (64 cycles each) before the FIRST poll round of a step: every member of a chain publishes at about the same time,
// a round issued right behind the own publish returns mostly stale granules and costs a full extra round
#ifndef AMS_RING_FWD_SLEEP
#define AMS_RING_FWD_SLEEP 0
#endif
#ifndef AMS_RING_FWD_STORE_LATE
#define AMS_RING_FWD_STORE_LATE 1      // 1: the stores only the backward pass needs are issued one step later, right behind the NEXT wait
#endif
#ifndef AMS_RING_BWD_SLEEP
#define AMS_RING_BWD_SLEEP 0
#endif
#ifndef AMS_RING_X6_DEFAULT
#define AMS_RING_X6_DEFAULT 1
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

constexpr int TB = 16;          // batch rows per chain
constexpr int UW = 12;          // hidden units per workgroup = 4 granules of 3
constexpr int MAXR = 7;         // producers per wave (K split over 4 waves): NW <= 28, i.e. H <= 336
constexpr int MAXT = 6;         // backward: 16-unit output tiles per wave: ceil(NW * 12 / 16) <= 24
constexpr int IDS_STRIDE = 32;  // per-chain slots in the id / flag tables (NW <= 28)
constexpr unsigned SPIN_LIMIT = 1u << 20;

constexpr int RING_AMAX_WORD = 2;     // sync head: word 0 = error word, word 2 = max |da| of a backward launch, bytes 64.. = trace
struct RingArgs {
    float* G; float* out; float* cst; float* tch; const float* dout;   // tch [B,T,2,H] = tanh(c_t), written by the forward ring
    float* dbpart;              // backward, optional: [B,2,4H] = sum over t of da[b,t,dir,:] (bias-gradient partials)
    const float* Uf; const float* Ub; long ldu;
    unsigned* err;              // word 0 of the sync buffer
    unsigned* sticky;           // optional: a word the CALLER owns and never zeroes per launch -- set whenever err is set, so that one
                                // check at the caller's next host sync covers every ring launch since its last check
    unsigned* ids;              // [n_chains][IDS_STRIDE]   XCC id + 1 of every workgroup (placement agreement)
    float* xbuf;                // forward: granules [n_chains][2][TB][NW*4] float4; backward: partial tiles [n_chains][2][NW][NW][UW*TB]
    int B, T, H, NW, n_chains, force_safe, trace;
    const float* amax_u;        // forward, fp16x3: device pointer to an upper bound of max |U| over both recurrent kernels
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

// Gate non-linearities of the ring epilogue: ~1 ulp like libm's, without its branches and special-case paths (the epilogue is on
// the hand-off cycle: 1300 -> ~800 cycles per step).  e^x = 2^(x log2 e) with the product split as n + r, |r| <= 1/2, r carried
// with the low word of log2(e) so that v_exp_f32 sees an argument good to ~2^-30; scaling by v_ldexp_f32.  tanh: an odd minimax
// polynomial (degree 11, max rel. error 1.35e-7) below 0.55, 1 - 2 / (1 + e^{2|x|}) above (abs. error ~1e-7 where tanh > 0.5).
// NOT __expf: its single-product argument is off by |x| * 2^-24 relative, which the 80-step recurrence amplifies.
__device__ __forceinline__ float ring_exp(float x) {
    x = fminf(fmaxf(x, -87.3f), 88.7f);
    const float n = rintf(x * 1.4426950408889634f);
    float r = fmaf(x, 1.4426950408889634f, -n);
    r = fmaf(x, 1.925963033500180e-8f, r);
    return ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
}
// 1 / d for d in [1, 2^127]: v_rcp_f32 (1 ulp) + one Newton step (~0.5 ulp) instead of the IEEE division sequence (scale / fmas /
// fixup, ~12 instructions) -- five of them per element sat on the hand-off cycle
__device__ __forceinline__ float ring_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.0f), r);
}
__device__ __forceinline__ float ring_sigmoid(float x) { return ring_rcp(1.0f + ring_exp(-x)); }
__device__ __forceinline__ float ring_tanh(float x) {
    const float ax = fabsf(x), y = x * x;
    float p = -0.006332100369036198f;
    p = fmaf(p, y, 0.02110716514289379f);
    p = fmaf(p, y, -0.053859058767557144f);
    p = fmaf(p, y, 0.1333262026309967f);
    p = fmaf(p, y, -0.33333316445350647f);
    p = fmaf(p, y, 1.0f);
    const float big = 1.0f - 2.0f * ring_rcp(1.0f + ring_exp(2.0f * ax));
    return ax < 0.55f ? x * p : copysignf(big, x);
}

// returns true when the wait must be abandoned (timeout here, or another ring already gave up)
__device__ __forceinline__ bool spin_check(unsigned& spins, unsigned* err, unsigned* sticky) {
    ++spins;
    if ((spins & 255u) == 0u) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (spins >= SPIN_LIMIT) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

// Phase anatomy (tools/ring_anatomy.py): when `trace` is set, thread 0 of (chain 0, member 0) accumulates shader-clock cycles per
// phase into words 8.. of the sync header.  One uniform branch per stamp; off in production launches.
struct Trace {
    unsigned long long* out; unsigned long long last; unsigned acc[8]; bool on;
    __device__ __forceinline__ void begin(unsigned long long* p, bool enable) {
        out = p; on = enable;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0;
        if (on) last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void stamp(int slot) {             // slot is a literal: acc[] stays in registers
        if (on) { const unsigned long long now = __builtin_readcyclecounter(); acc[slot] += (unsigned)(now - last); last = now; }
    }
    __device__ __forceinline__ void end() {
        if (on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) out[i] = acc[i];
        }
    }
};

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
            if (spin_check(spins, a.err, a.sticky)) { dead = true; break; }
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
// X6 (round 3): the recurrent product h_{t-1} . U on the bf16 pipe as f32 arithmetic -- exact three-way bf16 split of both operands, six
// partial products, the five small ones in their own accumulator (csrc/gemm.hip, SEP) -- on v_mfma_f32_16x16x32_bf16: 54 MFMAs of
// ~17 cycles instead of 63 of 32 per wave and step.  U's slice lives in registers as three bf16 images (108 VGPRs instead of 63);
// h_{t-1} is split on arrival: every lane already holds exactly the 8 k-slots per MFMA its A fragment needs, because an MFMA's k
// order is free as long as A and B agree (slot e = 3 i + j of lane group q <-> k = 12 r_i + 3 q + j on both sides).
typedef __bf16 rbf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 rbf16x2_t __attribute__((ext_vector_type(2)));
typedef float rf32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned ring_pk_bf16(float a, float b) {
    const rf32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rbf16x2_t));
}
__device__ __forceinline__ void ring_split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = ring_pk_bf16(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = ring_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = ring_pk_bf16(sa, sb);
}

typedef _Float16 rf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 rf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 rf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ring_split2h(float a, float b, unsigned& hi, unsigned& mid) {
    const rf32x2_t v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, rf16x2_t));
    const rf16x2_t h = __builtin_bit_cast(rf16x2_t, hi);
    const rf32x2_t r = {a - (float)h[0], b - (float)h[1]};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rf16x2_t));
}
// One value as the two fp16 terms of ring_split2h in ONE dword, hi | mid << 16: what the PRODUCER of h_t publishes in the fp16x3 forward
// ring (round 5).  A consumer then builds its MFMA operand planes with two v_perm_b32 per pair of k-slots; splitting on arrival was two
// conversions back, two subtractions and two packed conversions per pair in each of the 25 consumers' four waves, on the hand-off cycle.
__device__ __forceinline__ unsigned ring_pack_hm(float v) {
    const _Float16 hi = (_Float16)v;
    const _Float16 mid = (_Float16)(v - (float)hi);
    return (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, mid) << 16);
}
// 2^(13 - floor(log2(amax))); 1 for 0, denormals, Inf, NaN (csrc/gemm.hip::f16_scale)
__device__ __forceinline__ float ring_f16_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    const int se = 127 + 13 - (e - 127);
    return (se >= 1 && se <= 254) ? __uint_as_float((unsigned)se << 23) : 1.0f;
}

// the same scale and its exact inverse (both powers of two; (1, 1) where ring_f16_scale gives 1)
__device__ __forceinline__ void ring_f16_scale2(float amax, float& sc, float& inv) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    const int se = 127 + 13 - (e - 127);
    const bool ok = (e != 0 && e != 255 && se >= 1 && se <= 253);
    sc = ok ? __uint_as_float((unsigned)se << 23) : 1.0f;
    inv = ok ? __uint_as_float((unsigned)(254 - se) << 23) : 1.0f;
}

template <int NR, int ARITH = 0>      // 0: v_mfma_f32_16x16x4_f32, 1: bf16x6, 2: fp16x3 (a.amax_u = bound of the recurrent kernels)
__global__ __launch_bounds__(256) void lstm_ring_fwd_kernel(RingArgs a) {
    constexpr bool X6 = (ARITH == 1), F16 = (ARITH == 2);
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

    // X6: the same weights as three bf16 images, 8 k-slots per lane and MFMA: slot e = 3 i + j, MFMA m = e / 8 (zero beyond 3 NR)
    constexpr int NM = (X6 || F16) ? (3 * NR + 7) / 8 : 1;
    rbf16x8_t bq[NM][3][F16 ? 2 : 3];                              // [MFMA][column tile][plane hi / mid / lo]
    // fp16x3 (csrc/gemm.hip): U scaled by a power of two from its bound and split in two fp16 terms; h_{t-1} (|h| < 1) scaled by 2^13
    float sc_u = 1.0f, sc_inv = 1.0f;
    if constexpr (F16) {
        sc_u = ring_f16_scale(a.amax_u[0]);
        sc_inv = (1.0f / sc_u) * (1.0f / 8192.0f);
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                unsigned hi[4], mid[4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int e0 = 8 * m + 2 * pr, e1 = e0 + 1;
                    const float v0 = e0 < 3 * NR ? bw[e0 / 3][e0 % 3][t] : 0.f, v1 = e1 < 3 * NR ? bw[e1 / 3][e1 % 3][t] : 0.f;
                    ring_split2h(v0 * sc_u, v1 * sc_u, hi[pr], mid[pr]);
                }
                const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]};
                bq[m][t][0] = __builtin_bit_cast(rbf16x8_t, h4);
                bq[m][t][1] = __builtin_bit_cast(rbf16x8_t, m4);
            }
    }
    if constexpr (X6) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                unsigned hi[4], mid[4], lo[4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int e0 = 8 * m + 2 * pr, e1 = e0 + 1;
                    const float v0 = e0 < 3 * NR ? bw[e0 / 3][e0 % 3][t] : 0.f, v1 = e1 < 3 * NR ? bw[e1 / 3][e1 % 3][t] : 0.f;
                    ring_split3(v0, v1, hi[pr], mid[pr], lo[pr]);
                }
                const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]}, l4 = {lo[0], lo[1], lo[2], lo[3]};
                bq[m][t][0] = __builtin_bit_cast(rbf16x8_t, h4);
                bq[m][t][1] = __builtin_bit_cast(rbf16x8_t, m4);
                bq[m][t][2] = __builtin_bit_cast(rbf16x8_t, l4);
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

    // per-thread base pointers (time 0) and per-time-step strides, computed once: dead threads alias element (0, 0)
    const long bl_ = live ? b : 0, ul_ = live ? u : 0;
    float* const gp = a.G + ((bl_ * T) * 2 + dir) * (4 * H) + ul_;            // + t * 8H + gate * H
    float* const cp = a.cst + ((bl_ * T) * 2 + dir) * H + ul_;                // + t * 2H
    float* const tp_ = a.tch + ((bl_ * T) * 2 + dir) * H + ul_;               // + t * 2H
    float* const op = a.out + (bl_ * T) * (2 * H) + dir * H + ul_;            // + t * 2H
    const int gst = 8 * H, cst_st = 2 * H;
    float zq[4];
    {
        const float* g0 = gp + (dir ? T - 1 : 0) * gst;
#pragma unroll
        for (int g = 0; g < 4; ++g) zq[g] = g0[g * H];
    }

    Trace tr;
    tr.begin(reinterpret_cast<unsigned long long*>(a.err) + 8, a.trace && chain == 0 && w == 0 && tid == 0);
#if AMS_RING_FWD_STORE_LATE
    float pend_v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int pend_t = -1;
#endif
    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        const int par = s & 1;
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float zn[4];
        // Next step's pre-activations are requested AFTER this step's wait has ended and before its MFMA chain: vmcnt retires in
        // issue order, so a load issued ahead of the poll would put its own (HBM / MALL) latency in front of every poll round.
        // UNCONDITIONAL loads on clamped addresses (dead threads read element (0, 0), the last step re-reads its own row): a load
        // inside a branch is waited for at the join (s_waitcnt vmcnt(0) in front of the MFMA chain).
        {
            float4 hv[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s > 0 && !abort) {
                unsigned spins = 0;
                const unsigned want = (unsigned)s;              // h_{s-1} carries tag s
                const unsigned rbase = (unsigned)(((par ^ 1) * TB + rowl) * NG) * 16u;
                if (AMS_RING_FWD_SLEEP) __builtin_amdgcn_s_sleep(AMS_RING_FWD_SLEEP);
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
                    if (spin_check(spins, a.err, a.sticky)) { abort = true; break; }
                }
            }
            tr.stamp(0);                                        // wait for h_{s-1}
#if AMS_RING_FWD_STORE_LATE
            if (live && pend_t >= 0) {
                float* gr = gp + pend_t * gst;
                gr[0 * H] = pend_v[0]; gr[1 * H] = pend_v[1]; gr[2 * H] = pend_v[2]; gr[3 * H] = pend_v[3];
                cp[pend_t * cst_st] = pend_v[4]; tp_[pend_t * cst_st] = pend_v[5]; op[pend_t * cst_st] = pend_v[6];
            }
#endif
#if !AMS_RING_FWD_FETCH_LATE
            {
                const float* gn = gp + min(max(dir ? t - 1 : t + 1, 0), T - 1) * gst;
#pragma unroll
                for (int g = 0; g < 4; ++g) zn[g] = gn[g * H];
            }
#endif
            if (s > 0) {
                if constexpr (F16) {
                    f32x4 accs[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    rf16x8_t aq[NM][2];
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        unsigned hi[4], mid[4];
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) {
                            // the granules carry hi | mid << 16 per value (ring_pack_hm at the producer): plane dwords by byte permutes
                            const int e0 = 8 * m + 2 * pr, e1 = e0 + 1;
                            const unsigned p0 = e0 < 3 * NR ? __float_as_uint(e0 % 3 == 0 ? hv[e0 / 3].x : e0 % 3 == 1 ? hv[e0 / 3].y : hv[e0 / 3].z) : 0u;
                            const unsigned p1 = e1 < 3 * NR ? __float_as_uint(e1 % 3 == 0 ? hv[e1 / 3].x : e1 % 3 == 1 ? hv[e1 / 3].y : hv[e1 / 3].z) : 0u;
                            hi[pr] = __builtin_amdgcn_perm(p1, p0, 0x05040100u);
                            mid[pr] = __builtin_amdgcn_perm(p1, p0, 0x07060302u);
                        }
                        const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]};
                        aq[m][0] = __builtin_bit_cast(rf16x8_t, h4);
                        aq[m][1] = __builtin_bit_cast(rf16x8_t, m4);
                    }
                    constexpr int PA[3] = {1, 0, 0};                    // lo.hi, hi.lo | hi.hi: the cross terms in their own accumulator
                    constexpr int PB[3] = {0, 1, 0};
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                        for (int m = 0; m < NM; ++m)
#pragma unroll
                            for (int t3 = 0; t3 < 3; ++t3) {
                                const rf16x8_t fb = __builtin_bit_cast(rf16x8_t, bq[m][t3][PB[pp]]);
                                if (pp < 2) accs[t3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[m][PA[pp]], fb, accs[t3], 0, 0, 0);
                                else acc[t3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[m][PA[pp]], fb, acc[t3], 0, 0, 0);
                            }
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) acc[t3] = (acc[t3] + accs[t3]) * sc_inv;
                } else if constexpr (X6) {
                    f32x4 accs[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                    rbf16x8_t aq[NM][3];
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        unsigned hi[4], mid[4], lo[4];
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) {
                            const int e0 = 8 * m + 2 * pr, e1 = e0 + 1;
                            const float v0 = e0 < 3 * NR ? (e0 % 3 == 0 ? hv[e0 / 3].x : e0 % 3 == 1 ? hv[e0 / 3].y : hv[e0 / 3].z) : 0.f;
                            const float v1 = e1 < 3 * NR ? (e1 % 3 == 0 ? hv[e1 / 3].x : e1 % 3 == 1 ? hv[e1 / 3].y : hv[e1 / 3].z) : 0.f;
                            ring_split3(v0, v1, hi[pr], mid[pr], lo[pr]);
                        }
                        const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]}, l4 = {lo[0], lo[1], lo[2], lo[3]};
                        aq[m][0] = __builtin_bit_cast(rbf16x8_t, h4);
                        aq[m][1] = __builtin_bit_cast(rbf16x8_t, m4);
                        aq[m][2] = __builtin_bit_cast(rbf16x8_t, l4);
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};           // lo.hi, hi.lo, mid.mid, mid.hi, hi.mid | hi.hi: smallest first
                    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp)
#pragma unroll
                        for (int m = 0; m < NM; ++m)
#pragma unroll
                            for (int t3 = 0; t3 < 3; ++t3) {
                                if (pp < 5) accs[t3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[m][PA[pp]], bq[m][t3][PB[pp]], accs[t3], 0, 0, 0);
                                else acc[t3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[m][PA[pp]], bq[m][t3][PB[pp]], acc[t3], 0, 0, 0);
                            }
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) acc[t3] += accs[t3];
                } else {
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
            }
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) *reinterpret_cast<f32x4*>(&red[par][wave][t3][lane][0]) = acc[t3];
        tr.stamp(1);                                            // MFMA chain + accumulators to LDS
        __syncthreads();
        tr.stamp(2);                                            // barrier (slowest wave)

        float pre[4];
        const float* rp = &red[par][0][0][0][0];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = zq[g];
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) v += rp[wv * (3 * 64 * 4) + src_off[g]];
            pre[g] = v;
        }
        const float ig = ring_sigmoid(pre[0]);
        const float gg = ring_tanh(pre[1]);
        const float fg = ring_sigmoid(pre[2] + 1.0f);                 // forget_bias = 1.0
        const float og = ring_sigmoid(pre[3]);
        const float c = c_state * fg + ig * gg;
        const float tc = ring_tanh(c);
        const float h = live ? tc * og : 0.f;                         // dead rows / units publish zeros
        c_state = c;
        // hand h_s to the ring first: lanes ul = 0, 3, 6, 9 assemble {h[ul], h[ul+1], h[ul+2], tag}
        // h of the next two lanes of this 16-lane row: DPP row shifts on the VALU (the ds_bpermute form of __shfl_down goes through the LDS
        // crossbar: two more round trips on the hand-off's critical path)
        // fp16x3: the published value is h * 2^13 already split (ring_pack_hm) -- one split per value instead of one per consumer wave
        const int hb = F16 ? (int)ring_pack_hm(h * 8192.0f) : __float_as_int(h);
        const float h0 = __int_as_float(hb);
        const float h1 = __int_as_float(__builtin_amdgcn_update_dpp(0, hb, 0x101, 0xf, 0xf, true));     // row_shl:1
        const float h2 = __int_as_float(__builtin_amdgcn_update_dpp(0, hb, 0x102, 0xf, 0xf, true));     // row_shl:2
        if (ul < UW && ul % 3 == 0)
            st16(rs, xb, (unsigned)(((par * TB + row) * NG) + w * 4 + ul / 3) * 16u, make_float4(h0, h1, h2, __uint_as_float((unsigned)(s + 1))), fast);
        tr.stamp(3);                                            // gate epilogue up to the granule store
        // what the backward pass needs: kept in registers and stored one step later, right behind the NEXT wait (AMS_RING_FWD_STORE_LATE):
        // vmcnt counts loads and stores together, so seven scattered stores issued here sit in front of the poll that follows (round 4:
        // 4343 -> 4221 cycles per step; in round 2, with the 2300-cycle f32 MFMA chain, the same move had cost 0.26 us)
#if AMS_RING_FWD_STORE_LATE
        pend_v[0] = ig; pend_v[1] = gg; pend_v[2] = fg; pend_v[3] = og; pend_v[4] = c; pend_v[5] = tc; pend_v[6] = h; pend_t = t;
#else
        if (live) {
            float* gr = gp + t * gst;
            gr[0 * H] = ig;
            gr[1 * H] = gg;
            gr[2 * H] = fg;
            gr[3 * H] = og;
            cp[t * cst_st] = c;
            tp_[t * cst_st] = tc;
            op[t * cst_st] = h;
        }
#endif
#if AMS_RING_FWD_FETCH_LATE
        {
            const float* gn = gp + min(max(dir ? t - 1 : t + 1, 0), T - 1) * gst;
#pragma unroll
            for (int g = 0; g < 4; ++g) zn[g] = gn[g * H];
        }
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g) zq[g] = zn[g];
        tr.stamp(4);                                            // G / cst / tanh(c) / out stores
    }
#if AMS_RING_FWD_STORE_LATE
    if (live && pend_t >= 0) {
        float* gr = gp + pend_t * gst;
        gr[0 * H] = pend_v[0]; gr[1 * H] = pend_v[1]; gr[2 * H] = pend_v[2]; gr[3 * H] = pend_v[3];
        cp[pend_t * cst_st] = pend_v[4]; tp_[pend_t * cst_st] = pend_v[5]; op[pend_t * cst_st] = pend_v[6];
    }
#endif
    tr.end();
}

// ----------------------------------------------------------------------------------------------------- backward
// NI = ceil(NT / 4) output-tile rounds per wave (compile-time, as NR above); NP = 4 * ceil(NW / 4) partial tiles summed per element.
// Partial tiles are read with 16-byte loads: a dword load costs the CU's address unit the same 16 cycles per wave-instruction as
// a dwordx4 one, and 28 of them per thread (112 per workgroup and step) were ~1800 cycles of the step.  Thread (group g, slot n,
// row quad i), g < PG = 5, reads producers g, g + 5, ... (NPG = ceil(NW / 5) float4 loads), the PG group sums meet in LDS and are
// added in group order -- a fixed order, so the result is deterministic.
constexpr int PG = 5;

// Hand-off (round 4): the partial tiles carry their own validity.  Every 16-byte piece of a tile is one lane's store, and the low
// mantissa bit of each of its four floats holds the PHASE of the step that produced it: 1 for steps 0, 1, then 0 for 2, 3, 1 for
// 4, 5 ... (a tile buffer is reused every second step, so its phase alternates; the buffers start zeroed, phase 0).  A consumer simply
// loads the pieces it needs until all four bits of every one show the phase it expects -- one hop.  Until round 4 a producer waited
// for its stores to be acknowledged (s_waitcnt vmcnt(0): ~1150 cycles), met its workgroup at a barrier, raised a flag; the consumer
// polled the flags and only then loaded the tiles: three memory round trips on the critical path of each of the 80 steps, ~3000 of
// ~5900 cycles.  The stolen bit costs each partial 1 ulp (it is cleared before the sum); a piece cannot be seen half-written (one
// dwordx4 store inside one 64-byte line), and a phase cannot be mistaken for the one before last: a workgroup overwrites a buffer with
// step s only after it has consumed step s - 1 of every producer, each of which had consumed all of step s - 2 before producing that.
__device__ __forceinline__ float4 ring_tag4(float4 v, unsigned phase) {
    const unsigned x = __float_as_uint(v.x) & ~1u, y = __float_as_uint(v.y) & ~1u, z = __float_as_uint(v.z) & ~1u, w = __float_as_uint(v.w) & ~1u;
    return make_float4(__uint_as_float(x | phase), __uint_as_float(y | phase), __uint_as_float(z | phase), __uint_as_float(w | phase));
}
__device__ __forceinline__ bool ring_tagged4(float4 v, unsigned phase) {
    return ((__float_as_uint(v.x) & __float_as_uint(v.y) & __float_as_uint(v.z) & __float_as_uint(v.w) & 1u) == phase) &&
           (((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) & 1u) == phase);
}
__device__ __forceinline__ float ring_untag(float v) { return __uint_as_float(__float_as_uint(v) & ~1u); }
__device__ __forceinline__ unsigned ring_phase(int s) { return (unsigned)(((s >> 1) & 1) ^ 1); }

// ARITH = 2 (round 4): the recurrent product da . U^T on the 16-bit pipe as fp16x3 (csrc/gemm.hip), as the forward ring has run it
// since round 3.  The f32 MFMA chain was the longest phase of a step -- 60 v_mfma_f32_16x16x4_f32 of 32 cycles per wave, ~1900 of
// ~6400 cycles -- and da has no a-priori bound to scale it into fp16 range with.  So the product is TRANSPOSED:
//     dh^T [units x rows] = U-slice [units x 48] . da^T [48 x rows]
// U is the A operand (static, scaled once from the caller's bound amax_u, split into two fp16 planes, in registers) and da the B
// operand, whose COLUMNS are batch rows -- and an MFMA lane owns exactly one output column.  Every batch row therefore gets its own
// power-of-two scale, 2^(13 - floor(log2 max_k |da[row, k]|)), computed by the lane from the 12 values it reads anyway (the other three
// lane groups of its row: two v_permlane swaps), undone on the lane's own accumulator registers: no row can lose bits to a larger
// neighbour, which a per-tile or per-tensor scale would allow.  k order: slot e of lane group q <-> k = 12 q + e (gate q, local unit
// e) on both sides; 48 k = one K = 32 MFMA (slots 0..7) + one K = 16 MFMA (slots 8..11).  6 MFMAs of ~17 cycles per output tile instead
// of 12 of 32.
// The partial tiles change shape with it: a lane holds 4 consecutive UNITS of one row, so a tile is [row][slot] here ([slot][row] in
// the f32 form), still 16-byte stores and loads.
template <int NI, int NPG, int ARITH = 0>
__global__ __launch_bounds__(256) void lstm_ring_bwd_kernel(RingArgs a) {
    constexpr bool F16 = (ARITH == 2);
    constexpr int AP = F16 ? 4 * UW + 4 : 4 * UW + 1;          // row pitch of lds_a (fp16x3: 16-byte aligned rows for ds_read_b128)
    __shared__ __attribute__((aligned(16))) float lds_a[TB][AP];             // da of this workgroup: [row][gate * 12 + local unit]
    __shared__ __attribute__((aligned(16))) float psum[PG][UW * TB];         // partial dh sums of the PG producer groups
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
    float bw[F16 ? 1 : NI][UW];
    rf16x8_t wq[F16 ? NI : 1][2];                               // fp16x3: [tile][plane hi / mid], k-slots 0..7 of the lane group (K = 32 MFMA)
    rf16x4_t wr[F16 ? NI : 1][2];                               //         ... k-slots 8..11 (K = 16 MFMA)
    float sc_u_inv = 1.0f;
    if constexpr (F16) {
        float sc_u;
        ring_f16_scale2(a.amax_u[0], sc_u, sc_u_inv);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int tl = wave + 4 * i, unit_row = tl * 16 + n16;
            unsigned hi[6], mid[6];
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 2 * pr + h, ucol = w * UW + e;            // k = 12 q + e: gate q, local unit e
                    v[h] = (tl < NT && unit_row < H && ucol < H) ? U[(long)unit_row * a.ldu + q * H + ucol] * sc_u : 0.f;
                }
                ring_split2h(v[0], v[1], hi[pr], mid[pr]);
            }
            const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]};
            const uint2 h2 = {hi[4], hi[5]}, m2 = {mid[4], mid[5]};
            wq[i][0] = __builtin_bit_cast(rf16x8_t, h4);
            wq[i][1] = __builtin_bit_cast(rf16x8_t, m4);
            wr[i][0] = __builtin_bit_cast(rf16x4_t, h2);
            wr[i][1] = __builtin_bit_cast(rf16x4_t, m2);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int tl = wave + 4 * i, unit_row = tl * 16 + n16;
#pragma unroll
            for (int kk = 0; kk < UW; ++kk) {
                const int k = 4 * kk + q, gate = k / UW, ucol = w * UW + k % UW;
                bw[i][kk] = (tl < NT && unit_row < H && ucol < H) ? U[(long)unit_row * a.ldu + gate * H + ucol] : 0.f;
            }
        }
    }

    // element of this thread: (local unit n, row r) -- the order the partial tiles are stored in
    const int nl = tid >> 4, r = tid & 15;
    const int b = bt * TB + r, u = w * UW + nl;
    const bool live = (nl < UW && b < a.B && u < H);
    const size_t tile_f = (size_t)UW * TB;                      // floats per partial tile
    float* pb = a.xbuf + (size_t)chain * 2 * NW * NW * tile_f;
    const rsrc_t rs = make_rsrc(pb, (unsigned)((size_t)2 * NW * NW * tile_f * 4));
    // where this lane's column of output tile (wave + 4 i) goes: consumer = unit / 12, slot = unit % 12 (computed once)
    // (fp16x3: the lane holds units tile * 16 + 4 q .. + 3 of row n16 -- a 4-aligned group never straddles a consumer's 12 units)
    int toff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if constexpr (F16) {
            const int unit = (wave + 4 * i) * 16 + 4 * q;
            toff[i] = (unit < NW * UW) ? (int)((((size_t)(unit / UW) * NW + w) * tile_f + n16 * UW + unit % UW) * 4u) : -1;
        } else {
            const int unit = (wave + 4 * i) * 16 + n16;
            toff[i] = (unit < NW * UW) ? (int)((((size_t)(unit / UW) * NW + w) * tile_f + (unit % UW) * TB + 4 * q) * 4u) : -1;
        }
    }
    float dc_state = 0.f;
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};                       // running sum over t of this element's four da (bias gradient)
    float amax_f = 0.f;                                         // max |da| of this element over t: operand bound of the products that read dZ

    // per-thread base pointers (time 0) and per-time-step strides; dead threads alias element (0, 0)
    const long bl_ = live ? b : 0, ul_ = live ? u : 0;
    float* const gp = a.G + ((bl_ * T) * 2 + dir) * (4 * H) + ul_;            // + t * 8H + gate * H
    const float* const cp = a.cst + ((bl_ * T) * 2 + dir) * H + ul_;          // + t * 2H
    const float* const tp_ = a.tch + ((bl_ * T) * 2 + dir) * H + ul_;         // + t * 2H
    const float* const dp = a.dout + (bl_ * T) * (2 * H) + dir * H + ul_;     // + t * 2H
    const int gst = 8 * H, cst_st = 2 * H;
    struct Ops { float dh, ig, gg, fg, og, tc, c_prev; };
    // unconditional loads on clamped addresses (see the forward kernel): out-of-range times read their nearest valid neighbour,
    // validity is applied to the VALUES afterwards.  tanh(c_t) comes from the forward ring (tch), not from libm again.
    auto fetch = [&](int t_raw) {
        Ops o;
        const int t = min(max(t_raw, 0), T - 1);
        const int tp = dir ? t + 1 : t - 1, tpc = min(max(tp, 0), T - 1);
        o.dh = dp[t * cst_st];
        const float* gr = gp + t * gst;
        o.ig = gr[0 * H]; o.gg = gr[1 * H]; o.fg = gr[2 * H]; o.og = gr[3 * H];
        o.tc = tp_[t * cst_st];
        // x 0/1, not a select: hipcc sinks a load that only one side of a (uniform) branch uses INTO that branch and waits there
        o.c_prev = cp[tpc * cst_st] * ((tp >= 0 && tp < T) ? 1.0f : 0.0f);
        return o;
    };
    Ops cur = fetch(dir ? 0 : T - 1);

    Trace tr;
    tr.begin(reinterpret_cast<unsigned long long*>(a.err) + 8, a.trace && chain == 0 && w == 0 && tid == 0);
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : (T - 1 - s);
        const int par = s & 1;
        float dh = cur.dh;
        const float ig = cur.ig, gg = cur.gg, fg = cur.fg, og = cur.og, tc = cur.tc, c_prev = cur.c_prev;
        if (s > 0) {
            // thread -> (group pg, float4 index f4 = slot * 4 + row quad; fp16x3: row * 3 + slot quad); 240 of 256 threads take part, the
            // last 16 re-read group 4's pieces and drop them
            const int pg = tid / 48, f4 = tid - pg * 48, pgc = min(pg, PG - 1);
            const unsigned base = (unsigned)((((size_t)(par ^ 1) * NW + w) * NW) * tile_f + f4 * 4) * 4u;
            const unsigned want = ring_phase(s - 1);
            float4 pv[NPG];
            unsigned spins = 0;
#ifdef AMS_RING_DBG_ACK                          // anatomy build (-DAMS_RING_DBG_ACK): how long this wave's own stores take to be acknowledged
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // -- 750 cycles, then 1020 of polling: the two are serial in the shipped form
            tr.stamp(5);                                        // too (1680), vmcnt being one counter for loads and stores
#endif
            for (;;) {                                          // every wave waits for the pieces it sums itself
#pragma unroll
                for (int k = 0; k < NPG; ++k) pv[k] = ld16_l2(rs, base + (unsigned)(min(pgc + PG * k, NW - 1) * tile_f) * 4u);
                __builtin_amdgcn_sched_barrier(0);              // all requests in flight before the first piece is looked at
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NPG; ++k) ok &= ring_tagged4(pv[k], want);
                if (__all(ok) || abort) break;
                if (spin_check(spins, a.err, a.sticky)) { abort = true; break; }
                if (AMS_RING_BWD_SLEEP) __builtin_amdgcn_s_sleep(AMS_RING_BWD_SLEEP);
            }
            tr.stamp(0);                                        // wait for the partial tiles
            if (pg < PG) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < NPG; ++k) {
                    const float m = (pg + PG * k < NW) ? 1.0f : 0.0f;          // clamped duplicates count as zero
                    sum.x += ring_untag(pv[k].x) * m; sum.y += ring_untag(pv[k].y) * m;
                    sum.z += ring_untag(pv[k].z) * m; sum.w += ring_untag(pv[k].w) * m;
                }
                *reinterpret_cast<float4*>(&psum[pg][f4 * 4]) = sum;
            }
            __syncthreads();
            const int e = F16 ? r * UW + (nl < UW ? nl : 0) : (nl < UW ? nl : 0) * TB + r;
#pragma unroll
            for (int g = 0; g < PG; ++g) dh += psum[g][e];
        } else {
            tr.stamp(0);
        }
        tr.stamp(1);                                            // partial tiles loaded + summed
        // next step's operands: requested now (after this step's waits, ~2000 cycles of MFMA work ahead), not behind the publish --
        // vmcnt retires in issue order, so loads issued ahead of the next poll would put their HBM latency in front of it
#if !AMS_RING_FETCH_LATE
        const Ops nxt = fetch(dir ? t + 1 : t - 1);
#endif
        const float d_o = dh * tc;
        const float dcv = dc_state + dh * og * (1.0f - tc * tc);
        const float da0 = live ? dcv * gg * ig * (1.0f - ig) : 0.f;
        const float da1 = live ? dcv * ig * (1.0f - gg * gg) : 0.f;
        const float da2 = live ? dcv * c_prev * fg * (1.0f - fg) : 0.f;
        const float da3 = live ? d_o * og * (1.0f - og) : 0.f;
        dc_state = dcv * fg;
        dbs[0] += da0; dbs[1] += da1; dbs[2] += da2; dbs[3] += da3;
        amax_f = fmaxf(fmaxf(amax_f, fmaxf(fabsf(da0), fabsf(da1))), fmaxf(fabsf(da2), fabsf(da3)));
        if (nl < UW) {
            lds_a[r][0 * UW + nl] = da0;
            lds_a[r][1 * UW + nl] = da1;
            lds_a[r][2 * UW + nl] = da2;
            lds_a[r][3 * UW + nl] = da3;
        }
        tr.stamp(2);                                            // gate derivative math + LDS write
        __syncthreads();
        tr.stamp(3);                                            // barrier
        // da to memory BEFORE the recurrent product, not after it: the publish below waits for this wave's youngest store, and these four
        // (scattered dwords, the slowest stores of the step) then have the whole MFMA chain to complete in
        if (live) {
            float* gr = gp + t * gst;
            gr[0 * H] = da0;
            gr[1 * H] = da1;
            gr[2 * H] = da2;
            gr[3 * H] = da3;
        }
#ifdef AMS_RING_DBG_FINE                         // anatomy build: the last phase of a step in four pieces (tools/ring_anatomy.py, AMS_ANATOMY_FINE=1)
        tr.stamp(5);                                            // dZ stores issued
#endif
        if (s + 1 < T) {
            f32x4 acc[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (F16) {
                // this lane's B fragment: da[row n16][12 q .. 12 q + 11], three 16-byte reads
                const float4* brow = reinterpret_cast<const float4*>(&lds_a[n16][UW * q]);
                const float4 v0 = brow[0], v1 = brow[1], v2 = brow[2];
                float mx = fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                                 fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))));
                // max over the row's four lane groups (non-negative floats order like their bit patterns)
                unsigned mu = __float_as_uint(mx);
                const auto s16 = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
                mu = max(s16[0], s16[1]);
                const auto s32 = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                mu = max(s32[0], s32[1]);
                float sc, sc_inv;
                ring_f16_scale2(__uint_as_float(mu), sc, sc_inv);
                sc_inv *= sc_u_inv;
                rf16x8_t bq[2];                                 // plane hi / mid, k-slots 0..7
                rf16x4_t br[2];                                 //                 k-slots 8..11
                {
                    unsigned hi[6], mid[6];
                    ring_split2h(v0.x * sc, v0.y * sc, hi[0], mid[0]);
                    ring_split2h(v0.z * sc, v0.w * sc, hi[1], mid[1]);
                    ring_split2h(v1.x * sc, v1.y * sc, hi[2], mid[2]);
                    ring_split2h(v1.z * sc, v1.w * sc, hi[3], mid[3]);
                    ring_split2h(v2.x * sc, v2.y * sc, hi[4], mid[4]);
                    ring_split2h(v2.z * sc, v2.w * sc, hi[5], mid[5]);
                    const uint4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]};
                    const uint2 h2 = {hi[4], hi[5]}, m2 = {mid[4], mid[5]};
                    bq[0] = __builtin_bit_cast(rf16x8_t, h4);
                    bq[1] = __builtin_bit_cast(rf16x8_t, m4);
                    br[0] = __builtin_bit_cast(rf16x4_t, h2);
                    br[1] = __builtin_bit_cast(rf16x4_t, m2);
                }
                // ONE accumulator per tile (the workgroup must fit beside two residency-capped product workgroups of 160 registers:
                // <= 192 in all; with the cross terms in accumulators of their own it needed 230, and a ring launched behind such a
                // product then waited for its workgroups to retire -- 470 us instead of 300).  Smallest terms first.
#ifdef AMS_RING_DBG_FINE
                asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(br[0]), "+v"(br[1]));
                tr.stamp(6);                                    // B operand read, scaled, split
#endif
                constexpr int PA[3] = {1, 0, 0};                // U plane: mid.hi, hi.mid, hi.hi
                constexpr int PB[3] = {0, 1, 0};                // da plane
                // Per tile TWO chains that never meet in the matrix pipe: the k-slots 8..11 on v_mfma_f32_16x16x16_f16, the k-slots 0..7 on
                // v_mfma_f32_16x16x32_f16, each starting from the constant 0 and feeding only MFMAs of its OWN opcode; one VALU add
                // joins them.  (An accumulator handed from one MFMA shape to the other came back wrong -- back-to-back forwarding of
                // SrcC is a same-opcode affair -- and round 5 kept ONE chain per tile apart with a hand-counted `s_nop 15; s_nop 7`;
                // a second accumulator SET, or a VALU pass between the chains, cost 24 / 6 registers and with them the workgroup's
                // place beside two capped product workgroups: 2.63 -> 3.11 / 2.81 -> 3.28 ms per step, measured in round 6.)  Tile by
                // tile the two partial sums are the only extra live registers, the chains of a tile alternate (a dependent MFMA is two
                // issues behind its producer), and a finished tile leaves while the next one is in the pipe.
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                const unsigned pbase16 = (unsigned)((size_t)par * NW * NW * tile_f * 4u);
#pragma unroll
                for (int i = 0; i < NI; ++i) {                  // (two tiles at a time: 24 accumulator registers, 180 in all -- over the 176 the place beside the products allows)
                    f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(wr[i][PA[0]], br[PB[0]], zero4, 0, 0, 0);
                    f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[i][PA[0]], bq[PB[0]], zero4, 0, 0, 0);
                    r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(wr[i][PA[1]], br[PB[1]], r16, 0, 0, 0);
                    r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[i][PA[1]], bq[PB[1]], r32, 0, 0, 0);
                    r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(wr[i][PA[2]], br[PB[2]], r16, 0, 0, 0);
                    r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[i][PA[2]], bq[PB[2]], r32, 0, 0, 0);
                    acc[i] = (r16 + r32) * sc_inv;
#ifndef AMS_RING_DBG_FINE
                    if (toff[i] >= 0) st16(rs, pb, pbase16 + (unsigned)toff[i], ring_tag4(make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]), ring_phase(s)), fast);
#endif
                }
#ifdef AMS_RING_DBG_FINE
#pragma unroll
                for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(acc[i]));
                tr.stamp(7);                                    // MFMA chains + rescale
#endif
            } else {
                // partial dh_{next} = da_own [16 x 48] . U^T slice [48 x NT*16]; the 12 A fragments of this lane first, ONE wait
                float av[UW];
                const float* arow = &lds_a[lane & 15][q];
#pragma unroll
                for (int kk = 0; kk < UW; ++kk) av[kk] = arow[4 * kk];
#pragma unroll
                for (int kk = 0; kk < UW; ++kk) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bw[i][kk], acc[i], 0, 0, 0);
                }
            }
            // f32: tile column n16 of output tile tl = unit tl*16 + n16 -> consumer unit / 12, slot unit % 12; rows 4q..4q+3 contiguous
            const unsigned pbase = (unsigned)((size_t)par * NW * NW * tile_f * 4u);
#ifndef AMS_RING_DBG_FINE
            if constexpr (!F16)                                 // (the fp16x3 form stores every tile as it leaves the pipe, above)
#endif
#pragma unroll
            for (int i = 0; i < NI; ++i)
                if (toff[i] >= 0) st16(rs, pb, pbase + (unsigned)toff[i], ring_tag4(make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]), ring_phase(s)), fast);
            tr.stamp(4);                                        // MFMA chain + tile stores issued
        }
#if AMS_RING_FETCH_LATE
        const Ops nxt = fetch(dir ? t + 1 : t - 1);
#endif
        cur = nxt;
    }
    if (a.dbpart && live) {
        float* o = a.dbpart + ((long)b * 2 + dir) * (4 * H) + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) o[g * H] = dbs[g];
    }
    // word RING_AMAX_WORD of the sync head (zeroed before the launch): max |da| over the whole launch, as float bits
    amax_f = live ? amax_f : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax_f = fmaxf(amax_f, __shfl_xor(amax_f, o));
    if (lane == 0) atomicMax(a.err + RING_AMAX_WORD, __float_as_uint(amax_f));
    tr.end();
}

// Workgroups of the ring kernels this device can hold AT ONCE: CUs x 2 (what their registers admit), read from the device once.
// MI355X in SPX mode: 256 x 2 = 512.  A CPX / DPX partition, a CU-masked process or a smaller part has fewer, and a ring that
// is not fully resident would only ever time out.
inline long ring_capacity() {
    static const long cap = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 512L;                                        // no device visible (build / CPU-side symbol tests): MI355X geometry
        if (const char* f = getenv("AMS_LSTM_RING_CUS")) cus = atoi(f);      // testing aid: pretend a smaller partition
        return 2L * cus;
    }();
    return cap;
}
inline bool ring_shape(int B, int H, int& NW, int& n_chains) {
    NW = ceil_div(H, UW);
    n_chains = 2 * ceil_div(B, TB);
    // the grid is padded to whole groups of 8 chains (one chain per XCD id): count those workgroups, not only the live ones
    const long grid = 8L * NW * ceil_div(n_chains, 8);
    return NW <= 4 * MAXR && ceil_div(NW * UW, 16) <= 4 * MAXT && (long)n_chains * NW <= 512 && grid <= ring_capacity();
}

struct RingLayout { size_t ids, x, total, head; };

inline RingLayout ring_layout(int NW, int n_chains, int backward) {
    RingLayout L;
    L.ids = 256;
    L.x = (L.ids + (size_t)n_chains * IDS_STRIDE * 4 + 255) & ~(size_t)255;
    const size_t xbytes = backward ? (size_t)n_chains * 2 * NW * NW * UW * TB * 4 : (size_t)n_chains * 2 * TB * NW * 4 * 16;
    L.total = L.x + xbytes;
    L.head = L.total;                       // [0, head) is zeroed before every launch: the granule tags (forward) / phase bits (backward) start at 0
    return L;
}

// AMS_LSTM_RING_X6 (read once): 1 = the forward ring's recurrent product on the bf16 pipe (bf16x6), 0 = v_mfma_f32_16x16x4_f32
inline bool ring_fwd_x6() {
    static const bool v = getenv("AMS_LSTM_RING_X6") ? atoi(getenv("AMS_LSTM_RING_X6")) != 0 : AMS_RING_X6_DEFAULT;
    return v;
}

// AMS_LSTM_RING_F16 (read once): 0 = the forward ring never takes the fp16x3 form, whatever bound it is given
inline bool ring_fwd_f16() {
    static const bool v = !(getenv("AMS_LSTM_RING_F16") && atoi(getenv("AMS_LSTM_RING_F16")) == 0);
    return v;
}

// AMS_LSTM_RING_BWD_F16 (read once): 0 = the backward ring keeps v_mfma_f32_16x16x4_f32 whatever bound it is given (measurement aid)
inline bool ring_bwd_f16() {
    static const bool v = !(getenv("AMS_LSTM_RING_BWD_F16") && atoi(getenv("AMS_LSTM_RING_BWD_F16")) == 0);
    return v;
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

// bytes at the start of the sync buffer that must be zero when a ring launch starts (forward: the whole buffer, granule tags included)
size_t ams_blstm_ring_sync_head_bytes(int B, int H, int backward) {
    int NW, n_chains;
    if (B <= 0 || H <= 0 || !ring_shape(B, H, NW, n_chains)) return 0;
    return ring_layout(NW, n_chains, backward).head;
}

// Same contract as ams_blstm_recurrent_fwd (G: pre-activations in, activated gates out; out; cst), plus `sync`
// (ams_blstm_ring_sync_bytes(B, H, 0) bytes, word 0 = timeout flag) and `tch` [B,T,2,H] = tanh(c_t), which the backward ring reads
// instead of calling tanhf again.  safe bit 0 forces the placement-independent hand-off, bit 1 turns the phase trace on.
// Bit 2: the caller has ALREADY zeroed the sync buffer (all of it for the forward ring, its first ams_blstm_ring_sync_head_bytes()
// for the backward ring) in stream order before this launch -- no memset node in front of the ring (ops.py clears the buffers of a
// whole pass with one memset on the side stream while the pass starts).
// amax_u (optional): device pointer to an upper bound of max |U| over both recurrent kernels -> the recurrent product runs as fp16x3.
ams_status ams_blstm_ring_fwd(float* G, float* out, float* cst, float* tch, const float* Uf, const float* Ub, long ldu, const float* amax_u,
                              void* sync, size_t sync_bytes, void* sticky_err, int B, int T, int H, int safe, void* stream) {
    AMS_REQUIRE(G && out && cst && tch && Uf && Ub && sync && B > 0 && T > 0 && H > 0);
    int NW, n_chains;
    AMS_REQUIRE(ring_shape(B, H, NW, n_chains));
    const RingLayout L = ring_layout(NW, n_chains, 0);
    if (sync_bytes < L.total) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (!(safe & 4) && hipMemsetAsync(sync, 0, L.head, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;      // bit 2: the caller cleared [0, head)
    RingArgs a{};
    a.G = G; a.out = out; a.cst = cst; a.tch = tch; a.Uf = Uf; a.Ub = Ub; a.ldu = ldu;
    a.err = (unsigned*)sync; a.sticky = (unsigned*)sticky_err; a.ids = (unsigned*)((char*)sync + L.ids);
    a.xbuf = (float*)((char*)sync + L.x);
    a.B = B; a.T = T; a.H = H; a.NW = NW; a.n_chains = n_chains; a.force_safe = ((safe & 1) || ring_force_safe()) ? 1 : 0; a.trace = (safe & 2) ? 1 : 0;
    const dim3 grid(8 * NW * ceil_div(n_chains, 8));
    if (ring_fwd_x6() && amax_u && ring_fwd_f16()) {
        a.amax_u = amax_u;
        switch (ceil_div(NW, 4)) {
            case 1: hipLaunchKernelGGL((lstm_ring_fwd_kernel<1, 2>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((lstm_ring_fwd_kernel<2, 2>), grid, dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL((lstm_ring_fwd_kernel<3, 2>), grid, dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL((lstm_ring_fwd_kernel<4, 2>), grid, dim3(256), 0, st, a); break;
            case 5: hipLaunchKernelGGL((lstm_ring_fwd_kernel<5, 2>), grid, dim3(256), 0, st, a); break;
            case 6: hipLaunchKernelGGL((lstm_ring_fwd_kernel<6, 2>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((lstm_ring_fwd_kernel<7, 2>), grid, dim3(256), 0, st, a); break;
        }
        return ams_check_launch();
    }
    if (ring_fwd_x6()) {
        switch (ceil_div(NW, 4)) {
            case 1: hipLaunchKernelGGL((lstm_ring_fwd_kernel<1, 1>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((lstm_ring_fwd_kernel<2, 1>), grid, dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL((lstm_ring_fwd_kernel<3, 1>), grid, dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL((lstm_ring_fwd_kernel<4, 1>), grid, dim3(256), 0, st, a); break;
            case 5: hipLaunchKernelGGL((lstm_ring_fwd_kernel<5, 1>), grid, dim3(256), 0, st, a); break;
            case 6: hipLaunchKernelGGL((lstm_ring_fwd_kernel<6, 1>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((lstm_ring_fwd_kernel<7, 1>), grid, dim3(256), 0, st, a); break;
        }
        return ams_check_launch();
    }
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
// dbpart (optional, [B,2,4H]): receives sum_t da[b,t,dir,:] -- the bias gradient is then a column sum over B rows instead of B*T.
// On return float word 2 of `sync` holds max |da| (the operand bound of the three products that read dZ).
// amax_u (optional): device pointer to an upper bound of max |U| over both recurrent kernels -> the recurrent product runs as fp16x3
// with one scale per batch row (see lstm_ring_bwd_kernel); NULL keeps the f32 MFMA.
ams_status ams_blstm_ring_bwd(float* G, const float* cst, const float* tch, const float* dout, float* dbpart, const float* Uf, const float* Ub,
                              long ldu, const float* amax_u, void* sync, size_t sync_bytes, void* sticky_err, int B, int T, int H, int safe, void* stream) {
    AMS_REQUIRE(G && cst && tch && dout && Uf && Ub && sync && B > 0 && T > 0 && H > 0);
    int NW, n_chains;
    AMS_REQUIRE(ring_shape(B, H, NW, n_chains));
    const RingLayout L = ring_layout(NW, n_chains, 1);
    if (sync_bytes < L.total) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (!(safe & 4) && hipMemsetAsync(sync, 0, L.head, st) != hipSuccess) return AMS_E_LAUNCH_FAILED;      // bit 2: the caller cleared [0, head)
    RingArgs a{};
    a.G = G; a.cst = const_cast<float*>(cst); a.tch = const_cast<float*>(tch); a.dout = dout; a.dbpart = dbpart; a.Uf = Uf; a.Ub = Ub; a.ldu = ldu;
    a.err = (unsigned*)sync; a.sticky = (unsigned*)sticky_err; a.ids = (unsigned*)((char*)sync + L.ids);
    a.xbuf = (float*)((char*)sync + L.x);
    a.B = B; a.T = T; a.H = H; a.NW = NW; a.n_chains = n_chains; a.force_safe = ((safe & 1) || ring_force_safe()) ? 1 : 0; a.trace = (safe & 2) ? 1 : 0;
    const dim3 grid(8 * NW * ceil_div(n_chains, 8));
    // NW = 1..28 -> NT = ceil(12 NW / 16) = 1..21 -> NI = ceil(NT / 4) = 1..6; producers per group: NPG = ceil(NW / 5) = 1..6
    const int NI = ceil_div(ceil_div(NW * UW, 16), 4);
    if (ring_fwd_x6() && amax_u && ring_fwd_f16() && ring_bwd_f16()) {
        a.amax_u = amax_u;
        if (NW <= 5)       hipLaunchKernelGGL((lstm_ring_bwd_kernel<1, 1, 2>), grid, dim3(256), 0, st, a);
        else if (NW <= 10) hipLaunchKernelGGL((lstm_ring_bwd_kernel<2, 2, 2>), grid, dim3(256), 0, st, a);
        else if (NW <= 15) hipLaunchKernelGGL((lstm_ring_bwd_kernel<3, 3, 2>), grid, dim3(256), 0, st, a);
        else if (NW <= 20) hipLaunchKernelGGL((lstm_ring_bwd_kernel<4, 4, 2>), grid, dim3(256), 0, st, a);
        else if (NW <= 25) hipLaunchKernelGGL((lstm_ring_bwd_kernel<5, 5, 2>), grid, dim3(256), 0, st, a);
        else               hipLaunchKernelGGL((lstm_ring_bwd_kernel<6, 6, 2>), grid, dim3(256), 0, st, a);
        return ams_check_launch();
    }
    if (NW <= 5)       hipLaunchKernelGGL((lstm_ring_bwd_kernel<1, 1>), grid, dim3(256), 0, st, a);       // NT <= 4
    else if (NW <= 10) hipLaunchKernelGGL((lstm_ring_bwd_kernel<2, 2>), grid, dim3(256), 0, st, a);       // NT <= 8
    else if (NW <= 15) hipLaunchKernelGGL((lstm_ring_bwd_kernel<3, 3>), grid, dim3(256), 0, st, a);       // NT <= 12
    else if (NW <= 20) hipLaunchKernelGGL((lstm_ring_bwd_kernel<4, 4>), grid, dim3(256), 0, st, a);       // NT <= 15
    else if (NW <= 25) hipLaunchKernelGGL((lstm_ring_bwd_kernel<5, 5>), grid, dim3(256), 0, st, a);       // NT <= 19
    else               hipLaunchKernelGGL((lstm_ring_bwd_kernel<6, 6>), grid, dim3(256), 0, st, a);       // NT <= 21
    (void)NI;
    return ams_check_launch();
}

}  // extern "C"
