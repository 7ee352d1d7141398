// Batched k-means mask assignment (reference models/Kmeans_2.py:86-188), HBM-bound streaming kernels.
//
// The reference tiles X nb_tries times (2.1 GB at the benchmark shape) and broadcasts a [R,L,C,E] temporary per
// label pass (4.2 GB).  Here the un-tiled, normalised embeddings are streamed: row r = b*tries + try reads x[b]
// (tries index the centroid set only), one pass computes the assignment AND the centroid numerators/denominators
// for the next iteration, and nothing larger than [R, chunks, C*(E+1)] partial sums is written.
//
// Bit-exact hard labels.  tf.unsorted_segment_sum is order-nondeterministic; oracle/kmeans.py and these kernels
// share ONE summation order: 8192-point chunks; lane j (0..255) adds its 32 points j, j+256, .. sequentially; each wavefront
// (64 consecutive lanes) combines its lanes by a halving tree v[j] += v[j+s], s = 32..1; the wavefront totals are added sequentially
// in (chunk, wavefront) order.  (The SOFT modes, which are tolerance-checked, keep 2048-point chunks and one 256-lane tree: their
// 64 rows would not fill the device with 8192-point workgroups.)  Everything that is compared bit for bit
// (hard distances, normalisation, inertia) accumulates left-to-right with separate, individually rounded multiply and add (no contraction), sqrt is IEEE,
// ties pick the lowest cluster (tf.argmin).
//
// Algorithmic bytes per pass: L*E*4 per row (+ L*4 weights); x[b] is shared by the `tries` rows of an utterance
// through L2.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
// Bit-exact parity with oracle/kmeans.py needs IEEE mul/add (no FMA contraction), sqrt and divide: contraction is
// switched off for this translation unit (see Makefile) and sqrtf / operator/ are the correctly rounded forms
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  NB: without OCML_BASIC_ROUNDED_OPERATIONS the
// __fsqrt_rn / __fmul_rn intrinsics are NOT rounding-safe (native sqrt, contractable multiply).
#pragma clang fp contract(off)

namespace {

#ifndef AMS_KM_CHUNK_SOFT
#define AMS_KM_CHUNK_SOFT 2048
#endif
constexpr int CHUNK_SOFT = AMS_KM_CHUNK_SOFT, CHUNK_HARD = 8192, LANES = 256;
__host__ __device__ constexpr int chunk_of(bool soft) { return soft ? CHUNK_SOFT : CHUNK_HARD; }

enum { HARD_ACC = 0, SOFT_ACC = 1, HARD_FINAL = 2, SOFT_FINAL = 3, HARD_LABELS = 4 };     // HARD_LABELS: HARD_FINAL without the inertia

// x [nrows, E] -> xn: x * (1/sqrt(max(sum_e x^2, 1e-12))), sequential over e (tf.nn.l2_normalize, Kmeans_2.py:40-41).
// Rows are staged through LDS in slabs of 256 so global traffic is coalesced while each thread still sums ITS row left to right
// (the order oracle/kmeans.py uses): 1.59 ms -> HBM speed for the 210 MB embedding tensor of the benchmark shape.
__global__ __launch_bounds__(256) void kmeans_normalize_kernel(const float* __restrict__ x, float* __restrict__ xn, long nrows, int E) {
    extern __shared__ float tile[];                  // [256][E + 1]
    const int LD = E + 1, tid = threadIdx.x;
    for (long r0 = (long)blockIdx.x * 256; r0 < nrows; r0 += (long)gridDim.x * 256) {
        const int nr = (int)min((long)256, nrows - r0);
        const float* src = x + r0 * E;
        __syncthreads();
        for (int i = tid; i < nr * E; i += 256) tile[(i / E) * LD + (i % E)] = src[i];
        __syncthreads();
        if (tid < nr) {
            float* p = tile + tid * LD;
            float ss = 0.f;
            for (int e = 0; e < E; ++e) ss = __fadd_rn(ss, __fmul_rn(p[e], p[e]));
            const float inv = ((1.0f) / (sqrtf(fmaxf(ss, 1e-12f))));
            for (int e = 0; e < E; ++e) p[e] = __fmul_rn(p[e], inv);
        }
        __syncthreads();
        float* dst = xn + r0 * E;
        for (int i = tid; i < nr * E; i += 256) dst[i] = tile[(i / E) * LD + (i % E)];
    }
}

struct KmArgs {
    const float* xn;       // [b, L, E]
    const float* w;        // [b, L] or null (all ones)
    const float* cent;     // [R, C, E]
    float* part;           // [R, G, NV]
    int32_t* labels;       // [R, L] (hard final) or null
    float* soft;           // [R, L, C] (soft final) or null
    long L;
    int b, tries, G;       // G chunks per row; partial rows per row: G (soft modes) or 4 G (hard modes: one per wavefront)
    int w_mod_b;           // 1: weight row = r % b (reference tile quirk), 0: r / tries
    float beta;
    float one;             // 1.0f, opaque to the compiler (see the HARD modes of kmeans_pass_kernel)
    // in-launch finish (tickets != NULL): the workgroup that stores a row's LAST chunk partial adds the row's chunks up in chunk order and
    // writes the centroids (ACC passes: fin_out [R, C, E], fin_den [R, C] or NULL) / the inertia (FINAL passes: fin_out [R]) -- what
    // kmeans_reduce_kernel did in a launch of its own, ~5 us + a kernel boundary on the serial chain of each of the 11 passes
    unsigned* tickets;     // [R], zero on entry, left zero
    float* fin_out; float* fin_den;
};

typedef float f2 __attribute__((ext_vector_type(2)));

// HAS_W: silence weights present.  Without them the reference multiplies by w = 1 (x*1 == x exactly), so the multiplies are
// dropped.  Element-wise work is written on 2-vectors (v_pk_mul_f32 / v_pk_add_f32: IEEE per component, no FMA) while every
// running sum keeps its left-to-right scalar order, so the result is bit-identical to the scalar formulation.
// __launch_bounds__(256, 3): at least 3 waves per SIMD, i.e. <= 168 VGPRs -- without the bound hipcc settles at 222-256 registers
// (2 waves, or 1) by hoisting the slab's LDS reads and the centroids into registers.
#ifndef AMS_KM_SGPR_CENT
#define AMS_KM_SGPR_CENT 1
#endif
#ifndef AMS_KM_ACC_DEPTH
#define AMS_KM_ACC_DEPTH 1
#endif
#ifndef AMS_KM_DIST_DEPTH
#define AMS_KM_DIST_DEPTH 1
#endif
#ifndef AMS_KM_XREG
#define AMS_KM_XREG 0       // 1: HARD_ACC keeps the point (E floats) in registers between the distance and the accumulation (one LDS read per group instead of two)
#endif
template <int E_, int C_, int MODE, bool HAS_W>
#ifndef AMS_KM_WAVES
#define AMS_KM_WAVES 3
#endif
__global__ __launch_bounds__(256, (MODE == HARD_ACC || MODE == HARD_FINAL || MODE == HARD_LABELS) ? AMS_KM_WAVES : 1) void kmeans_pass_kernel(KmArgs a) {
    static_assert(E_ % 4 == 0, "rows are staged as 16-byte vectors");
    constexpr bool ACC = (MODE == HARD_ACC || MODE == SOFT_ACC);
    constexpr bool SOFT = (MODE == SOFT_ACC || MODE == SOFT_FINAL);
    constexpr int NV = ACC ? C_ * (E_ + 1) : 2 * C_;
    // labels alone have no summation order to keep: the small chunks, which fill the device when R = b (the re-assignment at the end)
    constexpr int CHUNK = chunk_of(SOFT || MODE == HARD_LABELS), PPL = CHUNK / LANES;
    constexpr int LD = E_ + 4;                     // 16-byte aligned rows; 16 lanes x 16 B at this pitch cover all 64 banks
    constexpr int V4 = E_ / 4, V2 = E_ / 2;
    constexpr int BUF = (256 * LD > 128 * 64) ? 256 * LD : 128 * 64;
    __shared__ __attribute__((aligned(16))) float buf[BUF];
    __shared__ __attribute__((aligned(16))) float scent[C_ * E_];
    // Work order.  The grid is flat; the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, so XCD x sees ids
    // x, x + 8, ...  Utterance bi = 8 * (n / M) + x is given to XCD x as a whole (M = G * tries workgroups, n = id / 8), chunk by
    // chunk with the `tries` rows of a chunk adjacent: one utterance's embeddings (3.3 MB at L = 20480, E = 40) then sit in THAT
    // XCD's 4 MB L2 while its 10 tries read them.  In (chunk, row) order every try's read went out to the fabric: 2.1 GB per pass
    // at the benchmark shape, a prefetched slab took ~6 us to arrive and the waves sat idle 55 % of the time (-DAMS_KM_TRACE).
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int r, g;
    {
        const int M = a.G * a.tries, xcd = blockIdx.x & 7, n = blockIdx.x >> 3;
        const int ub = (n / M) * 8 + xcd, m = n - (n / M) * M;
        if (ub >= a.b) return;                                     // padding of the last group of 8 utterances (whole workgroup)
        g = m / a.tries;
        r = ub * a.tries + (m - g * a.tries);
    }
    const int bi = r / a.tries;
    const float* xb = a.xn + (long)bi * a.L * E_;
    const float* wb = HAS_W ? a.w + (long)(a.w_mod_b ? (r % a.b) : bi) * a.L : nullptr;
    for (int i = tid; i < C_ * E_; i += 256) scent[i] = a.cent[(long)r * C_ * E_ + i];
#if AMS_KM_SGPR_CENT
    // HARD modes: the row's centroids are wave-uniform -> held in SGPRs (C*E = 80 scalars at E = 40, C = 2) and fed to the packed
    // VALU operations as scalar operands.  As LDS broadcast reads they were 40 ds_read_b64 per point and try beside the 20
    // ds_read_b128 of the point itself: the pass was LDS-issue-bound (~120 us of LDS pipe per pass against ~90 us of VALU).
    float cs[SOFT ? 1 : C_ * E_];
    if (!SOFT) {
        const float* cg = a.cent + (long)r * C_ * E_;
#pragma unroll
        for (int i = 0; i < C_ * E_; ++i) cs[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cg[i])));
    }
#endif

    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;

    // slab j+1 is fetched into registers while slab j is being processed: without it every 256-point slab exposed a full
    // HBM/L2 round trip (8 per workgroup) and the pass ran at 1/5 of its arithmetic rate
    // Each WAVE stages the 64 rows its own lanes consume (rows wave*64 .. +63 of the slab) and never waits for the other three
    // waves inside the slab loop: LDS operations of one wave retire in issue order, so a fence + wave barrier is all the
    // ordering a wave needs between its writes and its reads.  (Workgroup barriers per slab left 41 % of the wave cycles parked.)
    float4 pre[V4];
    auto fetch = [&](int j) {
        const long q0 = (long)g * CHUNK + (long)j * LANES + wave * 64;
        const int np = (int)max((long)0, min((long)64, a.L - q0));
        const float4* src = reinterpret_cast<const float4*>(xb + q0 * E_);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const int i = lane + 64 * k;
#ifdef AMS_KM_NOLOAD            /* timing experiment only: the pass without its global reads */
            pre[k] = make_float4((float)i, 1.f, (float)j, 0.5f);
#else
            pre[k] = (i < np * V4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
        }
    };
    // measured, b = 64, L = 20480, 10 iterations: hard (640 rows, 6400 workgroups) 4.31 -> 3.82 ms with the prefetch; soft
    // (64 rows, 640 workgroups = 2.5 per CU, exp-heavy) 0.80 -> 1.15 ms -- so it is on for the hard modes only
    constexpr bool PF = !SOFT;
    if (PF) fetch(0);
    float* wbuf = buf + wave * 64 * LD;
    for (int j = 0; j < PPL; ++j) {
        const long p0 = (long)g * CHUNK + (long)j * LANES;
        const int npts = (int)max((long)0, min((long)LANES, a.L - p0));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // this wave's reads of the previous slab are complete
        __builtin_amdgcn_wave_barrier();
        if (!PF) fetch(j);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const int i = lane + 64 * k;
            const int row = i / V4, c4 = i - row * V4;
            *reinterpret_cast<float4*>(&wbuf[row * LD + c4 * 4]) = pre[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (PF && j + 1 < PPL) fetch(j + 1);
        if (tid < npts) {
            // HARD modes: the point is NOT held in registers (40 VGPRs): it is re-read from this wave's LDS rows as 16-byte vectors
            // (conflict-free at the LD = E + 4 pitch) once for the distances and once for the accumulation -- with the 82
            // accumulators and the 40 prefetch registers that is what keeps the kernel under 168 VGPRs = 3 waves per SIMD
            // (it ran at 236 VGPRs / 2 waves with ~44 % of the wave cycles waiting).  Same operations in the same order as before.
            const float* xrow = &buf[tid * LD];
            // the centroids stay in LDS (broadcast reads): without this clobber hipcc hoists all C*E of them out of the slab loop into
            // VGPRs (80 registers at E = 40, C = 2)
            if (!SOFT) asm volatile("" ::: "memory");
            float x[SOFT ? E_ : 1];
            if (SOFT) {
#pragma unroll
                for (int q = 0; q < V4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(xrow + q * 4);
                    x[(q * 4 + 0) % (SOFT ? E_ : 1)] = v.x; x[(q * 4 + 1) % (SOFT ? E_ : 1)] = v.y;
                    x[(q * 4 + 2) % (SOFT ? E_ : 1)] = v.z; x[(q * 4 + 3) % (SOFT ? E_ : 1)] = v.w;
                }
            }
            // without silence weights the reference multiplies by w = 1 (x * 1 == x bit for bit): the multiplies are dropped.  (That form
            // used to spill 85 registers under the 168-VGPR bound -- see the note on the running sums below -- and the kernel
            // multiplied by an opaque 1.0f instead.)
            const float wv = HAS_W ? wb[p0 + tid] : 1.0f;
            const f2 wv2 = {wv, wv};
            float4 xv4[(AMS_KM_XREG && MODE == HARD_ACC) ? V4 : 1];
            float d2[C_];
            if (SOFT) {
#pragma unroll
                for (int c = 0; c < C_; ++c) {
                    // soft distances are checked to a tolerance: two packed partial sums per cluster (even / odd e) as fused chains and the
                    // weight applied to the sum -- 41 instead of 100 vector instructions per cluster
                    f2 dk = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < V2; ++q) {
                        const f2 xv = {x[(2 * q) % (SOFT ? E_ : 1)], x[(2 * q + 1) % (SOFT ? E_ : 1)]};
                        const f2 cv = *reinterpret_cast<const f2*>(&scent[c * E_ + 2 * q]);
                        const f2 diff = xv - cv;
                        dk = __builtin_elementwise_fma(diff, diff, dk);
                    }
                    d2[c] = HAS_W ? (dk.x + dk.y) * wv : (dk.x + dk.y);
                }
            } else {
                // d2[c] = sum over e, left to right, of (x_e - c_e)^2 w with the square ROUNDED before it is weighted and added
                // (Kmeans_2.py:187: tf.square, the multiply by notsilent and reduce_sum are three ops; oracle/kmeans.py sqdist).  Two
                // clusters share a packed register: per e one v_pk_add (x_e - {c0_e, c1_e}), one v_pk_mul, (one more with weights,) one
                // v_pk_add for both -- 120 packed instructions per point at E = 40, C = 2.  (Round 5 ran this as ONE v_pk_fma chain, 80
                // instructions; that is not an evaluation of the reference's graph -- tests/test_kmeans_distance_forms.py -- and went.)
                auto dist = [&](auto WT) {
                    constexpr bool W = decltype(WT)::value;
                    constexpr int CP = (C_ + 1) / 2;
                    f2 dp[CP];
#pragma unroll
                    for (int cp = 0; cp < CP; ++cp) dp[cp] = (f2){0.f, 0.f};
                    // the point's groups come from LDS AMS_KM_DIST_DEPTH ahead of the group being consumed: read just in time, each of the
                    // ten groups put a full LDS round trip (~120 cycles) in front of its eight dependent instructions -- with three waves
                    // per SIMD that latency, not issue, was the pass (2100 cycles per 64 point-tries against ~740 of issue)
                    constexpr int DD = AMS_KM_DIST_DEPTH;
                    float4 vd[DD + 1];
#pragma unroll
                    for (int d = 0; d < DD; ++d) vd[d] = *reinterpret_cast<const float4*>(xrow + (d < V4 ? d : V4 - 1) * 4);
#pragma unroll
                    for (int q4 = 0; q4 < V4; ++q4) {
                        asm volatile("" ::: "memory");              // no wholesale preload of the point into VGPRs
                        // the differences of all ten groups are independent while a pair's chain is serial: left alone the scheduler
                        // computes every difference first (80 live registers, spilled under the 168-VGPR bound) -- the chains are DUE here
#pragma unroll
                        for (int cp = 0; cp < CP; ++cp) asm volatile("" : "+v"(dp[cp]));
                        if (q4 + DD < V4) vd[(q4 + DD) % (DD + 1)] = *reinterpret_cast<const float4*>(xrow + (q4 + DD) * 4);
                        const float4 v = vd[q4 % (DD + 1)];
                        if (AMS_KM_XREG && MODE == HARD_ACC) xv4[q4 % ((AMS_KM_XREG && MODE == HARD_ACC) ? V4 : 1)] = v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xe = k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
                            const f2 xx = {xe, xe};
#pragma unroll
                            for (int cp = 0; cp < CP; ++cp) {
                                const int c0 = 2 * cp, c1 = (2 * cp + 1 < C_) ? 2 * cp + 1 : 2 * cp;
#if AMS_KM_SGPR_CENT
                                const f2 cc = {cs[(c0 * E_ + 4 * q4 + k) % (SOFT ? 1 : C_ * E_)], cs[(c1 * E_ + 4 * q4 + k) % (SOFT ? 1 : C_ * E_)]};
#else
                                const f2 cc = {scent[c0 * E_ + 4 * q4 + k], scent[c1 * E_ + 4 * q4 + k]};
#endif
                                const f2 df = xx - cc;
                                const f2 sq = df * df;                           // rounded on its own (contraction is off in this unit)
                                dp[cp] = dp[cp] + (W ? sq * wv2 : sq);
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C_; ++c) d2[c] = (c & 1) ? dp[c / 2].y : dp[c / 2].x;
                };
                // (a 0/1 weight could be folded into one multiply per point -- w * sum == sum of w * terms bit for bit -- with the
                // per-term loop kept for other values; both loops in one kernel spill 21-72 VGPRs under the 168 bound and the
                // pass gets slower, 3.54 vs 3.39 ms per 10 x 10 run: measured, not used)
                if (HAS_W) dist(std::true_type{}); else dist(std::false_type{});
            }
            if (!SOFT) {
                int lab = 0;
                float best = sqrtf(d2[0]);
#pragma unroll
                for (int c = 1; c < C_; ++c) {
                    const float dc = sqrtf(d2[c]);
                    if (dc < best) { best = dc; lab = c; }
                }
                if (MODE == HARD_ACC) {
                    f2 mm[C_];
#pragma unroll
                    for (int c = 0; c < C_; ++c) {
                        const float m = (lab == c) ? 1.0f : 0.0f;
                        mm[c] = (f2){m, m};
                        acc[C_ * E_ + c] = __fadd_rn(acc[C_ * E_ + c], m);
                    }
                    auto accum = [&](auto WT) {
                        constexpr bool W = decltype(WT)::value;
                        // ACC_DEPTH vector groups of the point are in flight ahead of the one being accumulated: the distance
                        // temporaries are dead here, so the extra registers are free, and a group's 4 packed FMAs (16 cycles)
                        // no longer wait out a full LDS round trip each
                        constexpr int D = AMS_KM_ACC_DEPTH;
                        float4 vq[D + 1];
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            asm volatile("" ::: "memory");
                            if (!AMS_KM_XREG) vq[d] = *reinterpret_cast<const float4*>(xrow + (d < V4 ? d : V4 - 1) * 4);
                        }
#pragma unroll
                        for (int q4 = 0; q4 < V4; ++q4) {
                            asm volatile("" ::: "memory");
                            if (!AMS_KM_XREG && q4 + D < V4) vq[(q4 + D) % (D + 1)] = *reinterpret_cast<const float4*>(xrow + (q4 + D) * 4);
                            const float4 v = AMS_KM_XREG ? xv4[q4 % (AMS_KM_XREG ? V4 : 1)] : vq[q4 % (D + 1)];
                            f2 t0 = {v.x, v.y}, t1 = {v.z, v.w};
                            if (W) { t0 = t0 * wv2; t1 = t1 * wv2; }
#pragma unroll
                            for (int c = 0; c < C_; ++c) {
                                f2 a0 = {acc[c * E_ + 4 * q4], acc[c * E_ + 4 * q4 + 1]};
                                f2 a1 = {acc[c * E_ + 4 * q4 + 2], acc[c * E_ + 4 * q4 + 3]};
                                // m is 0 or 1, so t * m is exact and the fused form rounds exactly like multiply-then-add
                                a0 = __builtin_elementwise_fma(t0, mm[c], a0);
                                a1 = __builtin_elementwise_fma(t1, mm[c], a1);
                                acc[c * E_ + 4 * q4] = a0.x; acc[c * E_ + 4 * q4 + 1] = a0.y;
                                acc[c * E_ + 4 * q4 + 2] = a1.x; acc[c * E_ + 4 * q4 + 3] = a1.y;
                            }
                        }
                    };
                    if (HAS_W) accum(std::true_type{}); else accum(std::false_type{});
                } else if (MODE == HARD_LABELS) {
                    a.labels[(long)r * a.L + p0 + tid] = lab;
                } else {
                    // inertia terms: unweighted distance to the assigned centroid (Kmeans_2.py:131-136)
                    float dist = 0.f;
#pragma unroll
                    for (int c = 0; c < C_; ++c) {
                        if (lab == c) {
                            float d = 0.f;
#pragma unroll
                            for (int e = 0; e < E_; ++e) {
                                const float diff = __fsub_rn(xrow[e], scent[c * E_ + e]);
                                d = __fadd_rn(d, __fmul_rn(diff, diff));
                            }
                            dist = d;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C_; ++c) {
                        const float m = (lab == c) ? 1.0f : 0.0f;
                        acc[c] = __fadd_rn(acc[c], __fmul_rn(dist, m));
                        acc[C_ + c] = __fadd_rn(acc[C_ + c], m);
                    }
                    if (a.labels) a.labels[(long)r * a.L + p0 + tid] = lab;
                }
            } else {
                float ex[C_], sum = 0.f;
#pragma unroll
                for (int c = 0; c < C_; ++c) { ex[c] = expf(-a.beta * d2[c]); sum += ex[c]; }
                const float inv = 1.0f / sum;
                if (MODE == SOFT_ACC) {
#pragma unroll
                    for (int c = 0; c < C_; ++c) {
                        const float lb = ex[c] * inv;
                        // x (w lab) as packed FMAs (the soft modes are checked to a tolerance, not to the bit): as (x w) lab with separate
                        // multiplies and adds the accumulation was 240 of the ~480 vector instructions per point
                        const float wl = wv * lb;
                        const f2 wl2 = {wl, wl};
#pragma unroll
                        for (int q = 0; q < V2; ++q) {
                            f2 aq = {acc[c * E_ + 2 * q], acc[c * E_ + 2 * q + 1]};
                            const f2 xq = {x[(2 * q) % (SOFT ? E_ : 1)], x[(2 * q + 1) % (SOFT ? E_ : 1)]};
                            aq = __builtin_elementwise_fma(xq, wl2, aq);
                            acc[c * E_ + 2 * q] = aq.x; acc[c * E_ + 2 * q + 1] = aq.y;
                        }
                        acc[C_ * E_ + c] += lb;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C_; ++c) {
                        const float lb = ex[c] * inv;
                        float d = 0.f;
#pragma unroll
                        for (int e = 0; e < E_; ++e) { const float diff = x[e] - scent[c * E_ + e]; d += diff * diff; }
                        acc[c] += d * lb;
                        acc[C_ + c] += lb;
                        if (a.soft) a.soft[((long)r * a.L + p0 + tid) * C_ + c] = lb;
                    }
                }
            }
        }
    }
    constexpr int NPW = SOFT ? 1 : 4;                              // partial rows per workgroup
    if (MODE == HARD_LABELS) return;
    if (!SOFT) {
        // HARD modes: every wavefront is a partial of its own -- lanes combined by the halving tree l += l + s (s = 32 .. 1) on the VALU
        // (gfx950's v_permlane32_swap / v_permlane16_swap bring the upper half / the odd 16-lane rows down, row_shl DPP does the rest);
        // the four totals are added by the finisher in wavefront order.  No LDS, no barrier, all four waves busy.
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float v = acc[i];
            v = __fadd_rn(v, lane_above<32>(v));
            v = __fadd_rn(v, lane_above<16>(v));
            v = __fadd_rn(v, lane_above<8>(v));
            v = __fadd_rn(v, lane_above<4>(v));
            v = __fadd_rn(v, lane_above<2>(v));
            v = __fadd_rn(v, lane_above<1>(v));
            if (lane == 0) {
                float* const dst = a.part + (((long)r * a.G + g) * 4 + wave) * NV + i;
                if (a.tickets) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through: read by the finisher
                else *dst = v;
            }
        }
        if (a.tickets) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partial is acknowledged before the barrier below
    } else {
    // SOFT modes: halving tree over the 256 lanes: s = 128, 64 through LDS (in batches of <= 64 values, VALUE-major so the 64 lanes of a
    // wave touch 64 consecutive words -- lane-major put every lane on one bank: 64-way conflicts), s = 32..1 by shuffles.
    // NV is a compile-time constant, so both loops unroll fully and `acc` stays in registers.
#pragma unroll
    for (int v0 = 0; v0 < NV; v0 += 64) {
        __syncthreads();
        if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (v0 + i < NV) buf[i * 128 + (wave - 2) * 64 + lane] = acc[v0 + i];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (v0 + i < NV) acc[v0 + i] = __fadd_rn(acc[v0 + i], buf[i * 128 + wave * 64 + lane]);
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (v0 + i < NV) buf[i * 64 + lane] = acc[v0 + i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i)
                if (v0 + i < NV) acc[v0 + i] = __fadd_rn(acc[v0 + i], buf[i * 64 + lane]);
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float v = acc[i];
            v = __fadd_rn(v, lane_above<32>(v));
            v = __fadd_rn(v, lane_above<16>(v));
            v = __fadd_rn(v, lane_above<8>(v));
            v = __fadd_rn(v, lane_above<4>(v));
            v = __fadd_rn(v, lane_above<2>(v));
            v = __fadd_rn(v, lane_above<1>(v));
            if (lane == 0) {
                float* const dst = a.part + ((long)r * a.G + g) * NV + i;
                if (a.tickets) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through: read by the finisher
                else *dst = v;
            }
        }
    }
    }
    if (a.tickets == nullptr) return;
    // the partials leave with agent-scope stores and the arrival is counted once they are acknowledged (workgroup-scope release =
    // s_waitcnt vmcnt(0); NOT __threadfence(): csrc/dpcl.hip); the last chunk's workgroup finishes the row
    __shared__ int last_sh;
    if (!SOFT) __syncthreads();                                    // all four wavefronts' partials are out
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int last = atomicAdd(a.tickets + r, 1u) == (unsigned)a.G - 1u;
        if (last) __hip_atomic_store(a.tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = last;
    }
    __syncthreads();
    if (!last_sh) return;
    const int NP = a.G * NPW;
    const float* pr = a.part + (long)r * NP * NV;
    // chunk partials fetched EIGHT AT A TIME and then added in chunk order: as one dependent chain of agent-scope loads per element the
    // finish put ~20 us at the end of every pass (225 us instead of 199 + a 5-us reduce launch)
    auto chunk_sum = [&](int k) {
        float s = 0.f;
        int gg = 0;
        for (; gg + 8 <= NP; gg += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(pr + (long)(gg + j) * NV + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 8; ++j) s = __fadd_rn(s, v[j]);
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (gg + j < NP) ? __hip_atomic_load(pr + (long)min(gg + j, NP - 1) * NV + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (gg + j < NP) s = __fadd_rn(s, v[j]);
        return s;
    };
    if (ACC) {
        if (tid < C_ * E_) {
            const int c = tid / E_;
            const float num = chunk_sum(tid), den = chunk_sum(C_ * E_ + c);
            a.fin_out[(long)r * C_ * E_ + tid] = ((num) / (den));
            if (a.fin_den && (tid % E_) == 0) a.fin_den[(long)r * C_ + c] = den;
        }
    } else if (tid < C_ && a.fin_out) {
        const float tot = chunk_sum(tid), cnt = chunk_sum(C_ + tid);
        __shared__ float q_sh[C_];
        q_sh[tid] = ((tot) / (cnt));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
            float inertia = 0.f;
#pragma unroll
            for (int c = 0; c < C_; ++c) inertia = __fadd_rn(inertia, q_sh[c]);
            a.fin_out[r] = inertia;
        }
    }
}

// Sum the G partial rows of a row in order; ACC passes: centroid = num / den.  FINAL passes: inertia[r] = sum_c tot_c/cnt_c.
__global__ void kmeans_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ den_out, int R,
                                     int G, int C, int E, int final_) {
    const int NV = final_ ? 2 * C : C * (E + 1);
    const int per = final_ ? 1 : C * E;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * per) return;
    const int r = (int)(i / per), k = (int)(i - (long)r * per);
    const float* pr = part + (long)r * G * NV;
    if (final_) {
        float inertia = 0.f;
        for (int c = 0; c < C; ++c) {
            float tot = 0.f, cnt = 0.f;
            for (int g = 0; g < G; ++g) { tot = __fadd_rn(tot, pr[(long)g * NV + c]); cnt = __fadd_rn(cnt, pr[(long)g * NV + C + c]); }
            inertia = __fadd_rn(inertia, ((tot) / (cnt)));
        }
        out[r] = inertia;
    } else {
        const int c = k / E;
        float num = 0.f, den = 0.f;
        for (int g = 0; g < G; ++g) { num = __fadd_rn(num, pr[(long)g * NV + k]); den = __fadd_rn(den, pr[(long)g * NV + C * E + c]); }
        out[(long)r * C * E + k] = ((num) / (den));
        if (den_out && (k % E) == 0) den_out[(long)r * C + c] = den;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// HARD_ACC for E = 40, C = 2, tries a multiple of 5 (HAS_W: with silence weights): TQ tries of an utterance from ONE read of its points.
//
// kmeans_pass_kernel gives every try a workgroup of its own: the point is staged and read per try (10 x the LDS traffic and L2 reads of
// the utterance) and each lane carries the 82 running sums of ITS points -- 3 waves per SIMD, a latency mix at 0.13 of HBM
// (profiles/r05_i_cfg_*: 205-210 us per pass).  Here a workgroup of TEN waves owns one 64-lane column of a chunk (the unit of the
// summation order, see the top of the file) for TQ = 5 tries, and every wave has two roles per iteration of two 64-point slabs:
//   labels   wave (try tt = wave % 5, slab k = wave / 5): lane = point, the point's 40 values from LDS, the try's centroids in SGPRs
//            for the whole launch, the packed distance chains of kmeans_pass_kernel, argmin -> ONE 64-bit ballot per (slab, try) in LDS;
//   sums     wave q owns components 4q .. 4q+3: its float4 of the point stays in registers and is accumulated into 5 tries x 2
//            clusters x 4 running sums (40 registers instead of 82) under the ballots of the previous iteration, as v_pk_fma with a
//            0/1 factor exactly like kmeans_pass_kernel; wave q < 5 also keeps try q's two counts.
// Same operations on the same operands in the same order per running sum as kmeans_pass_kernel => bit-identical partials; the
// centroids are read once per workgroup, the points once per 5 tries (420 MB per pass at the benchmark shape instead of 2.1 GB).
// One barrier per iteration: labels of iteration i and sums of iteration i-1 run between the same two barriers (x and the ballots are
// double-buffered).  <= 96 VGPRs: two workgroups (20 waves) per CU.
constexpr int TQ = 5;
typedef __attribute__((address_space(4))) float kt_cfloat;                // constant address space: uniform reads become s_load
constexpr size_t KT_LDS_BYTES = 3 * 128 * 40 * sizeof(float);      // kmeans_hard_tries_kernel's x buffers (dynamic LDS)
struct KtArgs {
    const float* xn; const float* cent; float* part; unsigned* tickets; float* fin_out; float* fin_den;
    long L; int b, tries, G;
    int32_t* labels;                   // kmeans_hard_tries_final_kernel: [R, L] or null
    const float* w; int w_mod_b;       // silence weights [b, L] or null; row of try r: r % b (the reference's tile quirk) or r / tries
    unsigned long long* dbg;           // AMS_KT_DBG builds: per (workgroup, wave) {HW_ID | XCC_ID << 32, start, end} (s_memrealtime)
};

// one component of the packed distance chains: df = {x - c0, x - c1} with x broadcast from the low (KT_LO) or high (KT_HI) half of a
// register pair and {c0, c1} a scalar pair; sq = df * df ROUNDED (tf.square is an op of its own, Kmeans_2.py:187); d <- d + sq
#define KT_LO "op_sel_hi:[0,1]"
#define KT_HI "op_sel:[1,0] op_sel_hi:[1,1]"
#define KT_DIST(D, X, SEL, C) do { f2 df_; asm("v_pk_add_f32 %1, %2, %3 " SEL " neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %1, %1, %1\n\t" \
                                              "v_pk_add_f32 %0, %0, %1" : "+v"(D), "=&v"(df_) : "v"(X), "s"(C)); } while (0)

// the same with silence weights: d <- d + (df * df) * w (w = {w, w}); KT_DIST2W also q <- q + df * df, the unweighted inertia distance
#define KT_DISTW(D, X, SEL, C, W) do { f2 df_; asm("v_pk_add_f32 %1, %2, %3 " SEL " neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %1, %1, %1\n\t" \
                                                  "v_pk_mul_f32 %1, %1, %4\n\tv_pk_add_f32 %0, %0, %1" : "+v"(D), "=&v"(df_) : "v"(X), "s"(C), "v"(W)); } while (0)
#define KT_DIST2W(D, Q, X, SEL, C, W) do { f2 df_; asm("v_pk_add_f32 %2, %3, %4 " SEL " neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_mul_f32 %2, %2, %2\n\t" \
                                                      "v_pk_add_f32 %1, %1, %2\n\tv_pk_mul_f32 %2, %2, %5\n\tv_pk_add_f32 %0, %0, %2" \
                                                      : "+v"(D), "+v"(Q), "=&v"(df_) : "v"(X), "s"(C), "v"(W)); } while (0)

__device__ __forceinline__ float mask_to_float(unsigned long long m) {       // 1.0f in the lanes whose bit of the (wave-uniform) mask is set
    float f;
    // the "s" constraint does not make a value scalar: pinned with readfirstlane (folded away where the compiler knows it is uniform)
    const unsigned long long ms = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)m);
    asm volatile("v_cndmask_b32_e64 %0, 0, 1.0, %1" : "=v"(f) : "s"(ms));
    return f;
}

// Halving trees l += l + s (s = 32 .. 1) of FOUR values with the additions of four lane_above trees: v_permlane32_swap exchanges the upper
// half of a with the lower half of b, so ONE add does s = 32 for both; v_permlane16_swap (odd 16-lane rows of the first <-> even rows of
// the second) does the same for s = 16 on the two sums; s = 8 .. 1 run inside the 16-lane rows on one register.  3 swaps + 7 adds for
// what was 8 swaps + 24 adds.  The totals of a / c / b / d are in lanes 0 / 16 / 32 / 48.
__device__ __forceinline__ float tree4(float a, float b, float c, float d) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float ab = __fadd_rn(a, b);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    float cd = __fadd_rn(c, d);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ab), "+v"(cd));
    float v = __fadd_rn(ab, cd);
    v = __fadd_rn(v, lane_above<8>(v));
    v = __fadd_rn(v, lane_above<4>(v));
    v = __fadd_rn(v, lane_above<2>(v));
    v = __fadd_rn(v, lane_above<1>(v));
    return v;
}

#ifndef AMS_KT_WAVES
#define AMS_KT_WAVES 6
#endif
template <bool HAS_W>
__global__ __launch_bounds__(640, AMS_KT_WAVES) void kmeans_hard_tries_kernel(KtArgs a) {
    constexpr int E_ = 40, C_ = 2, NV = C_ * (E_ + 1), LD = E_, V4 = E_ / 4, SL = CHUNK_HARD / LANES;
    // x is TRIPLE-buffered: the labels read slab pair `it`, the sums re-read their components of pair it - 1 (no copy kept in registers),
    // the staging writes pair it + 1 -- all between the same two barriers.  Rows are UNPADDED (3 x 20 KB: two workgroups fit beside each
    // other with room for the 0/1 factors and the weights) and the 16-byte groups of a row are rotated by one for rows 8..15 of every 16: group c of
    // row p sits at slot (c + ((p >> 3) & 1)) % 10, so that 16 lanes reading the same group of 16 consecutive rows (160 B apart: 10 p
    // mod 16 takes only the 8 even values) still cover all 64 banks.
    // (dynamic: with the size in sight hipcc sees that six waves per SIMD cannot be reached, settles for five and spends 87-99 VGPRs;
    // the workgroup's ten waves sit 3 + 3 + 2 + 2 on the SIMDs, so a second workgroup needs SIX slots on a SIMD: <= 80 VGPRs)
    extern __shared__ __attribute__((aligned(16))) float kt_dyn[];
    float (*xbuf)[128 * LD] = reinterpret_cast<float (*)[128 * LD]>(kt_dyn);
    __shared__ float mf[2][2][TQ][64];                             // 1.0 where (slab, try, lane) chose cluster 1 and is a point, else 0.0
    __shared__ float wf[HAS_W ? 2 : 1][2][TQ][HAS_W ? 64 : 1];    // silence weights: the weight of (slab, try, lane) (every try may have its own row)
    __shared__ int cbuf[TQ][C_];
    __shared__ int last_sh[TQ];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef AMS_KT_DBG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();                    // placement / lifetime probe (tools/probes/kt_probe.py)
#endif
    // workgroup -> (utterance, try group, column, chunk), the chunk slowest: a short last chunk's workgroups come last
    int id = blockIdx.x;
    const int ntg = a.tries / TQ;
    const int ub = id % a.b; id /= a.b;
    const int tg = id % ntg; id /= ntg;
    const int k4 = id & 3, g = id >> 2;
    const float* xb = a.xn + (long)ub * a.L * E_;
    const long base = (long)g * CHUNK_HARD + k4 * 64;              // slab i of this column: points base + 256 i .. + 63
    const long left = a.L - base;
    const int nsl = left <= 0 ? 0 : (int)min((long)SL, (left + LANES - 1) / LANES);
    const int nit = (nsl + 1) / 2;
    const int row0 = ub * a.tries + tg * TQ;
    const int left32 = (int)min(max(left, (long)0), (long)(SL * LANES));
    auto valid_of = [&](int slab) {                                 // lanes of a slab that are points of the utterance
        const int nv = left32 - slab * LANES;
        return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
    };

    // label role
    const int tt = wave % TQ, kk = wave / TQ;
    // the try's centroids as 40 scalar PAIRS {c0_e, c1_e}, fetched with scalar loads (uniform address, written by the previous launch):
    // as 80 vector loads + v_readfirstlane they were 160 vector instructions per wave in every workgroup's prologue
    unsigned long long cpair[E_];
    {
        const kt_cfloat* cg = (const kt_cfloat*)(a.cent + (long)(row0 + tt) * C_ * E_);
#pragma unroll
        for (int e = 0; e < E_; ++e) cpair[e] = ((unsigned long long)__float_as_uint(cg[E_ + e]) << 32) | __float_as_uint(cg[e]);
    }
    // the counts are integers (<= 2048 per column and chunk): every order of adding them in float gives the same float, so the label
    // wave counts the bits of its ballots on the scalar unit instead of the sums role adding 0/1 factors lane by lane
    int n0 = 0, n1 = 0;
    // silence weights of this wave's (try, slab parity): requested one iteration ahead, like x (zeros past L: bounds-checked buffer loads)
    __amdgpu_buffer_rsrc_t wrs;
    float w_nxt = 1.0f;
    if constexpr (HAS_W) {
        const int r1 = row0 + tt;
        wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w + (long)(a.w_mod_b ? (r1 % a.b) : ub) * a.L), (short)0, (int)(a.L * 4), 0x00020000);
        w_nxt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (unsigned)((base + (long)kk * LANES + lane) * 4), 0, 0));
    }
    // sums role
    float acc[TQ][C_][4];
#pragma unroll
    for (int t = 0; t < TQ; ++t)
#pragma unroll
        for (int c = 0; c < C_; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][c][j] = 0.f;

    // staging: thread -> (point pr of a slab, 16-byte group c4) of BOTH slabs of an iteration
    const int pr = tid / V4, c4 = tid - pr * V4;
    // the utterance as a buffer resource: 32-bit offsets, and rows past L come back as zeros without a test (raw buffer bounds check)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, (int)(a.L * E_ * 4), 0x00020000);
    float4 pf[2];
    unsigned foff = (unsigned)((base + pr) * E_ + c4 * 4) * 4u;   // byte offset of this thread's group in slab 0 of the column
    auto fetch = [&](int it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            pf[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, foff + (unsigned)((2 * it + k) * LANES * E_ * 4), 0, 0));
    };
    const int c4s = (c4 + ((pr >> 3) & 1)) % V4;                   // this group's slot in its row
    auto stage = [&](int bf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<float4*>(&xbuf[bf][(k * 64 + pr) * LD + c4s * 4]) = pf[k];
    };
    // reads: lane = row; group c at slot c + f (f = 0 / 1) except group 9 of the rotated rows, which wrapped to slot 0
    const int rot = (lane >> 3) & 1;
    const int off_lab = (kk * 64 + lane) * LD + rot * 4;            // + 4 c for c < 9
    const int off_lab9 = (kk * 64 + lane) * LD + (rot ? 0 : 36);
    const int off_sum = lane * LD + ((wave + rot) % V4) * 4;        // + 64 LD k
    if (nit > 0) {
        fetch(0);
        stage(0);
        if (nit > 1) fetch(1);
    }
    __syncthreads();
    int bprev = 2, bcur = 0, bnext = 1;
    for (int it = 0; it <= nit; ++it) {
        const int cur = it & 1;
        if (it > 0) {
            // ---- sums of iteration it - 1 under its labels: the 0/1 factors come from LDS as floats (as ballots they cost two v_readlane
            // and two v_cndmask per try and slab on the VALU, which is what bounds this kernel)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float vf = mask_to_float(valid_of(2 * (it - 1) + k));
                const float4 xs = *reinterpret_cast<const float4*>(&xbuf[bprev][off_sum + k * 64 * LD]);
                float m1[TQ], wt[HAS_W ? TQ : 1];
#pragma unroll
                for (int t = 0; t < TQ; ++t) m1[t] = mf[cur ^ 1][k][t][lane];
                if constexpr (HAS_W) {
#pragma unroll
                    for (int t = 0; t < TQ; ++t) wt[t] = wf[cur ^ 1][k][t][lane];
                }
                const f2 x0 = {xs.x, xs.y}, x1 = {xs.z, xs.w};
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    f2 t0 = x0, t1 = x1;
                    if constexpr (HAS_W) { const f2 w2 = {wt[t], wt[t]}; t0 = x0 * w2; t1 = x1 * w2; }      // x w first (rounded), as kmeans_pass_kernel
                    // {m0, m1} in ONE register pair: the packed FMAs of cluster c take component c for both halves (op_sel)
                    const f2 mm = {vf - m1[t], m1[t]};
                    const f2 mm0 = __builtin_shufflevector(mm, mm, 0, 0), mm1 = __builtin_shufflevector(mm, mm, 1, 1);
                    f2 a00 = {acc[t][0][0], acc[t][0][1]}, a01 = {acc[t][0][2], acc[t][0][3]};
                    f2 a10 = {acc[t][1][0], acc[t][1][1]}, a11 = {acc[t][1][2], acc[t][1][3]};
                    // m is 0 or 1, so t * m is exact and the fused form rounds exactly like multiply-then-add (kmeans_pass_kernel)
                    a00 = __builtin_elementwise_fma(t0, mm0, a00); a01 = __builtin_elementwise_fma(t1, mm0, a01);
                    a10 = __builtin_elementwise_fma(t0, mm1, a10); a11 = __builtin_elementwise_fma(t1, mm1, a11);
                    asm volatile("" : "+v"(a00), "+v"(a01), "+v"(a10), "+v"(a11));      // due HERE: left alone they sink below the labels
                    acc[t][0][0] = a00.x; acc[t][0][1] = a00.y; acc[t][0][2] = a01.x; acc[t][0][3] = a01.y;
                    acc[t][1][0] = a10.x; acc[t][1][1] = a10.y; acc[t][1][2] = a11.x; acc[t][1][3] = a11.y;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                          // sums and labels are independent streams: interleaved they need both register sets
        if (it < nit) {
            // ---- labels of (slab kk, try tt): d <- d + round((x_e - c_e)^2), both clusters in one packed register
            const float wv = w_nxt;
            const f2 wv2 = {wv, wv};
            if constexpr (HAS_W)                                    // (clamped inside the chunk's slabs: past them nothing is consumed)
                w_nxt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (unsigned)((base + (long)(2 * (it + 1) + kk) * LANES + lane) * 4), 0, 0));
            const float* xrow = &xbuf[bcur][off_lab];
            f2 dp = {0.f, 0.f};
            float4 vd[2];
            vd[0] = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
            for (int q4 = 0; q4 < V4; ++q4) {
                asm volatile("" ::: "memory");                      // no wholesale preload of the point into VGPRs
                asm volatile("" : "+v"(dp));
                if (q4 + 1 < V4) vd[(q4 + 1) & 1] = *reinterpret_cast<const float4*>(q4 + 1 < V4 - 1 ? xrow + (q4 + 1) * 4 : &xbuf[bcur][off_lab9]);
                const float4 v = vd[q4 & 1];
                // one component of both clusters' chains: the point's value broadcast by op_sel from its place in the float4 (left to
                // the compiler, every second broadcast was a v_mov), the centroid pair a scalar operand
                const f2 xlo = {v.x, v.y}, xhi = {v.z, v.w};
                if constexpr (HAS_W) {
                    KT_DISTW(dp, xlo, KT_LO, cpair[4 * q4 + 0], wv2); KT_DISTW(dp, xlo, KT_HI, cpair[4 * q4 + 1], wv2);
                    KT_DISTW(dp, xhi, KT_LO, cpair[4 * q4 + 2], wv2); KT_DISTW(dp, xhi, KT_HI, cpair[4 * q4 + 3], wv2);
                } else {
                    KT_DIST(dp, xlo, KT_LO, cpair[4 * q4 + 0]); KT_DIST(dp, xlo, KT_HI, cpair[4 * q4 + 1]);
                    KT_DIST(dp, xhi, KT_LO, cpair[4 * q4 + 2]); KT_DIST(dp, xhi, KT_HI, cpair[4 * q4 + 3]);
                }
            }
            const bool one = sqrtf(dp.y) < sqrtf(dp.x);            // ties pick cluster 0 (tf.argmin)
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(one);
            const unsigned long long valid = valid_of(2 * it + kk);
            mf[cur][kk][tt][lane] = mask_to_float(bal & valid);
            if constexpr (HAS_W) wf[cur][kk][tt][lane] = wv;
            n1 += __builtin_popcountll(bal & valid);
            n0 += __builtin_popcountll(~bal & valid);
            // ---- stage the next iteration's slabs, request the ones after it
            if (it + 1 < nit) {
                stage(bnext);
                if (it + 2 < nit) fetch(it + 2);
            }
        }
        __syncthreads();
        const int bt = bprev; bprev = bcur; bcur = bnext; bnext = bt;
    }

    // every wave's lanes by the halving tree, FOUR running sums per tree (tree4): components 4 wave .. + 3 of the TQ rows; always
    // write-through stores (the finisher may sit on another XCD)
    // The output pointers are read from the kernel-argument segment only HERE: held in SGPRs across the loop beside the 80 centroid
    // scalars they were spilled, and hipcc then shuffled forty SGPR pairs per iteration to make room for the buffer resource.
    const __attribute__((address_space(4))) KtArgs* ka = (const __attribute__((address_space(4))) KtArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    float* const a_part = ka->part; unsigned* const a_tickets = ka->tickets; float* const a_fin_out = ka->fin_out;
    float* const a_fin_den = ka->fin_den;
#ifdef AMS_KT_DBG
    unsigned long long* const a_dbg = ka->dbg;
#endif
    const int NP = 4 * ka->G, pi = g * 4 + k4;
    const int jl = ((lane >> 4) & 1) * 2 + (lane >> 5);            // lanes 0 / 16 / 32 / 48 hold components 0 / 2 / 1 / 3 of a group
#pragma unroll
    for (int t = 0; t < TQ; ++t)
#pragma unroll
        for (int c = 0; c < C_; ++c) {
            const float v = tree4(acc[t][c][0], acc[t][c][1], acc[t][c][2], acc[t][c][3]);
            if ((lane & 15) == 0)
                __hip_atomic_store(a_part + ((long)(row0 + t) * NP + pi) * NV + c * E_ + wave * 4 + jl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef AMS_KT_DBG
    if (a_dbg && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = a_dbg + ((long)blockIdx.x * 10 + wave) * 4;
        d[0] = hw | ((unsigned long long)xcc << 32);
        d[1] = dbg_t0;
        d[2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // counts of try tt: slab parity 1's wave hands its bits to parity 0's, which stores the totals
    if (kk == 1 && lane == 0) { cbuf[tt][0] = n0; cbuf[tt][1] = n1; }
    __syncthreads();
    if (kk == 0 && lane < C_) {
        const int tot = (lane == 0 ? n0 : n1) + cbuf[tt][lane];
        __hip_atomic_store(a_part + ((long)(row0 + tt) * NP + pi) * NV + C_ * E_ + lane, (float)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a_tickets == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's partials are acknowledged
    __syncthreads();
    if (tid < TQ) {
        const int r = row0 + tid;
        const int last = atomicAdd(a_tickets + r, 1u) == (unsigned)NP - 1u;
        if (last) __hip_atomic_store(a_tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh[tid] = last;
    }
    __syncthreads();
    // rows whose last partial this workgroup stored are finished here: 80 threads per row, partial rows in order, eight loads at a time
    const int ft = tid / (C_ * E_), fk = tid - ft * (C_ * E_);
    if (ft >= TQ || !last_sh[ft]) return;
    const int r = row0 + ft;
    const float* prr = a_part + (long)r * NP * NV;
    auto chunk_sum = [&](int k) {
        float s = 0.f;
        for (int gg = 0; gg < NP; gg += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(prr + (long)min(gg + j, NP - 1) * NV + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (gg + j < NP) s = __fadd_rn(s, v[j]);
        }
        return s;
    };
    const int c = fk / E_;
    const float num = chunk_sum(fk), den = chunk_sum(C_ * E_ + c);
    a_fin_out[(long)r * C_ * E_ + fk] = ((num) / (den));
    if (a_fin_den && (fk % E_) == 0) a_fin_den[(long)r * C_ + c] = den;
}

// HARD_FINAL in the same shape (E = 40, C = 2, tries a multiple of 5; HAS_W: the labels are weighted, the inertia distance is not): labels and inertia terms of TQ tries from one
// read of the points.  No sums role: wave (try tt, kk) owns COLUMN 2 cp + kk of the chunk for all its slabs, so its lanes' running sums
// tot_c += dist * [label == c] follow the summation order on their own; one slab of both columns (128 consecutive points) per iteration,
// x double-buffered, one barrier.  dist = sum_e (x_e - c_e)^2 of the assigned centroid with separate multiply and add (oracle
// inertia_hard), both clusters packed beside the (weighted) label chain, which shares the squares.  Counts by popcount (integers).
template <bool HAS_W>
__global__ __launch_bounds__(640, 6) void kmeans_hard_tries_final_kernel(KtArgs a) {
    constexpr int E_ = 40, C_ = 2, LD = E_, V4 = E_ / 4, SL = CHUNK_HARD / LANES, NVF = 2 * C_;
    extern __shared__ __attribute__((aligned(16))) float kt_dyn[];
    float (*xbuf)[128 * LD] = reinterpret_cast<float (*)[128 * LD]>(kt_dyn);
    __shared__ int last_sh[TQ];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int id = blockIdx.x;
    const int ntg = a.tries / TQ;
    const int ub = id % a.b; id /= a.b;
    const int tg = id % ntg; id /= ntg;
    const int cp = id & 1, g = id >> 1;
    const float* xb = a.xn + (long)ub * a.L * E_;
    const long base = (long)g * CHUNK_HARD + cp * 128;             // slab i of the column pair: points base + 256 i .. + 127
    const long left = a.L - base;
    const int nit = left <= 0 ? 0 : (int)min((long)SL, (left + LANES - 1) / LANES);
    const int left32 = (int)min(max(left, (long)0), (long)(SL * LANES));
    const int row0 = ub * a.tries + tg * TQ;
    const int tt = wave % TQ, kk = wave / TQ;
    auto valid_of = [&](int slab) {
        const int nv = left32 - slab * LANES - kk * 64;
        return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
    };
    // the try's centroids as 40 scalar PAIRS {c0_e, c1_e}, fetched with scalar loads (uniform address, written by the previous launch):
    // as 80 vector loads + v_readfirstlane they were 160 vector instructions per wave in every workgroup's prologue
    unsigned long long cpair[E_];
    {
        const kt_cfloat* cg = (const kt_cfloat*)(a.cent + (long)(row0 + tt) * C_ * E_);
#pragma unroll
        for (int e = 0; e < E_; ++e) cpair[e] = ((unsigned long long)__float_as_uint(cg[E_ + e]) << 32) | __float_as_uint(cg[e]);
    }
    int n0 = 0, n1 = 0;
    float tot0 = 0.f, tot1 = 0.f;
    __amdgpu_buffer_rsrc_t wrs;
    float w_nxt = 1.0f;
    if constexpr (HAS_W) {                                          // silence weights: the LABELS are weighted, the inertia distance is not
        const int r1 = row0 + tt;
        wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w + (long)(a.w_mod_b ? (r1 % a.b) : ub) * a.L), (short)0, (int)(a.L * 4), 0x00020000);
        w_nxt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (unsigned)((base + kk * 64 + lane) * 4), 0, 0));
    }
    const int pr = tid / V4, c4 = tid - pr * V4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, (int)(a.L * E_ * 4), 0x00020000);
    float4 pf[2];
    const unsigned foff = (unsigned)((base + pr) * E_ + c4 * 4) * 4u;
    auto fetch = [&](int it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            pf[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, foff + (unsigned)((it * LANES + k * 64) * E_ * 4), 0, 0));
    };
    const int c4s = (c4 + ((pr >> 3) & 1)) % V4;                   // rotated groups: see kmeans_hard_tries_kernel
    auto stage = [&](int bf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) *reinterpret_cast<float4*>(&xbuf[bf][(k * 64 + pr) * LD + c4s * 4]) = pf[k];
    };
    const int rot = (lane >> 3) & 1;
    const int off_lab = (kk * 64 + lane) * LD + rot * 4;
    const int off_lab9 = (kk * 64 + lane) * LD + (rot ? 0 : 36);
    if (nit > 0) {
        fetch(0);
        stage(0);
        if (nit > 1) fetch(1);
    }
    __syncthreads();
    int32_t* const lrow = a.labels ? a.labels + (long)(row0 + tt) * a.L + base + kk * 64 + lane : nullptr;
    for (int it = 0; it < nit; ++it) {
        const int cur = it & 1;
        const float wv = w_nxt;
        const f2 wv2 = {wv, wv};
        if constexpr (HAS_W)
            w_nxt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (unsigned)((base + (long)(it + 1) * LANES + kk * 64 + lane) * 4), 0, 0));
        const float* xrow = &xbuf[cur][off_lab];
        f2 dp = {0.f, 0.f}, dq = {0.f, 0.f};
        float4 vd[2];
        vd[0] = *reinterpret_cast<const float4*>(xrow);
#pragma unroll
        for (int q4 = 0; q4 < V4; ++q4) {
            asm volatile("" ::: "memory");
            asm volatile("" : "+v"(dp), "+v"(dq));
            if (q4 + 1 < V4) vd[(q4 + 1) & 1] = *reinterpret_cast<const float4*>(q4 + 1 < V4 - 1 ? xrow + (q4 + 1) * 4 : &xbuf[cur][off_lab9]);
            const float4 v = vd[q4 & 1];
            // label distance (weighted) and inertia distance (not) from the same rounded squares; without weights they are one chain
            const f2 xlo = {v.x, v.y}, xhi = {v.z, v.w};
            if constexpr (HAS_W) {
                KT_DIST2W(dp, dq, xlo, KT_LO, cpair[4 * q4 + 0], wv2); KT_DIST2W(dp, dq, xlo, KT_HI, cpair[4 * q4 + 1], wv2);
                KT_DIST2W(dp, dq, xhi, KT_LO, cpair[4 * q4 + 2], wv2); KT_DIST2W(dp, dq, xhi, KT_HI, cpair[4 * q4 + 3], wv2);
            } else {
                KT_DIST(dp, xlo, KT_LO, cpair[4 * q4 + 0]); KT_DIST(dp, xlo, KT_HI, cpair[4 * q4 + 1]);
                KT_DIST(dp, xhi, KT_LO, cpair[4 * q4 + 2]); KT_DIST(dp, xhi, KT_HI, cpair[4 * q4 + 3]);
            }
        }
        if constexpr (!HAS_W) dq = dp;
        const bool one = sqrtf(dp.y) < sqrtf(dp.x);
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(one);
        const unsigned long long valid = valid_of(it);
        n1 += __builtin_popcountll(bal & valid);
        n0 += __builtin_popcountll(~bal & valid);
        const float dist = one ? dq.y : dq.x;
        const float vf = mask_to_float(valid);
        // tot_c += dist * [label == c] for the lanes that are points (kmeans_pass_kernel skips the others: adding +0 is the same)
        const float m1 = one ? 1.0f : 0.0f, m0 = 1.0f - m1;
        const float t0v = __fmul_rn(dist, m0), t1v = __fmul_rn(dist, m1);
        tot0 = __fadd_rn(tot0, vf != 0.f ? t0v : 0.f);
        tot1 = __fadd_rn(tot1, vf != 0.f ? t1v : 0.f);
        if (lrow && vf != 0.f) lrow[(long)it * LANES] = one ? 1 : 0;
        if (it + 1 < nit) {
            stage(cur ^ 1);
            if (it + 2 < nit) fetch(it + 2);
        }
        __syncthreads();
    }
    const __attribute__((address_space(4))) KtArgs* ka = (const __attribute__((address_space(4))) KtArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    float* const a_part = ka->part; unsigned* const a_tickets = ka->tickets; float* const a_fin_out = ka->fin_out;
    const int NP = 4 * ka->G, pi = g * 4 + cp * 2 + kk;
    {
        float* const dst = a_part + ((long)(row0 + tt) * NP + pi) * NVF;
        const float v = tree4(tot0, tot1, 0.f, 0.f);               // lanes 0 / 32: tot of cluster 0 / 1
        if ((lane & 31) == 0) __hip_atomic_store(dst + (lane >> 5), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < C_) __hip_atomic_store(dst + C_ + lane, (float)(lane == 0 ? n0 : n1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a_tickets == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < TQ) {
        const int r = row0 + tid;
        const int last = atomicAdd(a_tickets + r, 1u) == (unsigned)(NP / 2) - 1u;      // one arrival per (column pair, chunk)
        if (last) __hip_atomic_store(a_tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh[tid] = last;
    }
    __syncthreads();
    // inertia[r] = sum_c tot_c / cnt_c, partial rows in order (kmeans_pass_kernel's finish): one lane pair per finished row
    const int ft = tid >> 1, c = tid & 1;
    if (ft >= TQ || !last_sh[ft]) return;
    const int r = row0 + ft;
    const float* prr = a_part + (long)r * NP * NVF;
    auto chunk_sum = [&](int k) {
        float s = 0.f;
        for (int gg = 0; gg < NP; gg += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(prr + (long)min(gg + j, NP - 1) * NVF + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (gg + j < NP) s = __fadd_rn(s, v[j]);
        }
        return s;
    };
    const float q = chunk_sum(c) / chunk_sum(C_ + c);
    const float q1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0x101, 0xf, 0xf, true));     // row_shl:1: lane pair's cluster 1
    if (c == 0) a_fin_out[r] = __fadd_rn(__fadd_rn(0.f, q), q1);
}

// ------------------------------------------------------------------------------------------------------------------------------
// SOFT_ACC the way csrc/kmeans_soft.hip runs its backward passes (round 5; E = 40, tolerance-checked like every soft mode):
//   * the row's centroids are SCALAR operands (s_load through the constant address space) of packed FMAs, not LDS broadcast reads
//     (40 ds_read_b64 per point in kmeans_pass_kernel's soft branch);
//   * d_c = w (|x|^2 - 2 <x, c> + |c|^2): one dot product per cluster (20 v_pk_fma) instead of subtract / multiply / add per element;
//   * sums as v_pk_fma with the factor w lab_c; slab j + 1 is requested while slab j is worked on; ONE resident round of workgroups
//     (512 / R chunks per row).
// ~120 vector instructions per point instead of ~480: 66 -> 42 us per pass at 64 rows x 20480 points (fine-tuning step 4.54 -> 4.33 ms).
struct KsfArgs {
    const float* xn; const float* w; const float* cent; float* part; unsigned* tickets; float* cent_out; float* den_out;
    long L; int b, tries, nG, spw, w_mod_b; float beta;
};
typedef const f2 __attribute__((address_space(4))) kt_cf2;

template <int C_, bool HAS_W>
__global__ __launch_bounds__(256) void kmeans_soft_acc_kernel(KsfArgs a) {
    constexpr int E_ = 40, CE = C_ * E_, NV = CE + C_, LD = E_ + 4, V4 = E_ / 4, H = E_ / 2;
    __shared__ __attribute__((aligned(16))) float buf[256 * LD];
    __shared__ int last_sh;
    const int r = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bi = r / a.tries;
    const float* xb = a.xn + (long)bi * a.L * E_;
    const float* wb = HAS_W ? a.w + (long)(a.w_mod_b ? (r % a.b) : bi) * a.L : nullptr;
    const kt_cfloat* cen = (const kt_cfloat*)(a.cent + (long)r * CE);
    kt_cf2* cen2 = (kt_cf2*)cen;
    float cc[C_];                                               // |c|^2 (uniform: scalar operands)
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        float v = 0.f;
#pragma unroll
        for (int e = 0; e < E_; ++e) v = fmaf(cen[c * E_ + e], cen[c * E_ + e], v);
        cc[c] = v;
    }
    f2 acc2[C_][H];
    float den[C_];
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        den[c] = 0.f;
#pragma unroll
        for (int q = 0; q < H; ++q) acc2[c][q] = (f2){0.f, 0.f};
    }
    float4 pre[V4];
    float wpre = 1.0f;
    auto fetch = [&](int j) {
        const long q0 = ((long)g * a.spw + j) * LANES;
        const long lim = (a.L - q0) * V4;
        const float4* src = reinterpret_cast<const float4*>(xb + q0 * E_);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const long i = tid + 256 * k;
            pre[k] = src[i < lim ? i : (lim > 0 ? lim - 1 : -q0 * V4)];       // clamped inside the utterance's rows
        }
        if (HAS_W) wpre = wb[min(q0 + tid, a.L - 1)];
    };
    fetch(0);
    for (int j = 0; j < a.spw; ++j) {
        const long p0 = ((long)g * a.spw + j) * LANES;
        const int npts = (int)max((long)0, min((long)LANES, a.L - p0));
        if (npts <= 0) break;                                   // (workgroup-uniform)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const int i = tid + 256 * k;
            const int row = i / V4, c4 = i - row * V4;
            *reinterpret_cast<float4*>(&buf[row * LD + c4 * 4]) = (i < npts * V4) ? pre[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const float wv = wpre;
        if (j + 1 < a.spw) fetch(j + 1);
        if (tid < npts) {
            f2 x2[H];
            f2 xs = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < V4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&buf[tid * LD + q * 4]);
                x2[2 * q] = (f2){v.x, v.y}; x2[2 * q + 1] = (f2){v.z, v.w};
                xs = __builtin_elementwise_fma(x2[2 * q], x2[2 * q], xs);
                xs = __builtin_elementwise_fma(x2[2 * q + 1], x2[2 * q + 1], xs);
            }
            const float xx = xs[0] + xs[1];
            float lab[C_], sum = 0.f;
#pragma unroll
            for (int c = 0; c < C_; ++c) {
                f2 sd = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < H; ++q) sd = __builtin_elementwise_fma(x2[q], (f2)cen2[c * H + q], sd);
                const float d = fmaxf(xx - 2.0f * (sd[0] + sd[1]) + cc[c], 0.f) * (HAS_W ? wv : 1.0f);
                lab[c] = expf(-a.beta * d);
                sum += lab[c];
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int c = 0; c < C_; ++c) {
                const float lb = lab[c] * inv, wl = HAS_W ? wv * lb : lb;
                const f2 wl2 = {wl, wl};
#pragma unroll
                for (int q = 0; q < H; ++q) acc2[c][q] = __builtin_elementwise_fma(x2[q], wl2, acc2[c][q]);
                den[c] += lb;
            }
        }
    }
    // workgroup sum of the NV values: a lane tree per wave (VALU), the four wave totals meet in LDS
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C_; ++c) {
#pragma unroll
        for (int q = 0; q < H; ++q) {
            const float v0 = wave_sum_lane0(acc2[c][q][0]), v1 = wave_sum_lane0(acc2[c][q][1]);
            if (lane == 0) { buf[wave * NV + c * E_ + 2 * q] = v0; buf[wave * NV + c * E_ + 2 * q + 1] = v1; }
        }
        const float vd = wave_sum_lane0(den[c]);
        if (lane == 0) buf[wave * NV + CE + c] = vd;
    }
    __syncthreads();
    float* const mypart = a.part + ((long)r * a.nG + g) * NV;
    for (int i = tid; i < NV; i += 256) {
        const float v = (buf[i] + buf[NV + i]) + (buf[2 * NV + i] + buf[3 * NV + i]);
        if (a.tickets) __hip_atomic_store(mypart + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else mypart[i] = v;
    }
    if (a.tickets == nullptr) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        const int last = atomicAdd(a.tickets + r, 1u) == (unsigned)a.nG - 1u;
        if (last) __hip_atomic_store(a.tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = last;
    }
    __syncthreads();
    if (!last_sh || tid >= CE) return;
    const int c = tid / E_;
    float num = 0.f, dn = 0.f;
    for (int q = 0; q < a.nG; ++q) {
        num += __hip_atomic_load(a.part + ((long)r * a.nG + q) * NV + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dn += __hip_atomic_load(a.part + ((long)r * a.nG + q) * NV + CE + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a.cent_out[(long)r * CE + tid] = num / dn;
    if (a.den_out && (tid % E_) == 0) a.den_out[(long)r * C_ + c] = dn;
}

// centroids[r,c,:] = xn[r/tries, idx[r,c], :]                 (Kmeans_2.py:61-71)
__global__ void kmeans_init_kernel(const float* __restrict__ xn, const int32_t* __restrict__ idx, float* __restrict__ cent, int R,
                                   int C, int E, long L, int tries) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * C * E) return;
    const int e = (int)(i % E);
    const long rc = i / E;
    const int r = (int)(rc / C);
    cent[i] = xn[((long)(r / tries) * L + idx[rc]) * E + e];
}

// best[b] = argmin_try inertia[b*tries + try] (first minimum); sel[b] = centroids[b*tries + best]
__global__ void kmeans_select_kernel(const float* __restrict__ inertia, const float* __restrict__ cent, int32_t* __restrict__ best,
                                     float* __restrict__ sel, int b, int tries, int CE) {
    const int bi = blockIdx.x;
    if (bi >= b) return;
    int bt = 0;
    float bv = inertia[(long)bi * tries];
    for (int t = 1; t < tries; ++t) {
        const float v = inertia[(long)bi * tries + t];
        if (v < bv) { bv = v; bt = t; }
    }
    if (threadIdx.x == 0) best[bi] = bt;
    for (int i = threadIdx.x; i < CE; i += blockDim.x) sel[(long)bi * CE + i] = cent[((long)bi * tries + bt) * CE + i];
}


template <int MODE>
ams_status launch_pass(const KmArgs& a, int R, int E, int C, hipStream_t st) {
    dim3 grid((unsigned)(ceil_div(a.b, 8) * 8 * a.G * a.tries));   // flat: see the work order at the top of kmeans_pass_kernel
    const bool hw = a.w != nullptr;
#define AMS_KM(EE, CC) do { if (hw) hipLaunchKernelGGL((kmeans_pass_kernel<EE, CC, MODE, true>), grid, dim3(256), 0, st, a); \
                            else hipLaunchKernelGGL((kmeans_pass_kernel<EE, CC, MODE, false>), grid, dim3(256), 0, st, a); } while (0)
    if (E == 40 && C == 2) AMS_KM(40, 2);
    else if (E == 40 && C == 3) AMS_KM(40, 3);
    else if (E == 40 && C == 4) AMS_KM(40, 4);
    else if (E == 8 && C == 2) AMS_KM(8, 2);
    else if (E == 8 && C == 3) AMS_KM(8, 3);
    else if (E == 8 && C == 4) AMS_KM(8, 4);
    else if (E == 20 && C == 2) AMS_KM(20, 2);
    else if (E == 20 && C == 3) AMS_KM(20, 3);
    else return AMS_E_INVALID_ARG;
#undef AMS_KM
    return ams_check_launch();
}

}  // namespace

extern "C" {

ams_status ams_kmeans_normalize(const float* x, float* xn, long nrows, int E, void* stream) {
    AMS_REQUIRE(x && xn && nrows > 0 && E > 0);
    AMS_REQUIRE(E <= 60);                                   // 256 x (E+1) floats of LDS
    long blocks = (nrows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kmeans_normalize_kernel, dim3((int)blocks), dim3(256), (size_t)256 * (E + 1) * sizeof(float), (hipStream_t)stream,
                       x, xn, nrows, E);
    return ams_check_launch();
}

#ifdef AMS_KT_DBG
static unsigned long long* g_kt_dbg = nullptr;
unsigned long long* ams_dbg_kt_buffer(size_t words) {       // device buffer the next launches write their placement / times to
    if (!g_kt_dbg) { hipMalloc((void**)&g_kt_dbg, words * 8); hipMemset(g_kt_dbg, 0, words * 8); }
    return g_kt_dbg;
}
int ams_dbg_kt_occupancy() {
    int n = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kmeans_hard_tries_kernel<false>, 640, KT_LDS_BYTES);
    return n;
}
#endif

size_t ams_kmeans_workspace_bytes(int R, long L, int E, int C) {
    const int NP = max(ceil_div(L, CHUNK_SOFT), 4 * ceil_div(L, CHUNK_HARD));      // partial rows per row, whichever mode runs
    return sizeof(float) * (size_t)R * NP * C * (E + 1);
}

ams_status ams_kmeans_init(const float* xn, const int32_t* init_idx, float* centroids, int b, int tries, long L, int E, int C,
                           void* stream) {
    AMS_REQUIRE(xn && init_idx && centroids && b > 0 && tries > 0 && L > 0);
    const long n = (long)b * tries * C * E;
    hipLaunchKernelGGL(kmeans_init_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, xn, init_idx, centroids,
                       b * tries, C, E, L, tries);
    return ams_check_launch();
}

// One Lloyd iteration for all R = b*tries rows: labels from `cent_in`, new centroids to `cent_out`.
// beta < 0: hard assignment (argmin), else soft assignment softmax(-beta d^2).
ams_status ams_kmeans_iterate(const float* xn, const float* w, const float* cent_in, float* cent_out, float* den_out, int b, int tries,
                              long L, int E, int C, float beta, int w_mod_b, void* ws, size_t ws_bytes, void* tickets, void* stream) {
    AMS_REQUIRE(xn && cent_in && cent_out && ws && b > 0 && tries > 0 && L > 0 && C >= 2 && C <= 4);
    const int R = b * tries;
    if (ws_bytes < ams_kmeans_workspace_bytes(R, L, E, C)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    KmArgs a{};
    a.xn = xn; a.w = w; a.cent = cent_in; a.part = (float*)ws; a.L = L; a.b = b; a.tries = tries; a.G = ceil_div(L, chunk_of(beta >= 0.f));
    a.w_mod_b = w_mod_b; a.beta = beta; a.one = 1.0f;
    a.tickets = (unsigned*)tickets; a.fin_out = cent_out; a.fin_den = den_out;
    // AMS_KM_TRIES=0: every try a workgroup of its own (kmeans_pass_kernel) also where kmeans_hard_tries_kernel applies -- same bits
    static const bool tries_kernel = [] { const char* e = getenv("AMS_KM_TRIES"); return !(e && e[0] == '0'); }();
    if (tries_kernel && beta < 0.f && E == 40 && C == 2 && tries % TQ == 0 && L * E * 4 < (1L << 31))   /* 32-bit buffer offsets */ {
        KtArgs k{};
        k.xn = xn; k.cent = cent_in; k.part = (float*)ws; k.tickets = (unsigned*)tickets; k.fin_out = cent_out; k.fin_den = den_out;
        k.L = L; k.b = b; k.tries = tries; k.G = a.G; k.w = w; k.w_mod_b = w_mod_b;
#ifdef AMS_KT_DBG
        k.dbg = g_kt_dbg;
#endif
        if (w) hipLaunchKernelGGL(kmeans_hard_tries_kernel<true>, dim3((unsigned)(b * (tries / TQ) * 4 * a.G)), dim3(640), KT_LDS_BYTES, st, k);
        else hipLaunchKernelGGL(kmeans_hard_tries_kernel<false>, dim3((unsigned)(b * (tries / TQ) * 4 * a.G)), dim3(640), KT_LDS_BYTES, st, k);
        ams_status s2 = ams_check_launch();
        if (s2 != AMS_OK || tickets) return s2;
        hipLaunchKernelGGL(kmeans_reduce_kernel, dim3(ceil_div((long)R * C * E, 256)), dim3(256), 0, st, (const float*)ws, cent_out, den_out, R,
                           4 * a.G, C, E, 0);
        return ams_check_launch();
    }
    // AMS_KM_SOFT=0: the soft accumulation pass of kmeans_pass_kernel also where kmeans_soft_acc_kernel applies
    static const bool soft_kernel = [] { const char* e = getenv("AMS_KM_SOFT"); return !(e && e[0] == '0'); }();
    if (soft_kernel && beta >= 0.f && E == 40) {
        KsfArgs k{};
        k.xn = xn; k.w = w; k.cent = cent_in; k.part = (float*)ws; k.tickets = (unsigned*)tickets; k.cent_out = cent_out; k.den_out = den_out;
        k.L = L; k.b = b; k.tries = tries; k.w_mod_b = w_mod_b; k.beta = beta;
        const int slabs = ceil_div(L, LANES);
        int nG = 512 / R; if (nG < 1) nG = 1; if (nG > slabs) nG = slabs;
        if (nG > a.G) nG = a.G;                                 // the workspace holds a.G partial rows per row
        k.spw = ceil_div(slabs, nG); k.nG = ceil_div(slabs, k.spw);
        const dim3 grid(k.nG, R);
#define AMS_KSF(CC) do { if (w) hipLaunchKernelGGL((kmeans_soft_acc_kernel<CC, true>), grid, dim3(256), 0, st, k); \
                         else hipLaunchKernelGGL((kmeans_soft_acc_kernel<CC, false>), grid, dim3(256), 0, st, k); } while (0)
        if (C == 2) AMS_KSF(2); else if (C == 3) AMS_KSF(3); else AMS_KSF(4);
#undef AMS_KSF
        ams_status s2 = ams_check_launch();
        if (s2 != AMS_OK || tickets) return s2;
        hipLaunchKernelGGL(kmeans_reduce_kernel, dim3(ceil_div((long)R * C * E, 256)), dim3(256), 0, st, (const float*)ws, cent_out, den_out, R,
                           k.nG, C, E, 0);
        return ams_check_launch();
    }
    ams_status s = beta < 0.f ? launch_pass<HARD_ACC>(a, R, E, C, st) : launch_pass<SOFT_ACC>(a, R, E, C, st);
    if (s != AMS_OK || tickets) return s;
    hipLaunchKernelGGL(kmeans_reduce_kernel, dim3(ceil_div((long)R * C * E, 256)), dim3(256), 0, st, (const float*)ws, cent_out, den_out, R,
                       beta < 0.f ? 4 * a.G : a.G, C, E, 0);
    return ams_check_launch();
}

// Labels for `cent` (+ per-row inertia).  hard: labels int32 [R,L]; soft: soft [R,L,C].  Either output may be NULL.
ams_status ams_kmeans_assign(const float* xn, const float* w, const float* cent, int32_t* labels, float* soft, float* inertia, int b,
                             int tries, long L, int E, int C, float beta, int w_mod_b, void* ws, size_t ws_bytes, void* tickets, void* stream) {
    AMS_REQUIRE(xn && cent && ws && b > 0 && tries > 0 && L > 0 && C >= 2 && C <= 4);
    const int R = b * tries;
    if (ws_bytes < ams_kmeans_workspace_bytes(R, L, E, C)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    KmArgs a{};
    a.xn = xn; a.w = w; a.cent = cent; a.part = (float*)ws; a.labels = labels; a.soft = soft; a.L = L; a.b = b; a.tries = tries;
    const bool labels_only = beta < 0.f && !inertia && labels;
    a.G = ceil_div(L, chunk_of(beta >= 0.f || labels_only)); a.w_mod_b = w_mod_b; a.beta = beta; a.one = 1.0f;
    a.tickets = inertia ? (unsigned*)tickets : nullptr; a.fin_out = inertia; a.fin_den = nullptr;
    static const bool tries_kernel = [] { const char* e = getenv("AMS_KM_TRIES"); return !(e && e[0] == '0'); }();
    if (tries_kernel && beta < 0.f && inertia && E == 40 && C == 2 && tries % TQ == 0 && L * E * 4 < (1L << 31))   /* 32-bit buffer offsets */ {
        KtArgs k{};
        k.xn = xn; k.cent = cent; k.part = (float*)ws; k.tickets = (unsigned*)tickets; k.fin_out = inertia; k.labels = labels;
        k.L = L; k.b = b; k.tries = tries; k.G = a.G; k.w = w; k.w_mod_b = w_mod_b;
        if (w) hipLaunchKernelGGL(kmeans_hard_tries_final_kernel<true>, dim3((unsigned)(b * (tries / TQ) * 2 * a.G)), dim3(640), 2 * 128 * 40 * sizeof(float), st, k);
        else hipLaunchKernelGGL(kmeans_hard_tries_final_kernel<false>, dim3((unsigned)(b * (tries / TQ) * 2 * a.G)), dim3(640), 2 * 128 * 40 * sizeof(float), st, k);
        ams_status s2 = ams_check_launch();
        if (s2 != AMS_OK || tickets) return s2;
        hipLaunchKernelGGL(kmeans_reduce_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, st, (const float*)ws, inertia, (float*)nullptr, R, 4 * a.G, C, E, 1);
        return ams_check_launch();
    }
    ams_status s = labels_only ? launch_pass<HARD_LABELS>(a, R, E, C, st)
                 : beta < 0.f ? launch_pass<HARD_FINAL>(a, R, E, C, st) : launch_pass<SOFT_FINAL>(a, R, E, C, st);
    if (s != AMS_OK) return s;
    if (inertia && !tickets) {
        hipLaunchKernelGGL(kmeans_reduce_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, st, (const float*)ws, inertia, (float*)nullptr, R, beta < 0.f ? 4 * a.G : a.G, C, E, 1);
        s = ams_check_launch();
    }
    return s;
}

// One backward pass of the unrolled soft k-means for b rows (the selected tries, already gathered by the caller):
ams_status ams_kmeans_select(const float* inertia, const float* centroids, int32_t* best, float* selected, int b, int tries, int E,
                             int C, void* stream) {
    AMS_REQUIRE(inertia && centroids && best && selected && b > 0 && tries > 0);
    hipLaunchKernelGGL(kmeans_select_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, inertia, centroids, best, selected, b, tries, C * E);
    return ams_check_launch();
}

}  // extern "C"
