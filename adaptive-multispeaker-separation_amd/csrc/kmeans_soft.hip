// Backward of the unrolled SOFT k-means (reference models/Kmeans_2.py:145-188, beta branch): the gradient of the
// selected try's soft labels / centroids with respect to the (normalised) embeddings -- what the front_*_finetuning recipes
// differentiate through (models/network.py:554-582 -> models/adapt.py:339-372).
//
// One entry point, ams_kmeans_soft_bwd, enqueues the whole chain:
//   init            G[n] = dsel (or 0), record n = final centroids, arrival tickets cleared
//   n + 1 passes    (the final assignment, then iterations n-1 .. 0): one streaming read of xn each; every workgroup leaves its
//                   partial centroid-gradient sums, the LAST workgroup of an utterance to arrive adds them in chunk order, writes
//                   G[i] and the constants record of the NEXT pass (dnum = G/den, dden = -<G, c_next>/den) -- no reduce launches
//   dx              ONE pass over xn writes dx for the final assignment and all n iterations
// Round 3 measured this chain at 2.1 ms of the 6.1 ms cfg3(ii) step (11 passes of 84 us + 22 reduce launches, dx 525 us at 0.8 TB/s).
// What was wrong, and what this file does instead:
//   * the per-utterance constants (centroids, dnum, dden: 2 C E + C floats per iteration) were LDS broadcast reads -- ~320 ds_read per
//     point and iteration next to ~700 VALU: here they are wave-uniform global loads from a constants RECORD (s_load -> SGPR operands of
//     the FMAs: the scalar unit fetches them beside the vector pipe);
//   * the translation unit of the bit-exact hard k-means is compiled without FMA contraction; this one is not (tolerance-tested
//     against float64 autograd, tests/test_gpu_kmeans_soft.py): half the VALU instructions;
//   * the dx update is regrouped: dx_e = x_e * (sum_c a_c) - sum_c a_c cent_ce + sum_c b_c dnum_ce with a_c = 2 dd2_c, b_c = w lab_c:
//     five FMAs per element and iteration;
//   * rows are staged through LDS as 16-byte vectors both ways (the old dx kernel moved 4 bytes per load with a divide per element).
#include "common.h"

namespace {

constexpr int KS_LANES = 256;
// chunks per utterance of a phase-1 pass: ONE resident round of workgroups (2 per CU by registers: 512) when the batch allows it --
// the pass is latency-bound (a workgroup's slabs are a serial chain load -> LDS -> ~1 us of arithmetic), not arithmetic-bound
inline int ks_chunks(int b, long L) {
    const int slabs = ceil_div(L, KS_LANES);
    int nG = 512 / (b > 0 ? b : 1);
    if (nG < 1) nG = 1;
    if (nG > slabs) nG = slabs;
    return nG;
}

// A constants record is read through the CONSTANT address space: wave-uniform loads from it are scalar loads (s_load_dwordxN into
// SGPRs, fetched by the scalar unit beside the vector pipe) whatever the compiler can prove about clobbers -- through a plain global
// pointer hipcc issued 255 broadcast global_load_dword per point group instead.  Legal here: every record a kernel reads was written by
// an EARLIER launch (the scalar cache is invalidated at kernel boundaries); the record a pass writes is read by later launches only.
typedef const float __attribute__((address_space(4))) ks_cfloat;
__device__ __forceinline__ ks_cfloat* ks_const(const float* p) { return (ks_cfloat*)p; }

// constants record of one (utterance, iteration): [cent C*E | dnum C*E | dden C, padded to 4 | |cent_c|^2 C, padded to 4]
template <int E_, int C_> struct KsRec { static constexpr int CE = C_ * E_, N = 2 * CE + 8; };
typedef float ks_f2 __attribute__((ext_vector_type(2)));
typedef const ks_f2 __attribute__((address_space(4))) ks_cf2;

struct KsArgs {
    const float* xn; const float* w; const float* w_final; const float* cents; const float* dens; const float* dsel; const float* dout;
    float* dx; float* g0;
    const float* inv;   // [b, L] or NULL: 1 / |x| of the l2-normalise that produced xn -- dx then leaves as the gradient w.r.t. its INPUT
    const float* inv0;  // [b, L] or NULL (needs inv): that input was itself v = u / |u|, inv0 = 1 / |u| -- dx leaves as the gradient w.r.t. u
    const int* seed;    // [b, C] or NULL: the points c_0 was picked from -- g0 is added onto their rows here instead of by the caller
    float* part;        // [b, G, C*E + C]
    float* rec;         // [b, n_it + 1, N]
    float* G;           // [n_it + 1, b, C*E]
    unsigned* ticket;   // [b]
    unsigned* amax;     // max |dx| as float bits (ams_kmeans_soft_bwd_amax_offset): cleared by ks_init_kernel, folded by ks_dx_kernel / ks_seed_kernel
    long L; int b, n_it, nG, spw; float beta;      // nG chunks per utterance, spw slabs of 256 points per workgroup
};

// G[n] = dsel or 0; record n = final centroids; tickets = 0
template <int E_, int C_>
__global__ void ks_init_kernel(KsArgs a) {
    using R = KsRec<E_, C_>;
    const int r = blockIdx.x, tid = threadIdx.x;
    float* rec = a.rec + ((long)r * (a.n_it + 1) + a.n_it) * R::N;
    const float* cf = a.cents + ((long)a.n_it * a.b + r) * R::CE;
    for (int i = tid; i < R::N; i += blockDim.x) {
        float v = i < R::CE ? cf[i] : 0.f;
        if (i >= 2 * R::CE + 4 && i < 2 * R::CE + 4 + C_) { v = 0.f; for (int e = 0; e < E_; ++e) v = fmaf(cf[(i - 2 * R::CE - 4) * E_ + e], cf[(i - 2 * R::CE - 4) * E_ + e], v); }
        rec[i] = v;
    }
    for (int i = tid; i < R::CE; i += blockDim.x) a.G[((long)a.n_it * a.b + r) * R::CE + i] = a.dsel ? a.dsel[(long)r * R::CE + i] : 0.f;
    if (tid == 0) a.ticket[r] = 0u;
    if (tid == 0 && r == 0) *a.amax = 0u;
}

// softmax labels and d/d(d2) of one point for one constants record.  FINAL: dlab comes from dout, no dnum / dden.
// x2: the point as E/2 pairs (every product below is a v_pk_fma_f32 with the record pair as its scalar operand); xx = |x|^2.
// |x - c|^2 = |x|^2 - 2 <x, c> + |c|^2: one dot product per cluster instead of a subtract and a multiply-add per element (|c|^2 sits
// in the record; the cancellation costs ~1e-7 absolute on d2, i.e. ~beta 1e-7 relative on the labels).
template <int E_, int C_, bool FINAL>
__device__ __forceinline__ void ks_point(const ks_f2 (&x2)[E_ / 2], float xx, ks_cfloat* rec, float wv, float beta, const float* dlab_in,
                                         float (&lab)[C_], float (&dd2)[C_]) {
    constexpr int CE = C_ * E_, H = E_ / 2;
    ks_cf2* rec2 = (ks_cf2*)rec;
    float d[C_], dlab[C_];
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        ks_f2 s = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < H; ++q) s = __builtin_elementwise_fma(x2[q], (ks_f2)rec2[c * H + q], s);
        d[c] = wv * (xx - 2.0f * (s[0] + s[1]) + rec[2 * CE + 4 + c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C_; ++c) { lab[c] = __expf(-beta * d[c]); sum += lab[c]; }
    const float inv = 1.0f / sum;
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        lab[c] *= inv;
        if (FINAL) dlab[c] = dlab_in[c];
        else {
            ks_f2 s = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < H; ++q) s = __builtin_elementwise_fma(x2[q], (ks_f2)rec2[(CE + c * E_) / 2 + q], s);
            dlab[c] = fmaf(wv, s[0] + s[1], rec[2 * CE + c]);
        }
        mean = fmaf(lab[c], dlab[c], mean);
    }
#pragma unroll
    for (int c = 0; c < C_; ++c) dd2[c] = -beta * wv * lab[c] * (dlab[c] - mean);
}

// One pass of phase 1: centroid gradient of iteration `it` (FINAL: of the returned assignment).  grid (nG, b), 256 threads.
template <int E_, int C_, bool FINAL>
__global__ __launch_bounds__(256) void ks_pass_kernel(KsArgs a, int it) {
    using R = KsRec<E_, C_>;
    constexpr int CE = R::CE, NV = CE + C_, LD = E_ + 4, V4 = E_ / 4;
    static_assert(E_ % 4 == 0, "rows are staged as 16-byte vectors");
    constexpr int BUF = (256 * LD > 128 * 64) ? 256 * LD : 128 * 64;
    __shared__ __attribute__((aligned(16))) float buf[BUF];
    __shared__ int last_sh;
    const int r = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* xb = a.xn + (long)r * a.L * E_;
    const float* wsel = FINAL ? a.w_final : a.w;
    const float* wb = wsel ? wsel + (long)r * a.L : nullptr;
    ks_cfloat* rec = ks_const(a.rec + ((long)r * (a.n_it + 1) + (FINAL ? a.n_it : it)) * R::N);

    // acc[c*E+e] = sum_l dd2[l,c] x[l,e];  acc[CE+c] = sum_l dd2[l,c]   ->   g[c,e] = -2 (acc[c,e] - cent[c,e] acc[CE+c])
    float acc[NV];
    ks_f2 acc2[C_][E_ / 2];
#pragma unroll
    for (int c = 0; c < C_; ++c) {
        acc[CE + c] = 0.f;
#pragma unroll
        for (int q = 0; q < E_ / 2; ++q) acc2[c][q] = ks_f2{0.f, 0.f};
    }
    // slab j + 1 is requested into registers before slab j is worked on (unconditional loads on clamped addresses, validity applied
    // at the LDS write): a workgroup's slabs were a serial chain of exposed memory round trips
    float4 pre[V4];
    float wpre = 1.0f, dpre[C_];                                // the point's weight (and d/d label, FINAL) travel with its slab
    auto fetch = [&](int j) {
        const long q0 = ((long)g * a.spw + j) * KS_LANES;
        const long lim = (a.L - q0) * V4;                       // float4s of this slab that exist (may be <= 0)
        const float4* src = reinterpret_cast<const float4*>(xb + q0 * E_);
#pragma unroll
        for (int k = 0; k < V4; ++k) {
            const long i = tid + 256 * k;
            pre[k] = src[i < lim ? i : (lim > 0 ? lim - 1 : -q0 * V4)];       // clamped inside the utterance's rows
        }
        const long pt = min(q0 + tid, a.L - 1);
        if (wb) wpre = wb[pt];
        if (FINAL) {
#pragma unroll
            for (int c = 0; c < C_; ++c) dpre[c] = a.dout[((long)r * a.L + pt) * C_ + c];
        }
    };
    if (a.L > 0) fetch(0);
    for (int j = 0; j < a.spw; ++j) {
        const long p0 = ((long)g * a.spw + j) * KS_LANES;
        const int npts = (int)max((long)0, min((long)KS_LANES, a.L - p0));
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
        float dl[C_];
#pragma unroll
        for (int c = 0; c < C_; ++c) dl[c] = FINAL ? dpre[c] : 0.f;
        if (j + 1 < a.spw) fetch(j + 1);
        if (tid < npts) {
            ks_f2 x2[E_ / 2];
            ks_f2 xs = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < V4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&buf[tid * LD + q * 4]);
                x2[2 * q] = ks_f2{v.x, v.y}; x2[2 * q + 1] = ks_f2{v.z, v.w};
                xs = __builtin_elementwise_fma(x2[2 * q], x2[2 * q], xs);
                xs = __builtin_elementwise_fma(x2[2 * q + 1], x2[2 * q + 1], xs);
            }
            float lab[C_], dd2[C_];
            ks_point<E_, C_, FINAL>(x2, xs[0] + xs[1], rec, wv, a.beta, dl, lab, dd2);
#pragma unroll
            for (int c = 0; c < C_; ++c) {
                const ks_f2 dd = {dd2[c], dd2[c]};
#pragma unroll
                for (int q = 0; q < E_ / 2; ++q) acc2[c][q] = __builtin_elementwise_fma(dd, x2[q], acc2[c][q]);
                acc[CE + c] += dd2[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C_; ++c)
#pragma unroll
        for (int q = 0; q < E_ / 2; ++q) { acc[c * E_ + 2 * q] = acc2[c][q][0]; acc[c * E_ + 2 * q + 1] = acc2[c][q][1]; }
    // workgroup sum of the NV accumulators: a lane tree per wave (VALU, no LDS), the four wave totals meet in LDS in wave order
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v = wave_sum_lane0(acc[i]);
        if (lane == 0) buf[wave * NV + i] = v;
    }
    __syncthreads();
    float* const mypart = a.part + ((long)r * a.nG + g) * NV;
    // partials leave as agent-scope (write-through) stores and are read back the same way below; the arrival is counted once this
    // workgroup's stores are acknowledged (workgroup-scope release = s_waitcnt vmcnt(0), then the barrier) -- not __threadfence(), whose
    // agent-scope release writes the XCD's whole L2 back (csrc/dpcl.hip, round 4)
    for (int i = tid; i < NV; i += 256)
        __hip_atomic_store(mypart + i, (buf[i] + buf[NV + i]) + (buf[2 * NV + i] + buf[3 * NV + i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the last workgroup of this utterance to arrive adds the chunks in chunk order (the result does not depend on who is last),
    // writes G[it] (FINAL: adds to G[n]) and the constants record of the pass that consumes it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) last_sh = (atomicAdd(a.ticket + r, 1u) == (unsigned)a.nG - 1u) ? 1 : 0;
    __syncthreads();
    if (!last_sh) return;
    float* const sum = buf;                                    // [NV]
    for (int i = tid; i < NV; i += 256) {
        float s = 0.f;
        for (int q = 0; q < a.nG; ++q) s += __hip_atomic_load(a.part + ((long)r * a.nG + q) * NV + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum[i] = s;
    }
    __syncthreads();
    const int gi = FINAL ? a.n_it : it;                        // G index written
    float* const Gout = a.G + ((long)gi * a.b + r) * CE;
    float* const gfin = buf + NV + 4;                          // [CE] the finished gradient, for the record below
    for (int i = tid; i < CE; i += 256) {
        const int c = i / E_;
        float gv = -2.0f * (sum[i] - rec[i] * sum[CE + c]);
        if (FINAL) gv += Gout[i];
        Gout[i] = gv;
        gfin[i] = gv;
        if (gi == 0 && a.g0) a.g0[(long)r * CE + i] = gv;
    }
    __syncthreads();
    if (gi > 0) {
        // record gi-1: cent_{gi-1}, dnum = G[gi] / den_{gi-1}, dden = -<G[gi], cent_gi> / den_{gi-1}
        float* const nrec = a.rec + ((long)r * (a.n_it + 1) + (gi - 1)) * R::N;
        const float* cprev = a.cents + ((long)(gi - 1) * a.b + r) * CE;
        const float* ccur = a.cents + ((long)gi * a.b + r) * CE;
        const float* den = a.dens + ((long)(gi - 1) * a.b + r) * C_;
        for (int i = tid; i < CE; i += 256) { nrec[i] = cprev[i]; nrec[CE + i] = gfin[i] / den[i / E_]; }
        if (tid < C_) {
            float dsum = 0.f, cc = 0.f;
            for (int e = 0; e < E_; ++e) { dsum += gfin[tid * E_ + e] * ccur[tid * E_ + e]; cc = fmaf(cprev[tid * E_ + e], cprev[tid * E_ + e], cc); }
            nrec[2 * CE + tid] = -dsum / den[tid];
            nrec[2 * CE + 4 + tid] = cc;
        }
    }
    if (tid == 0) a.ticket[r] = 0u;                            // for the next pass (stream order)
}

// Phase 2: dx for the final assignment and all iterations in one pass over xn.  grid (chunks, b), 256 threads.
template <int E_, int C_>
__global__ __launch_bounds__(256) void ks_dx_kernel(KsArgs a, int chunks_per_wg) {
    using R = KsRec<E_, C_>;
    constexpr int CE = R::CE, LD = E_ + 4, V4 = E_ / 4;
    __shared__ __attribute__((aligned(16))) float buf[256 * LD];
    const int r = blockIdx.y, tid = threadIdx.x;
    const float* xb = a.xn + (long)r * a.L * E_;
    float* dxb = a.dx + (long)r * a.L * E_;
    const float* wb = a.w ? a.w + (long)r * a.L : nullptr;
    const float* wf = a.w_final ? a.w_final + (long)r * a.L : nullptr;
    ks_cfloat* rec0 = ks_const(a.rec + (long)r * (a.n_it + 1) * R::N);
    float amax_l = 0.f;                                         // max |dx| of the rows this thread wrote (the bound of the products that read dx)
    for (int s = 0; s < chunks_per_wg; ++s) {
        const long p0 = ((long)blockIdx.x * chunks_per_wg + s) * 256;
        if (p0 >= a.L) break;
        const int npts = (int)min((long)256, a.L - p0);
        // the point's weights and d/d label are requested with the slab, not at their first use
        const long pt = min(p0 + tid, a.L - 1);
        const float wit_pre = wb ? wb[pt] : 1.0f, wfin_pre = wf ? wf[pt] : 1.0f;
        float dl_pre[C_];
#pragma unroll
        for (int c = 0; c < C_; ++c) dl_pre[c] = a.dout ? a.dout[((long)r * a.L + pt) * C_ + c] : 0.f;
        __syncthreads();
        {
            const float4* src = reinterpret_cast<const float4*>(xb + p0 * E_);
#pragma unroll
            for (int k = 0; k < V4; ++k) {
                const int i = tid + 256 * k;
                const float4 v = (i < npts * V4) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int row = i / V4, c4 = i - row * V4;
                *reinterpret_cast<float4*>(&buf[row * LD + c4 * 4]) = v;
            }
        }
        __syncthreads();
        ks_f2 dxl[E_ / 2];
#pragma unroll
        for (int q = 0; q < E_ / 2; ++q) dxl[q] = ks_f2{0.f, 0.f};
        if (tid < npts) {
            ks_f2 x2[E_ / 2];
            ks_f2 xs = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < V4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&buf[tid * LD + q * 4]);
                x2[2 * q] = ks_f2{v.x, v.y}; x2[2 * q + 1] = ks_f2{v.z, v.w};
                xs = __builtin_elementwise_fma(x2[2 * q], x2[2 * q], xs);
                xs = __builtin_elementwise_fma(x2[2 * q + 1], x2[2 * q + 1], xs);
            }
            const float xx = xs[0] + xs[1];
            const float wit = wit_pre;
            float ksum = 0.f;                                   // sum over passes of sum_c 2 dd2_c: dx += x * ksum at the end
            if (a.dout) {
                ks_cfloat* rec = rec0 + (long)a.n_it * R::N;
                ks_cf2* rec2 = (ks_cf2*)rec;
                const float wv = wfin_pre;
                float lab[C_], dd2[C_];
                ks_point<E_, C_, true>(x2, xx, rec, wv, a.beta, dl_pre, lab, dd2);
#pragma unroll
                for (int c = 0; c < C_; ++c) {
                    const float a2 = 2.0f * dd2[c];
                    ksum += a2;
                    const ks_f2 na = {-a2, -a2};
#pragma unroll
                    for (int q = 0; q < E_ / 2; ++q) dxl[q] = __builtin_elementwise_fma(na, (ks_f2)rec2[c * (E_ / 2) + q], dxl[q]);
                }
            }
            for (int it = 0; it < a.n_it; ++it) {
                ks_cfloat* rec = rec0 + (long)it * R::N;
                ks_cf2* rec2 = (ks_cf2*)rec;
                float dl[C_], lab[C_], dd2[C_];
                ks_point<E_, C_, false>(x2, xx, rec, wit, a.beta, dl, lab, dd2);
#pragma unroll
                for (int c = 0; c < C_; ++c) {
                    const float a2 = 2.0f * dd2[c], b2 = wit * lab[c];
                    ksum += a2;
                    const ks_f2 na = {-a2, -a2}, bb = {b2, b2};
#pragma unroll
                    for (int q = 0; q < E_ / 2; ++q) {
                        dxl[q] = __builtin_elementwise_fma(na, (ks_f2)rec2[c * (E_ / 2) + q], dxl[q]);
                        dxl[q] = __builtin_elementwise_fma(bb, (ks_f2)rec2[(CE + c * E_) / 2 + q], dxl[q]);
                    }
                }
            }
            const ks_f2 kk = {ksum, ksum};
#pragma unroll
            for (int q = 0; q < E_ / 2; ++q) dxl[q] = __builtin_elementwise_fma(x2[q], kk, dxl[q]);
            if (a.inv) {
                // Jacobian of xn = u / |u| applied here (tf.nn.l2_normalize, utils/ops.py:323): du = (dx - xn <xn, dx>) / |u| -- the point
                // is in registers; as a pass of its own (l2norm_bwd_slab_kernel) it was 630 MB of traffic, 151 us of a fine-tuning step
                ks_f2 ds = {0.f, 0.f};
#pragma unroll
                for (int q = 0; q < E_ / 2; ++q) ds = __builtin_elementwise_fma(x2[q], dxl[q], ds);
                const float dot = ds[0] + ds[1], iv = a.inv[(long)r * a.L + pt];
                const ks_f2 nd = {-dot, -dot}, i2 = {iv, iv};
#pragma unroll
                for (int q = 0; q < E_ / 2; ++q) dxl[q] = __builtin_elementwise_fma(nd, x2[q], dxl[q]) * i2;
                if (a.inv0) {
                    // ... and of v = u / |u| before it (the embedding network's Normalize layer): v = xn |v| = xn / inv
                    const float nv = 1.0f / iv, iv0 = a.inv0[(long)r * a.L + pt];
                    const ks_f2 n2 = {nv, nv};
                    ks_f2 d0 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < E_ / 2; ++q) d0 = __builtin_elementwise_fma(x2[q] * n2, dxl[q], d0);
                    const float dot0 = (d0[0] + d0[1]) * nv;                     // <v, dv> |v|: the factor v below carries the other |v|
                    const ks_f2 m0 = {-dot0, -dot0}, j0 = {iv0, iv0};
#pragma unroll
                    for (int q = 0; q < E_ / 2; ++q) dxl[q] = __builtin_elementwise_fma(m0, x2[q], dxl[q]) * j0;
                }
            }
        }
        __syncthreads();
        if (tid < npts) {
#pragma unroll
            for (int q = 0; q < V4; ++q)
                *reinterpret_cast<float4*>(&buf[tid * LD + q * 4]) = make_float4(dxl[2 * q][0], dxl[2 * q][1], dxl[2 * q + 1][0], dxl[2 * q + 1][1]);
#pragma unroll
            for (int q = 0; q < E_ / 2; ++q) amax_l = fmaxf(amax_l, fmaxf(fabsf(dxl[q][0]), fabsf(dxl[q][1])));
        }
        __syncthreads();
        {
            float4* dst = reinterpret_cast<float4*>(dxb + p0 * E_);
#pragma unroll
            for (int k = 0; k < V4; ++k) {
                const int i = tid + 256 * k;
                if (i < npts * V4) {
                    const int row = i / V4, c4 = i - row * V4;
                    dst[i] = *reinterpret_cast<const float4*>(&buf[row * LD + c4 * 4]);
                }
            }
        }
    }
    // one atomic per wave (a NaN in dx is dropped by fmaxf here and reaches the products through dx itself)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax_l = fmaxf(amax_l, __shfl_xor(amax_l, o, 64));
    if ((tid & 63) == 0) atomicMax(a.amax, __float_as_uint(amax_l));
}

// c_0 = xn[seed]: the remaining centroid gradient g0 goes onto the rows it was picked from (the C picks of an utterance are distinct --
// np.random.choice without replacement, Kmeans_2.py:63 -- so no two updates meet), through the same Jacobian as every other row.
// grid b, C_ * 64 threads (one wave per cluster).
template <int E_, int C_>
__global__ void ks_seed_kernel(KsArgs a) {
    const int r = blockIdx.x, c = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = a.seed[r * C_ + c];
    if (row < 0 || row >= a.L) return;
    const float* g = a.g0 + ((long)r * C_ + c) * E_;
    const float* x = a.xn + ((long)r * a.L + row) * E_;
    float* d = a.dx + ((long)r * a.L + row) * E_;
    float gv = lane < E_ ? g[lane] : 0.f;
    if (a.inv) {
        const float xv = lane < E_ ? x[lane] : 0.f;
        const float iv = a.inv[(long)r * a.L + row];
        gv = (gv - xv * wave_sum(xv * gv)) * iv;
        if (a.inv0) {
            const float vv = xv / iv;
            gv = (gv - vv * wave_sum(vv * gv)) * a.inv0[(long)r * a.L + row];
        }
    }
    float nv = 0.f;
    if (lane < E_) { nv = d[lane] + gv; d[lane] = nv; }
    nv = fabsf(nv);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nv = fmaxf(nv, __shfl_xor(nv, o, 64));
    if (lane == 0) atomicMax(a.amax, __float_as_uint(nv));
}

template <int E_, int C_>
ams_status ks_run(KsArgs a, hipStream_t st) {
    hipLaunchKernelGGL((ks_init_kernel<E_, C_>), dim3(a.b), dim3(128), 0, st, a);
    dim3 grid(a.nG, a.b);
    if (a.dout) hipLaunchKernelGGL((ks_pass_kernel<E_, C_, true>), grid, dim3(256), 0, st, a, a.n_it);
    else {
        // no gradient through the returned labels: G[n] = dsel alone.  The record of iteration n-1 (and g0 when nothing was unrolled)
        // still has to be written: a zero-work FINAL pass, one workgroup per utterance
        KsArgs z = a;
        z.nG = 1; z.L = 0; z.spw = 0;
        hipLaunchKernelGGL((ks_pass_kernel<E_, C_, true>), dim3(1, a.b), dim3(256), 0, st, z, a.n_it);
    }
    for (int it = a.n_it - 1; it >= 0; --it) hipLaunchKernelGGL((ks_pass_kernel<E_, C_, false>), grid, dim3(256), 0, st, a, it);
    const int chunks = ceil_div(a.L, 256);
    int per = ceil_div(chunks, max(1, 4096 / a.b));
    if (per < 1) per = 1;
    hipLaunchKernelGGL((ks_dx_kernel<E_, C_>), dim3(ceil_div(chunks, per), a.b), dim3(256), 0, st, a, per);
    if (a.seed) hipLaunchKernelGGL((ks_seed_kernel<E_, C_>), dim3(a.b), dim3(C_ * 64), 0, st, a);
    return ams_check_launch();
}

inline size_t ks_align(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" {

size_t ams_kmeans_soft_bwd_workspace_bytes(int b, long L, int E, int C, int n_it) {
    if (b <= 0 || L <= 0 || E <= 0 || C <= 0 || n_it < 0) return 0;
    const size_t nG = (size_t)ks_chunks(b, L), CE = (size_t)C * E;
    return ks_align((size_t)b * nG * (CE + C) * 4) + ks_align((size_t)b * (n_it + 1) * (2 * CE + 8) * 4) + ks_align((size_t)(n_it + 1) * b * CE * 4) +
           ks_align((size_t)b * 4) + 256;
}
// byte offset, inside that workspace, of max |dx| (a float): valid after ams_kmeans_soft_bwd -- the operand bound of the products that read dx
size_t ams_kmeans_soft_bwd_amax_offset(int b, long L, int E, int C, int n_it) {
    const size_t n = ams_kmeans_soft_bwd_workspace_bytes(b, L, E, C, n_it);
    return n ? n - 256 : 0;
}

// Gradient of the selected try of the unrolled soft k-means w.r.t. the normalised embeddings (Kmeans_2.py:145-188).
//   xn [b,L,E]; w [b,L] silence weights of the iterations or NULL; w_final: weights of the returned assignment or NULL (end-assign: 1);
//   cents [n_it+1,b,C,E] = c_0 .. c_n of the selected rows; dens [n_it,b,C] = sum_l lab_i; dsel [b,C,E] = d loss / d c_n or NULL;
//   dout [b,L,C] = d loss / d returned labels or NULL.   Out: dx [b,L,E] (fully written), g0 [b,C,E] = d loss / d c_0.
//   seed [b,C] (or NULL): the points c_0 was taken from -> g0 is added onto those rows of dx here; inv [b,L] (or NULL): 1/|u| of the
//   l2-normalise that produced xn -> dx is the gradient w.r.t. its input; inv0 [b,L] (or NULL): that input was itself a normalised
//   v = u / |u| with inv0 = 1/|u| (ams_l2norm2_fwd) -> dx is the gradient w.r.t. u.  All in the pass that has the point in registers.
ams_status ams_kmeans_soft_bwd(const float* xn, const float* w, const float* w_final, const float* cents, const float* dens, const float* dsel,
                               const float* dout, const float* inv, const float* inv0, const int32_t* seed, float* dx, float* g0, int b, long L,
                               int E, int C, float beta, int n_it, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(xn && cents && dx && g0 && ws && b > 0 && L > 0 && C >= 2 && C <= 4 && beta >= 0.f && n_it >= 0 && (n_it == 0 || dens));
    AMS_REQUIRE((!inv || seed) && (!inv0 || inv));  // with a Jacobian applied to dx the caller can no longer add g0 to it
    if (ws_bytes < ams_kmeans_soft_bwd_workspace_bytes(b, L, E, C, n_it)) return AMS_E_WORKSPACE_TOO_SMALL;
    KsArgs a{};
    a.xn = xn; a.w = w; a.w_final = w_final; a.cents = cents; a.dens = dens; a.dsel = dsel; a.dout = dout; a.dx = dx; a.g0 = g0;
    a.inv = inv; a.inv0 = inv0; a.seed = seed;
    a.L = L; a.b = b; a.n_it = n_it; a.nG = ks_chunks(b, L); a.spw = ceil_div(ceil_div(L, KS_LANES), a.nG); a.nG = ceil_div(ceil_div(L, KS_LANES), a.spw); a.beta = beta;
    const size_t CE = (size_t)C * E;
    char* p = (char*)ws;
    a.part = (float*)p; p += ks_align((size_t)b * a.nG * (CE + C) * 4);
    a.rec = (float*)p; p += ks_align((size_t)b * (n_it + 1) * (2 * CE + 8) * 4);
    a.G = (float*)p; p += ks_align((size_t)(n_it + 1) * b * CE * 4);
    a.ticket = (unsigned*)p;
    // (the partial sums above are laid out for the chunk count this launch uses, which may be below the one the size query assumed: the
    // bound's word is addressed from the END of the workspace, where ams_kmeans_soft_bwd_amax_offset says it is)
    a.amax = (unsigned*)((char*)ws + ams_kmeans_soft_bwd_amax_offset(b, L, E, C, n_it));
    hipStream_t st = (hipStream_t)stream;
    if (E == 40 && C == 2) return ks_run<40, 2>(a, st);
    if (E == 40 && C == 3) return ks_run<40, 3>(a, st);
    if (E == 40 && C == 4) return ks_run<40, 4>(a, st);
    if (E == 8 && C == 2) return ks_run<8, 2>(a, st);
    if (E == 8 && C == 3) return ks_run<8, 3>(a, st);
    if (E == 20 && C == 2) return ks_run<20, 2>(a, st);
    return AMS_E_INVALID_ARG;
}

}  // extern "C"
