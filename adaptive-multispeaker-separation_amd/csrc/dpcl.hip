// Deep-clustering affinity loss (reference models/dpcl.py:41-87), forward and backward.
//
//   cost = mean_b( ||V^T D V||_F - 2 ||V^T D Y||_F + ||Y^T D Y||_F ),  D_i = 1/sqrt((Y (Y^T 1))_i)
//
// The reference materialises DV, DY and runs three batched matmuls with K = T*F.  Here V ([B,TF,E], 210 MB
// at the benchmark shape) is streamed ONCE: each point is the augmented vector z_i = [v_i | y_i] (E+S <= 64
// entries) and a single Gram  Z^T D Z  accumulated with v_mfma_f32_16x16x4_f32 contains all three blocks
// (V^T D V, V^T D Y, Y^T D Y).  HBM-bound: algorithmic bytes = TF*(E+S)*4 per utterance.
//
// Backward streams V again: dV_i = D_i (Gn v_i - An y_i) with Gn = 2G/(|G| B), An = 2A/(|A| B), fused with the
// l2-normalise backward so dU is written directly (one read of V, Y; one write of dU).  Thread-per-point
// VALU matvec with the small matrices fetched through the scalar cache; points are transposed through LDS so
// global traffic stays float4-coalesced.
#include "common.h"

// timing anatomy builds (WRONG results): 1 = no MFMAs, 2 = no norm / no epilogue arithmetic, 4 = no global loads, 8 = no global stores
#ifndef AMS_DPCL_AMAX
#define AMS_DPCL_AMAX 2
#endif
#ifndef AMS_DPCL_PTS
#define AMS_DPCL_PTS 256
#endif
#ifndef AMS_DPCL_DBG
#define AMS_DPCL_DBG 0
#endif
namespace {

constexpr int CHUNK = 2048;            // points per workgroup in the Gram pass

__global__ void dpcl_count_kernel(const float* __restrict__ Y, float* __restrict__ cnt, long TF, int S) {
    __shared__ float sm[4][8];
    const int b = blockIdx.x;
    float acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.f;
    const float* y = Y + (long)b * TF * S;
    for (long i = threadIdx.x; i < TF; i += blockDim.x)
        for (int s = 0; s < S; ++s) acc[s] += y[i * S + s];
    for (int s = 0; s < S; ++s) {
        const float v = wave_sum(acc[s]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < S) cnt[(long)b * S + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

template <int NT>
__global__ __launch_bounds__(256) void dpcl_gram_kernel(const float* __restrict__ V, const float* __restrict__ Y,
                                                        const float* __restrict__ cnt, float* __restrict__ part, long TF,
                                                        int E, int S, int nchunk) {
    constexpr int Z = NT * 16;
    constexpr int ZP = Z + 1;                       // LDS row pitch (odd: the 4 point-slots of a wave land on distinct banks)
    constexpr int PTS = 64;                         // points staged per iteration (16 per wave)
    __shared__ float zt[PTS * ZP];                  // augmented points [v | y | 0]
    __shared__ float dsh[PTS];                      // D_i
    __shared__ float red[Z * Z];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e_lo = lane & 15, slot = lane >> 4;
    for (int i = tid; i < Z * Z; i += 256) red[i] = 0.f;
    for (int i = tid; i < PTS * ZP; i += 256) zt[i] = 0.f;       // padding columns stay zero

    float cn[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) cn[s] = (s < S) ? cnt[(long)b * S + s] : 0.f;

    // upper-triangular tile pairs only: the Gram is symmetric
    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long p_begin = (long)c * CHUNK, p_end = min(TF, p_begin + CHUNK);
    const float* Vb = V + (long)b * TF * E;
    const float* Yb = Y + (long)b * TF * S;
    for (long p0 = p_begin; p0 < p_end; p0 += PTS) {
        const int npts = (int)min((long)PTS, p_end - p0);
        __syncthreads();
        // coalesced staging: the PTS*E floats of this slab are contiguous in memory
        for (int i = tid; i < PTS * E; i += 256) {
            const int p = i / E, e = i - p * E;
            zt[p * ZP + e] = (p < npts) ? Vb[p0 * E + i] : 0.f;
        }
        if (tid < PTS) {
            float diag = 0.f;
            if (tid < npts)
                for (int s = 0; s < S; ++s) {
                    const float yv = Yb[(p0 + tid) * S + s];
                    zt[tid * ZP + E + s] = yv;
                    diag += yv * cn[s];
                }
            else
                for (int s = 0; s < S; ++s) zt[tid * ZP + E + s] = 0.f;
            dsh[tid] = (tid < npts && diag > 0.f) ? 1.0f / sqrtf(diag) : 0.f;    // all-zero Y row: reference has D = inf
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int p = wave * 16 + g * 4 + slot;
            const float d = dsh[p];
            float a[NT], bb[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                a[ti] = zt[p * ZP + ti * 16 + e_lo];
                bb[ti] = a[ti] * d;
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj) {
                    if (AMS_DPCL_DBG & 1) acc[ti][tj][0] += a[ti] * bb[tj];
                    else acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], bb[tj], acc[ti][tj], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    // C/D layout: row = (lane>>4)*4 + r, col = lane&15.  4 waves add into LDS one after another (fixed order);
    // off-diagonal tiles are mirrored.
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ti * 16 + slot * 4 + r, col = tj * 16 + e_lo;
                        red[row * Z + col] += acc[ti][tj][r];
                        if (tj != ti) red[col * Z + row] += acc[ti][tj][r];
                    }
        }
        __syncthreads();
    }
    float* out = part + ((long)b * nchunk + c) * (Z * Z);
    for (int i = tid; i < Z * Z; i += 256) out[i] = red[i];
}


// ---------------------------------------------------------------------------------------------------------------
// Fused pass for training: reads the dense layer's output U (BEFORE l2-normalise) once and produces the per-point
// 1/|u| and the augmented Gram of the NORMALISED points in the same sweep -- V is never written unless the caller
// asks for it (V_out).  Algorithmic HBM bytes per utterance: TF*(E+S)*4 read + TF*4 written.
//   * label counts come from CP fixed-order partial sums per utterance (dpcl_count_part_kernel);
//   * a workgroup owns UCHUNK points and stages PTS of them per iteration with 16-byte loads that are issued one
//     iteration ahead (registers), so the MFMA phase of slab i covers the HBM latency of slab i+1;
//   * normalisation is applied on the MFMA operand fetch (per-lane factor: 1/|u| for embedding columns, 1 for
//     label columns), the rows in LDS stay raw.
constexpr int CP = 8;                  // label-count partials per utterance
#ifndef AMS_DPCL_BWD_F16
#define AMS_DPCL_BWD_F16 1          // dpcl_bwd_u2_kernel: the product on the 16-bit pipe as fp16x3 (0: v_mfma_f32_16x16x4_f32)
#endif
#ifndef AMS_DPCL_UCHUNK
#define AMS_DPCL_UCHUNK 2560
#endif
constexpr int UCHUNK = AMS_DPCL_UCHUNK;           // points per workgroup in the fused pass
#ifndef AMS_DPCL_BCHUNK
#define AMS_DPCL_BCHUNK 2560
#endif
constexpr int BCHUNK = AMS_DPCL_BCHUNK;           // ... in the backward pass (no per-workgroup partials there: may be smaller)

// MAKE: the labels do not exist yet -- src holds the per-source representations, rows (b, s) of TF values each, and the pass writes
// Y[b, p, s] = a where s = argmax_s' |src[b, s', p]| (first index wins ties), b_ elsewhere, exactly as make_masks_kernel
// (csrc/elementwise.hip; reference models/network.py:369-378) and counts what it writes.  The summation order is the same in both
// forms, so the counts carry the same bits whichever produced them.
constexpr int CNT_THREADS = 1024;       // 16 waves per workgroup: 8 workgroups per utterance are few, the pass is latency-bound
template <bool MAKE>
__global__ __launch_bounds__(CNT_THREADS) void dpcl_count_part_kernel(const float* __restrict__ src, float* __restrict__ cntp, long TF, int S,
                                                              unsigned* __restrict__ ticket = nullptr, float* __restrict__ Yout = nullptr,
                                                              float a = 1.f, float b_ = 0.f, int take_abs = 0) {
    __shared__ float sm[CNT_THREADS / 64][8];
    const int b = blockIdx.y, c = blockIdx.x;
    // arrival counters of the Gram pass that follows: ticket[0] counts finished utterances, ticket[1 + b] the chunks of utterance b
    if (ticket && c == 0 && threadIdx.x == 0) { ticket[1 + b] = 0u; if (b == 0) ticket[0] = 0u; }
    const long per = (TF + CP - 1) / CP;
    const long lo = (long)c * per, hi = min(TF, lo + per);
    float acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = 0.f;
    if (MAKE) {
        const float* r = src + (long)b * S * TF;
        float* y = Yout + (long)b * TF * S;
        for (long i = lo + threadIdx.x; i < hi; i += CNT_THREADS) {
            float best = 0.f;
            int bi = 0;
            for (int s = 0; s < S; ++s) {
                float v = r[(long)s * TF + i];
                if (take_abs) v = fabsf(v);
                if (s == 0 || v > best) { best = v; bi = s; }
            }
            for (int s = 0; s < S; ++s) {
                const float v = (s == bi) ? a : b_;
                y[i * S + s] = v;
                acc[s] += v;
            }
        }
    } else {
        const float* y = src + (long)b * TF * S;
        for (long i = lo + threadIdx.x; i < hi; i += CNT_THREADS)
            for (int s = 0; s < S; ++s) acc[s] += y[i * S + s];
    }
    for (int s = 0; s < S; ++s) {
        const float v = wave_sum(acc[s]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < S) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CNT_THREADS / 64; ++w) t += sm[w][threadIdx.x];
        cntp[((long)b * CP + c) * S + threadIdx.x] = t;
    }
}

__device__ __forceinline__ void load_counts(const float* __restrict__ cntp, int b, int S, float (&cn)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float t = 0.f;
        if (s < S)
            for (int c = 0; c < CP; ++c) t += cntp[((long)b * CP + c) * S + s];
        cn[s] = t;
    }
}

template <bool IN_LAUNCH>
__device__ __forceinline__ void dpcl_finish_utterance(const float* __restrict__ part, float* __restrict__ per_utt, float* __restrict__ mats,
                                                      int E, int S, int Z, int nchunk, int B, int b, float* __restrict__ gram,
                                                      float (*sm)[16]);
__device__ __forceinline__ void dpcl_finish_batch(const float* __restrict__ per_utt, float* __restrict__ out, unsigned* __restrict__ ticket,
                                                  unsigned* __restrict__ clear, int B, int* last_sh, float* __restrict__ stage, int cap);

// fin: what the pass needs to finish the loss itself (no dpcl_finish_kernel launch: 19 us + a launch boundary on the critical path of
// a training step): the workgroup that stores the LAST chunk partial of an utterance reduces them (chunk order, as the separate
// kernel does) and the one that finishes the last utterance writes the batch means.
struct GramFinish {
    float* per_utt; float* mats; float* out;
    unsigned* ticket;                                // [0]: utterances finished, [1 + b]: chunks of utterance b stored (zeroed by the count pass)
    unsigned* clear;                                 // the backward's max |dU| slot
    int B;
};

template <int NT, int EC, int SC>                  // EC / SC: compile-time E / S (0 = use the run-time value; EC != 0 implies 16-byte rows)
__global__ __launch_bounds__(256) void dpcl_gram_u_kernel(const float* __restrict__ U, const float* __restrict__ Y,
                                                          const float* __restrict__ cntp, float* __restrict__ inv_out,
                                                          float* __restrict__ V_out, float* __restrict__ part, long TF, int E_rt,
                                                          int S_rt, int nchunk, GramFinish fin) {
    const int E = EC ? EC : E_rt;
    const int S = SC ? SC : S_rt;
    constexpr int Z = NT * 16;
    constexpr int ZP = Z + 4;                       // row pitch: 16-byte aligned rows, 16 lanes x 16 B cover all banks
    constexpr int PTS = NT <= 3 ? AMS_DPCL_PTS : 128;        // points staged per iteration
    constexpr int NV = PTS * Z / 4 / 256;           // 16-byte loads per thread and slab (upper bound, E <= Z)
    __shared__ __attribute__((aligned(16))) float zt[PTS * ZP];   // raw points [u | y | 0]; reused for the final reduce
    __shared__ float dsh[PTS], ivs[PTS];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e_lo = lane & 15, slot = lane >> 4;
    for (int i = tid; i < PTS * ZP; i += 256) zt[i] = 0.f;       // padding columns stay zero

    float cn[8];
    load_counts(cntp, b, S, cn);

    f32x4 acc[NT][NT];                              // upper-triangular tile pairs only: the Gram is symmetric
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long p_begin = (long)c * UCHUNK, p_end = min(TF, p_begin + UCHUNK);
    const float* Ub = U + (long)b * TF * E;
    const float* Yb = Y + (long)b * TF * S;
    const bool vec = EC ? true : ((E % 4 == 0) && (((uintptr_t)U & 15) == 0));      // EC path: the launcher checked alignment
    const int nvec = PTS * E / 4;                   // 16-byte groups per full slab (vec path)

    float4 pre[NV];
    float yv[8];
    // Loads are UNCONDITIONAL with clamped indices (out-of-range lanes re-read the last valid vector and are zeroed when the
    // slab is stored to LDS): exec-masked branches around the loads made hipcc fall back to s_waitcnt vmcnt(0) right after
    // issuing them, which serialised the "prefetch" with the HBM round trip.
    auto fetch = [&](long p0) {
        const int npts = (int)min((long)PTS, p_end - p0);
        if (vec) {
            const float4* src = reinterpret_cast<const float4*>(Ub + p0 * E);
            const int last = npts * E / 4 - 1;
#pragma unroll
            for (int j = 0; j < NV; ++j) pre[j] = src[min(tid + 256 * j, last)];
        } else {
            const float* src = Ub + p0 * E;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = tid + 256 * (4 * j + q);
                    t[q] = (i < npts * E) ? src[i] : 0.f;
                }
                pre[j] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
        if (tid < PTS) {
            const float* yr = Yb + (p0 + min(tid, npts - 1)) * S;
#pragma unroll
            for (int s = 0; s < 8; ++s) yv[s] = (s < S) ? yr[s] : 0.f;
        }
    };

    if (p_begin < p_end) fetch(p_begin);
    for (long p0 = p_begin; p0 < p_end; p0 += PTS) {
        const int npts = (int)min((long)PTS, p_end - p0);
        __syncthreads();                            // MFMA phase of the previous slab is done with zt
        if (vec) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i4 = tid + 256 * j;
                if (i4 < nvec) {
                    const int pnt = (i4 * 4) / E, e = i4 * 4 - pnt * E;
                    *reinterpret_cast<float4*>(&zt[pnt * ZP + e]) = (i4 * 4 < npts * E) ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float t[4] = {pre[j].x, pre[j].y, pre[j].z, pre[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = tid + 256 * (4 * j + q);
                    if (i < PTS * E) zt[(i / E) * ZP + (i % E)] = t[q];
                }
            }
        }
        if (tid < PTS) {
            float diag = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < S) { const float y = tid < npts ? yv[s] : 0.f; zt[tid * ZP + E + s] = y; diag += y * cn[s]; }
            dsh[tid] = (tid < npts && diag > 0.f) ? 1.0f / sqrtf(diag) : 0.f;    // all-zero Y row: reference has D = inf
        }
        __syncthreads();
        if (p0 + PTS < p_end && !(AMS_DPCL_DBG & 4)) fetch(p0 + PTS);      // in flight during everything below
        if (tid < PTS && !(AMS_DPCL_DBG & 2)) {
            float ss = 0.f;
            const float* row = &zt[tid * ZP];
            for (int e = 0; e < E; ++e) ss += row[e] * row[e];
            const float iv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));   // tf.nn.l2_normalize epsilon (utils/ops.py:323)
            ivs[tid] = iv;
            if (inv_out && tid < npts) inv_out[(long)b * TF + p0 + tid] = iv;
        }
        __syncthreads();
        if (V_out) {
            float* dst = V_out + ((long)b * TF + p0) * E;
            for (int i = tid; i < npts * E; i += 256) {
                const int pnt = i / E, e = i - pnt * E;
                dst[i] = zt[pnt * ZP + e] * ivs[pnt];
            }
        }
#pragma unroll 4
        for (int g = 0; g < PTS / 16; ++g) {
            const int pnt = wave * (PTS / 4) + g * 4 + slot;
            const float d = dsh[pnt], iv = ivs[pnt];
            float a[NT], bb[NT];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) {
                const float raw = zt[pnt * ZP + ti * 16 + e_lo];
                a[ti] = (ti * 16 + e_lo < E) ? raw * iv : raw;
                bb[ti] = a[ti] * d;
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj) {
                    if (AMS_DPCL_DBG & 1) acc[ti][tj][0] += a[ti] * bb[tj];
                    else acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], bb[tj], acc[ti][tj], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    float* red = zt;                                // Z*Z <= PTS*ZP
    for (int i = tid; i < Z * Z; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {                   // fixed order: waves add one after another, mirrored tiles
        if (wave == w) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ti * 16 + slot * 4 + r, col = tj * 16 + e_lo;
                        red[row * Z + col] += acc[ti][tj][r];
                        if (tj != ti) red[col * Z + row] += acc[ti][tj][r];
                    }
        }
        __syncthreads();
    }
    // Partials leave with agent-scope (write-through) stores and are read back the same way by whoever arrives last, and the
    // arrival is counted once this workgroup's stores are acknowledged (workgroup-scope release = s_waitcnt vmcnt(0), then the
    // barrier).  NOT __threadfence(): an agent-scope release writes the XCD's whole L2 back, and with this kernel's 1/|u| stream
    // dirty in it 512 such fences doubled the kernel (129 us against 62).
    float* out = part + ((long)b * nchunk + c) * (Z * Z);
    for (int i = tid; i < Z * Z; i += 256) __hip_atomic_store(out + i, red[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    __shared__ float sm[3][16];
    __shared__ int last_sh;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) last_sh = (atomicAdd(fin.ticket + 1 + b, 1u) == (unsigned)nchunk - 1u) ? 1 : 0;
    __syncthreads();
    if (!last_sh) return;                           // uniform over the workgroup
    dpcl_finish_utterance<true>(part, fin.per_utt, fin.mats, E, S, Z, nchunk, fin.B, b, red, sm);
    dpcl_finish_batch(fin.per_utt, fin.out, fin.ticket, fin.clear, fin.B, &last_sh, red, PTS * ZP);
}

// ---------------------------------------------------------------------------------------------------------------
// The fused forward pass with its Gram on the 16-bit matrix pipe (round 6; E = 40, S <= 4, 16-byte aligned U).
// dpcl_gram_u_kernel above spends 29 of its 74 us in v_mfma_f32_16x16x4_f32 (four points per instruction, MfmaUtil 0.32) and the
// memory side another 45: neither saturated, and the phases of a slab -- stage, norm, MFMA -- add.  Here the augmented points
//   z'_p = sqrt(D_p) [v_p | y_p]      (Z^T D Z = Z'^T Z': ONE operand, |z'| <= 1: a constant scale 2^13, no bound to measure)
// are cut ONCE per element into two fp16 terms (fp16x3: hi.hi + hi.lo + lo.hi, f32 accumulation, as csrc/gemm.hip) and laid down
// TRANSPOSED -- per plane [feature][point], 2 points per dword -- so that an operand of v_mfma_f32_16x16x32_f16 (lane: feature
// lane & 15, eight consecutive points) is one ds_read_b128 and 32 points cost a wave 18 MFMAs of 4 passes instead of 48 of 8.
// Two threads per point do norm and cut together: each sums 20 features, the pair meets by DPP, and pairs of neighbouring points swap
// what the other one writes (DPP again), so every thread writes ten dwords per plane.  128 points per slab, one 32-point group per
// wave; everything around it (label counts, chunk partials in chunk order, the in-launch finish) is dpcl_gram_u_kernel's.
typedef _Float16 dp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dp_f16x2 __attribute__((ext_vector_type(2)));
typedef float dp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned dp_pk_f16(float a, float b) {
    const dp_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dp_f16x2));
}
__device__ __forceinline__ void dp_split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = dp_pk_f16(a, b);
    const dp_f16x2 h = __builtin_bit_cast(dp_f16x2, hi);
    lo = dp_pk_f16(a - (float)h[0], b - (float)h[1]);
}
__device__ __forceinline__ float dp_lane_xor1(float v) {      // quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dp_lane_xor2(float v) {      // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}

template <int EC, int SC>
__global__ __launch_bounds__(256) void dpcl_gram_u16_kernel(const float* __restrict__ U, const float* __restrict__ Y,
                                                            const float* __restrict__ cntp, float* __restrict__ inv_out,
                                                            float* __restrict__ V_out, float* __restrict__ part, long TF, int nchunk,
                                                            GramFinish fin) {
    constexpr int E = EC, S = SC, NT = 3, Z = NT * 16, ZP = Z + 4;
    constexpr int PTS = 128;                        // points per slab: one 32-point group per wave
    constexpr int NV = PTS * E / 4 / 256;           // 16-byte loads per thread and slab
    constexpr int HP = PTS * 2 + 16;                // bytes per feature row of a plane: + 16 -> the 16 rows of a ds_read_b128 group are 16 different slots
    constexpr int HF = E / 2, QF = HF / 2;          // features per thread of a point's pair; features a thread WRITES (for two points)
    static_assert(E % 8 == 0 && (PTS * E / 4) % 256 == 0 && E + S <= Z && S <= 4, "dpcl_gram_u16_kernel: E = 40-like shapes only");
    static_assert(Z * Z <= PTS * ZP, "the final reduce reuses the raw slab");
    __shared__ __attribute__((aligned(16))) float zt[PTS * ZP];                 // raw points [u | y | 0]; reused for the final reduce
    __shared__ __attribute__((aligned(16))) unsigned char zh[2 * Z * HP];       // planes hi | lo of z' * 2^13, [feature][point]
    __shared__ float dsh[PTS], ivs[PTS];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e_lo = lane & 15, slot = lane >> 4;
    for (int i = tid; i < PTS * ZP; i += 256) zt[i] = 0.f;                     // padding columns stay zero
    for (int i = tid; i < 2 * Z * HP / 4; i += 256) reinterpret_cast<unsigned*>(zh)[i] = 0u;     // padding features stay zero

    float cn[8];
    load_counts(cntp, b, S, cn);

    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long p_begin = (long)c * UCHUNK, p_end = min(TF, p_begin + UCHUNK);
    const float* Ub = U + (long)b * TF * E;
    const float* Yb = Y + (long)b * TF * S;

    float4 pre[NV];
    float yv[S];
    auto fetch = [&](long p0) {                     // unconditional loads on clamped indices (see dpcl_gram_u_kernel)
        const int npts = (int)min((long)PTS, p_end - p0);
        const float4* src = reinterpret_cast<const float4*>(Ub + p0 * E);
        const int last = npts * E / 4 - 1;
#pragma unroll
        for (int j = 0; j < NV; ++j) pre[j] = src[min(tid + 256 * j, last)];
        if (tid < PTS) {
            const float* yr = Yb + (p0 + min(tid, npts - 1)) * S;
#pragma unroll
            for (int s = 0; s < S; ++s) yv[s] = yr[s];
        }
    };

    const int pnt2 = tid >> 1, half = tid & 1, odd = pnt2 & 1;
    if (p_begin < p_end) fetch(p_begin);
    for (long p0 = p_begin; p0 < p_end; p0 += PTS) {
        const int npts = (int)min((long)PTS, p_end - p0);
        __syncthreads();                            // MFMA phase of the previous slab is done with zh, the cut with zt
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i4 = tid + 256 * j;
            const int pnt = (i4 * 4) / E, e = i4 * 4 - pnt * E;
            *reinterpret_cast<float4*>(&zt[pnt * ZP + e]) = (i4 * 4 < npts * E) ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < PTS) {
            float diag = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { const float y = tid < npts ? yv[s] : 0.f; zt[tid * ZP + E + s] = y; diag += y * cn[s]; }
            dsh[tid] = (tid < npts && diag > 0.f) ? 1.0f / sqrtf(diag) : 0.f;    // all-zero Y row: reference has D = inf
        }
        __syncthreads();
        if (p0 + PTS < p_end) fetch(p0 + PTS);      // in flight during everything below
        {
            // norm + cut: thread (point pnt2, half): features HF half .. HF half + HF - 1 of its point
            float v[HF];
            const float4* row = reinterpret_cast<const float4*>(&zt[pnt2 * ZP + half * HF]);
#pragma unroll
            for (int q = 0; q < HF / 4; ++q) { const float4 t = row[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
            float ss = 0.f;
#pragma unroll
            for (int k = 0; k < HF; ++k) ss += v[k] * v[k];
            ss += dp_lane_xor1(ss);
            const float iv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));   // tf.nn.l2_normalize epsilon (utils/ops.py:323)
            if (half == 0) {
                ivs[pnt2] = iv;
                if (inv_out && pnt2 < npts) inv_out[(long)b * TF + p0 + pnt2] = iv;
            }
            const float sd = sqrtf(dsh[pnt2]) * 8192.0f;        // sqrt(D_p) * 2^13 (0 for points past the end and all-zero label rows)
            const float sc = iv * sd;
#pragma unroll
            for (int k = 0; k < HF; ++k) v[k] *= sc;
            unsigned char* const wbase = zh + (half * HF + odd * QF) * HP + (pnt2 >> 1) * 4;
#pragma unroll
            for (int k = 0; k < QF; ++k) {
                const float own = odd ? v[QF + k] : v[k];       // what this thread writes: its point's feature ...
                const float give = odd ? v[k] : v[QF + k];      // ... and what the neighbouring point's thread writes
                const float got = dp_lane_xor2(give);
                unsigned hi, lo;
                dp_split2(odd ? got : own, odd ? own : got, hi, lo);           // (even point, odd point)
                *reinterpret_cast<unsigned*>(wbase + k * HP) = hi;
                *reinterpret_cast<unsigned*>(wbase + k * HP + Z * HP) = lo;
            }
            if (half == 1) {                                    // the label columns: S features, written by the even point's thread
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float own = zt[pnt2 * ZP + E + s] * sd;
                    const float got = dp_lane_xor2(own);
                    unsigned hi, lo;
                    dp_split2(own, got, hi, lo);
                    if (!odd) {
                        *reinterpret_cast<unsigned*>(zh + (E + s) * HP + (pnt2 >> 1) * 4) = hi;
                        *reinterpret_cast<unsigned*>(zh + (E + s) * HP + (pnt2 >> 1) * 4 + Z * HP) = lo;
                    }
                }
            }
        }
        __syncthreads();
        if (V_out) {
            float* dst = V_out + ((long)b * TF + p0) * E;
            for (int i = tid; i < npts * E; i += 256) {
                const int pnt = i / E, e = i - pnt * E;
                dst[i] = zt[pnt * ZP + e] * ivs[pnt];
            }
        }
        {
            // wave w: points 32 w .. 32 w + 31; lane (feature e_lo of a tile, points 8 slot .. 8 slot + 7)
            const unsigned char* const rb = zh + e_lo * HP + (32 * wave + 8 * slot) * 2;
            dp_f16x8 a[NT][2];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int p = 0; p < 2; ++p) a[ti][p] = *reinterpret_cast<const dp_f16x8*>(rb + ti * 16 * HP + p * Z * HP);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj) {
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ti][1], a[tj][0], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ti][0], a[tj][1], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ti][0], a[tj][0], acc[ti][tj], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    float* red = zt;                                // Z*Z <= PTS*ZP
    for (int i = tid; i < Z * Z; i += 256) red[i] = 0.f;
    __syncthreads();
    constexpr float UNSCALE = 1.0f / (8192.0f * 8192.0f);
    for (int w = 0; w < 4; ++w) {                   // fixed order: waves add one after another, mirrored tiles
        if (wave == w) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ti * 16 + slot * 4 + r, col = tj * 16 + e_lo;
                        const float g = acc[ti][tj][r] * UNSCALE;
                        red[row * Z + col] += g;
                        if (tj != ti) red[col * Z + row] += g;
                    }
        }
        __syncthreads();
    }
    float* out = part + ((long)b * nchunk + c) * (Z * Z);
    for (int i = tid; i < Z * Z; i += 256) __hip_atomic_store(out + i, red[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    __shared__ float sm[3][16];
    __shared__ int last_sh;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) last_sh = (atomicAdd(fin.ticket + 1 + b, 1u) == (unsigned)nchunk - 1u) ? 1 : 0;
    __syncthreads();
    if (!last_sh) return;                           // uniform over the workgroup
    dpcl_finish_utterance<true>(part, fin.per_utt, fin.mats, E, S, Z, nchunk, fin.B, b, red, sm);
    dpcl_finish_batch(fin.per_utt, fin.out, fin.ticket, fin.clear, fin.B, &last_sh, red, PTS * ZP);
}

// Per utterance: reduce chunk partials (fixed order), Frobenius norms, cost_b, normalised matrices for bwd.  Called by all
// threads of a workgroup (any size that is a multiple of 64, <= 1024); gram = Z*Z floats of LDS, sm = 3 x 16 floats of LDS.
// IN_LAUNCH: the partials were stored by other workgroups of the SAME launch (agent-scope stores): read them with agent-scope loads.
template <bool IN_LAUNCH>
__device__ __forceinline__ void dpcl_finish_utterance(const float* __restrict__ part, float* __restrict__ per_utt, float* __restrict__ mats,
                                                      int E, int S, int Z, int nchunk, int B, int b, float* __restrict__ gram,
                                                      float (*sm)[16]) {
    const int tid = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6;
    for (int i = tid; i < Z * Z; i += nthr) {
        // chunk partials summed in chunk order; loads issued 8 at a time (a serial chain of ~1 us round trips made this
        // 23 us for 8 chunks of a 48x48 Gram)
        const float* pp = part + (long)b * nchunk * (Z * Z) + i;
        float s = 0.f;
        int c = 0;
        for (; c + 8 <= nchunk; c += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = IN_LAUNCH ? __hip_atomic_load(pp + (long)(c + j) * (Z * Z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : pp[(long)(c + j) * (Z * Z)];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; c < nchunk; ++c)
            s += IN_LAUNCH ? __hip_atomic_load(pp + (long)c * (Z * Z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pp[(long)c * (Z * Z)];
        gram[i] = s;
    }
    __syncthreads();
    float sg = 0.f, sa = 0.f, sc = 0.f;
    for (int i = tid; i < Z * Z; i += nthr) {
        const int r = i / Z, c = i - r * Z;
        const float v = gram[i];
        if (r < E && c < E) sg += v * v;
        else if (r < E && c >= E && c < E + S) sa += v * v;
        else if (r >= E && r < E + S && c >= E && c < E + S) sc += v * v;
    }
    sg = wave_sum(sg); sa = wave_sum(sa); sc = wave_sum(sc);
    if ((tid & 63) == 0) { sm[0][tid >> 6] = sg; sm[1][tid >> 6] = sa; sm[2][tid >> 6] = sc; }
    __syncthreads();
    float tg = 0.f, ta = 0.f, tc = 0.f;
    for (int w = 0; w < nw; ++w) { tg += sm[0][w]; ta += sm[1][w]; tc += sm[2][w]; }
    const float nG = sqrtf(tg), nA = sqrtf(ta), nC = sqrtf(tc);
    if (tid == 0) {                                 // agent-scope stores: dpcl_finish_batch reads them from another workgroup
        __hip_atomic_store(per_utt + b * 4 + 0, nG - 2.0f * nA + nC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(per_utt + b * 4 + 1, nG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(per_utt + b * 4 + 2, -2.0f * nA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(per_utt + b * 4 + 3, nC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // mats[b]: Gn [E,E] then An [E,S]
    float* m = mats + (long)b * (E * E + E * S);
    const float kg = 2.0f / (nG * B), ka = 2.0f / (nA * B);
    for (int i = tid; i < E * E; i += nthr) m[i] = gram[(i / E) * Z + (i % E)] * kg;
    for (int i = tid; i < E * S; i += nthr) m[E * E + i] = gram[(i / S) * Z + E + (i % S)] * ka;
}

// The LAST workgroup to finish an utterance also writes the batch means out[0..3] (what dpcl_mean_kernel does in a launch of its own:
// 8 us on the critical path of a training step) and clears the backward's max |dU| slot; utterances are summed in index order whoever
// arrives last, so the result does not depend on the arrival order.  Called by all threads; last_sh = one int of LDS, stage = cap >= 4
// floats of LDS (may be the Gram's: the utterance is finished by then).
__device__ __forceinline__ void dpcl_finish_batch(const float* __restrict__ per_utt, float* __restrict__ out, unsigned* __restrict__ ticket,
                                                  unsigned* __restrict__ clear, int B, int* last_sh, float* __restrict__ stage, int cap) {
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // per_utt[b] (this thread's agent-scope stores) acknowledged
        *last_sh = (atomicAdd(ticket, 1u) == (unsigned)B - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (*last_sh) {
        // all 4 B values fetched in one round trip (one load per thread into LDS), then summed in utterance order from there: as a loop
        // of dependent agent-scope loads this was 64 round trips at the very end of the launch
        float t = 0.f;
        const int capu = cap / 4;                               // utterances per round
        for (int u0 = 0; u0 < B; u0 += capu) {
            const int n = min(capu, B - u0);
            if (u0) __syncthreads();
            for (int i = tid; i < 4 * n; i += blockDim.x)
                stage[i] = __hip_atomic_load(&per_utt[4 * u0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid < 4)
                for (int u = 0; u < n; ++u) t += stage[u * 4 + tid];
        }
        if (tid < 4) out[tid] = t / B;
        if (tid == 4 && clear) clear[0] = 0u;
    }
}

__global__ __launch_bounds__(1024) void dpcl_finish_kernel(const float* __restrict__ part, float* __restrict__ per_utt,
                                                           float* __restrict__ mats, int E, int S, int Z, int nchunk, int B,
                                                           float* __restrict__ out = nullptr, unsigned* __restrict__ ticket = nullptr,
                                                           unsigned* __restrict__ clear = nullptr) {
    extern __shared__ float gram[];
    __shared__ float sm[3][16];
    __shared__ int last_sh;
    dpcl_finish_utterance<false>(part, per_utt, mats, E, S, Z, nchunk, B, blockIdx.x, gram, sm);
    if (ticket) dpcl_finish_batch(per_utt, out, ticket, clear, B, &last_sh, gram, Z * Z);
}

__global__ void dpcl_mean_kernel(const float* __restrict__ per_utt, float* __restrict__ out, int B, unsigned* __restrict__ clear = nullptr) {
    // out[0] = cost, out[1..3] = the three summary terms (dpcl.py:82-85)
    const int k = threadIdx.x;
    if (k == 4 && clear) clear[0] = 0u;             // the backward's max |dU| slot
    if (k < 4) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += per_utt[b * 4 + k];
        out[k] = s / B;
    }
}

// Backward, fused with l2norm backward.  One thread per point.
// FROM_U: V points at U (pre-normalisation), cnt at the CP count partials, inv is mandatory; v = u * inv on the fly.
template <int E_, bool FROM_U>
__global__ __launch_bounds__(256) void dpcl_bwd_kernel(const float* __restrict__ V, const float* __restrict__ Y,
                                                       const float* __restrict__ cnt, const float* __restrict__ mats,
                                                       const float* __restrict__ inv, const float* __restrict__ upstream,
                                                       float* __restrict__ dU, long TF, int S, int fuse_l2norm) {
    constexpr int LDS_STRIDE = E_ + 1;
    __shared__ float tile[256 * LDS_STRIDE];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long p0 = (long)blockIdx.x * 256;
    const int npts = (int)min((long)256, TF - p0);
    const float* Vb = V + ((long)b * TF + p0) * E_;
    // coalesced load of npts*E floats -> LDS [point][E+1]
    for (int i = tid; i < npts * E_; i += 256) tile[(i / E_) * LDS_STRIDE + (i % E_)] = Vb[i];
    __syncthreads();
    const float* G = mats + (long)b * (E_ * E_ + E_ * S);
    const float* A = G + E_ * E_;
    float v[E_], dv[E_];
    float cn[8];
    if (FROM_U) load_counts(cnt, b, S, cn);
    else {
#pragma unroll
        for (int s = 0; s < 8; ++s) cn[s] = (s < S) ? cnt[(long)b * S + s] : 0.f;
    }
    if (tid < npts) {
        const long p = p0 + tid;
        const float iv_u = FROM_U ? inv[(long)b * TF + p] : 1.0f;
#pragma unroll
        for (int e = 0; e < E_; ++e) { v[e] = tile[tid * LDS_STRIDE + e] * iv_u; dv[e] = 0.f; }
        const float* y = Y + ((long)b * TF + p) * S;
        float diag = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (s < S) diag += y[s] * cn[s];
        const float d = (1.0f / sqrtf(diag)) * (upstream ? upstream[0] : 1.0f);
        // dv = Gn^T-free: G is symmetric; dv[f] = sum_e v[e] G[e][f]
        for (int e = 0; e < E_; ++e) {
            const float ve = v[e];
#pragma unroll
            for (int f = 0; f < E_; ++f) dv[f] += ve * G[e * E_ + f];
        }
        for (int s = 0; s < S; ++s) {
            const float ys = y[s];
            if (ys != 0.f) {
#pragma unroll
                for (int f = 0; f < E_; ++f) dv[f] -= ys * A[f * S + s];
            }
        }
        const bool dz = !(diag > 0.f);           // all-zero Y row: D = inf in the reference; contributes 0 * inf
#pragma unroll
        for (int f = 0; f < E_; ++f) dv[f] = dz ? 0.f : dv[f] * d;
        if (fuse_l2norm) {
            float dot = 0.f;
#pragma unroll
            for (int f = 0; f < E_; ++f) dot += v[f] * dv[f];
            const float iv = FROM_U ? iv_u : inv[(long)b * TF + p];
            const bool active = iv < 0.999999e6f;
#pragma unroll
            for (int f = 0; f < E_; ++f) dv[f] = active ? (dv[f] - v[f] * dot) * iv : dv[f] * iv;
        }
    }
    __syncthreads();
    if (tid < npts) {
#pragma unroll
        for (int f = 0; f < E_; ++f) tile[tid * LDS_STRIDE + f] = dv[f];
    }
    __syncthreads();
    float* Ub = dU + ((long)b * TF + p0) * E_;
    for (int i = tid; i < npts * E_; i += 256) Ub[i] = tile[(i / E_) * LDS_STRIDE + (i % E_)];
}


// Backward from U on the matrix cores.  dV_p = D_p * ([v_p | y_p] . M) with M = [Gn ; -An^T] (Z x E, zero padded), i.e.
// a [points x Z] . [Z x E] product per utterance: 16-point groups run v_mfma_f32_16x16x4_f32 chains against M held in
// registers as B fragments (one load per workgroup), the A fragments come from the staged rows in LDS scaled by 1/|u|
// on the fly.  The l2-normalise Jacobian (a 16-lane dot per point) is applied on the accumulators and dU goes back
// through LDS so global traffic is 16-byte coalesced both ways; slab i+1 is fetched while slab i is in the MFMA phase.
// Algorithmic HBM bytes per utterance: TF*(2E+S+1)*4.
template <int NT, int EC, int SC>                  // EC / SC: compile-time E / S (0 = run-time; EC != 0 implies 16-byte rows)
__global__ __launch_bounds__(256, 2) void dpcl_bwd_u_kernel(const float* __restrict__ U, const float* __restrict__ Y,
                                                         const float* __restrict__ cntp, const float* __restrict__ mats,
                                                         const float* __restrict__ inv, const float* __restrict__ upstream,
                                                         float* __restrict__ dU, long TF, int E_rt, int S_rt, unsigned* __restrict__ amax_out) {
    const int E = EC ? EC : E_rt;
    float amax_f = 0.f;                             // max |dU| seen by this lane (v_max3_f32 with |.| modifiers)
    const int S = SC ? SC : S_rt;
    constexpr int Z = NT * 16, ZP = Z + 4, KT = Z / 4;
    constexpr int PTS = NT <= 3 ? AMS_DPCL_PTS : 128;
    constexpr int NV = PTS * Z / 4 / 256;
    __shared__ __attribute__((aligned(16))) float zt[PTS * ZP];
    __shared__ float dsh[PTS], ivs[PTS];
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e_lo = lane & 15, slot = lane >> 4;
    for (int i = tid; i < PTS * ZP; i += 256) zt[i] = 0.f;

    float cn[8];
    load_counts(cntp, b, S, cn);
    const float up = upstream ? upstream[0] : 1.0f;

    // B fragments: lane (slot, e_lo) holds M[kt*4 + slot][ft*16 + e_lo]
    float bm[KT][NT];
    {
        const float* G = mats + (long)b * (E * E + E * S);
        const float* A = G + E * E;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int ft = 0; ft < NT; ++ft) {
                const int k = kt * 4 + slot, f = ft * 16 + e_lo;
                float v = 0.f;
                if (f < E) {
                    if (k < E) v = G[k * E + f];
                    else if (k < E + S) v = -A[f * S + (k - E)];
                }
                bm[kt][ft] = v;
            }
    }

    const long p_begin = (long)c * BCHUNK, p_end = min(TF, p_begin + BCHUNK);
    const float* Ub = U + (long)b * TF * E;
    const float* Yb = Y + (long)b * TF * S;
    const float* ib = inv + (long)b * TF;
    float* dUb = dU + (long)b * TF * E;
    const bool vec = EC ? true : ((E % 4 == 0) && (((uintptr_t)U & 15) == 0) && (((uintptr_t)dU & 15) == 0));
    const int nvec = PTS * E / 4;

    float4 pre[NV];
    float yv[8];
    float ivp = 0.f;
    // Loads are UNCONDITIONAL with clamped indices (out-of-range lanes re-read the last valid vector and are zeroed when the
    // slab is stored to LDS): exec-masked branches around the loads made hipcc fall back to s_waitcnt vmcnt(0) right after
    // issuing them, which serialised the "prefetch" with the HBM round trip.
    auto fetch = [&](long p0) {
        const int npts = (int)min((long)PTS, p_end - p0);
        if (vec) {
            const float4* src = reinterpret_cast<const float4*>(Ub + p0 * E);
            const int last = npts * E / 4 - 1;
#pragma unroll
            for (int j = 0; j < NV; ++j) pre[j] = src[min(tid + 256 * j, last)];
        } else {
            const float* src = Ub + p0 * E;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = tid + 256 * (4 * j + q);
                    t[q] = (i < npts * E) ? src[i] : 0.f;
                }
                pre[j] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
        if (tid < PTS) {
            const int pt = min(tid, npts - 1);
            const float* yr = Yb + (p0 + pt) * S;
#pragma unroll
            for (int s = 0; s < 8; ++s) yv[s] = (s < S) ? yr[s] : 0.f;
            ivp = ib[p0 + pt];
        }
    };

    if (p_begin < p_end) fetch(p_begin);
    for (long p0 = p_begin; p0 < p_end; p0 += PTS) {
        const int npts = (int)min((long)PTS, p_end - p0);
        __syncthreads();                            // write-out of the previous slab is done with zt
        if (vec) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i4 = tid + 256 * j;
                if (i4 < nvec) {
                    const int pnt = (i4 * 4) / E, e = i4 * 4 - pnt * E;
                    *reinterpret_cast<float4*>(&zt[pnt * ZP + e]) = (i4 * 4 < npts * E) ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float t[4] = {pre[j].x, pre[j].y, pre[j].z, pre[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = tid + 256 * (4 * j + q);
                    if (i < PTS * E) zt[(i / E) * ZP + (i % E)] = t[q];
                }
            }
        }
        if (tid < PTS) {
            float diag = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (s < S) { const float y = tid < npts ? yv[s] : 0.f; zt[tid * ZP + E + s] = y; diag += y * cn[s]; }
            dsh[tid] = (tid < npts && diag > 0.f) ? up / sqrtf(diag) : 0.f;     // all-zero Y row contributes nothing
            ivs[tid] = tid < npts ? ivp : 0.f;
        }
        __syncthreads();
        if (p0 + PTS < p_end && !(AMS_DPCL_DBG & 4)) fetch(p0 + PTS);

#pragma unroll 1
        for (int gi = 0; gi < PTS / 64; ++gi) {
            const int pb = wave * (PTS / 4) + gi * 16;
            const float iv_a = ivs[pb + e_lo];
            f32x4 acc[NT];
#pragma unroll
            for (int ft = 0; ft < NT; ++ft) acc[ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* arow = &zt[(pb + e_lo) * ZP + slot];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                float a = arow[kt * 4];
                a = (kt * 4 + slot < E) ? a * iv_a : a;
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) {
                    if (AMS_DPCL_DBG & 1) acc[ft][kt & 3] += a * bm[kt][ft];
                    else acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bm[kt][ft], acc[ft], 0, 0, 0);
                }
            }
            // accumulator layout: point = pb + slot*4 + r, column f = ft*16 + e_lo
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pnt = pb + slot * 4 + r;
                if (AMS_DPCL_DBG & 2) {
#pragma unroll
                    for (int ft = 0; ft < NT; ++ft) if (ft * 16 + e_lo < E) zt[pnt * ZP + ft * 16 + e_lo] = acc[ft][r];
                    continue;
                }
                const float d = dsh[pnt], iv = ivs[pnt];
                float vv[NT], dd[NT], dot = 0.f;
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) {
                    const int f = ft * 16 + e_lo;
                    vv[ft] = (f < E) ? zt[pnt * ZP + f] * iv : 0.f;
                    dd[ft] = acc[ft][r] * d;
                    dot += vv[ft] * dd[ft];
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 16);
                const bool active = iv < 0.999999e6f;
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) {
                    const int f = ft * 16 + e_lo;
                    if (f < E) zt[pnt * ZP + f] = active ? (dd[ft] - vv[ft] * dot) * iv : dd[ft] * iv;
                }
            }
        }
        __syncthreads();
        if (vec) {
            float4* dst = reinterpret_cast<float4*>(dUb + p0 * E);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i4 = tid + 256 * j;
                if (i4 < nvec && i4 * 4 < npts * E && (!(AMS_DPCL_DBG & 8) || zt[0] == 12345.f)) {
                    const int pnt = (i4 * 4) / E, e = i4 * 4 - pnt * E;
                    const float4 o = *reinterpret_cast<const float4*>(&zt[pnt * ZP + e]);
                    dst[i4] = o;
                    amax_f = fmaxf(fmaxf(amax_f, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                }
            }
        } else {
            float* dst = dUb + p0 * E;
            for (int i = tid; i < npts * E; i += 256) { const float o = zt[(i / E) * ZP + (i % E)]; dst[i] = o; amax_f = fmaxf(amax_f, fabsf(o)); }
        }
    }
    if (amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax_f = fmaxf(amax_f, __shfl_xor(amax_f, o));
        if (lane == 0) atomicMax(amax_out, __float_as_uint(amax_f));    // a NaN in dU is dropped here and reaches the products through dU itself
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the fused BACKWARD pass without the LDS transposes (E % 4 == 0, E + S <= 48, S <= 4: the recipes' E = 40).
// The anatomy of the staged kernels (AMS_DPCL_DBG builds, cold caches, 64 x 20480 x 40): backward 137 us, without its MFMAs 129,
// without its epilogue 119, without ANY global traffic 107, with nothing but the staging and its four barriers per slab 37 --
// neither HBM nor the matrix pipe but the chain load -> LDS -> barrier -> MFMA -> LDS -> barrier -> store, two workgroups per CU.
// The backward product can take its operands straight from global memory in the layout the MFMA wants, because an MFMA's k
// order is free as long as A and B agree:
//   dV^T = M^T . Z^T  (features x points): B operand lane (slot, e) = point e, k = 16 j + 4 slot + i for MFMA (j, i) -- exactly
//   the three float4 a lane loads from its point's row; the D registers of feature tile ft are features 16 ft + 4 slot + r of
//   point e: the SAME float4 positions, so the l2-normalise Jacobian needs the lane's own loads and two cross-slot shuffles,
//   and dU leaves as float4.  No LDS, no barrier, waves independent.
// (The Gram pass was tried the same way -- A lane (slot, e) = feature e of point slot from 4-byte loads, |u|^2 as a 16-lane
// reduction -- and was slower, 163 vs 110 us: 1/|u| and D are then computed once per FOUR points per instruction instead of once
// per 64, and that VALU work outweighs the barriers it removes.  It keeps the staged form.)
constexpr int BCH2 = 1024;             // points per 256-thread workgroup, backward (16 groups of 16 points per wave)

template <int EC, int SC>
__global__ __launch_bounds__(256) void dpcl_bwd_u2_kernel(const float* __restrict__ U, const float* __restrict__ Y,
                                                          const float* __restrict__ cntp, const float* __restrict__ mats,
                                                          const float* __restrict__ inv, const float* __restrict__ upstream,
                                                          float* __restrict__ dU, long TF, unsigned* __restrict__ amax_out) {
    constexpr int E = EC, S = SC, NT = 3, NJ = 3;
    float amax_f = 0.f;                             // max |dU| seen by this lane (v_max3_f32 with |.| modifiers)
    static_assert(E % 4 == 0 && E + S <= 48 && S <= 4 && E > 32, "layout of dpcl_bwd_u2_kernel");
    const int b = blockIdx.y, c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e_lo = lane & 15, slot = lane >> 4;
    float cn[8];
    load_counts(cntp, b, S, cn);
    const float up = upstream ? upstream[0] : 1.0f;
    // A fragments: MFMA (j, i) of feature tile ft: M[k = 16 j + 4 slot + i][f = 16 ft + e_lo],  M = [Gn ; -An^T]
    float am[NJ][4][NT];
    {
        const float* G = mats + (long)b * (E * E + E * S);
        const float* A = G + E * E;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) {
                    const int k = 16 * j + 4 * slot + i, f = ft * 16 + e_lo;
                    float v = 0.f;
                    if (f < E) {
                        if (k < E) v = G[k * E + f];
                        else if (k < E + S) v = -A[f * S + (k - E)];
                    }
                    am[j][i][ft] = v;
                }
    }
#if AMS_DPCL_BWD_F16
    // fp16x3 (csrc/gemm.hip): M scaled by a power of two from its own maximum and split exactly into two fp16 terms, once per workgroup;
    // z (|u / |u|| <= 1, labels 0 / 1) scaled by 2^13 and split per group.  A lane's four k-slots of block j are exactly its A / B
    // fragment of a v_mfma_f32_16x16x16_f16, blocks 0 and 1 together those of a 16x16x32: 18 MFMAs of ~16 cycles per group of 16 points
    // instead of 36 v_mfma_f32_16x16x4_f32 of 32 (38 us of matrix pipe per launch at 64 x 20480 points).
    typedef _Float16 dh8 __attribute__((ext_vector_type(8)));
    typedef _Float16 dh4 __attribute__((ext_vector_type(4)));
    dh8 a8[NT][2];                                  // [feature tile][plane hi / lo], k-blocks 0 and 1
    dh4 a4[NT][2];                                  //                                 k-block 2
    float m_inv;
    {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) mx = fmaxf(mx, fabsf(am[j][i][ft]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const int ex = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        const int se = 127 + 13 - (ex - 127);
        const bool ok = ex != 0 && ex != 255 && se >= 1 && se <= 253;
        const float m_sc = ok ? __uint_as_float((unsigned)se << 23) : 1.0f;
        m_inv = (ok ? __uint_as_float((unsigned)(254 - se) << 23) : 1.0f) * (1.0f / 8192.0f);
#pragma unroll
        for (int ft = 0; ft < NT; ++ft) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = am[e >> 2][e & 3][ft] * m_sc;
                const _Float16 h = (_Float16)v;
                a8[ft][0][e] = h; a8[ft][1][e] = (_Float16)(v - (float)h);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = am[2][e][ft] * m_sc;
                const _Float16 h = (_Float16)v;
                a4[ft][0][e] = h; a4[ft][1][e] = (_Float16)(v - (float)h);
            }
        }
    }
#endif
    const long p_begin = (long)c * BCH2, p_end = min(TF, p_begin + BCH2);
    const float* Ub = U + (long)b * TF * E;
    const float* Yb = Y + (long)b * TF * S;
    const float* ib = inv + (long)b * TF;
    float* dUb = dU + (long)b * TF * E;
    constexpr int NG = BCH2 / 16 / 4;               // groups per wave
    // which of the lane's three float4 are embedding values, which one is the label vector
    const bool uval[NJ] = {true, true, 32 + 4 * slot < E};
    const bool ylane = (32 + 4 * slot == E);

    struct Grp { float4 q[NJ]; float y[4]; float iv; };
    auto fetch = [&](long p, Grp& g) {
        const long pc = min(p, p_end - 1);          // clamped: lanes past the end redo the last point and are discarded
        const float4* row = reinterpret_cast<const float4*>(Ub + pc * E);
        g.q[0] = row[slot];
        g.q[1] = row[4 + slot];
        g.q[2] = row[min(8 + slot, E / 4 - 1)];
#pragma unroll
        for (int s = 0; s < 4; ++s) g.y[s] = (s < S) ? Yb[pc * S + s] : 0.f;
        g.iv = ib[pc];
    };
    auto compute = [&](long g0, const Grp& g) {
        const long p = g0 + e_lo;
        const bool live = p < p_end;
        float diag = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) diag += g.y[s] * cn[s];
        const float d = (live && diag > 0.f) ? up / sqrtf(diag) : 0.f;          // all-zero Y row contributes nothing
        const float iv = g.iv;
        float z[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float t[4] = {g.q[j].x, g.q[j].y, g.q[j].z, g.q[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) z[j][i] = uval[j] ? t[i] * iv : 0.f;
        }
        if (ylane) {
#pragma unroll
            for (int i = 0; i < 4; ++i) z[NJ - 1][i] = g.y[i];
        }
        float dd[NT][4];
#if !AMS_DPCL_BWD_F16
        f32x4 acc[NT];
#pragma unroll
        for (int ft = 0; ft < NT; ++ft) acc[ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
#endif
#if AMS_DPCL_BWD_F16
        {
            dh8 z8[2];
            dh4 z4[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = z[e >> 2][e & 3] * 8192.0f;
                const _Float16 h = (_Float16)v;
                z8[0][e] = h; z8[1][e] = (_Float16)(v - (float)h);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = z[2][e] * 8192.0f;
                const _Float16 h = (_Float16)v;
                z4[0][e] = h; z4[1][e] = (_Float16)(v - (float)h);
            }
            // per feature tile two chains from 0, one per MFMA shape, alternating, joined by one add (an accumulator must not pass from one
            // shape to the other: csrc/lstm_ring.hip); smallest terms first: lo.hi, hi.lo, hi.hi.  Tile by tile only two partial sums
            // are live: 160 registers instead of 176 -- three waves per SIMD again (round 6)
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ft = 0; ft < NT; ++ft) {
                f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[ft][PA[0]], z8[PB[0]], zero4, 0, 0, 0);
                f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[ft][PA[0]], z4[PB[0]], zero4, 0, 0, 0);
                r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[ft][PA[1]], z8[PB[1]], r32, 0, 0, 0);
                r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[ft][PA[1]], z4[PB[1]], r16, 0, 0, 0);
                r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8[ft][PA[2]], z8[PB[2]], r32, 0, 0, 0);
                r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[ft][PA[2]], z4[PB[2]], r16, 0, 0, 0);
                const f32x4 t = (r32 + r16) * (m_inv * d);
                dd[ft][0] = t[0]; dd[ft][1] = t[1]; dd[ft][2] = t[2]; dd[ft][3] = t[3];
            }
        }
#else
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ft = 0; ft < NT; ++ft) {
                    if (AMS_DPCL_DBG & 1) acc[ft][i] += am[j][i][ft] * z[j][i];
                    else acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[j][i][ft], z[j][i], acc[ft], 0, 0, 0);
                }
#endif
        // acc[ft][r]: feature 16 ft + 4 slot + r of point e_lo -- the positions of the lane's own float4 number ft
        float dot = 0.f;
#pragma unroll
        for (int ft = 0; ft < NT; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#if !AMS_DPCL_BWD_F16
                dd[ft][r] = acc[ft][r] * d;
#endif
                if (uval[ft]) dot += z[ft][r] * dd[ft][r];
            }
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        const bool active = iv < 0.999999e6f;       // |u|^2 was clamped to 1e-12: the norm's gradient is zero there
#pragma unroll
        for (int ft = 0; ft < NT; ++ft) {
            if (uval[ft] && live && !(AMS_DPCL_DBG & 8)) {
                float4 o;
                o.x = active ? (dd[ft][0] - z[ft][0] * dot) * iv : dd[ft][0] * iv;
                o.y = active ? (dd[ft][1] - z[ft][1] * dot) * iv : dd[ft][1] * iv;
                o.z = active ? (dd[ft][2] - z[ft][2] * dot) * iv : dd[ft][2] * iv;
                o.w = active ? (dd[ft][3] - z[ft][3] * dot) * iv : dd[ft][3] * iv;
                reinterpret_cast<float4*>(dUb + p * E)[4 * ft + slot] = o;
                amax_f = fmaxf(fmaxf(amax_f, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    };
    // three register buffers in rotation: while group n is in the MFMAs, groups n + 1 and n + 2 are in flight (one wave keeps
    // 5 KB outstanding, a CU's 16 waves 80 KB -- what 6 TB/s x 3 us of loaded latency asks of each of the 256 CUs)
    Grp g0b, g1b, g2b;
    const long w0 = p_begin + (long)wave * (NG * 16);
    const long w_end = min(p_end, w0 + (long)NG * 16);
    if (w0 < w_end) fetch(w0 + e_lo, g0b);
    if (w0 + 16 < w_end) fetch(w0 + 16 + e_lo, g1b);
#pragma unroll 1
    for (long g = w0; g < w_end; g += 48) {
        if (g + 32 < w_end) fetch(g + 32 + e_lo, g2b);
        compute(g, g0b);
        if (g + 16 >= w_end) break;
        if (g + 48 < w_end) fetch(g + 48 + e_lo, g0b);
        compute(g + 16, g1b);
        if (g + 32 >= w_end) break;
        if (g + 64 < w_end) fetch(g + 64 + e_lo, g1b);
        compute(g + 32, g2b);
    }
    if (amax_out && AMS_DPCL_AMAX) {                // max |dU|: the operand bound of the products that read dU
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax_f = fmaxf(amax_f, __shfl_xor(amax_f, o));
        if (AMS_DPCL_AMAX == 2) {                   // one atomic per workgroup
            __shared__ float wm[4];
            if (lane == 0) wm[wave] = amax_f;
            __syncthreads();
            if (tid == 0) atomicMax(amax_out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
        } else if (lane == 0) atomicMax(amax_out, __float_as_uint(amax_f));    // a NaN in dU is dropped here and reaches the products through dU itself
    }
}

// AMS_DPCL_GRAM_F16=0 (read once): the fused forward keeps its Gram on v_mfma_f32_16x16x4_f32 (A/B runs; the tests hold both)
inline bool dpcl_gram_f16() {
    static const bool v = !(getenv("AMS_DPCL_GRAM_F16") && atoi(getenv("AMS_DPCL_GRAM_F16")) == 0);
    return v;
}

// AMS_DPCL_LDS=1 (read once): the LDS-staged passes of round 2 also where the direct ones apply (A/B runs, tests hold both)
inline bool dpcl_direct() {
    static const bool v = !(getenv("AMS_DPCL_LDS") && atoi(getenv("AMS_DPCL_LDS")) != 0);
    return v;
}

}  // namespace

extern "C" {

size_t ams_dpcl_workspace_bytes(int B, long TF, int E, int S) {
    const int NT = ceil_div(E + S, 16), Z = NT * 16;
    const int nchunk = ceil_div(TF, CHUNK);
    // cnt [B,S] | per_utt [B,4] | mats [B, E*E+E*S] | partials [B, nchunk, Z*Z]
    return sizeof(float) * ((size_t)B * S + (size_t)B * 4 + (size_t)B * (E * E + E * S) + (size_t)B * nchunk * Z * Z);
}

// out[0] = cost, out[1..3] = mean of the three terms.  ws keeps cnt/mats for ams_dpcl_loss_bwd.
ams_status ams_dpcl_loss_fwd(const float* V, const float* Y, float* out, int B, long TF, int E, int S, void* ws, size_t ws_bytes,
                             void* stream) {
    AMS_REQUIRE(V && Y && out && ws && B > 0 && TF > 0 && E > 0 && S > 0 && S <= 8 && E + S <= 64);
    if (ws_bytes < ams_dpcl_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int NT = ceil_div(E + S, 16), Z = NT * 16, nchunk = ceil_div(TF, CHUNK);
    float* cnt = (float*)ws;
    float* per_utt = cnt + (size_t)B * S;
    float* mats = per_utt + (size_t)B * 4;
    float* part = mats + (size_t)B * (E * E + E * S);
    hipLaunchKernelGGL(dpcl_count_kernel, dim3(B), dim3(256), 0, st, Y, cnt, TF, S);
    dim3 grid(nchunk, B);
    switch (NT) {
        case 1: hipLaunchKernelGGL((dpcl_gram_kernel<1>), grid, dim3(256), 0, st, V, Y, cnt, part, TF, E, S, nchunk); break;
        case 2: hipLaunchKernelGGL((dpcl_gram_kernel<2>), grid, dim3(256), 0, st, V, Y, cnt, part, TF, E, S, nchunk); break;
        case 3: hipLaunchKernelGGL((dpcl_gram_kernel<3>), grid, dim3(256), 0, st, V, Y, cnt, part, TF, E, S, nchunk); break;
        default: hipLaunchKernelGGL((dpcl_gram_kernel<4>), grid, dim3(256), 0, st, V, Y, cnt, part, TF, E, S, nchunk); break;
    }
    hipLaunchKernelGGL(dpcl_finish_kernel, dim3(B), dim3(1024), Z * Z * sizeof(float), st, part, per_utt, mats, E, S, Z, nchunk, B);
    hipLaunchKernelGGL(dpcl_mean_kernel, dim3(1), dim3(64), 0, st, per_utt, out, B);
    return ams_check_launch();
}

// dU (or dV when inv == NULL) from the state a preceding ams_dpcl_loss_fwd left in ws; upstream (device scalar,
// may be NULL) multiplies the result (chain rule for d loss / d cost).
ams_status ams_dpcl_loss_bwd(const float* V, const float* Y, const float* inv, const float* upstream, float* dU, int B, long TF,
                             int E, int S, const void* ws, void* stream) {
    AMS_REQUIRE(V && Y && dU && ws && B > 0 && TF > 0 && S > 0 && S <= 8);
    hipStream_t st = (hipStream_t)stream;
    const float* cnt = (const float*)ws;
    const float* mats = cnt + (size_t)B * S + (size_t)B * 4;
    dim3 grid(ceil_div(TF, 256), B);
    const int fuse = inv ? 1 : 0;
#define AMS_DPCL_BWD(EE) \
    hipLaunchKernelGGL((dpcl_bwd_kernel<EE, false>), grid, dim3(256), 0, st, V, Y, cnt, mats, inv, upstream, dU, TF, S, fuse)
    switch (E) {
        case 40: AMS_DPCL_BWD(40); break;
        case 32: AMS_DPCL_BWD(32); break;
        case 20: AMS_DPCL_BWD(20); break;
        case 16: AMS_DPCL_BWD(16); break;
        case 8: AMS_DPCL_BWD(8); break;
        case 4: AMS_DPCL_BWD(4); break;
        case 3: AMS_DPCL_BWD(3); break;
        default: return AMS_E_INVALID_ARG;
    }
#undef AMS_DPCL_BWD
    return ams_check_launch();
}

// byte offset, inside the ams_dpcl_loss_fwd_u workspace, of max |dU| (as a float): cleared by the forward, filled by ams_dpcl_loss_bwd_u --
// the operand bound of the two dense-layer products that read dU
size_t ams_dpcl_u_amax_offset(int B, long TF, int E, int S) {
    const int NT = ceil_div(E + S, 16), Z = NT * 16;
    const int nchunk = ceil_div(TF, UCHUNK);
    const size_t n = sizeof(float) * ((size_t)B * CP * S + (size_t)B * 4 + (size_t)B * (E * E + E * S) + (size_t)B * nchunk * Z * Z);
    return (n + 15) / 16 * 16;
}

size_t ams_dpcl_u_workspace_bytes(int B, long TF, int E, int S) {
    // cntp [B,CP,S] | per_utt [B,4] | mats [B, E*E+E*S] | partials [B, nchunk, Z*Z] | bound of dU (1 float) | B + 1 arrival counters
    return ams_dpcl_u_amax_offset(B, TF, E, S) + ((size_t)(B + 2) * 4 + 15) / 16 * 16;
}

// The label counts of ams_dpcl_loss_fwd_u, as a call of their own: Y is known long before U (it comes from the front end, U from the
// whole recurrent stack), so a training step runs this beside the recurrence and passes counts_ready = 1 later.
ams_status ams_dpcl_u_count_labels(const float* Y, int B, long TF, int E, int S, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(Y && ws && B > 0 && TF > 0 && E > 0 && S > 0 && S <= 8 && E + S <= 64);
    if (ws_bytes < ams_dpcl_u_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    unsigned* const slot = (unsigned*)((char*)ws + ams_dpcl_u_amax_offset(B, TF, E, S));
    hipLaunchKernelGGL(dpcl_count_part_kernel<false>, dim3(CP, B), dim3(CNT_THREADS), 0, (hipStream_t)stream, Y, (float*)ws, TF, S, slot + 1);
    return ams_check_launch();
}

// ams_make_masks (without the argmax output) and ams_dpcl_u_count_labels in ONE pass: the labels are counted as they are written.
ams_status ams_dpcl_u_make_masks(const float* rep_non_mix, float* Y, int B, int S, long TF, int E, float a, float b, int take_abs,
                                 void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(rep_non_mix && Y && ws && B > 0 && TF > 0 && E > 0 && S > 0 && S <= 8 && E + S <= 64);
    if (ws_bytes < ams_dpcl_u_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    unsigned* const slot = (unsigned*)((char*)ws + ams_dpcl_u_amax_offset(B, TF, E, S));
    hipLaunchKernelGGL(dpcl_count_part_kernel<true>, dim3(CP, B), dim3(CNT_THREADS), 0, (hipStream_t)stream, rep_non_mix, (float*)ws, TF, S,
                       slot + 1, Y, a, b, take_abs);
    return ams_check_launch();
}

// Fused l2-normalise + loss forward on the dense output U [B,TF,E].  inv [B,TF] receives 1/|u| (needed by the
// backward); V_out (may be NULL) receives the normalised embeddings.  ws keeps counts/mats for ams_dpcl_loss_bwd_u.
ams_status ams_dpcl_loss_fwd_u(const float* U, const float* Y, float* inv, float* V_out, float* out, int B, long TF, int E, int S,
                               int counts_ready, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(U && Y && inv && out && ws && B > 0 && TF > 0 && E > 0 && S > 0 && S <= 8 && E + S <= 64);
    if (ws_bytes < ams_dpcl_u_workspace_bytes(B, TF, E, S)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    const int NT = ceil_div(E + S, 16), nchunk = ceil_div(TF, UCHUNK);
    float* cntp = (float*)ws;
    float* per_utt = cntp + (size_t)B * CP * S;
    float* mats = per_utt + (size_t)B * 4;
    float* part = mats + (size_t)B * (E * E + E * S);
    unsigned* const slot = (unsigned*)((char*)ws + ams_dpcl_u_amax_offset(B, TF, E, S));      // word 0: max |dU|, words 1 .. B + 1: arrival counters
    if (!counts_ready) hipLaunchKernelGGL(dpcl_count_part_kernel<false>, dim3(CP, B), dim3(CNT_THREADS), 0, st, Y, cntp, TF, S, slot + 1);
    const GramFinish fin = {per_utt, mats, out, slot + 1, slot, B};
    dim3 grid(nchunk, B);
    switch (NT) {
        case 1: hipLaunchKernelGGL((dpcl_gram_u_kernel<1, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin); break;
        case 2: hipLaunchKernelGGL((dpcl_gram_u_kernel<2, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin); break;
        case 3: {
            const bool al = (((uintptr_t)U & 15) == 0);
            if (E == 40 && S == 2 && al && dpcl_gram_f16()) hipLaunchKernelGGL((dpcl_gram_u16_kernel<40, 2>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, nchunk, fin);
            else if (E == 40 && S == 3 && al && dpcl_gram_f16()) hipLaunchKernelGGL((dpcl_gram_u16_kernel<40, 3>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, nchunk, fin);
            else if (E == 40 && S == 2 && al) hipLaunchKernelGGL((dpcl_gram_u_kernel<3, 40, 2>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin);
            else if (E == 40 && S == 3 && al) hipLaunchKernelGGL((dpcl_gram_u_kernel<3, 40, 3>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin);
            else hipLaunchKernelGGL((dpcl_gram_u_kernel<3, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin);
            break;
        }
        default: hipLaunchKernelGGL((dpcl_gram_u_kernel<4, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, inv, V_out, part, TF, E, S, nchunk, fin); break;
    }
    return ams_check_launch();
}

// dU from U, 1/|u| and the state ams_dpcl_loss_fwd_u left in ws.
ams_status ams_dpcl_loss_bwd_u(const float* U, const float* Y, const float* inv, const float* upstream, float* dU, int B, long TF,
                               int E, int S, const void* ws, void* stream) {
    AMS_REQUIRE(U && Y && inv && dU && ws && B > 0 && TF > 0 && S > 0 && S <= 8);
    hipStream_t st = (hipStream_t)stream;
    const float* cntp = (const float*)ws;
    const float* mats = cntp + (size_t)B * CP * S + (size_t)B * 4;
    const int NT = ceil_div(E + S, 16);
    unsigned* const amax_out = (unsigned*)((char*)const_cast<void*>(ws) + ams_dpcl_u_amax_offset(B, TF, E, S));     // cleared by ams_dpcl_loss_fwd_u
    if (E + S > 64) return AMS_E_INVALID_ARG;
    dim3 grid(ceil_div(TF, BCHUNK), B);
    switch (NT) {
        case 1: hipLaunchKernelGGL((dpcl_bwd_u_kernel<1, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out); break;
        case 2: hipLaunchKernelGGL((dpcl_bwd_u_kernel<2, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out); break;
        case 3: {
            const bool al = (((uintptr_t)U & 15) == 0) && (((uintptr_t)dU & 15) == 0);
            const dim3 grid2(ceil_div(TF, BCH2), B);
            if (E == 40 && S == 2 && al && dpcl_direct()) hipLaunchKernelGGL((dpcl_bwd_u2_kernel<40, 2>), grid2, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, amax_out);
            else if (E == 40 && S == 3 && al && dpcl_direct()) hipLaunchKernelGGL((dpcl_bwd_u2_kernel<40, 3>), grid2, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, amax_out);
            else if (E == 40 && S == 2 && al) hipLaunchKernelGGL((dpcl_bwd_u_kernel<3, 40, 2>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out);
            else if (E == 40 && S == 3 && al) hipLaunchKernelGGL((dpcl_bwd_u_kernel<3, 40, 3>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out);
            else hipLaunchKernelGGL((dpcl_bwd_u_kernel<3, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out);
            break;
        }
        default: hipLaunchKernelGGL((dpcl_bwd_u_kernel<4, 0, 0>), grid, dim3(256), 0, st, U, Y, cntp, mats, inv, upstream, dU, TF, E, S, amax_out); break;
    }
    return ams_check_launch();
}

}  // extern "C"
