// BSS-eval (SDR / SIR / SAR) for gfx950 -- reference utils/bss_eval.py:586-748 (cupy path).
//
// The reference recomputes, for every (estimate, reference) pair, the FFTs of the references, the Toeplitz Gram matrix of
// their delayed copies (flen = 512 lags) and a dense LU solve.  None of that depends on the estimate, so here per utterance:
//   1. ONE batched real FFT of the nsrc references and nsrc estimates (zero padded to n = 2^ceil(log2(nsampl+flen-1)));
//   2. cross-spectra -> ONE batched inverse FFT -> all auto/cross-correlations (reference pairs and reference x estimate);
//   3. the (nsrc*flen)^2 Gram matrix assembled ONCE from them (its diagonal blocks are the single-reference Grams), with the
//      reference's block write order (:696-702: the later write wins) reproduced exactly;
//   4. Cholesky factorisations (hipSOLVER potrf): 1 full + nsrc single, then potrs with all estimates as right-hand sides;
//   5. the distortion filters applied by spectral multiplication (one batched FFT each way), sums of squares in float64 with a
//      fixed two-stage reduction order, criteria with the reference's 10 log10(num / (den + 1e-12)).
// float64 throughout, as the reference (:595-596).
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <hipsolver/hipsolver.h>
#include <math.h>
#include <stdlib.h>
#include "../../../include/ams_bss.h"

struct ams_bss_ctx {
    int S, L, F, n, nc, Lp, NP;
    hipfftHandle fwd_in, inv_corr, fwd_c, inv_proj;
    hipsolverHandle_t solver;
    int lwork;
    size_t ws_bytes;
    // offsets (in bytes) into the caller's workspace
    size_t o_tpad, o_spec, o_xs, o_corr, o_G, o_Gj, o_Df, o_Dj, o_cpad, o_cspec, o_pspec, o_proj, o_part, o_work, o_info;
};

namespace {

constexpr int RB = 256;            // blocks per (e,j) pair in the first reduction stage

__global__ void pad_kernel(const double* __restrict__ ref, const double* __restrict__ est, double* __restrict__ tpad, int S, int L,
                           int n) {
    const int r = blockIdx.y;
    const double* src = r < S ? ref + (long)r * L : est + (long)(r - S) * L;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) tpad[(long)r * n + t] = t < L ? src[t] : 0.0;
}

// pair p < S(S+1)/2: references (i >= j), row-major over i then j;  p >= that: (reference i, estimate e)
__device__ __forceinline__ void pair_of(int p, int S, int& a, int& b) {
    const int nrr = S * (S + 1) / 2;
    if (p < nrr) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= p) ++i;
        a = i; b = p - i * (i + 1) / 2;
    } else {
        const int q = p - nrr;
        a = q / S; b = S + q % S;
    }
}
__global__ void cross_kernel(const hipfftDoubleComplex* __restrict__ spec, hipfftDoubleComplex* __restrict__ xs, int S, int nc) {
    int a, b;
    pair_of(blockIdx.y, S, a, b);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nc; k += gridDim.x * blockDim.x) {
        const hipfftDoubleComplex u = spec[(long)a * nc + k], v = spec[(long)b * nc + k];
        xs[(long)blockIdx.y * nc + k] = make_hipDoubleComplex(u.x * v.x + u.y * v.y, u.y * v.x - u.x * v.y);      // u * conj(v)
    }
}

__device__ __forceinline__ int rr_index(int i, int j) { return i * (i + 1) / 2 + j; }      // i >= j

// G[(I,a),(J,b)] with the reference's overwrite order (see header comment); column-major == row-major (symmetric up to rounding,
// and the solver only reads the lower triangle of the column-major view = the upper triangle of this row-major fill).
__global__ void gram_kernel(const double* __restrict__ corr, double* __restrict__ G, double* __restrict__ Gj, int S, int F, int n) {
    const int N = S * F;
    const double sc = 1.0 / n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)N * N; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / N), col = (int)(idx % N);
        const int I = row / F, a = row % F, J = col / F, b = col % F;
        double v;
        if (I > J) v = corr[(long)rr_index(I, J) * n + ((b - a) % n + n) % n];
        else if (I < J) v = corr[(long)rr_index(J, I) * n + ((a - b) % n + n) % n];
        else v = corr[(long)rr_index(I, I) * n + ((a - b) % n + n) % n];
        v *= sc;
        // hipSOLVER is column-major: element (row, col) of the mathematical matrix lives at col*N + row
        G[(long)col * N + row] = v;
        if (I == J) Gj[(long)I * F * F + (long)b * F + a] = v;
    }
}

// right-hand sides: Dfull column e (length S*F), Dj[j] column e (length F):  D[i*F + k] = corr_(i,e)[(n - k) mod n] / n
__global__ void rhs_kernel(const double* __restrict__ corr, double* __restrict__ Df, double* __restrict__ Dj, int S, int F, int n) {
    const int nrr = S * (S + 1) / 2;
    const int N = S * F;
    const double sc = 1.0 / n;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < S * N; idx += gridDim.x * blockDim.x) {
        const int e = idx / N, r = idx % N, i = r / F, k = r % F;
        const double v = corr[(long)(nrr + i * S + e) * n + (n - k) % n] * sc;
        Df[(long)e * N + r] = v;
        Dj[((long)i * S + e) * F + k] = v;
    }
}

// filters, zero padded for the spectral product: rows [0, S*S): (full, e, i);  rows [S*S, 2*S*S): (single, j, e)
__global__ void cpad_kernel(const double* __restrict__ Cf, const double* __restrict__ Cj, double* __restrict__ cpad, int S, int F, int n) {
    const int r = blockIdx.y;
    const double* src;
    if (r < S * S) { const int e = r / S, i = r % S; src = Cf + (long)e * S * F + (long)i * F; }
    else { const int q = r - S * S; src = Cj + (long)q * F; }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) cpad[(long)r * n + t] = t < F ? src[t] : 0.0;
}

__device__ __forceinline__ hipfftDoubleComplex cmul(hipfftDoubleComplex u, hipfftDoubleComplex v) {
    return make_hipDoubleComplex(u.x * v.x - u.y * v.y, u.x * v.y + u.y * v.x);
}
// rows [0,S): projection of estimate e on ALL references;  rows [S, S+S*S): (single reference j, estimate e)
__global__ void prod_kernel(const hipfftDoubleComplex* __restrict__ spec, const hipfftDoubleComplex* __restrict__ cspec,
                            hipfftDoubleComplex* __restrict__ pspec, int S, int nc) {
    const int r = blockIdx.y;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nc; k += gridDim.x * blockDim.x) {
        hipfftDoubleComplex acc = make_hipDoubleComplex(0.0, 0.0);
        if (r < S) {
            for (int i = 0; i < S; ++i) {                                  // same order as :723-729 (i ascending)
                const hipfftDoubleComplex t = cmul(cspec[(long)(r * S + i) * nc + k], spec[(long)i * nc + k]);
                acc.x += t.x; acc.y += t.y;
            }
        } else {
            const int q = r - S, j = q / S;
            acc = cmul(cspec[(long)(S * S + q) * nc + k], spec[(long)j * nc + k]);
        }
        pspec[(long)r * nc + k] = acc;
    }
}

__device__ __forceinline__ double block_sum(double v, double* sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    __syncthreads();
    return t;
}

// the four components exactly as :641-663, five sums of squares per (estimate e, reference j) pair; stage 1 of 2
__global__ __launch_bounds__(256) void sums_kernel(const double* __restrict__ tpad, const double* __restrict__ proj,
                                                   double* __restrict__ part, int S, int n, int Lp) {
    __shared__ double sm[4];
    const int pair = blockIdx.y, e = pair / S, j = pair % S;
    const double sc = 1.0 / n;
    const double* ref = tpad + (long)j * n;
    const double* est = tpad + (long)(S + e) * n;
    const double* pf = proj + (long)e * n;
    const double* pj = proj + (long)(S + j * S + e) * n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Lp; t += gridDim.x * blockDim.x) {
        const double s_true = ref[t];
        const double e_spat = pj[t] * sc - s_true;
        const double e_interf = pf[t] * sc - s_true - e_spat;
        const double e_artif = -s_true - e_spat - e_interf + est[t];
        const double s_filt = s_true + e_spat;
        a0 += s_filt * s_filt;
        a1 += (e_interf + e_artif) * (e_interf + e_artif);
        a2 += e_interf * e_interf;
        a3 += (s_filt + e_interf) * (s_filt + e_interf);
        a4 += e_artif * e_artif;
    }
    double* out = part + ((long)pair * RB + blockIdx.x) * 5;
    double v;
    v = block_sum(a0, sm); if (threadIdx.x == 0) out[0] = v;
    v = block_sum(a1, sm); if (threadIdx.x == 0) out[1] = v;
    v = block_sum(a2, sm); if (threadIdx.x == 0) out[2] = v;
    v = block_sum(a3, sm); if (threadIdx.x == 0) out[3] = v;
    v = block_sum(a4, sm); if (threadIdx.x == 0) out[4] = v;
}

// stage 2 + criteria (:732-748).  crit[k][e][j]; NaN when a factorisation failed.
__global__ void crit_kernel(const double* __restrict__ part, const int* __restrict__ info, int ninfo, double* __restrict__ crit,
                            int* __restrict__ info_out, int S) {
    const int pair = blockIdx.x;
    if (threadIdx.x != 0) return;
    int bad = 0;
    for (int i = 0; i < ninfo; ++i) bad |= (info[i] != 0);
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < RB; ++b)
        for (int k = 0; k < 5; ++k) s[k] += part[((long)pair * RB + b) * 5 + k];
    const double nanv = nan("");
    crit[0 * S * S + pair] = bad ? nanv : 10.0 * log10(s[0] / (s[1] + 1e-12));
    crit[1 * S * S + pair] = bad ? nanv : 10.0 * log10(s[0] / (s[2] + 1e-12));
    crit[2 * S * S + pair] = bad ? nanv : 10.0 * log10(s[3] / (s[4] + 1e-12));
    if (pair == 0) info_out[0] = bad;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int ams_bss_abi_version(void) { return 1; }

int ams_bss_create(ams_bss_ctx** out, int nsrc, int nsampl, int flen) {
    if (!out || nsrc < 1 || nsrc > 6 || nsampl < 1 || flen < 1) return -1;
    ams_bss_ctx* c = (ams_bss_ctx*)calloc(1, sizeof(ams_bss_ctx));
    if (!c) return -3;
    c->S = nsrc; c->L = nsampl; c->F = flen;
    c->Lp = nsampl + flen - 1;
    int n = 1;
    while (n < c->Lp) n <<= 1;                                   // :689
    c->n = n; c->nc = n / 2 + 1;
    const int S = nsrc, F = flen;
    c->NP = S * (S + 1) / 2 + S * S;
    bool ok = hipfftPlan1d(&c->fwd_in, n, HIPFFT_D2Z, 2 * S) == HIPFFT_SUCCESS;
    ok = ok && hipfftPlan1d(&c->inv_corr, n, HIPFFT_Z2D, c->NP) == HIPFFT_SUCCESS;
    ok = ok && hipfftPlan1d(&c->fwd_c, n, HIPFFT_D2Z, 2 * S * S) == HIPFFT_SUCCESS;
    ok = ok && hipfftPlan1d(&c->inv_proj, n, HIPFFT_Z2D, S + S * S) == HIPFFT_SUCCESS;
    ok = ok && hipsolverCreate(&c->solver) == HIPSOLVER_STATUS_SUCCESS;
    if (!ok) { free(c); return -3; }
    int lw1 = 0, lw2 = 0, lw3 = 0, lw4 = 0;
    hipsolverDpotrf_bufferSize(c->solver, HIPSOLVER_FILL_MODE_LOWER, S * F, nullptr, S * F, &lw1);
    hipsolverDpotrf_bufferSize(c->solver, HIPSOLVER_FILL_MODE_LOWER, F, nullptr, F, &lw2);
    hipsolverDpotrs_bufferSize(c->solver, HIPSOLVER_FILL_MODE_LOWER, S * F, S, nullptr, S * F, nullptr, S * F, &lw3);
    hipsolverDpotrs_bufferSize(c->solver, HIPSOLVER_FILL_MODE_LOWER, F, S, nullptr, F, nullptr, F, &lw4);
    c->lwork = lw1;
    if (lw2 > c->lwork) c->lwork = lw2;
    if (lw3 > c->lwork) c->lwork = lw3;
    if (lw4 > c->lwork) c->lwork = lw4;
    size_t o = 0;
    const size_t d = sizeof(double), z = 2 * sizeof(double);
    c->o_tpad = o;  o = align256(o + (size_t)2 * S * n * d);
    c->o_spec = o;  o = align256(o + (size_t)2 * S * c->nc * z);
    c->o_xs = o;    o = align256(o + (size_t)c->NP * c->nc * z);
    c->o_corr = o;  o = align256(o + (size_t)c->NP * n * d);
    c->o_G = o;     o = align256(o + (size_t)S * F * S * F * d);
    c->o_Gj = o;    o = align256(o + (size_t)S * F * F * d);
    c->o_Df = o;    o = align256(o + (size_t)S * S * F * d);
    c->o_Dj = o;    o = align256(o + (size_t)S * S * F * d);
    c->o_cpad = o;  o = align256(o + (size_t)2 * S * S * n * d);
    c->o_cspec = o; o = align256(o + (size_t)2 * S * S * c->nc * z);
    c->o_pspec = o; o = align256(o + (size_t)(S + S * S) * c->nc * z);
    c->o_proj = o;  o = align256(o + (size_t)(S + S * S) * n * d);
    c->o_part = o;  o = align256(o + (size_t)S * S * RB * 5 * d);
    c->o_work = o;  o = align256(o + (size_t)(c->lwork > 0 ? c->lwork : 1) * d);
    c->o_info = o;  o = align256(o + (size_t)(2 * (1 + S)) * sizeof(int));
    c->ws_bytes = o;
    *out = c;
    return 0;
}

void ams_bss_destroy(ams_bss_ctx* c) {
    if (!c) return;
    hipfftDestroy(c->fwd_in); hipfftDestroy(c->inv_corr); hipfftDestroy(c->fwd_c); hipfftDestroy(c->inv_proj);
    hipsolverDestroy(c->solver);
    free(c);
}

size_t ams_bss_workspace_bytes(const ams_bss_ctx* c) { return c ? c->ws_bytes : 0; }

int ams_bss_eval_pairs(ams_bss_ctx* c, const double* ref, const double* est, double* crit, int* info, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!c || !ref || !est || !crit || !info || !ws) return -1;
    if (ws_bytes < c->ws_bytes) return -2;
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    const int S = c->S, F = c->F, n = c->n, nc = c->nc, N = S * F;
    double* tpad = (double*)(w + c->o_tpad);
    hipfftDoubleComplex* spec = (hipfftDoubleComplex*)(w + c->o_spec);
    hipfftDoubleComplex* xs = (hipfftDoubleComplex*)(w + c->o_xs);
    double* corr = (double*)(w + c->o_corr);
    double* G = (double*)(w + c->o_G);
    double* Gj = (double*)(w + c->o_Gj);
    double* Df = (double*)(w + c->o_Df);
    double* Dj = (double*)(w + c->o_Dj);
    double* cpad = (double*)(w + c->o_cpad);
    hipfftDoubleComplex* cspec = (hipfftDoubleComplex*)(w + c->o_cspec);
    hipfftDoubleComplex* pspec = (hipfftDoubleComplex*)(w + c->o_pspec);
    double* proj = (double*)(w + c->o_proj);
    double* part = (double*)(w + c->o_part);
    double* work = (double*)(w + c->o_work);
    int* dinfo = (int*)(w + c->o_info);

    hipfftSetStream(c->fwd_in, st); hipfftSetStream(c->inv_corr, st); hipfftSetStream(c->fwd_c, st); hipfftSetStream(c->inv_proj, st);
    hipsolverSetStream(c->solver, st);

    hipLaunchKernelGGL(pad_kernel, dim3(64, 2 * S), dim3(256), 0, st, ref, est, tpad, S, c->L, n);
    if (hipfftExecD2Z(c->fwd_in, tpad, spec) != HIPFFT_SUCCESS) return -3;
    hipLaunchKernelGGL(cross_kernel, dim3(32, c->NP), dim3(256), 0, st, spec, xs, S, nc);
    if (hipfftExecZ2D(c->inv_corr, xs, corr) != HIPFFT_SUCCESS) return -3;
    hipLaunchKernelGGL(gram_kernel, dim3(1024), dim3(256), 0, st, corr, G, Gj, S, F, n);
    hipLaunchKernelGGL(rhs_kernel, dim3(16), dim3(256), 0, st, corr, Df, Dj, S, F, n);
    // factorise + solve: full system with the nsrc estimates as right-hand sides, then each single-reference system
    int k = 0;
    if (hipsolverDpotrf(c->solver, HIPSOLVER_FILL_MODE_LOWER, N, G, N, work, c->lwork, dinfo + k++) != HIPSOLVER_STATUS_SUCCESS) return -3;
    if (hipsolverDpotrs(c->solver, HIPSOLVER_FILL_MODE_LOWER, N, S, G, N, Df, N, work, c->lwork, dinfo + k++) != HIPSOLVER_STATUS_SUCCESS) return -3;
    for (int j = 0; j < S; ++j) {
        double* A = Gj + (long)j * F * F;
        double* B = Dj + (long)j * S * F;
        if (hipsolverDpotrf(c->solver, HIPSOLVER_FILL_MODE_LOWER, F, A, F, work, c->lwork, dinfo + k++) != HIPSOLVER_STATUS_SUCCESS) return -3;
        if (hipsolverDpotrs(c->solver, HIPSOLVER_FILL_MODE_LOWER, F, S, A, F, B, F, work, c->lwork, dinfo + k++) != HIPSOLVER_STATUS_SUCCESS) return -3;
    }
    hipLaunchKernelGGL(cpad_kernel, dim3(64, 2 * S * S), dim3(256), 0, st, Df, Dj, cpad, S, F, n);
    if (hipfftExecD2Z(c->fwd_c, cpad, cspec) != HIPFFT_SUCCESS) return -3;
    hipLaunchKernelGGL(prod_kernel, dim3(32, S + S * S), dim3(256), 0, st, spec, cspec, pspec, S, nc);
    if (hipfftExecZ2D(c->inv_proj, pspec, proj) != HIPFFT_SUCCESS) return -3;
    hipLaunchKernelGGL(sums_kernel, dim3(RB, S * S), dim3(256), 0, st, tpad, proj, part, S, n, c->Lp);
    hipLaunchKernelGGL(crit_kernel, dim3(S * S), dim3(64), 0, st, part, dinfo, k, crit, info, S);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
