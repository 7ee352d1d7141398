// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 157 TF peak).
//
// One templated kernel, C[M,N] (+)= op(A)[M,K] . op(B)[K,N] (+ bias[N]), with A-operand loaders:
//   A_ROW    A[m,k] = A[m*lda + k]                      (activations x weights: in-proj, dense fwd)
//   A_COL    A[m,k] = A[k*lda + m]                      (X^T . dY weight gradients), optional row mask
//   A_FRAMES A[m,k] = x[b*L + t*hop + k - pl], m = b*T+t (adaptive analysis filterbank = strided conv;
//                                                         reference models/adapt.py:122)
//   A_FRAMES_T  A[m=k_w, k=(b,t)] = frame element        (filter gradient of the same conv)
// and B-operand loaders B_ROW (B[k*ldb+n]) / B_COL (B[n*ldb+k]).
//
// Tiling: 128x128x8 block tile (BK = 8 measured best: 16.8 KB LDS and 95 VGPRs give 5 waves/SIMD, +4..10 % over BK = 16), 256 threads = 4 waves in 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs).  LDS holds k-major operand panels As[16][128+pad], Bs[16][128+pad] so the
// MFMA operand fetch (lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) is a conflict-free ds_read_b32.
// Register-staged double buffering: tile t+1 is fetched into VGPRs before the MFMAs of tile t and
// written to the other LDS buffer after them (one barrier per k-tile).
// Split-K (gridDim.z) writes fp32 partial slabs to a caller workspace; a second kernel reduces them in
// fixed order (deterministic) and applies bias / accumulate.
#include "common.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

// per-launch options of the product entry points (arguments of the C ABI, no state survives a call):
//   amax_a / amax_b: device pointers to upper bounds of max|A|, max|B| (both set -> fp16x3 arithmetic)
//   lds_pad: residency cap of a launch that is meant to run BESIDE a latency-critical kernel (unused dynamic LDS limits how many
//            of these workgroups a CU admits); 0 = uncapped, critical-path launch
//   sk_scratch / sk_bytes: stream-K scratch of the caller (ams_gemm_sk_scratch_bytes(); first AMS_SK_FLAG_BYTES zero on entry, left zero;
//            launches that share it must be ordered on one stream); NULL = no stream-K
//   amax_out: the launch leaves max |C| there (bit pattern; the caller zeroes it)
struct LaunchOpt { const float* amax_a = nullptr; const float* amax_b = nullptr; int lds_pad = 0; void* sk_scratch = nullptr; size_t sk_bytes = 0;
                   unsigned* amax_out = nullptr; int force_cfg = -1; };
constexpr size_t AMS_SK_FLAG_BYTES = 8192;       // ints [0, 512) stream-K flags, [512, 1024) per-workgroup output maxima, [1024] their ticket
constexpr size_t AMS_SK_SLOT_BYTES = (128 * 256 + 256) * sizeof(float);     // the largest tile (128 x 256) + its column sums
std::atomic<int> g_gemm_arith{-1};        // -1: not chosen yet (AMS_GEMM_X6, default 1), 0: native f32 MFMA, 1: bf16x6

// Tuning overrides (A/B runs only): read from the environment ONCE per process, never on the launch path.
struct GemmTuning { int group_m, splits, x6cfg, x6rule, x6persist, sk; bool noprio, novec, f16x3; double x6waste; bool c_vec; };
inline const GemmTuning& tuning() {
    static const GemmTuning t = [] {
        GemmTuning v{0, 0, -1, 2, 1, 1, false, false, true, 1.30, true};
        if (const char* f = getenv("AMS_GEMM_CVEC")) v.c_vec = atoi(f) != 0;    // 0: the x6 epilogue stores columns as the MFMA leaves them (dword stores)
        if (const char* f = getenv("AMS_GEMM_SK")) v.sk = atoi(f);        // stream-K: 0 off, 1 the tail of multi-round launches where the cost model prefers it (default), 2 wherever it applies
        if (const char* f = getenv("AMS_GEMM_X6WASTE")) v.x6waste = atof(f);
        if (const char* f = getenv("AMS_GEMM_GROUP_M")) v.group_m = atoi(f);
        if (const char* f = getenv("AMS_GEMM_SPLITS")) v.splits = atoi(f);
        v.noprio = getenv("AMS_GEMM_NOPRIO") != nullptr;
        { const char* e = getenv("AMS_X6_PERSIST"); v.x6persist = e ? atoi(e) : 1; }
        { const char* e = getenv("AMS_GEMM_F16X3"); v.f16x3 = !(e && atoi(e) == 0); }
        v.novec = getenv("AMS_GEMM_NOVEC") != nullptr;
        if (const char* f = getenv("AMS_GEMM_X6CFG")) v.x6cfg = atoi(f);       // force one bf16x6 tile configuration (0, 1 or 3: X6Cfg)
        if (const char* f = getenv("AMS_GEMM_X6RULE")) v.x6rule = atoi(f);
        return v;
    }();
    return t;
}

#ifndef AMS_GEMM_BK
#define AMS_GEMM_BK 8
#endif
#ifndef AMS_GEMM_XCD_FLAT
#define AMS_GEMM_XCD_FLAT 1
#endif
// Arithmetic of the 16-byte-fetch products: 1 = bf16x6 (exact 3-way bf16 split of both f32 operands, six bf16 MFMA products, f32
// accumulation: f32-level error at 2.7x the f32 MFMA ceiling), 0 = native v_mfma_f32_32x32x2_f32.  Process-wide; the default comes
// from AMS_GEMM_X6 (read once), ams_gemm_set_arith() changes it (tests and A/B runs hold both kernels against float64).
inline bool use_x6() {
    int v = g_gemm_arith.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* f = getenv("AMS_GEMM_X6");
        v = (f && atoi(f) == 0) ? 0 : 1;
        g_gemm_arith.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
constexpr int BM = 128, BN = 128, BK = AMS_GEMM_BK;
constexpr int X6_BK = 32;           // k-tile of the bf16x6 kernel
// split-K cost model: microseconds per 16 k and workgroup of the bf16x6 tile configurations (measured: 128 x 128 0.84 at 4096^3)
#ifndef AMS_GEMM_X6_US16
#define AMS_GEMM_X6_US16 0.84
#endif
#ifndef AMS_X6_OCC1
#define AMS_X6_OCC1 1
#endif
#ifndef AMS_GEMM_X6_US16_2
#define AMS_GEMM_X6_US16_2 1.35
#endif
#ifndef AMS_GEMM_X6_US16_1
#define AMS_GEMM_X6_US16_1 2.35
#endif
constexpr int PAD_T = 2;   // k-contiguous source, transposed scalar LDS writes: stride 130 -> conflict-free
constexpr int PAD_V = 4;   // m/n-contiguous source, float4 LDS writes: stride 132 keeps 16B alignment

enum { A_ROW = 0, A_COL = 1, A_FRAMES = 2, A_FRAMES_T = 3 };
enum { B_ROW = 0, B_COL = 1 };
enum { EPI_STORE = 0, EPI_MAXPOOL = 1 };

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long lda, ldb, ldc;
    int accumulate;            // C += result
    int nbatch;                // batch count (flat grid: batch x split x tile)
    // frames loader
    int fr_L, fr_T, fr_hop, fr_pl, fr_W;
    int group_m;                  // tile rows per band of the XCD-aware tile order
    // A_COL row mask: element (m, k) is zero when (k % mask_period) == mask_skip  (period 0 = off)
    int mask_period, mask_skip;
    // split-K
    int splits, k_per_split;
    float* partial;            // [splits, M, N] when splits > 1
    int a_vec, b_vec;          // 16-byte vector loads legal
    int32_t* pidx;             // EPI_MAXPOOL: per (row-tile, column) arg-max row
    // batch (grid.z): element offsets added per batch index to A, B, C (any sign); partial slabs are z-major
    long a_zs, b_zs, c_zs;
    long bias_zs;              // batch index z shifts bias by z * bias_zs
    int hiprio;                // 1: launch is on the critical path (not residency-capped) -> waves raise their issue priority
    // column sums of B (B_ROW, VEC path): the tile_m == 0 workgroups add up the B rows they stage anyway and write one partial
    // row per split to bsum_part [splits, N]; a finishing kernel adds the splits into bsum_out (bias gradient = colsum(dY))
    float* bsum_part;
    // bsum_out / bsum_accumulate: where the column sums of B go directly when the launch has ONE k-split (no finishing kernel)
    float* bsum_out; int bsum_accumulate;
    // fp16x3 arithmetic: device pointers to an upper bound of max|A|, max|B| over the WHOLE operand tensors (all batches); the kernel
    // derives the power-of-two operand scales from them
    const float* amax_a; const float* amax_b;
    // stream-K (x6 kernels, splits == 1): sk_rounds >= 0 -> every workgroup runs sk_rounds whole tiles, then its share of the k-tiles of
    // the tiles that are left (x6_body); sk_flags [gridDim.x] ints (zero on entry, left zero), sk_slots gridDim.x partial-tile slots of
    // sk_slot_floats floats each
    int sk_rounds; int* sk_flags; float* sk_slots; int sk_slot_floats;
    int c_vec;                 // C (and the partial slabs, bias) are 16-byte addressable along n: the x6 epilogue stores rows as float4 through LDS
    int amax_fold;             // amax_out is folded inside the launch through sk_flags[512 ..] (else: atomicMax on a word the entry point cleared)
    unsigned* amax_out;        // != NULL: atomicMax of the bit patterns of |C| as stored (one atomic per workgroup and tile)
};

template <int AMODE>
__device__ __forceinline__ float loadA1(const GemmArgs& g, int m, int k) {
    if (m >= g.M || k >= g.K) return 0.f;
    if (AMODE == A_ROW) return g.A[(long)m * g.lda + k];
    if (AMODE == A_COL) {
        if (g.mask_period && (k % g.mask_period) == g.mask_skip) return 0.f;
        return g.A[(long)k * g.lda + m];
    }
    if (AMODE == A_FRAMES) {
        int b = m / g.fr_T, t = m - b * g.fr_T;
        int p = t * g.fr_hop + k - g.fr_pl;
        return (p >= 0 && p < g.fr_L) ? g.A[(long)b * g.fr_L + p] : 0.f;
    }
    // A_FRAMES_T: m = filter tap, k = frame index (b,t)
    int b = k / g.fr_T, t = k - b * g.fr_T;
    int p = t * g.fr_hop + m - g.fr_pl;
    return (p >= 0 && p < g.fr_L) ? g.A[(long)b * g.fr_L + p] : 0.f;
}

template <int BMODE>
__device__ __forceinline__ float loadB1(const GemmArgs& g, int k, int n) {
    if (n >= g.N || k >= g.K) return 0.f;
    if (BMODE == B_ROW) return g.B[(long)k * g.ldb + n];
    return g.B[(long)n * g.ldb + k];
}

// Work item of this workgroup: (batch z, k-split, output tile).  Shifts the operand pointers of a batched launch.
__device__ __forceinline__ void locate_tile(GemmArgs& g, int& split, int& tile_m, int& tile_n, int bm = BM, int bn = BN, int vbid = -1) {
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = (g.N + bn - 1) / bn;
    const int ntiles = tiles_m * tiles_n;
    int bid, zb;
    {
        const int nz = g.nbatch > 1 ? g.nbatch : 1;
        const int items = ntiles * g.splits * nz;               // == gridDim.x
        int item = vbid >= 0 ? vbid : (int)blockIdx.x;
#if AMS_GEMM_XCD_FLAT
        const int q = items / 8, r = items % 8, xcd = item % 8, idx = item / 8;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        zb = item / (ntiles * g.splits);
        item -= zb * (ntiles * g.splits);
        split = item / ntiles;
        bid = item - split * ntiles;
#else
        zb = item / (ntiles * g.splits);                       // round-1 order: per-(z, split) plane, tiles dealt by x % 8
        item -= zb * (ntiles * g.splits);
        split = item / ntiles;
        bid = item - split * ntiles;
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
#endif
    }
    if (g.nbatch > 1) {                             // batched launch: same shape, shifted operands
        const long z = zb;
        g.A += z * g.a_zs; g.B += z * g.b_zs; g.C += z * g.c_zs;
        if (g.bias) g.bias += z * g.bias_zs;
        if (g.partial) g.partial += z * (long)g.splits * g.M * g.N;
    }
    const int GROUP_M = g.group_m > 0 ? g.group_m : 1;        // chosen per launch (choose_group_m)
    const int band = bid / (GROUP_M * tiles_n), within = bid - band * (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    tile_n = within / band_rows;
    tile_m = band * GROUP_M + (within - tile_n * band_rows);
}

// Position `pos` of the flat (batch, split, tile) order -> tile; the second half of locate_tile for callers that enumerate positions
// themselves (stream-K).  XCD x owns the positions [xcd_start(x), xcd_start(x) + xcd_len(x)).
__device__ __forceinline__ int xcd_len(int items, int x) { return items / 8 + (x < items % 8 ? 1 : 0); }
__device__ __forceinline__ int xcd_start(int items, int x) { const int q = items / 8, r = items % 8; return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; }
__device__ __forceinline__ void locate_pos(GemmArgs& g, int pos, int& split, int& tile_m, int& tile_n, int bm, int bn) {
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = (g.N + bn - 1) / bn;
    const int ntiles = tiles_m * tiles_n;
    const int zb = pos / (ntiles * g.splits);
    pos -= zb * (ntiles * g.splits);
    split = pos / ntiles;
    const int bid = pos - split * ntiles;
    if (g.nbatch > 1) {
        const long z = zb;
        g.A += z * g.a_zs; g.B += z * g.b_zs; g.C += z * g.c_zs;
        if (g.bias) g.bias += z * g.bias_zs;
        if (g.partial) g.partial += z * (long)g.splits * g.M * g.N;
    }
    const int GROUP_M = g.group_m > 0 ? g.group_m : 1;
    const int band = bid / (GROUP_M * tiles_n), within = bid - band * (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    tile_n = within / band_rows;
    tile_m = band * GROUP_M + (within - tile_n * band_rows);
}

// Epilogue shared by the f32 and the bf16x6 kernel (the C/D layout of the 32x32 MFMAs does not depend on the input type):
// col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16 (&acc)[2][2], int split, int m0, int n0, int wm, int wn,
                                           int l31, int lk) {
    float* out = g.splits > 1 ? g.partial + (long)split * g.M * g.N : g.C;
    const long ldo = g.splits > 1 ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (g.splits == 1 && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    float* p = out + (long)row * ldo + col;
                    if (g.splits == 1 && g.accumulate) v += *p;
                    *p = v;
                }
            }
        }
}

// Fused max-pool partial, shared by the f32 and the bf16x6 kernel (2 x WNC waves of 64 x 64: WNC = 2, or 4 for the 128 x 256 tile):
template <int WNC = 2>
__device__ __forceinline__ void maxpool_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2], float* smem, int tile_m, int m0, int n0,
                                                 int wm, int wn, int l31, int lk) {
        // Fused max-pool partial (reference models/adapt.py:115-117: stride-1 conv + max_pool_with_argmax): the [Bt,L,N]
        // conv output is never written; each 128-row tile emits, per column, its maximum and the row that holds it
        // (first maximum wins ties).  A second small kernel combines tiles into pooling windows.
        float* sred = smem;                                     // [WNC wn][2 j][32] values then rows
        int* srow = reinterpret_cast<int*>(smem + WNC * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float best = -3.4e38f;
            int brow = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const float v = acc[i][j][r];
                    if (row < g.M && (v > best || (v == best && row < brow))) { best = v; brow = row; }
                }
            const float ob = __shfl_xor(best, 32, 64);
            const int orow = __shfl_xor(brow, 32, 64);
            if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
            if (wm == 1 && lk == 0) { sred[(wn * 2 + j) * 32 + l31] = best; srow[(wn * 2 + j) * 32 + l31] = brow; }
            __syncthreads();
            if (wm == 0 && lk == 0) {
                const float ob2 = sred[(wn * 2 + j) * 32 + l31];
                const int or2 = srow[(wn * 2 + j) * 32 + l31];
                if (ob2 > best || (ob2 == best && or2 < brow)) { best = ob2; brow = or2; }
                const int col = n0 + wn * 64 + j * 32 + l31;
                if (col < g.N) { g.C[(long)tile_m * g.N + col] = best; g.pidx[(long)tile_m * g.N + col] = brow; }
            }
            __syncthreads();
        }
}

// Is operand A contiguous along k (-> transposed LDS writes) ?
template <int AMODE> struct AKContig { static constexpr bool v = (AMODE == A_ROW || AMODE == A_FRAMES); };


#ifndef AMS_GEMM_WPE
#define AMS_GEMM_WPE 2
#endif
// BKT = k-depth of one LDS tile.  8 is best at ~5 workgroups per CU (+4..10 % over 16).  A deeper tile for the residency-capped
// side-stream launches was measured too (BKT = 16 at 1-2 workgroups per CU): the capped products themselves got 0-19 % faster
// but the recurrent step kernels sharing the CU slowed from 9 to 21 us, 9.14 k vs 9.60 k mixtures/s overall -- not used.
// VEC: every operand fetch is ONE unconditional 16-byte load on a clamped address; validity (k beyond the split, masked rows)
// is a predicate applied when the registers are written to LDS.  The guarded form below merges a vector path and a scalar
// path per operand; each merge is a USE of the loaded value, so hipcc put `s_waitcnt vmcnt(0)` between the A fetch and the B
// fetch and the MFMAs started only after A had landed: two exposed memory round trips per k-tile, hidden only when ~5
// workgroups share a CU.  The host picks VEC whenever both operands are 16-byte addressable along their contiguous axis.
// PF (VEC only) = register stages of operand prefetch: with 2, a fetch issued behind the MFMAs of tile kt is consumed behind
// those of tile kt+2 (112-124 VGPRs -> 4 waves/SIMD).  Measured (-DAMS_GEMM_PF=2): products alone get faster (dense forward
// 567 -> 542 us = 116 TFLOP/s, 1 workgroup/CU 72 -> 92 TFLOP/s) but the whole step does not (9.91-9.95 k vs 9.95-9.96 k
// mixtures/s: the recurrent step kernels ran 8-10 % slower in the same replay), so the default stays 1.
#ifndef AMS_GEMM_PF
#define AMS_GEMM_PF 2      // round 2: beside the ring recurrence (csrc/lstm_ring.hip) the deeper prefetch pays in the step too: +5..6 % on every product alone, 13.47 -> 13.58 k mixtures/s
#endif
#ifndef AMS_GEMM_FRAG
#define AMS_GEMM_FRAG 0
#endif
template <int AMODE, int BMODE, int EPI = EPI_STORE, int BKT = AMS_GEMM_BK, bool VEC = false, int PF = 1>
__global__ __launch_bounds__(256, AMS_GEMM_WPE) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BK = BKT, NLD = BK / 8, KQ = BK / 4;
    if (g.hiprio) __builtin_amdgcn_s_setprio(2);    // above a capped side-stream product sharing the CU, below the LSTM step kernels (3)
    constexpr bool AK = AKContig<AMODE>::v;
    constexpr bool BKc = (BMODE == B_COL);
    constexpr int LDA_S = BM + (AK ? PAD_T : PAD_V);
    constexpr int LDB_S = BN + (BKc ? PAD_T : PAD_V);
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA_S + 2 * BK * LDB_S];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA_S;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: consecutive block ids land on different XCDs (b % 8); give each XCD a contiguous run of a
    // BANDED order (bands of group_m tile rows, walked column by column) so the ~32-64 tiles an XCD has in flight
    // form a near-square patch: they share group_m A row-panels and a few B column-panels in that XCD's private 4 MB L2.
    // Row-major runs re-fetched every B panel once per tile row: 971 MB of fabric reads for the 5120x10240x600 dense
    // product whose operands are 37 MB (rocprofv3 FETCH_SIZE, profiles/r01_c_hbm_traffic.txt).
    //
    // The grid is FLAT (1-D) over work items = (batch z, k-split, tile), batch outermost, tile innermost: the dispatcher deals
    // consecutive workgroup ids round-robin to the XCDs, so XCD x receives the x-th contiguous run of that order -- with 8
    // splits each XCD owns one k-slice of both operands (read once: ~1x algorithmic), with fewer splits a patch of one slice.
    // (A (tiles, splits, batch) grid broke the b % 8 assumption for every y, z > 0 whenever tiles % 8 != 0: the weight-gradient
    // products re-fetched their panels ~3x, profiles/r02_c_hbm_traffic.txt.)
    int split, tile_m, tile_n;
    locate_tile(g, split, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int k_begin = split * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const int nk = (k_end - k_begin + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NLD], rb[NLD], ra2[NLD], rb2[NLD];
    long arow[NLD];                                 // A_ROW: element offset of this thread's operand row(s), mapped once
#pragma unroll
    for (int h = 0; h < NLD; ++h) arow[h] = (AMODE == A_ROW) ? (long)min(m0 + (tid + h * 256) / KQ, g.M - 1) * g.lda : 0;
    int fp0[NLD];                                   // A_FRAMES (VEC): first sample of this thread's frame, relative to its row
    if (AMODE == A_FRAMES) {
#pragma unroll
        for (int h = 0; h < NLD; ++h) {
            const int m = min(m0 + (tid + h * 256) / KQ, g.M - 1);
            const int b = m / g.fr_T, t = m - b * g.fr_T;
            arow[h] = (long)b * g.fr_L;
            fp0[h] = t * g.fr_hop - g.fr_pl;
        }
    }

    bool va[NLD], vb[NLD], va2[NLD], vb2[NLD];      // VEC: validity of the staged registers
    auto fetch_v = [&](int kt, float4 (&ra)[NLD], float4 (&rb)[NLD], bool (&va)[NLD], bool (&vb)[NLD]) {
        const int k0 = k_begin + kt * BK;
        {
#pragma unroll
            for (int h = 0; h < NLD; ++h) {
                const int q = tid + h * 256;
                if (AMODE == A_FRAMES) {                // frame taps: float4 along k; hop, pad and L are multiples of 4, so a
                    const int k = k0 + (q % KQ) * 4;    // float4 is entirely inside the signal or entirely in the zero padding
                    const int p = fp0[h] + k;
                    va[h] = k < k_end && p >= 0 && p < g.fr_L;
                    ra[h] = *reinterpret_cast<const float4*>(g.A + arow[h] + min(max(p, 0), g.fr_L - 4));
                } else if (AK) {                        // A_ROW: float4 along k (K % 4 == 0, so k < k_end covers all four)
                    const int k = k0 + (q % KQ) * 4;
                    va[h] = k < k_end;
                    ra[h] = *reinterpret_cast<const float4*>(g.A + arow[h] + min(k, g.K - 4));
                } else if (AMODE == A_FRAMES_T) {        // filter gradient: m = tap (float4 along taps), k = frame (b, t)
                    const int k = k0 + (q >> 5), m = m0 + (q & 31) * 4;
                    const int kc = min(k, g.K - 1);
                    const int b = kc / g.fr_T, t = kc - b * g.fr_T;
                    const int p = t * g.fr_hop + m - g.fr_pl;          // multiple of 4: inside the signal or inside the padding
                    va[h] = k < k_end && m < g.M && p >= 0 && p < g.fr_L;
                    ra[h] = *reinterpret_cast<const float4*>(g.A + (long)b * g.fr_L + min(max(p, 0), g.fr_L - 4));
                } else {                                // A_COL: float4 along m (M % 4 == 0)
                    const int k = k0 + (q >> 5), m = m0 + (q & 31) * 4;
                    va[h] = k < k_end && !(g.mask_period && (k % g.mask_period) == g.mask_skip);
                    ra[h] = *reinterpret_cast<const float4*>(g.A + (long)min(k, g.K - 1) * g.lda + min(m, g.M - 4));
                }
                if (BKc) {                              // B_COL: float4 along k
                    const int n = n0 + (q / KQ), k = k0 + (q % KQ) * 4;
                    vb[h] = k < k_end;
                    rb[h] = *reinterpret_cast<const float4*>(g.B + (long)min(n, g.N - 1) * g.ldb + min(k, g.K - 4));
                } else {                                // B_ROW: float4 along n (N % 4 == 0)
                    const int k = k0 + (q >> 5), n = n0 + (q & 31) * 4;
                    vb[h] = k < k_end;
                    rb[h] = *reinterpret_cast<const float4*>(g.B + (long)min(k, g.K - 1) * g.ldb + min(n, g.N - 4));
                }
            }
        }
    };
    auto fetch = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if (VEC) { fetch_v(kt, ra, rb, va, vb); return; }
#pragma unroll
        for (int h = 0; h < NLD; ++h) {
            const int q = tid + h * 256;
            if (AK) {
                const int m = m0 + (q / KQ), k = k0 + (q % KQ) * 4;
                bool fast = g.a_vec && m < g.M && k + 3 < k_end;
                if (AMODE == A_FRAMES && fast) {
                    int b = m / g.fr_T, t = m - b * g.fr_T;
                    int p = t * g.fr_hop + k - g.fr_pl;
                    if (p >= 0 && p + 3 < g.fr_L) ra[h] = *reinterpret_cast<const float4*>(g.A + (long)b * g.fr_L + p);
                    else fast = false;
                } else if (fast) {
                    ra[h] = *reinterpret_cast<const float4*>(g.A + arow[h] + k);
                }
                if (!fast) {
                    ra[h].x = (k + 0 < k_end) ? loadA1<AMODE>(g, m, k + 0) : 0.f;
                    ra[h].y = (k + 1 < k_end) ? loadA1<AMODE>(g, m, k + 1) : 0.f;
                    ra[h].z = (k + 2 < k_end) ? loadA1<AMODE>(g, m, k + 2) : 0.f;
                    ra[h].w = (k + 3 < k_end) ? loadA1<AMODE>(g, m, k + 3) : 0.f;
                }
            } else {
                const int k = k0 + (q >> 5), m = m0 + (q & 31) * 4;
                bool fast = g.a_vec && k < k_end && m + 3 < g.M && AMODE == A_COL &&
                            !(g.mask_period && (k % g.mask_period) == g.mask_skip);
                if (fast) ra[h] = *reinterpret_cast<const float4*>(g.A + (long)k * g.lda + m);
                else {
                    const bool kv = k < k_end;
                    ra[h].x = kv ? loadA1<AMODE>(g, m + 0, k) : 0.f;
                    ra[h].y = kv ? loadA1<AMODE>(g, m + 1, k) : 0.f;
                    ra[h].z = kv ? loadA1<AMODE>(g, m + 2, k) : 0.f;
                    ra[h].w = kv ? loadA1<AMODE>(g, m + 3, k) : 0.f;
                }
            }
            if (BKc) {
                const int n = n0 + (q / KQ), k = k0 + (q % KQ) * 4;
                if (g.b_vec && n < g.N && k + 3 < k_end) rb[h] = *reinterpret_cast<const float4*>(g.B + (long)n * g.ldb + k);
                else {
                    rb[h].x = (k + 0 < k_end) ? loadB1<BMODE>(g, k + 0, n) : 0.f;
                    rb[h].y = (k + 1 < k_end) ? loadB1<BMODE>(g, k + 1, n) : 0.f;
                    rb[h].z = (k + 2 < k_end) ? loadB1<BMODE>(g, k + 2, n) : 0.f;
                    rb[h].w = (k + 3 < k_end) ? loadB1<BMODE>(g, k + 3, n) : 0.f;
                }
            } else {
                const int k = k0 + (q >> 5), n = n0 + (q & 31) * 4;
                if (g.b_vec && k < k_end && n + 3 < g.N) rb[h] = *reinterpret_cast<const float4*>(g.B + (long)k * g.ldb + n);
                else {
                    const bool kv = k < k_end;
                    rb[h].x = kv ? loadB1<BMODE>(g, k, n + 0) : 0.f;
                    rb[h].y = kv ? loadB1<BMODE>(g, k, n + 1) : 0.f;
                    rb[h].z = kv ? loadB1<BMODE>(g, k, n + 2) : 0.f;
                    rb[h].w = kv ? loadB1<BMODE>(g, k, n + 3) : 0.f;
                }
            }
        }
    };

    const bool do_bsum = VEC && !BKc && g.bsum_part != nullptr && tile_m == 0;      // workgroup-uniform
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto stash_s = [&](int buf, float4 (&ra)[NLD], float4 (&rb)[NLD], bool (&va)[NLD], bool (&vb)[NLD]) {
        float* as = As + buf * BK * LDA_S;
        float* bs = Bs + buf * BK * LDB_S;
#pragma unroll
        for (int h = 0; h < NLD; ++h) {
            if (VEC) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!va[h]) ra[h] = z;
                if (!vb[h]) rb[h] = z;
                if (!BKc && do_bsum) { bsum4.x += rb[h].x; bsum4.y += rb[h].y; bsum4.z += rb[h].z; bsum4.w += rb[h].w; }
            }
            const int q = tid + h * 256;
            if (AK) {
                const int mi = q / KQ, kq = (q % KQ) * 4;
                as[(kq + 0) * LDA_S + mi] = ra[h].x;
                as[(kq + 1) * LDA_S + mi] = ra[h].y;
                as[(kq + 2) * LDA_S + mi] = ra[h].z;
                as[(kq + 3) * LDA_S + mi] = ra[h].w;
            } else {
                const int ki = q >> 5, mi = (q & 31) * 4;
                *reinterpret_cast<float4*>(as + ki * LDA_S + mi) = ra[h];
            }
            if (BKc) {
                const int ni = q / KQ, kq = (q % KQ) * 4;
                bs[(kq + 0) * LDB_S + ni] = rb[h].x;
                bs[(kq + 1) * LDB_S + ni] = rb[h].y;
                bs[(kq + 2) * LDB_S + ni] = rb[h].z;
                bs[(kq + 3) * LDB_S + ni] = rb[h].w;
            } else {
                const int ki = q >> 5, ni = (q & 31) * 4;
                *reinterpret_cast<float4*>(bs + ki * LDB_S + ni) = rb[h];
            }
        }
    };
    auto stash = [&](int buf) { stash_s(buf, ra, rb, va, vb); };

    const int l31 = lane & 31, lk = lane >> 5;
    // AMS_GEMM_FRAG: 0 = fragments of one k-pair are read right before its four MFMAs (a wave alone stalls ~LDS latency per pair and
    // relies on its 4 SIMD neighbours to fill the pipe); 1 = all fragments of the LDS tile first, then BK/2 x 4 MFMAs back to back.
    auto mfma_tile = [&](int buf) {
        const float* as = As + buf * BK * LDA_S + wm * 64 + l31;
        const float* bs = Bs + buf * BK * LDB_S + wn * 64 + l31;
#if AMS_GEMM_FRAG
        float fa0[BK / 2], fa1[BK / 2], fb0[BK / 2], fb1[BK / 2];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            fa0[kk / 2] = as[(kk + lk) * LDA_S];
            fa1[kk / 2] = as[(kk + lk) * LDA_S + 32];
            fb0[kk / 2] = bs[(kk + lk) * LDB_S];
            fb1[kk / 2] = bs[(kk + lk) * LDB_S + 32];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[kk], fb0[kk], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[kk], fb1[kk], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[kk], fb0[kk], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[kk], fb1[kk], acc[1][1], 0, 0, 0);
        }
#else
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = as[(kk + lk) * LDA_S];
            const float a1 = as[(kk + lk) * LDA_S + 32];
            const float b0 = bs[(kk + lk) * LDB_S];
            const float b1 = bs[(kk + lk) * LDB_S + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#endif
    };

    if (nk > 0) {
        fetch(0);
        stash(0);
    }
    if (VEC && PF == 2) {
        // No conditionals around the MFMAs or the staging: tiles past the end of the split are fetched from clamped
        // addresses with their validity predicate false, i.e. staged as zeros (an odd tile count runs one zero tile).
        fetch_v(1, ra, rb, va, vb);
        fetch_v(2, ra2, rb2, va2, vb2);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            mfma_tile(0);
            stash_s(1, ra, rb, va, vb);
            fetch_v(kt + 3, ra, rb, va, vb);
            __syncthreads();
            mfma_tile(1);
            stash_s(0, ra2, rb2, va2, vb2);
            fetch_v(kt + 4, ra2, rb2, va2, vb2);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) fetch(kt + 1);
            mfma_tile(buf);
            if (kt + 1 < nk) stash(buf ^ 1);
            __syncthreads();
        }
    }

    if (VEC && !BKc && do_bsum) {
        // thread q staged row (q >> 5) of every k-tile, columns (q & 31) * 4 .. + 3: the 8 threads of a column group meet in LDS
        // (fixed order -> deterministic).  PF == 2 runs up to two zero tiles past the end: they add nothing.
        float4* sb = reinterpret_cast<float4*>(smem);
        __syncthreads();
        sb[tid] = bsum4;
        __syncthreads();
        if (tid < 32) {
            float4 t = sb[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) { const float4 v = sb[tid + 32 * j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            const int n = n0 + tid * 4;
            if (n < g.N) *reinterpret_cast<float4*>(g.bsum_part + (long)split * g.N + n) = t;      // N % 4 == 0 on the VEC path
        }
        __syncthreads();
    }
    if (EPI == EPI_MAXPOOL) {
        maxpool_epilogue(g, acc, smem, tile_m, m0, n0, wm, wn, l31, lk);
        return;
    }
    store_tile(g, acc, split, m0, n0, wm, wn, l31, lk);
}

// ---- f32 products on the bf16 matrix pipe: exact 3-way operand split, six bf16 MFMA products, f32 accumulation ----------------------
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s against 157 TFLOP/s).  Every f32 x
// is EXACTLY hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even; 3 x 8 significant
// bits cover the 24 of an f32, both subtractions are exact), and every bf16 x bf16 product is exact in f32.  Of the nine partial
// products of a.b the kernel accumulates six in f32 -- lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi, smallest first -- and drops
// mid.lo, lo.mid, lo.lo, which are <= 2^-26 |a.b| together: a quarter of the half-ulp an f32 product rounding would cost.  The
// result is an f32 product with f32-level error (tests/test_gpu_gemm_x6.py holds it against float64 next to the native f32 MFMA
// kernel), at 16/6 = 2.7x the f32 MFMA ceiling.  Inf/NaN inputs give NaN (inf - inf in the split).
//
// Block tile BMX x BNX x 32 (X6Cfg below), waves of TM x TN MFMA tiles; per k-tile a wave issues 2 k-steps x 6 products x TM x TN
// MFMAs.  The operand fetch is the f32 kernel's (one unconditional 16-byte load per operand quarter on a clamped address, validity
// applied at the LDS write); the split happens ONCE per element and workgroup, between the staging registers and LDS.  LDS image
// per operand and part: four planes (one per group of 8 k) of R rows x 16 bytes, so that an MFMA operand (lane l: row l & 31,
// k-group l >> 5) is ONE ds_read_b128.  Sources that are contiguous along k (A_ROW, A_FRAMES, B_COL) are written as 16-byte rows by
// threads holding 8 consecutive k of a row; sources contiguous along m/n (A_COL, A_FRAMES_T, B_ROW) by threads holding a 4 (k) x 4
// (m) block, 8 bytes per m, into rows permuted by x6_slot() so that neither those writes nor the 16-lane groups of the reads pile up
// on a bank.  One LDS buffer, register-staged prefetch of the next k-tile, two barriers per k-tile.
//
// What was measured on the way (profiles/r02_i_*; 4096^3, 128 x 128 tile): the MFMAs of a k-tile take 0.83 us per workgroup and
// everything else (fetch, ~4.5 VALU per element of split arithmetic, LDS traffic) 0.86-1.1 us, and the two ADD instead of
// overlapping -- (a) two co-resident workgroups run in lock-step (876 us = 340 + 438 with the MFMAs compiled out); (b) a
// wave-specialised form (4 MFMA waves + 4 split waves, two LDS buffers) was slower: a split wave progresses ~2 VALU instructions
// per MFMA of its SIMD partner (phase trace: 1.6 us per tile for ~220 instructions); (c) a fused stream (one wave = MFMA + one
// slice of the next tile's split per MFMA, hand-laid behind sched_barrier fences, 1 wave per SIMD) measured the same as the plain
// form (873 us; its MFMA-only stream 426 us, its split-only stream 567 us).  Both forms are in the history (commits 810b95b,
// 232c519).  The kernel draws the board's power limit (1385-1393 W at 2135 MHz; its MFMA-only stream 1050 W at 2400 MHz) and gains
// 0-4 % with the split arithmetic compiled out: at that limit its run time is the energy of a product, not its instruction schedule
// (DESIGN.md 4.0).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#ifndef AMS_SK_CAPPED
#define AMS_SK_CAPPED 0
#endif
#ifndef AMS_X6_DBG
#define AMS_X6_DBG 0        // timing anatomy only (WRONG results): 1 no split arithmetic, 2 no LDS writes, 4 no MFMAs, 8 no LDS reads, 16 no fetch in the loop
#endif

// Tile configurations: block BMX x BNX, WMC x WNC waves of (TM x 32) x (TN x 32).
template <int CFG> struct X6Cfg;
template <> struct X6Cfg<0> { static constexpr int BMX = 128, BNX = 128, WMC = 2, WNC = 2, TM = 2, TN = 2; };   // 4 waves, 48.75 KB
template <> struct X6Cfg<1> { static constexpr int BMX = 256, BNX = 256, WMC = 2, WNC = 4, TM = 4, TN = 2; };   // 8 waves, 96.75 KB
template <> struct X6Cfg<3> { static constexpr int BMX = 128, BNX = 256, WMC = 2, WNC = 4, TM = 2, TN = 2; };   // 8 waves, 72.75 KB
constexpr int x6_plane(int rows) { return rows * 16 + 32; }     // bytes; +32: the four planes start 8 banks apart (16-byte row writes of one wave hit all four)
constexpr int x6_oper(int rows) { return 3 * 4 * x6_plane(rows); }
constexpr int x6_bm(int cfg) { return cfg == 1 ? 256 : 128; }
constexpr int x6_bn(int cfg) { return (cfg == 1 || cfg == 3) ? 256 : 128; }
constexpr int x6_lds(int cfg) { return x6_oper(x6_bm(cfg)) + x6_oper(x6_bn(cfg)); }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32: a -> bits 0..15, b -> bits 16..31
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// fp16x3 (round 3): two fp16 planes per operand instead of three bf16 planes, three products instead of six.
//   a * s = h0 + h1 + e,  h0 = fp16(a s),  h1 = fp16(a s - h0),  |e| <= 2^-22 |a s|   (both conversions round to nearest; h0 * h0' is exact in f32)
//   a.b ~ [h0.h0' + (h0.h1' + h1.h0')] / (s s')   -- dropped: h1.h1' (2^-22) and e: the same 2^-22 level as the f32 accumulation itself
// s = 2^(13 - floor(log2(amax))): the largest operand entry lands in [2^13, 2^14) (fp16 overflows at 65504), entries down to amax * 2^-17
// keep all 22 bits, smaller ones an absolute error of amax * 2^-39.  bf16x6 needs no scale (bf16 has the f32 exponent) and stays the
// arithmetic of every launch that does not supply the bounds.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& mid) {
    hi = pk_f16(a, b);
    const f16x2_t h = __builtin_bit_cast(f16x2_t, hi);
    mid = pk_f16(a - (float)h[0], b - (float)h[1]);
}
// 2^(13 - floor(log2(amax))) for a finite positive amax; 1 for 0, denormals, Inf and NaN (which then propagate as they would in f32)
__device__ __forceinline__ float f16_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    const int se = 127 + 13 - (e - 127);            // biased exponent of the scale
    return (se >= 1 && se <= 254) ? __uint_as_float((unsigned)se << 23) : 1.0f;
}
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
#if AMS_X6_DBG & 1
    hi = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); mid = hi; lo = hi; return;
#endif
    hi = pk_bf16(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = pk_bf16(sa, sb);
}
// LDS row of operand row n (0..R-1) for m/n-contiguous sources: the four rows a thread's float4 covers go to four R/4-row blocks,
// rotated by 4 rows per block (read groups {0-3,12-15,20-27} / {4-11,16-19,28-31} of ds_read_b128 then touch 16 different slots).
template <int R>
__device__ __forceinline__ int x6_slot(int n) { return (n & 3) * (R / 4) + (((n >> 2) + 4 * (n & 3)) & (R / 4 - 1)); }
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

#ifndef AMS_X6_STAMP
#define AMS_X6_STAMP 0       // 1: timing-anatomy build (tools/gemm_anatomy.py): thread 0 of every workgroup stamps the phases of its first 8 items
#endif
#if AMS_X6_STAMP
__device__ long long g_x6_stamp[1024 * 8 * 8];
#define X6_STAMP(ph) do { if (threadIdx.x == 0 && wi < 8) g_x6_stamp[((int)blockIdx.x * 8 + wi) * 8 + (ph)] = wall_clock64(); } while (0)
#else
#define X6_STAMP(ph) do { } while (0)
#endif
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// Raw buffer accesses with aux bit 16 = sc1: stores write through to the memory side, loads bypass this CU's L1 -- the publish form of
// MI355X_MICROARCH.md ("publish-large": sc1 payload -> s_waitcnt vmcnt(0) -> agent-scope flag, sc1 loads on the reader)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x6_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rs, off, 0, 16);
}

template <int AMODE, int BMODE, int CFG, int EPI, bool SEP, bool F16>
// PERSISTENT over work items (round 3): the grid is at most one resident set of workgroups (ams_gemm launch: 256 CUs x the
// configuration's workgroups per CU) and a workgroup walks items blockIdx.x, + gridDim.x, ... .  The operands of the NEXT item's first
// k-tile are requested BEFORE the epilogue stores of the finished one, so the stores (210 MB for the dense forward product: ~10 us per
// round of 256 tiles with every workgroup storing at once, and nothing else resident on the CU to hide them) drain under the next
// item's main loop instead of in front of the next workgroup's launch.  gridDim.x is a multiple of 8 (or the whole item count), so
// an item stays on the XCD the flat order meant it for.
__device__ __forceinline__ void x6_body(const GemmArgs& g0, unsigned char* const smem) {
    GemmArgs g = g0;
    using C = X6Cfg<CFG>;
    constexpr int BK = X6_BK, BMX = C::BMX, BNX = C::BNX, TM = C::TM, TN = C::TN;
    constexpr int NT = C::WMC * C::WNC * 64;
    constexpr bool AK = AKContig<AMODE>::v;
    constexpr bool BKc = (BMODE == B_COL);
    constexpr int A_PLANE = x6_plane(BMX), B_PLANE = x6_plane(BNX), A_PART = 4 * A_PLANE, B_PART = 4 * B_PLANE;
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + x6_oper(BMX);
    if (g.hiprio) __builtin_amdgcn_s_setprio(2);

    // Lane constants are RE-DERIVED at the top of every work item from a thread id the compiler cannot see through (lane_consts):
    // a workgroup that walks several items would otherwise keep ~30 loop-invariant registers (fragment addresses, LDS slots, block
    // coordinates) alive across the epilogue and the stream-K hand-off of each item, which is where these kernels run out of registers.
    int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WNC, wn = wave % C::WNC;
    int l31 = lane & 31, lk = lane >> 5;

    // only the two-accumulator (uncapped, alone-on-the-CU) variants request the NEXT work item's first k-tile before the stores of the
    // finished one: the capped single-accumulator ones are sized to sit beside a recurrence ring and that costs them
    // 20-40 VGPRs (next item's staging registers live over the epilogue); they fetch it after their stores.
    constexpr bool PERSIST = SEP && EPI == EPI_STORE;
    const int n_items = (int)((long)((g0.M + BMX - 1) / BMX) * ((g0.N + BNX - 1) / BNX) * g0.splits * (g0.nbatch > 1 ? g0.nbatch : 1));
    int split, tile_m, tile_n, m0, n0, k_begin, k_end, nk;

    // ---- the work list of this workgroup.  Plain launches: items blockIdx.x, + gridDim.x, ... (one item when not PERSIST).
    // STREAM-K (g0.sk_rounds >= 0; splits == 1, gridDim.x a multiple of 8): sk_rounds whole tiles like that, and then the tiles that
    // do not fill another round are shared BY K-TILE: XCD x (workgroups x, x + 8, ...: Gx of them) owns n_x left-over positions of the
    // flat order (its run minus the sk_rounds * Gx whole tiles it has done), U = n_x * nk k-tile units, and its workgroup w takes
    // the units [w U / Gx, (w + 1) U / Gx) -- less than one tile's worth, so at most the tail of one tile and the head of the next.
    // A segment that ends where its tile ends makes its workgroup the tile's OWNER: it adds the partial tiles the earlier
    // segments' workgroups left in their slots (fixed order: deterministic) and runs the epilogue; every other segment is a
    // PARTIAL: accumulators to the workgroup's slot as write-through stores, then a flag.  A workgroup runs its partial segment
    // FIRST and never waits before it has published, and an owner only waits for workgroups with a lower index on its own XCD.
    enum { W_FULL = 0, W_PART = 1, W_OWNER = 2 };
    // The capped single-accumulator variants take ONE item and no stream-K share unless built with -DAMS_SK_CAPPED=1: the walk and the
    // hand-off cost them ~40 VGPRs, and above 168 they no longer share a CU with a recurrence ring.
    constexpr bool WALK = PERSIST || (AMS_SK_CAPPED && EPI == EPI_STORE);
    const bool sk = WALK && g0.sk_rounds >= 0;
    const int nk_full = (g0.K + BK - 1) / BK;
    const int G = (int)gridDim.x;
    const int sk_x = (int)blockIdx.x & 7, sk_w = (int)blockIdx.x >> 3, sk_Gx = G >> 3;
    int sk_U = 0, sk_base = 0;
    // segment A: tail of tile segA_t, k-tiles [segA_k0, segA_k1) (owner / whole tile, or a partial when the range stays inside one tile);
    // segment B: head of tile segA_t + 1, k-tiles [0, segB_k1): always a partial, run first
    int segA_t = 0, segA_k0 = 0, segA_k1 = 0, segA_role = W_FULL, segB_k1 = 0;
    int nseg = 0;
    int dp_count = 1;
    if (sk) {
        dp_count = g0.sk_rounds;
        sk_base = xcd_start(n_items, sk_x) + dp_count * sk_Gx;
        sk_U = (xcd_len(n_items, sk_x) - dp_count * sk_Gx) * nk_full;
        const int u0 = sk_w * sk_U / sk_Gx, u1 = (sk_w + 1) * sk_U / sk_Gx;
        if (u1 > u0) {
            const int t0 = u0 / nk_full, ka = u0 - t0 * nk_full;
            const int e1 = min(u1 - t0 * nk_full, nk_full), e2 = u1 - (t0 + 1) * nk_full;
            segA_t = t0; segA_k0 = ka; segA_k1 = e1;
            segA_role = e1 == nk_full ? (ka == 0 ? W_FULL : W_OWNER) : W_PART;
            segB_k1 = e2 > 0 ? e2 : 0;                  // (< nk_full: a range is shorter than a tile or aligned with one)
            nseg = e2 > 0 ? 2 : 1;
        }
    } else if (PERSIST) dp_count = (n_items - (int)blockIdx.x + G - 1) / G;
    const int n_work = dp_count + nseg;
    if (n_work <= 0 && !(EPI == EPI_STORE && g0.amax_out != nullptr && g0.amax_fold)) return;
    int role = W_FULL, own_tile = 0;
    float vmax_wg = 0.f;                            // max |C| over this workgroup's stores (g0.amax_out)

    // SEP: two accumulator sets (64 x 64 waves, launches that are not residency-capped -- with 64 more VGPRs a workgroup no longer
    // shares a CU with a recurrence ring): hi.hi goes to `acc`, the five small partial products to `accs`, added once in the epilogue.  The bf16 MFMA adds its 16 products to the accumulator with the bits below its internal
    // guard bits TRUNCATED (two's complement, i.e. toward -inf) at a level set by the largest addend: a bias of ~1e-3 ulp of the
    // accumulator per MFMA, -0.3 .. -1 ulp per output when all six products go into one accumulator (measured, K = 600 .. 5120;
    // the f32 MFMA rounds to nearest: 0.00).  Small products into their own accumulator are truncated 2^-8 lower: 6x less bias.
    static_assert(!SEP || TM * TN <= 4, "no registers for a second accumulator set on 128 x 64 waves");
    f32x16 acc[TM][TN], accs[SEP ? TM : 1][SEP ? TN : 1];

    // Operand of R rows, NT threads.  k-contiguous source: slot = (row, k-group of 8), R * 4 slots, thread -> rows (tid >> 2) + (NT / 4) h,
    // k-group tid & 3, two float4 per slot.  m/n-contiguous source: 4 (k) x 4 (m) blocks, R / 4 x 8 of them, thread -> block
    // (tid % (R / 4), tid / (R / 4)) while tid < R * 2, four float4.  Either way at most four float4 per thread and operand.
    constexpr int NSA = BMX * 4 / NT, NSB = BNX * 4 / NT;          // k-contiguous slots per thread (1 or 2)
    int kgrp = tid & 3, krow = tid >> 2;
    int mbA = tid % (BMX / 4), kbA = tid / (BMX / 4);
    int mbB = tid % (BNX / 4), kbB = tid / (BNX / 4);
    bool actA = AK || tid < BMX * 2, actB = BKc || tid < BNX * 2;     // wave-uniform (multiples of 64)
    long arow[2] = {0, 0};
    int fp0[2] = {0, 0};
    long brow[2] = {0, 0};
    // per-item state of work item i of this workgroup: (batch, split, tile), k range, role, operand row pointers
    auto setup = [&](int i) {
        g = g0;
        if (i < dp_count) {
            locate_tile(g, split, tile_m, tile_n, BMX, BNX, (int)blockIdx.x + i * G);
            role = W_FULL;
            k_begin = split * g.k_per_split;
            k_end = min(g.K, k_begin + g.k_per_split);
        } else {
            const bool headB = nseg == 2 && i == dp_count;              // the partial head of the next tile comes first
            own_tile = headB ? segA_t + 1 : segA_t;
            locate_pos(g, sk_base + own_tile, split, tile_m, tile_n, BMX, BNX);
            role = headB ? W_PART : segA_role;
            k_begin = (headB ? 0 : segA_k0) * BK;
            k_end = min(g.K, (headB ? segB_k1 : segA_k1) * BK);
        }
        m0 = tile_m * BMX; n0 = tile_n * BNX;
        nk = (k_end - k_begin + BK - 1) / BK;
        if (AMODE == A_ROW) {
#pragma unroll
            for (int h = 0; h < NSA; ++h) arow[h] = (long)min(m0 + krow + (NT / 4) * h, g.M - 1) * g.lda;
        }
        if (AMODE == A_FRAMES) {
#pragma unroll
            for (int h = 0; h < NSA; ++h) {
                const int m = min(m0 + krow + (NT / 4) * h, g.M - 1);
                const int b = m / g.fr_T, t = m - b * g.fr_T;
                arow[h] = (long)b * g.fr_L;
                fp0[h] = t * g.fr_hop - g.fr_pl;
            }
        }
        if (BKc) {
#pragma unroll
            for (int h = 0; h < NSB; ++h) brow[h] = (long)min(n0 + krow + (NT / 4) * h, g.N - 1) * g.ldb;
        }
    };
    int wi = 0;
    setup(0);

    float4 ra[4], rb[4];
    bool va[4], vb[4];
    auto fetch = [&](int kt) {
        const int k0 = k_begin + kt * BK;
        if (AK) {
#pragma unroll
            for (int h = 0; h < NSA; ++h)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int k = k0 + kgrp * 8 + 4 * c;
                    if (AMODE == A_FRAMES) {            // hop, pad and L are multiples of 4: a float4 is all signal or all padding
                        const int p = fp0[h] + k;
                        va[2 * h + c] = k < k_end && p >= 0 && p < g.fr_L;
                        ra[2 * h + c] = *reinterpret_cast<const float4*>(g.A + arow[h] + min(max(p, 0), g.fr_L - 4));
                    } else {                            // A_ROW (K % 4 == 0)
                        va[2 * h + c] = k < k_end;
                        ra[2 * h + c] = *reinterpret_cast<const float4*>(g.A + arow[h] + min(k, g.K - 4));
                    }
                }
        } else if (actA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 4 * kbA + r, m = m0 + 4 * mbA;
                const int kc = min(k, g.K - 1);
                if (AMODE == A_FRAMES_T) {              // filter gradient: m = tap, k = frame (b, t)
                    const int b = kc / g.fr_T, t = kc - b * g.fr_T;
                    const int p = t * g.fr_hop + m - g.fr_pl;
                    va[r] = k < k_end && m < g.M && p >= 0 && p < g.fr_L;
                    ra[r] = *reinterpret_cast<const float4*>(g.A + (long)b * g.fr_L + min(max(p, 0), g.fr_L - 4));
                } else {                                // A_COL (M % 4 == 0)
                    va[r] = k < k_end && !(g.mask_period && (k % g.mask_period) == g.mask_skip);
                    ra[r] = *reinterpret_cast<const float4*>(g.A + (long)kc * g.lda + min(m, g.M - 4));
                }
            }
        }
        if (BKc) {
#pragma unroll
            for (int h = 0; h < NSB; ++h)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int k = k0 + kgrp * 8 + 4 * c;
                    vb[2 * h + c] = k < k_end;
                    rb[2 * h + c] = *reinterpret_cast<const float4*>(g.B + brow[h] + min(k, g.K - 4));
                }
        } else if (actB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 4 * kbB + r, n = n0 + 4 * mbB;
                vb[r] = k < k_end;
                rb[r] = *reinterpret_cast<const float4*>(g.B + (long)min(k, g.K - 1) * g.ldb + min(n, g.N - 4));
            }
        }
    };

    bool do_bsum = !BKc && g.bsum_part != nullptr && tile_m == 0;            // workgroup-uniform, per item
    float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // split the staged f32 values and write the three bf16 images of one operand (R rows; plane / part strides PL / PT)
    // fp16x3: the power-of-two operand scales (workgroup-uniform) and the factor that undoes both on the accumulators
    float sc_a = 1.0f, sc_b = 1.0f, sc_inv = 1.0f;
    if constexpr (F16) {
        sc_a = f16_scale(g0.amax_a[0]);
        sc_b = f16_scale(g0.amax_b[0]);
        sc_inv = (1.0f / sc_a) * (1.0f / sc_b);     // exact: powers of two, |exponent| <= 140 each way -- may leave the f32 range only where the f32 result would
    }
    auto stash_k = [&](unsigned char* base, int PL, int PT, int NS, float4 (&rv)[4], bool (&vv)[4], float sc) {      // k-contiguous source
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h >= NS) break;
            if (!vv[2 * h]) rv[2 * h] = z;
            if (!vv[2 * h + 1]) rv[2 * h + 1] = z;
            uint4 hi, mid, lo;
            if constexpr (F16) {
                split2h(rv[2 * h].x * sc, rv[2 * h].y * sc, hi.x, mid.x);
                split2h(rv[2 * h].z * sc, rv[2 * h].w * sc, hi.y, mid.y);
                split2h(rv[2 * h + 1].x * sc, rv[2 * h + 1].y * sc, hi.z, mid.z);
                split2h(rv[2 * h + 1].z * sc, rv[2 * h + 1].w * sc, hi.w, mid.w);
                lo = hi;
            } else {
            split3(rv[2 * h].x, rv[2 * h].y, hi.x, mid.x, lo.x);
            split3(rv[2 * h].z, rv[2 * h].w, hi.y, mid.y, lo.y);
            split3(rv[2 * h + 1].x, rv[2 * h + 1].y, hi.z, mid.z, lo.z);
            split3(rv[2 * h + 1].z, rv[2 * h + 1].w, hi.w, mid.w, lo.w);
            }
            unsigned char* p = base + kgrp * PL + (krow + (NT / 4) * h) * 16;
            if (AMS_X6_DBG & 2) { asm volatile("" :: "v"(hi.x ^ hi.y ^ hi.z ^ hi.w ^ mid.x ^ mid.y ^ mid.z ^ mid.w ^ lo.x ^ lo.y ^ lo.z ^ lo.w)); continue; }
            *reinterpret_cast<uint4*>(p) = hi;
            *reinterpret_cast<uint4*>(p + PT) = mid;
            if (!F16) *reinterpret_cast<uint4*>(p + 2 * PT) = lo;
        }
    };
    auto stash_m = [&](unsigned char* base, int PL, int PT, int kb, int slot0, int slot1, int slot2, int slot3, float4 (&rv)[4],
                       bool (&vv)[4], float sc) {                                                           // m/n-contiguous source
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (!vv[q]) rv[q] = z;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 hi, mid, lo;
            if constexpr (F16) {
                split2h(comp4(rv[0], j) * sc, comp4(rv[1], j) * sc, hi.x, mid.x);
                split2h(comp4(rv[2], j) * sc, comp4(rv[3], j) * sc, hi.y, mid.y);
                lo = hi;
            } else {
            split3(comp4(rv[0], j), comp4(rv[1], j), hi.x, mid.x, lo.x);
            split3(comp4(rv[2], j), comp4(rv[3], j), hi.y, mid.y, lo.y);
            }
            const int slot = j == 0 ? slot0 : j == 1 ? slot1 : j == 2 ? slot2 : slot3;
            unsigned char* p = base + (kb >> 1) * PL + slot * 16 + (kb & 1) * 8;
            if (AMS_X6_DBG & 2) { asm volatile("" :: "v"(hi.x ^ hi.y ^ mid.x ^ mid.y ^ lo.x ^ lo.y)); continue; }
            *reinterpret_cast<uint2*>(p) = hi;
            *reinterpret_cast<uint2*>(p + PT) = mid;
            if (!F16) *reinterpret_cast<uint2*>(p + 2 * PT) = lo;
        }
    };
    int sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    auto stash = [&]() {
        if (AK) stash_k(As, A_PLANE, A_PART, NSA, ra, va, sc_a);
        else if (actA) stash_m(As, A_PLANE, A_PART, kbA, sa0, sa1, sa2, sa3, ra, va, sc_a);
        if (BKc) stash_k(Bs, B_PLANE, B_PART, NSB, rb, vb, sc_b);
        else if (actB) {
            stash_m(Bs, B_PLANE, B_PART, kbB, sb0, sb1, sb2, sb3, rb, vb, sc_b);
            if (do_bsum) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { bsum4.x += rb[r].x; bsum4.y += rb[r].y; bsum4.z += rb[r].z; bsum4.w += rb[r].w; }
            }
        }
    };

    // MFMA operand addresses: lane l reads row (l & 31) of a 32-row tile, k-group 2 * kstep + (l >> 5)
    const unsigned char* ap[TM];
    const unsigned char* bp[TN];
    auto lane_consts = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));                  // opaque: nothing derived from it is loop-invariant to the compiler
        tid = t; lane = t & 63; l31 = lane & 31; lk = lane >> 5;
        kgrp = t & 3; krow = t >> 2;
        mbA = t % (BMX / 4); kbA = t / (BMX / 4);
        mbB = t % (BNX / 4); kbB = t / (BNX / 4);
        actA = AK || __builtin_amdgcn_readfirstlane(t) < BMX * 2; actB = BKc || __builtin_amdgcn_readfirstlane(t) < BNX * 2;
        sa0 = x6_slot<BMX>(4 * mbA); sa1 = x6_slot<BMX>(4 * mbA + 1); sa2 = x6_slot<BMX>(4 * mbA + 2); sa3 = x6_slot<BMX>(4 * mbA + 3);
        sb0 = x6_slot<BNX>(4 * mbB); sb1 = x6_slot<BNX>(4 * mbB + 1); sb2 = x6_slot<BNX>(4 * mbB + 2); sb3 = x6_slot<BNX>(4 * mbB + 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ar = (wm * TM + i) * 32 + l31;
            ap[i] = As + (AK ? ar : x6_slot<BMX>(ar)) * 16 + lk * A_PLANE;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int br = (wn * TN + j) * 32 + l31;
            bp[j] = Bs + (BKc ? br : x6_slot<BNX>(br)) * 16 + lk * B_PLANE;
        }
    };
    auto frag = [&](const unsigned char* p) {
        if (AMS_X6_DBG & 8) { const uint4 c = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; return __builtin_bit_cast(bf16x8_t, c); }
        return *reinterpret_cast<const bf16x8_t*>(p);
    };
    auto mfma_tile = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            constexpr int NP = F16 ? 2 : 3;
            bf16x8_t b[TN][NP];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[j][p] = frag(bp[j] + p * B_PART + ks * 2 * B_PLANE);
#pragma unroll
            for (int ip = 0; ip < TM; ip += 2) {        // two m-tiles at a time: 24 MFMAs on four alternating accumulators
                bf16x8_t a[2][NP];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p) a[i][p] = frag(ap[ip + i] + p * A_PART + ks * 2 * A_PLANE);
                // smallest partial products first
                constexpr int NPROD = F16 ? 3 : 6;
                constexpr int PA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0};
                constexpr int PB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < NPROD; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (AMS_X6_DBG & 4) { asm volatile("" :: "v"(a[i][PA[t]]), "v"(b[j][PB[t]])); continue; }
                            if constexpr (F16) {
                                const f16x8_t fa = __builtin_bit_cast(f16x8_t, a[i][PA[t]]), fb = __builtin_bit_cast(f16x8_t, b[j][PB[t]]);
                                if (SEP && t < NPROD - 1) accs[ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, accs[ip + i][j], 0, 0, 0);
                                else acc[ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[ip + i][j], 0, 0, 0);
                            } else if (SEP && t < 5)
                                accs[ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], accs[ip + i][j], 0, 0, 0);
                            else
                                acc[ip + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[ip + i][j], 0, 0, 0);
                        }
            }
        }
    };

    fetch(0);
    for (; n_work > 0;) {                           // one work item per trip; every branch below is workgroup-uniform
        lane_consts();
        X6_STAMP(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (SEP) accs[i][j][r] = 0.f; }
        do_bsum = !BKc && g.bsum_part != nullptr && tile_m == 0;
        bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
        stash();
        fetch(1);                                   // tiles past the split's end: clamped addresses, staged as zeros if ever used
        __syncthreads();
        X6_STAMP(1);
        for (int kt = 0;; ++kt) {
            mfma_tile();                            // tile kt
            if (kt + 1 >= nk) break;
            __syncthreads();
            stash();                                // tile kt + 1 (fetched one iteration ago)
            if (!(AMS_X6_DBG & 16)) fetch(kt + 2);
            __syncthreads();
        }

        X6_STAMP(2);
        float4 bs_t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!BKc && do_bsum) {
            // thread (kb, mb) summed rows 4 kb .. 4 kb + 3 of every k-tile, columns 4 mb .. + 3: the 8 threads of a column group meet
            // in LDS in a fixed order (deterministic)
            float4* sbuf = reinterpret_cast<float4*>(smem);
            __syncthreads();
            if (actB) sbuf[tid] = bsum4;
            __syncthreads();
            if (tid < BNX / 4) {
                float4 t = sbuf[tid];
#pragma unroll
                for (int j = 1; j < 8; ++j) { const float4 v = sbuf[tid + (BNX / 4) * j]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
                bs_t = t;
            }
        }
        if (SEP) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += accs[i][j];
        }
        const bool more = WALK && wi + 1 < n_work;
        const int cur_role = WALK ? role : W_FULL;
        // stream-K slots: the accumulators in FRAGMENT order (lane l's float4 q of MFMA tile (i, j) of wave w at
        // ((w TM TN + i TN + j) 4 + q) KB + 16 l: 64 lanes move 1 KB contiguous), then BNX column sums
        constexpr unsigned TILE_B = (unsigned)BMX * BNX * 4u;
        const unsigned loff = (unsigned)(wave * TM * TN * 4) * 1024u + (unsigned)lane * 16u;
        if (EPI == EPI_STORE && cur_role == W_OWNER) {
            // OWNER: the workgroups sk_w - 1, sk_w - 2, ... whose unit ranges reach into this tile hold its earlier k ranges
            const int tile_start = own_tile * nk_full;
            if (tid == 0) {
                for (int wp = sk_w - 1; wp >= 0 && (wp + 1) * sk_U / sk_Gx > tile_start; --wp) {
                    const int* const f = g0.sk_flags + (wp * 8 + sk_x);
                    int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 26)) __builtin_trap();      // seconds: the producer is gone -- fail loudly, never continue with a hole
                    }
                }
            }
            __syncthreads();
            for (int wp = sk_w - 1; wp >= 0 && (wp + 1) * sk_U / sk_Gx > tile_start; --wp) {
                const int src = wp * 8 + sk_x;
                const __amdgpu_buffer_rsrc_t rs = x6_rsrc(g0.sk_slots + (long)src * g0.sk_slot_floats, (unsigned)g0.sk_slot_floats * 4u);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float4 v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = ld16_sc1(rs, loff + (unsigned)((i * TN + j) * 4 + q) * 1024u);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[i][j][4 * q] += v[q].x; acc[i][j][4 * q + 1] += v[q].y;
                            acc[i][j][4 * q + 2] += v[q].z; acc[i][j][4 * q + 3] += v[q].w;
                        }
                    }
                if (!BKc && g0.bsum_part != nullptr && tid < BNX / 4) {
                    const float4 t = ld16_sc1(rs, TILE_B + (unsigned)tid * 16u);
                    bs_t.x += t.x; bs_t.y += t.y; bs_t.z += t.z; bs_t.w += t.w;
                }
                if (tid == 0) __hip_atomic_store(g0.sk_flags + src, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
            }
        }
        if (!BKc && do_bsum && cur_role != W_PART && tid < BNX / 4) {
            const int n = n0 + tid * 4;
            if (n < g.N) {
                if (g.splits > 1 || !g.bsum_out) *reinterpret_cast<float4*>(g.bsum_part + (long)split * g.N + n) = bs_t;
                else {                                  // the whole k range is here: this workgroup holds the column sums -- no finishing launch
                    float4* const po = reinterpret_cast<float4*>(g.bsum_out + n);
                    if (g.bsum_accumulate) { const float4 o = *po; bs_t.x += o.x; bs_t.y += o.y; bs_t.z += o.z; bs_t.w += o.w; }
                    *po = bs_t;
                }
            }
        }
        if constexpr (EPI == EPI_MAXPOOL) {             // stride-1 conv + max_pool_with_argmax (models/adapt.py:115-117), 128 x 128 tile only
            static_assert(CFG == 0 || CFG == 3, "the max-pool epilogue is written for 2 x WNC waves of 64 x 64");
            if constexpr (F16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] *= sc_inv;
            }
            __syncthreads();                            // every wave is done with the LDS images
            maxpool_epilogue<C::WNC>(g, acc, reinterpret_cast<float*>(smem), tile_m, m0, n0, wm, wn, l31, lk);
            return;                                     // (launched with one workgroup per item)
        }
        X6_STAMP(3);
        // what the stores of THIS item need, saved before the per-item state moves on
        float* const out = g.splits > 1 ? g.partial + (long)split * g.M * g.N : g.C;
        const long ldo = g.splits > 1 ? g.N : g.ldc;
        const float* const ebias = g.bias;
        const int em0 = m0, en0 = n0;
        if (PERSIST && more) {
            __syncthreads();                            // every wave has read its last fragments: the staging registers and LDS are free
            setup(wi + 1);
            fetch(0);                                   // the next item's first k-tile is in flight BEFORE this item's stores
        }
        X6_STAMP(4);
        if (EPI == EPI_STORE && cur_role == W_PART) {
            // PARTIAL segment: publish (16-byte write-through stores -> every wave's stores acknowledged -> barrier -> flag).  In the
            // PERSIST form the next segment's first k-tile is in flight here: its loads are older than these stores and vmcnt retires
            // in order, so the wait below covers both -- the stash that follows would have waited for them anyway.
            const int me = (int)blockIdx.x;
            const __amdgpu_buffer_rsrc_t rs = x6_rsrc(g0.sk_slots + (long)me * g0.sk_slot_floats, (unsigned)g0.sk_slot_floats * 4u);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        st16_sc1(rs, loff + (unsigned)((i * TN + j) * 4 + q) * 1024u,
                                 make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]));
            if (!BKc && g0.bsum_part != nullptr && tid < BNX / 4) st16_sc1(rs, TILE_B + (unsigned)tid * 16u, bs_t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(g0.sk_flags + me, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // Epilogue.  C/D layout of a 32x32 MFMA (any input type): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
            float vmax = 0.f;
            if (TN == 2 && g0.c_vec) {
                // A lane holds ONE column of 16 rows per MFMA tile: stored as they lie that is 64 dword stores per lane, and issuing
                // them took a workgroup 9.3 us per 128 x 256 tile with nothing else on the CU (tools/gemm_anatomy.py: 18 % of a K = 600
                // tile).  Each wave turns its 64-column band round in a PRIVATE 8 KB patch of the (now idle) operand LDS, 32 rows at a
                // time, and stores rows: 16 lanes x 16 bytes = one 256-byte row segment, 8 float4 stores per lane and MFMA row tile.
                float* const wl = reinterpret_cast<float*>(smem) + wave * 2048;
                if (!(PERSIST && more)) __syncthreads();    // every wave has read its last fragments (the PERSIST path has just met for that)
                const int rr = lane >> 4, c4 = (lane & 15) * 4;
                const int col = en0 + wn * 64 + c4;
                const bool cok = col < g0.N;                // (N % 4 == 0: a float4 is inside or outside)
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g0.splits == 1 && ebias && cok) bv = *reinterpret_cast<const float4*>(ebias + col);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (F16) acc[i][j] *= sc_inv;
#pragma unroll
                        for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + l31] = acc[i][j][r];
                    }
#pragma unroll
                    for (int p8 = 0; p8 < 8; ++p8) {
                        const int rl = p8 * 4 + rr;
                        float4 v = *reinterpret_cast<const float4*>(wl + rl * 64 + c4);
                        const int row = em0 + (wm * TM + i) * 32 + rl;
                        if (row < g0.M && cok) {
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            float4* const p = reinterpret_cast<float4*>(out + (long)row * ldo + col);
                            if (g0.splits == 1 && g0.accumulate) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            *p = v;
                            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                        }
                    }
                }
                if (more) __syncthreads();                  // the next item's images overwrite the patches
            } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = en0 + (wn * TN + j) * 32 + l31;
                    if (col >= g0.N) continue;
                    if constexpr (F16) acc[i][j] *= sc_inv;
                    const float bv = (g0.splits == 1 && ebias) ? ebias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = em0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        if (row < g0.M) {
                            float v = acc[i][j][r] + bv;
                            float* p = out + (long)row * ldo + col;
                            if (g0.splits == 1 && g0.accumulate) v += *p;
                            *p = v;
                            vmax = fmaxf(vmax, fabsf(v));
                        }
                    }
                }
            }
            vmax_wg = fmaxf(vmax_wg, vmax);
        }
        X6_STAMP(5);
#if AMS_X6_STAMP
        if (threadIdx.x == 0 && wi < 8) { g_x6_stamp[((int)blockIdx.x * 8 + wi) * 8 + 6] = cur_role; g_x6_stamp[((int)blockIdx.x * 8 + wi) * 8 + 7] = nk; }
#endif
        if (!more) break;
        if (!PERSIST) { __syncthreads(); setup(wi + 1); fetch(0); }
        ++wi;
    }
    if (EPI == EPI_STORE && g0.amax_out != nullptr) {
        // max |C| of the launch.  With the caller's scratch (sk_flags): every workgroup leaves its maximum as a write-through store,
        // counts itself in, and the LAST one folds them into amax_out[0] and zeroes words and ticket again -- nothing to clear in
        // front of the launch, gridDim.x stores instead of that many atomics on one word.  Without scratch: atomicMax on a word the
        // entry point cleared.  (NaN: fmaxf drops it -- a NaN output reaches the caller through C itself.)
        __syncthreads();
        float* const sm = reinterpret_cast<float*>(smem);
        int* const sl = reinterpret_cast<int*>(smem) + 16;
        const float wmax = wave_max(vmax_wg);
        if (lane == 0) sm[wave] = wmax;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < NT / 64; ++w) t = fmaxf(t, sm[w]);
            if (!g0.amax_fold) { atomicMax(g0.amax_out, __float_as_uint(t)); *sl = 0; }
            else {
                unsigned* const part = reinterpret_cast<unsigned*>(g0.sk_flags) + 512;
                __hip_atomic_store(part + blockIdx.x, __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *sl = __hip_atomic_fetch_add(part + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
            }
        }
        __syncthreads();
        if (*sl) {
            unsigned* const part = reinterpret_cast<unsigned*>(g0.sk_flags) + 512;
            unsigned t = 0u;
            for (int b = tid; b < G; b += NT) {
                t = max(t, __hip_atomic_load(part + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                __hip_atomic_store(part + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o));
            __syncthreads();
            if (lane == 0) reinterpret_cast<unsigned*>(sm)[wave] = t;
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NT / 64; ++w) t = max(t, reinterpret_cast<unsigned*>(sm)[w]);
                g0.amax_out[0] = t;
                __hip_atomic_store(part + 512, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int AMODE, int BMODE, int CFG, int EPI = EPI_STORE, bool SEP = false, bool F16 = false>
__global__ __launch_bounds__(X6Cfg<CFG>::WMC * X6Cfg<CFG>::WNC * 64, CFG == 0 ? 2 : ((AMS_SK_CAPPED && !SEP && EPI == EPI_STORE) ? 3 : 1)) void gemm_x6_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[x6_lds(CFG)];
    x6_body<AMODE, BMODE, CFG, EPI, SEP, F16>(g, smem);
}

__global__ void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C, const float* __restrict__ bias,
                                     int M, int N, long ldc, int splits, int accumulate, long c_zs, long bias_zs) {
    const long total = (long)M * N;
    partial += (long)blockIdx.y * splits * total;
    C += (long)blockIdx.y * c_zs;
    if (bias) bias += (long)blockIdx.y * bias_zs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (long)m * N);
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += partial[(long)k * total + i];
        if (bias) s += bias[n];
        float* p = C + (long)m * ldc + n;
        if (accumulate) s += *p;
        *p = s;
    }
}

// 16-byte form of the same reduction (N, ldc multiples of 4, 16-byte aligned C / bias, M*N < 2^31): one float4
// per thread and slab, 32-bit index arithmetic.  The scalar kernel above ran at 1.8 TB/s (55 us for the 3-slab dense weight
// gradient); the slabs are summed in the same order, so the result is bit-identical.
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                                const float* __restrict__ bias, int M, int N, long ldc, int splits,
                                                                int accumulate, long c_zs, long bias_zs) {
    const int n4 = N >> 2, total4 = M * n4;
    const long total = (long)M * N;
    partial += (long)blockIdx.y * splits * total;
    C += (long)blockIdx.y * c_zs;
    if (bias) bias += (long)blockIdx.y * bias_zs;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int m = i / n4, c4 = i - m * n4;
        const float4* src = reinterpret_cast<const float4*>(partial) + i;
        float4 s = src[0];
        for (int k = 1; k < splits; ++k) {
            const float4 v = src[(long)k * (total >> 2)];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) {
            const float4 bv = reinterpret_cast<const float4*>(bias)[c4];
            s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w;
        }
        float4* p = reinterpret_cast<float4*>(C + (long)m * ldc) + c4;
        if (accumulate) { const float4 o = *p; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        *p = s;
    }
}

__global__ void bsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int splits, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(long)k * N + n];
    out[n] = accumulate ? out[n] + s : s;
}

// Split-K factor from a small cost model calibrated on MI355X (scratch sweep, round 1):
//   t(s) = n * (k_iters * 1.02us + 5us) / occ(n)  +  (s+1)*M*N*4 B / 2.5 TB/s        [s > 1]
// n = ceil(tiles*s/256) workgroups end up on the busiest CU (equal-work workgroups time-share a CU's four
// SIMDs, so the launch ends when that CU drains); occ(n) discounts CUs holding only 1-3 workgroups, whose
// barrier stalls are not covered by a neighbour's MFMAs; the last term is the fp32 partial-slab round trip.
struct TilePlan { int bm, bn, bk; double us16; bool alone; };   // block tile, the cost of 16 k of it for one workgroup (microseconds), one workgroup per CU by construction
inline TilePlan f32_plan() { return {BM, BN, BK, 1.024, false}; }
// bf16x6 tile configuration (X6Cfg) of an M x N output.  Default (rule 2): 128 x 256 (8 waves of 64 x 64) where it wastes under
// 30 % of the columns it covers (AMS_GEMM_X6WASTE = 1.30; 1.10 until the fp16x3 products: with half the MFMAs per k-tile the
// 8-wave tile wins even at N = 600 -> 768 -- LSTM dX 98 -> 87 us, dense dX 289 -> 277 us alone, the B = 64 step 3.14 -> 3.09 ms,
// tools/probes/cfg_sweep.sh + tools/probes/ab_bench.sh), 128 x 128 otherwise -- both carry the second accumulator set (SEP) when they are not
// residency-capped.  Rule 1 (AMS_GEMM_X6RULE=1) adds the 256 x 256 tile (8 waves of 128 x 64) where both sides fit: +0.7 % on the
// step (projections 107 vs 121 us), but no registers for SEP -- its outputs carry the bf16 MFMA's truncation bias (-0.3 .. -1 ulp
// each, coherent: the LSTM bias gradients, sums over 5120 rows downstream of it, were 1.2e-5 off the oracle instead of < 1e-6), so
// it is not the default.  A residency-capped launch never takes it (8 waves x 256 VGPRs leave no room for a ring workgroup).
// Rule 0: {256 x 256, 128 x 128} only.  (256 x 128 was measured too -- profiles/r02_i_gemm_x6_tile_configs.txt -- and lost to
// 128 x 128 on every shape it suits.)
inline int x6_choose_cfg(int M, int N, bool capped) {
    if (tuning().x6cfg == 0 || tuning().x6cfg == 1 || tuning().x6cfg == 3) return tuning().x6cfg;
    const bool m256 = (double)ceil_div(M, 256) * 256 <= 1.10 * M, n256 = (double)ceil_div(N, 256) * 256 <= tuning().x6waste * N;
    const int rule = tuning().x6rule;
    if (m256 && n256 && !capped && rule <= 1) return 1;
    if (n256 && rule >= 1) return 3;
    return 0;
}
inline TilePlan x6_plan(int cfg) {
    return {x6_bm(cfg), x6_bn(cfg), X6_BK, cfg == 1 ? AMS_GEMM_X6_US16_1 : cfg == 0 ? AMS_GEMM_X6_US16 : AMS_GEMM_X6_US16_2, cfg != 0 && AMS_X6_OCC1};
}
inline int choose_splits(int M, int N, int K, int nbatch, const TilePlan& tp, double* t_out = nullptr, int smax = 32) {
    const int tiles = ceil_div(M, tp.bm) * ceil_div(N, tp.bn) * nbatch;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= smax; ++s) {
        if (s > 1 && K / s < 128) break;
        const int kps = ceil_div(ceil_div(K, s), tp.bk) * tp.bk;
        const int s2 = ceil_div(K, kps);
        const int n = ceil_div((long)tiles * s2, 256);
        // an 8-wave configuration is alone on its CU whatever the grid: no discount for few workgroups per CU
        const double occ = tp.alone ? 1.0 : n <= 1 ? 0.62 : (n == 2 ? 0.80 : (n == 3 ? 0.92 : 1.0));
        double t = n * ((kps / 16.0) * tp.us16 + 5.0) / occ;
        if (s2 > 1) t += (double)(s2 + 1) * M * N * nbatch * 4.0 / 2.5e6;
        if (t < best_t - 1e-9) { best_t = t; best = s2; }
    }
    if (t_out) *t_out = best_t;
    return best;
}
// what a workspace query assumes: the process-wide arithmetic (a launch whose operands are not 16-byte addressable falls back to
// the f32 kernel and re-plans within the workspace it is given)
inline int choose_splits(int M, int N, int K, int nbatch, bool capped) {
    return choose_splits(M, N, K, nbatch, use_x6() ? x6_plan(x6_choose_cfg(M, N, capped)) : f32_plan());
}
inline size_t slab_bytes(int M, int N, int nbatch, int splits) { return splits <= 1 ? 0 : (size_t)nbatch * splits * M * N * sizeof(float); }

// Band height of the tile order: the patch of tiles one XCD works on at a time (its share of the grid, at most ~64 in
// flight) should be as square as the tile grid allows, so every operand panel is fetched by as few XCDs as possible.
inline int choose_group_m(int tiles_m, int tiles_n) {
    int c = ceil_div((long)tiles_m * tiles_n, 8);
    if (c > 64) c = 64;
    int gm = (int)(sqrt((double)c) + 0.5);
    if (gm < 1) gm = 1;
    if (gm > tiles_m) gm = tiles_m;
    if (ceil_div(c, gm) > tiles_n) gm = ceil_div(c, tiles_n);
    if (gm > tiles_m) gm = tiles_m;
    if (tuning().group_m > 0) gm = tuning().group_m;
    return gm;
}

// raise a kernel's dynamic-LDS limit once per KERNEL (the memo is a static of the instantiation for that kernel's address) and thread
template <auto Kernel>
inline void raise_dyn_lds(int bytes) {
    static thread_local int raised = 0;
    if (raised < bytes) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        raised = bytes;
    }
}

inline int device_cus() {
    static int cus = 0;
    if (!cus) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); if (cus <= 0) cus = 256; }
    return cus;
}

// Stream-K plan of an x6 launch (x6_body): the resident grid G (a multiple of 8), the whole-tile rounds every workgroup runs, and the
// model's time for it in the units of choose_splits (microseconds at the bf16x6 calibration).  rounds < 0: does not apply.
struct SkPlan { int rounds = -1; unsigned grid = 0; double t = 1e30; };
inline SkPlan sk_plan(int tiles_all, int K, const TilePlan& tp, int wgcu) {
    SkPlan p;
    const int G = (device_cus() + 7) / 8 * 8 * wgcu;
    if (G > 512 || tiles_all < 8) return p;
    const int Gx = G / 8, q = tiles_all / 8, nkf = ceil_div(K, tp.bk);
    const int R = q / Gx;
    const int n_max = q - R * Gx + (tiles_all % 8 ? 1 : 0);                // left-over tiles of the fullest XCD
    if (n_max <= 0 || tiles_all - R * G == 0) return p;                    // whole rounds only: nothing to share
    const int sk_kt = ceil_div((long)n_max * nkf, Gx);                     // k-tiles of a workgroup's share
    if (sk_kt < 2) return p;
    const int contrib = ceil_div(nkf, sk_kt);                              // partial tiles an owner adds up
    const double tk = tp.us16 * (tp.bk / 16.0);
    const double occ = tp.alone ? 1.0 : wgcu <= 1 ? 0.62 : (wgcu == 2 ? 0.80 : 0.92);
    p.rounds = R;
    p.grid = (unsigned)G;
    p.t = wgcu * ((R * nkf + sk_kt) * tk + (R + 1) * 5.0 + 3.0 + 2.0 * contrib) / occ;
    return p;
}

template <int AMODE, int BMODE>
ams_status launch(GemmArgs& g, const LaunchOpt& opt, void* ws, size_t ws_bytes, hipStream_t st, int nbatch = 1,
                  float* bsum_out = nullptr, int bsum_accumulate = 0, float* bsum_ws = nullptr) {
    constexpr bool AKc = (AMODE == A_ROW), BKcc = (BMODE == B_COL);
    const int lds_pad = opt.lds_pad < 0 ? 0 : opt.lds_pad;
    const bool capped = lds_pad > 0;
    const bool vec_off = tuning().novec;
    const bool vec = g.a_vec && g.b_vec && !vec_off &&
                     (AMODE == A_FRAMES ? (g.K % 4 == 0 && g.fr_L >= 4) : AMODE == A_FRAMES_T ? (g.M % 4 == 0 && g.fr_L >= 4) :
                      AKc ? (g.K % 4 == 0 && g.K >= 4) : (g.M % 4 == 0 && g.M >= 4)) &&
                     (BKcc ? (g.K % 4 == 0 && g.K >= 4) : (g.N % 4 == 0 && g.N >= 4));
    const bool x6 = vec && use_x6();
    const int cfg = x6 ? ((opt.force_cfg == 0 || opt.force_cfg == 3) && tuning().x6cfg < 0 ? opt.force_cfg : x6_choose_cfg(g.M, g.N, capped)) : 0;
    // fp16x3 when the caller supplied bounds for both operands
    const bool f16 = x6 && cfg != 1 && opt.amax_a && opt.amax_b && tuning().f16x3;
    g.amax_a = f16 ? opt.amax_a : nullptr;
    g.amax_b = f16 ? opt.amax_b : nullptr;
    g.amax_out = x6 ? opt.amax_out : nullptr;
    g.c_vec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (g.c_zs % 4 == 0) && (g.bias_zs % 4 == 0) &&
              (((uintptr_t)g.C | (uintptr_t)ws | (uintptr_t)g.bias) & 15) == 0 && tuning().c_vec;
    if (opt.amax_out && !x6) return AMS_E_INVALID_ARG;                    // only the x6 epilogue measures its output
    const TilePlan tp = x6 ? x6_plan(cfg) : f32_plan();
    const int tiles = ceil_div(g.M, tp.bm) * ceil_div(g.N, tp.bn);
    g.group_m = choose_group_m(ceil_div(g.M, tp.bm), ceil_div(g.N, tp.bn));
    int splits = 1;
    double t_split = 1e30;
    if (ws && !opt.amax_out) {
        splits = choose_splits(g.M, g.N, g.K, nbatch, tp, &t_split);
        if (tuning().splits > 0) splits = tuning().splits;
        while (splits > 1 && (size_t)nbatch * splits * g.M * g.N * sizeof(float) > ws_bytes) --splits;
    } else choose_splits(g.M, g.N, g.K, nbatch, tp, &t_split, 1);
    // workgroups of this variant a CU holds: 8-wave configurations 1; 128 x 128: 2 by registers, a residency cap may ask for 1
    int wgcu = 1;
    if (x6 && cfg == 0) { wgcu = capped ? 163840 / (17152 + lds_pad) : 2; if (wgcu < 1) wgcu = 1; if (wgcu > 2) wgcu = 2; }
    // stream-K instead of whole-tile rounds / split-K slabs (x6_body), where the caller lent scratch and the model prefers it
    SkPlan sk;
    g.sk_rounds = -1; g.sk_flags = nullptr; g.sk_slots = nullptr; g.sk_slot_floats = 0; g.amax_fold = 0;
    if (x6 && cfg != 1 && (!capped || AMS_SK_CAPPED) && opt.sk_scratch && tuning().sk != 0 && tuning().splits <= 0 && AMS_GEMM_XCD_FLAT) {
        sk = sk_plan(tiles * nbatch, g.K, tp, wgcu);
        const size_t slot = (size_t)(tp.bm * tp.bn + tp.bn) * sizeof(float);
        if (sk.rounds >= 0 && (opt.sk_bytes < AMS_SK_FLAG_BYTES + (size_t)sk.grid * slot || (((uintptr_t)opt.sk_scratch) & 15))) sk.rounds = -1;
        if (sk.rounds >= 0 && tuning().sk == 1 && !(sk.t < 0.97 * t_split)) sk.rounds = -1;
        // only the TAIL of a launch with whole rounds (mode 1): a launch of fewer tiles than workgroups gains ~6 % of k-tiles from an even
        // share and pays it back in a second item per workgroup, the owner's wait and 256 instead of 240 busy CUs (tools/gemm_anatomy.py:
        // dense dX 322 vs 317 us, LSTM dX 90 vs 75; in the step 2.861 ms either way, 2.90 with stream-K everywhere, 2.886 without)
        // (a launch that measures its output cannot be cut into k-split slabs: there the even share is the only way to fill the chip)
        if (sk.rounds == 0 && tuning().sk == 1 && !opt.amax_out) sk.rounds = -1;
        if (sk.rounds >= 0) {
            splits = 1;
            g.sk_rounds = sk.rounds;
            g.sk_flags = (int*)opt.sk_scratch;
            g.sk_slots = (float*)((char*)opt.sk_scratch + AMS_SK_FLAG_BYTES);
            g.sk_slot_floats = (int)(slot / sizeof(float));
        }
    }
    int kps = ceil_div(g.K, splits);
    kps = ceil_div(kps, tp.bk) * tp.bk;
    splits = ceil_div(g.K, kps);
    g.splits = splits;
    g.k_per_split = kps;
    g.partial = (float*)ws;
    g.bsum_part = bsum_out ? bsum_ws : nullptr;
    const bool bsum_in_launch = x6 && bsum_out && splits == 1 && (((uintptr_t)bsum_out) & 15) == 0;
    if (sk.rounds >= 0 && bsum_out && !bsum_in_launch) return AMS_E_INVALID_ARG;     // (16-byte aligned column sums are an entry requirement)
    g.bsum_out = bsum_in_launch ? bsum_out : nullptr;
    g.bsum_accumulate = bsum_accumulate;
    g.nbatch = nbatch;
    dim3 grid((unsigned)((long)tiles * splits * nbatch));
    const bool prio_off = tuning().noprio;
    g.hiprio = (!capped && !prio_off) ? 1 : 0;
    if (g.amax_out) {
        // the launch folds its output maximum itself when the caller lent scratch (words [512, 1025) of its flag area) and the grid fits
        unsigned gx = grid.x;
        if (sk.rounds >= 0) gx = sk.grid;
        else if (x6 && tuning().x6persist > 0 && !capped && cfg != 1) { const long cap = (long)((device_cus() + 7) / 8 * 8) * (cfg == 0 ? 2 : 1) * tuning().x6persist; if ((long)gx > cap) gx = (unsigned)cap; }
        if (opt.sk_scratch && opt.sk_bytes >= AMS_SK_FLAG_BYTES && gx <= 512 && (((uintptr_t)opt.sk_scratch) & 15) == 0) {
            g.amax_fold = 1;
            g.sk_flags = (int*)opt.sk_scratch;
        } else if (hipMemsetAsync(g.amax_out, 0, sizeof(float), st) != hipSuccess) return AMS_E_LAUNCH_FAILED;
    }
    // Occupancy cap for launches that are meant to run BESIDE latency-critical kernels (weight-gradient products on
    // the side stream): unused dynamic LDS limits how many of these workgroups a CU admits, leaving registers/slots
    // for the recurrence.  An explicit argument of the entry points (LaunchOpt::lds_pad).
    if (x6) {
        // persistent grid (x6_body, uncapped two-accumulator variants): one resident set of workgroups -- 256 CUs x (2 for the 4-wave configuration, 1 for the 8-wave
        // ones) x AMS_X6_PERSIST (default 1; 0 = one workgroup per item as before) -- walks the items.
        if (sk.rounds >= 0) grid.x = sk.grid;
        else if (tuning().x6persist > 0 && !capped && cfg != 1) {      // the variants launched with SEP = true below
            const long cap = (long)((device_cus() + 7) / 8 * 8) * (cfg == 0 ? 2 : 1) * tuning().x6persist;
            if ((long)grid.x > cap) grid.x = (unsigned)cap;
        }
        // residency: the 128 x 128 configuration holds 48.75 KB of LDS (3 workgroups per CU by LDS, 2 by registers); a pad asks for
        // what it asks of the 16.8 KB f32 kernel (workgroups per CU), restated.  The 8-wave configurations are alone on a CU anyway.
        if (cfg == 0) {
            int pad = 0;
            if (capped) {
                int wg = 163840 / (17152 + lds_pad);
                if (wg < 1) wg = 1;
                pad = 163840 / wg - x6_lds(0) - 1024;
                if (pad < 0) pad = 0;
            }
            if (x6_lds(0) + pad > 64 * 1024) {
                raise_dyn_lds<&gemm_x6_kernel<AMODE, BMODE, 0>>(pad);
                raise_dyn_lds<&gemm_x6_kernel<AMODE, BMODE, 0, EPI_STORE, false, true>>(pad);
            }
            if (f16) {
                if (capped) hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 0, EPI_STORE, false, true>), grid, dim3(256), (size_t)pad, st, g);
                else hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 0, EPI_STORE, true, true>), grid, dim3(256), 0, st, g);
            } else if (capped) hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 0>), grid, dim3(256), (size_t)pad, st, g);
            else hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 0, EPI_STORE, true>), grid, dim3(256), 0, st, g);
        } else if (cfg == 1) hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 1>), grid, dim3(512), 0, st, g);
        else if (f16) {
            if (capped) hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 3, EPI_STORE, false, true>), grid, dim3(512), 0, st, g);
            else hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 3, EPI_STORE, true, true>), grid, dim3(512), 0, st, g);
        } else if (capped) hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 3>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((gemm_x6_kernel<AMODE, BMODE, 3, EPI_STORE, true>), grid, dim3(512), 0, st, g);
    } else if (vec) {
        if (lds_pad > 40 * 1024) raise_dyn_lds<&gemm_f32_kernel<AMODE, BMODE, 0, AMS_GEMM_BK, true, AMS_GEMM_PF>>(lds_pad);      // beyond the default 64 KB static + dynamic limit
        hipLaunchKernelGGL((gemm_f32_kernel<AMODE, BMODE, 0, AMS_GEMM_BK, true, AMS_GEMM_PF>), grid, dim3(256), (size_t)lds_pad, st, g);
    } else {
        if (lds_pad > 40 * 1024) raise_dyn_lds<&gemm_f32_kernel<AMODE, BMODE, 0>>(lds_pad);
        hipLaunchKernelGGL((gemm_f32_kernel<AMODE, BMODE>), grid, dim3(256), (size_t)lds_pad, st, g);
    }
    ams_status s = ams_check_launch();
    if (s != AMS_OK) return s;
    if (splits > 1) {
        const long total = (long)g.M * g.N;
        const bool rvec = (g.N % 4 == 0) && (g.ldc % 4 == 0) && (g.c_zs % 4 == 0) && (g.bias_zs % 4 == 0) &&
                          (((uintptr_t)g.C | (uintptr_t)g.partial | (uintptr_t)g.bias) & 15) == 0 && total < (1L << 31);
        if (rvec) {
            int blocks = (int)((total / 4 + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3(blocks, nbatch), dim3(256), 0, st, g.partial, g.C, g.bias, g.M, g.N, g.ldc,
                               splits, g.accumulate, g.c_zs, g.bias_zs);
        } else {
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, nbatch), dim3(256), 0, st, g.partial, g.C, g.bias, g.M, g.N, g.ldc,
                               splits, g.accumulate, g.c_zs, g.bias_zs);
        }
        s = ams_check_launch();
    }
    if (s == AMS_OK && bsum_out && !bsum_in_launch) {
        if (!vec || BMODE != B_ROW) return AMS_E_INVALID_ARG;          // only the 16-byte B_ROW fetch path accumulates the sums
        hipLaunchKernelGGL(bsum_finish_kernel, dim3(ceil_div(g.N, 256)), dim3(256), 0, st, bsum_ws, bsum_out, g.N, splits, bsum_accumulate);
        s = ams_check_launch();
    }
    return s;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

void ams_gemm_set_arith(int mode) { g_gemm_arith.store(mode ? 1 : 0, std::memory_order_relaxed); }
int ams_gemm_get_arith(void) { return use_x6() ? 1 : 0; }

// scratch of the stream-K launches: flags + one partial-tile slot per resident workgroup of the larger configuration
size_t ams_gemm_sk_scratch_bytes(void) {
    const size_t g8 = (size_t)(device_cus() + 7) / 8 * 8;
    const size_t a = g8 * 2 * (128 * 128 + 128) * sizeof(float), b = g8 * AMS_SK_SLOT_BYTES;
    return AMS_SK_FLAG_BYTES + (a > b ? a : b);
}

size_t ams_gemm_workspace_bytes(int M, int N, int K, int nbatch, int lds_pad) {
    if (M <= 0 || N <= 0 || K <= 0 || nbatch <= 0) return 0;
    int splits = choose_splits(M, N, K, nbatch, lds_pad > 0);
    if (tuning().splits > 0) splits = tuning().splits;
    return slab_bytes(M, N, nbatch, splits);
}

// C (+)= A^T . B  AND  bsum_out[N] (+)= column sums of B, in one pass over B (reference: the weight and bias gradients of a
// width-1 Conv1D, utils/ops.py:501-503 under tf.gradients).  bsum_ws: 32 * N floats (one row per possible split).
ams_status ams_gemm_f32_at_b_colsum(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                                    int accumulate, float* bsum_out, int bsum_accumulate, float* bsum_ws, const float* amax_a,
                                    const float* amax_b, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch, size_t sk_bytes,
                                    void* stream) {
    AMS_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && bsum_out && bsum_ws);
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    AMS_REQUIRE(g.a_vec && g.b_vec && M % 4 == 0 && N % 4 == 0 && aligned16(bsum_ws) && !tuning().novec);
    LaunchOpt o; o.amax_a = amax_a; o.amax_b = amax_b; o.lds_pad = lds_pad; o.sk_scratch = sk_scratch; o.sk_bytes = sk_bytes;
    return launch<A_COL, B_ROW>(g, o, ws, ws_bytes, (hipStream_t)stream, 1, bsum_out, bsum_accumulate, bsum_ws);
}

ams_status ams_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                        float* C, long ldc, const float* bias, int accumulate, int mask_period, int mask_skip, const float* amax_a,
                        const float* amax_b, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch, size_t sk_bytes, void* stream) {
    AMS_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C);
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate;
    g.mask_period = mask_period; g.mask_skip = mask_skip;
    g.a_vec = aligned16(A) && (lda % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    LaunchOpt o; o.amax_a = amax_a; o.amax_b = amax_b; o.lds_pad = lds_pad; o.sk_scratch = sk_scratch; o.sk_bytes = sk_bytes;
    if (!transA && !transB) return launch<A_ROW, B_ROW>(g, o, ws, ws_bytes, st);
    if (!transA && transB) return launch<A_ROW, B_COL>(g, o, ws, ws_bytes, st);
    if (transA && !transB) return launch<A_COL, B_ROW>(g, o, ws, ws_bytes, st);
    return launch<A_COL, B_COL>(g, o, ws, ws_bytes, st);
}

// nbatch products of one shape in ONE launch: operand z is at A + z*a_zs etc. (element offsets, any sign).
// Used for the two directions' recurrent-kernel gradients: 2 x 30 tiles fill the chip better than 30 twice.
ams_status ams_gemm_f32_batched(int transA, int transB, int M, int N, int K, const float* A, long lda, long a_zs, const float* B,
                                long ldb, long b_zs, float* C, long ldc, long c_zs, int nbatch, int accumulate, int mask_period,
                                int mask_skip, const float* amax_a, const float* amax_b, int lds_pad, void* ws, size_t ws_bytes,
                                void* sk_scratch, size_t sk_bytes, void* stream) {
    AMS_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C && nbatch >= 1 && nbatch <= 64);
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.bias = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.accumulate = accumulate;
    g.mask_period = mask_period; g.mask_skip = mask_skip;
    g.a_zs = a_zs; g.b_zs = b_zs; g.c_zs = c_zs;
    g.a_vec = aligned16(A) && (lda % 4 == 0) && (a_zs % 4 == 0);
    g.b_vec = aligned16(B) && (ldb % 4 == 0) && (b_zs % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    LaunchOpt o; o.amax_a = amax_a; o.amax_b = amax_b; o.lds_pad = lds_pad; o.sk_scratch = sk_scratch; o.sk_bytes = sk_bytes;
    if (!transA && !transB) return launch<A_ROW, B_ROW>(g, o, ws, ws_bytes, st, nbatch);
    if (!transA && transB) return launch<A_ROW, B_COL>(g, o, ws, ws_bytes, st, nbatch);
    if (transA && !transB) return launch<A_COL, B_ROW>(g, o, ws, ws_bytes, st, nbatch);
    return launch<A_COL, B_COL>(g, o, ws, ws_bytes, st, nbatch);
}

// Adaptive analysis filterbank, path A (reference models/adapt.py:122): y[b,t,n] = sum_k xpad[b,t*hop+k-pl] f[k,n]
size_t ams_front_conv_fwd_workspace_bytes(int Bt, int L, int W, int N, int hop) {
    if (Bt <= 0 || L <= 0 || W <= 0 || N <= 0 || hop <= 0) return 0;
    return ams_gemm_workspace_bytes(Bt * ((L + hop - 1) / hop), N, W, 1, 0);
}

// ws (may be NULL: no split-K) lets the few-tile benchmark shape (5120 x 256 output = 80 tiles) fill 256 CUs.
// amax_x / amax_f (both or neither): operand bounds -> fp16x3, and the tile configuration that needs no split-K at the benchmark
// shape (128 x 128: 240 tiles).  amax_y (optional, 16-bit-pipe launches only): the launch leaves max |y| there (cleared by a 4-byte
// memset node in front of it when the caller lends no scratch) -- the bound the next product wants, without a pass over y.
// the 16-byte-fetch (and with it the 16-bit-pipe) form of the strided analysis product applies: mirrors launch<A_FRAMES, B_ROW>'s test
static bool front_conv_is_x6(const float* x, const float* f, int L, int W, int N, int hop) {
    const int T = (L + hop - 1) / hop;
    int pad_total = (T - 1) * hop + W - L;
    if (pad_total < 0) pad_total = 0;
    const bool a_vec = aligned16(x) && (L % 4 == 0) && (hop % 4 == 0) && ((pad_total / 2) % 4 == 0);
    const bool b_vec = aligned16(f) && (N % 4 == 0);
    return use_x6() && a_vec && b_vec && !tuning().novec && W % 4 == 0 && L >= 4 && N >= 4;
}
ams_status ams_front_conv_fwd(const float* x, const float* f, float* y, int Bt, int L, int W, int N, int hop, const float* amax_x,
                              const float* amax_f, float* amax_y, int lds_pad, void* ws, size_t ws_bytes, void* sk_scratch,
                              size_t sk_bytes, void* stream) {
    AMS_REQUIRE(x && f && y && Bt > 0 && L > 0 && W > 0 && N > 0 && hop > 0);
    const int T = (L + hop - 1) / hop;
    int pad_total = (T - 1) * hop + W - L;
    if (pad_total < 0) pad_total = 0;
    GemmArgs g{};
    g.A = x; g.B = f; g.C = y; g.bias = nullptr;
    g.M = Bt * T; g.N = N; g.K = W; g.lda = 0; g.ldb = N; g.ldc = N;
    g.fr_L = L; g.fr_T = T; g.fr_hop = hop; g.fr_pl = pad_total / 2; g.fr_W = W;
    g.a_vec = aligned16(x) && (L % 4 == 0) && (hop % 4 == 0) && (g.fr_pl % 4 == 0);
    g.b_vec = aligned16(f) && (N % 4 == 0);
    LaunchOpt o; o.lds_pad = lds_pad; o.amax_a = amax_x; o.amax_b = amax_f; o.sk_scratch = sk_scratch; o.sk_bytes = sk_bytes;
    if (use_x6() && g.a_vec && g.b_vec && tuning().x6cfg < 0) {
        // few-tile product (15360 x 256 at the benchmark shape): take the tile configuration the cost model likes better, whole tiles,
        // split-K or stream-K alike (128 x 128: 240 tiles, one round, nothing to reduce; 128 x 256: 120 tiles, 3 k-splits + a reduce launch)
        double best = 1e30;
        for (int c : {0, 3}) {
            const TilePlan tp = x6_plan(c);
            double t = 1e30;
            choose_splits(g.M, g.N, g.K, 1, tp, &t, (ws && !amax_y) ? 32 : 1);     // (a measured output is a whole output: no k-split slabs)
            if (sk_scratch && lds_pad <= 0 && tuning().sk != 0) {
                const SkPlan sp = sk_plan(ceil_div(g.M, tp.bm) * ceil_div(g.N, tp.bn), g.K, tp, c == 0 ? 2 : 1);
                if (sp.rounds >= 0 && sp.t < t) t = sp.t;
            }
            if (t < best) { best = t; o.force_cfg = c; }
        }
    }
    if (amax_y) o.amax_out = reinterpret_cast<unsigned*>(amax_y);
    return launch<A_FRAMES, B_ROW>(g, o, ws, ws_bytes, (hipStream_t)stream);
}
int ams_front_conv_fwd_measures_output(const float* x, const float* f, int L, int W, int N, int hop) {
    return (x && f && L > 0 && W > 0 && N > 0 && hop > 0 && front_conv_is_x6(x, f, L, W, N, hop)) ? 1 : 0;
}

// Generic framed product: out[(r,t), n] = sum_k xpad[r, t*hop + k - pad_left] * Bm[k, n]   (STFT as a DFT product)
ams_status ams_frames_matmul(const float* x, const float* Bm, float* out, int R, int L, int W, int N, int hop, int T, int pad_left,
                             void* stream) {
    AMS_REQUIRE(x && Bm && out && R > 0 && L > 0 && W > 0 && N > 0 && hop > 0 && T > 0 && pad_left >= 0);
    GemmArgs g{};
    g.A = x; g.B = Bm; g.C = out; g.bias = nullptr;
    g.M = R * T; g.N = N; g.K = W; g.lda = 0; g.ldb = N; g.ldc = N;
    g.fr_L = L; g.fr_T = T; g.fr_hop = hop; g.fr_pl = pad_left; g.fr_W = W;
    g.a_vec = aligned16(x) && (L % 4 == 0) && (hop % 4 == 0) && (pad_left % 4 == 0);
    g.b_vec = aligned16(Bm) && (N % 4 == 0);
    return launch<A_FRAMES, B_ROW>(g, LaunchOpt{}, nullptr, 0, (hipStream_t)stream);
}

// Filter gradient of a framed product with explicit geometry: dB[k,n] = sum_{r,t} xpad[r, t*hop + k - pad_left] * dy[(r,t), n]
size_t ams_frames_matmul_bwd_filter_workspace_bytes(int R, int W, int N, int T) { return ams_gemm_workspace_bytes(W, N, R * T, 1, 0); }

ams_status ams_frames_matmul_bwd_filter(const float* x, const float* dy, float* dB, int R, int L, int W, int N, int hop, int T,
                                        int pad_left, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(x && dy && dB && R > 0 && L > 0 && W > 0 && N > 0 && hop > 0 && T > 0 && pad_left >= 0);
    GemmArgs g{};
    g.A = x; g.B = dy; g.C = dB; g.bias = nullptr;
    g.M = W; g.N = N; g.K = R * T; g.lda = 0; g.ldb = N; g.ldc = N;
    g.fr_L = L; g.fr_T = T; g.fr_hop = hop; g.fr_pl = pad_left; g.fr_W = W;
    g.a_vec = aligned16(x) && (L % 4 == 0) && (hop % 4 == 0) && (pad_left % 4 == 0);
    g.b_vec = aligned16(dy) && (N % 4 == 0);
    return launch<A_FRAMES_T, B_ROW>(g, LaunchOpt{}, ws, ws_bytes, (hipStream_t)stream);
}

size_t ams_front_conv_bwd_filter_workspace_bytes(int Bt, int L, int W, int N, int hop) {
    const int T = (L + hop - 1) / hop;
    return ams_gemm_workspace_bytes(W, N, Bt * T, 1, 0);
}

// df[k,n] = sum_{b,t} xpad[b,t*hop+k-pl] * dy[b,t,n]   
ams_status ams_front_conv_bwd_filter(const float* x, const float* dy, float* df, int Bt, int L, int W, int N, int hop,
                                     void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(x && dy && df && Bt > 0 && L > 0 && W > 0 && N > 0 && hop > 0);
    const int T = (L + hop - 1) / hop;
    int pad_total = (T - 1) * hop + W - L;
    if (pad_total < 0) pad_total = 0;
    GemmArgs g{};
    g.A = x; g.B = dy; g.C = df; g.bias = nullptr;
    g.M = W; g.N = N; g.K = Bt * T; g.lda = 0; g.ldb = N; g.ldc = N;
    g.fr_L = L; g.fr_T = T; g.fr_hop = hop; g.fr_pl = pad_total / 2; g.fr_W = W;
    g.a_vec = aligned16(x) && (L % 4 == 0) && (hop % 4 == 0) && (g.fr_pl % 4 == 0);
    g.b_vec = aligned16(dy) && (N % 4 == 0);
    return launch<A_FRAMES_T, B_ROW>(g, LaunchOpt{}, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"

// ---- max-pool front (path B) --------------------------------------------------------------------------------------
namespace {

// Combine per-tile partial maxima into pooling windows: window t covers positions [t*hop, t*hop+P) of row b.
// argmax = l*N + n (TF-1.x GPU convention: no batch term; reference utils/ops.py:111-116 relies on it).
__global__ void maxpool_window_kernel(const float* __restrict__ pmax, const int32_t* __restrict__ pidx, float* __restrict__ y,
                                      long long* __restrict__ argmax, int Bt, int L, int N, int P, int hop, int T) {
    const long total = (long)Bt * T * N;
    const int tiles_per_row = L / BM;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long bt = i / N;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int j0 = (t * hop) / BM, j1 = (t * hop + P) / BM;
        float best = -3.4e38f;
        int brow = 0;
        for (int j = j0; j < j1; ++j) {
            const long k = ((long)b * tiles_per_row + j) * N + n;
            const float v = pmax[k];
            if (v > best) { best = v; brow = pidx[k]; }          // tiles scanned in increasing position: first max wins
        }
        y[i] = best;
        argmax[i] = (long long)(brow - b * L) * N + n;
    }
}

// Generic (any P / hop / L) fallback: one thread per (b,t,n); O(P*W) each -- used for small or unaligned shapes.
__global__ void maxpool_naive_kernel(const float* __restrict__ x, const float* __restrict__ f, float* __restrict__ y,
                                     long long* __restrict__ argmax, int Bt, int L, int W, int N, int P, int hop, int T, int pl) {
    const long total = (long)Bt * T * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long bt = i / N;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float best = -3.4e38f;
        int bl = 0;
        for (int l = t * hop; l < t * hop + P; ++l) {
            float s = 0.f;
            for (int k = 0; k < W; ++k) {
                const int p = l + k - pl;
                if (p >= 0 && p < L) s += x[(long)b * L + p] * f[(long)k * N + n];
            }
            if (s > best) { best = s; bl = l; }
        }
        y[i] = best;
        argmax[i] = (long long)bl * N + n;
    }
}

// ---- sparse (argmax-position) kernels of path B.  Positions are int32 sample indices (= TF's flattened int64 argmax / N,
// converted ONCE by argmax_pos_kernel: a 64-bit divide in these inner loops cost more than the arithmetic) and the synthesis
// filter is read TRANSPOSED, f2t [N, W], so consecutive taps are consecutive addresses (f2[k*N+n] put every lane on its own
// cache line).  Measured at cfg2 (B=64, S=2, L=20480, W=1024, N=256): 6.8 / 9.5 / 9.3 ms -> see profiles/.
__global__ void argmax_pos_kernel(const long long* __restrict__ argmax, int32_t* __restrict__ pos, long count, int N) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        pos[i] = (int32_t)(argmax[i] / N);
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[(long)c * rows + r] = tile[threadIdx.x][j];
    }
}

// df[k,n] = sum_{r,t} xpad[r, pos[r/rdiv,t,n] + k - pl] * v[r,t,n].  Block = 256 consecutive taps k of one filter n and one
// slice of rows (grid.z): pos / v are uniform per iteration (scalar loads), x is a coalesced 1 KB read.
__global__ __launch_bounds__(256) void gather_filter_grad_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                                 const int32_t* __restrict__ pos, float* __restrict__ part, int R,
                                                                 int L, int W, int N, int T, int pl, int rdiv, int rows_per_z) {
    const int n = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int r_lo = blockIdx.z * rows_per_z, r_hi = min(R, r_lo + rows_per_z);
    float s = 0.f;
    for (int r = r_lo; r < r_hi; ++r) {
        const float* xr = x + (long)r * L;
        const int32_t* pr = pos + ((long)(r / rdiv) * T) * N + n;
        const float* vr = v + ((long)r * T) * N + n;
        // eight (position, value) pairs and their eight window samples in flight per round: the loop was one dependent chain of
        // scalar load -> vector load -> FMA per iteration (~1.5 us each, 3.1 ms per launch at the path-B shape); same summation order
        for (int t = 0; t < T; t += 8) {
            int p[8];
            float val[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = min(t + u, T - 1);
                p[u] = pr[(long)tt * N] + k - pl;
                val[u] = vr[(long)tt * N];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = xr[min(max(p[u], 0), L - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t + u < T && p[u] >= 0 && p[u] < L) s += xv[u] * val[u];
        }
    }
    if (k < W) part[((long)blockIdx.z * W + k) * N + n] = s;
}
// LDS-staged form of the same sum (the default).  For one (row r, frame t) the windows of ALL filters start inside one pooling
// window, so they lie inside one short segment of the row: [min_n pos, max_n pos + W).  The kernel above re-reads that segment
// from L2 once per filter (4 KB x R*T*N = 16 GB per launch at the path-B shape, 2.2 ms); here a workgroup = 256 taps x 64 filters
// x a slice of the (r, t) pairs stages the part of the segment its taps need ONCE in LDS and every thread accumulates 4 taps x 16
// filters in registers from it (8-byte LDS reads; a second copy of the segment shifted by one sample serves the odd shifts).
constexpr int GF_TAPS = 256, GF_FILT = 64, GF_SEG = 1536;      // segment capacity: spread of the positions + 256 taps + 1
constexpr int GF_QB = 4;                                        // (row, frame) pairs per round: wave w prepares pair w, ONE barrier triple
                                                                // per four pairs (round 5: one pair per round was three barriers around 64
                                                                // FMAs per thread -- 0.86 ms per launch at the path-B shape, latency-bound)
__global__ __launch_bounds__(256) void gather_filter_grad_lds_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                                     const int32_t* __restrict__ pos, float* __restrict__ part, int R,
                                                                     int L, int W, int N, int T, int pl, int rdiv, long pairs_per_z) {
    __shared__ __attribute__((aligned(16))) float seg0[GF_QB][GF_SEG + 8], seg1[GF_QB][GF_SEG + 8];
    __shared__ float sv[GF_QB][GF_FILT];
    __shared__ int ss[GF_QB][GF_FILT], smm[GF_QB][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * GF_TAPS, n0 = blockIdx.y * GF_FILT;
    const long q_lo = (long)blockIdx.z * pairs_per_z, q_hi = min((long)R * T, q_lo + pairs_per_z);
    float acc[16][4];
#pragma unroll
    for (int a = 0; a < 16; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (long q0 = q_lo; q0 < q_hi; q0 += GF_QB) {
        {   // wave w: positions and values of pair q0 + w for the workgroup's 64 filters, the spread of the positions
            const long q = q0 + wave;
            const bool pair = q < q_hi;
            const int r = pair ? (int)(q / T) : 0, t = pair ? (int)(q - (long)r * T) : 0;
            const int n = n0 + lane;
            const bool live = pair && n < N;
            const int p = live ? pos[((long)(r / rdiv) * T + t) * N + n] : 0;
            sv[wave][lane] = live ? v[((long)r * T + t) * N + n] : 0.f;          // a missing pair contributes zeros
            int mn = live ? p : 0x7fffffff, mx = live ? p : -0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
            if (mn > mx) { mn = 0; mx = 0; }
            ss[wave][lane] = live ? p - mn : 0;
            if (lane == 0) { smm[wave][0] = mn; smm[wave][1] = mx; }
        }
        __syncthreads();
        bool staged[GF_QB];
#pragma unroll
        for (int j = 0; j < GF_QB; ++j) {
            const long q = min(q0 + j, q_hi - 1);
            const int r = (int)(q / T);
            const int mn = smm[j][0], span = smm[j][1] - mn;          // workgroup-uniform
            const float* xr = x + (long)r * L;
            staged[j] = span + GF_TAPS + 1 <= GF_SEG;
            if (staged[j]) {
                const int base = mn - pl + k0, len = span + GF_TAPS + 1;
                for (int i = tid; i < len; i += 256) {
                    const int pp = base + i;
                    const float val = (pp >= 0 && pp < L) ? xr[pp] : 0.f;
                    seg0[j][i] = val;
                    if (i > 0) seg1[j][i - 1] = val;                  // seg1[i] = seg0[i + 1]
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GF_QB; ++j) {
            if (staged[j]) {
#pragma unroll
                for (int nn = 0; nn < 16; ++nn) {
                    const int f = wave * 16 + nn;
                    const int sh = ss[j][f];
                    const float vv = sv[j][f];
                    // taps 2 lane, 2 lane + 1 and + 128: seg[sh + kk], seg[sh + kk + 1] as ONE 8-byte read (even index into seg0, or seg1 shifted)
                    const float* sb = (sh & 1) ? (seg1[j] + (sh - 1)) : (seg0[j] + sh);
                    const float2 a = *reinterpret_cast<const float2*>(sb + 2 * lane);
                    const float2 b = *reinterpret_cast<const float2*>(sb + 2 * lane + 128);
                    acc[nn][0] = fmaf(vv, a.x, acc[nn][0]);
                    acc[nn][1] = fmaf(vv, a.y, acc[nn][1]);
                    acc[nn][2] = fmaf(vv, b.x, acc[nn][2]);
                    acc[nn][3] = fmaf(vv, b.y, acc[nn][3]);
                }
            } else {                                              // positions spread wider than the staging buffer: direct reads
                const long q = min(q0 + j, q_hi - 1);
                const float* xr = x + (long)(q / T) * L;
                const int mn = smm[j][0];
#pragma unroll
                for (int nn = 0; nn < 16; ++nn) {
                    const int f = wave * 16 + nn;
                    const int p0 = mn + ss[j][f] - pl + k0;
                    const float vv = sv[j][f];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int pp = p0 + 2 * lane + (jj & 1) + 128 * (jj >> 1);
                        acc[nn][jj] = fmaf(vv, (pp >= 0 && pp < L) ? xr[pp] : 0.f, acc[nn][jj]);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 2 * lane + (j & 1) + 128 * (j >> 1);
        if (k >= W) continue;
#pragma unroll
        for (int nn = 0; nn < 16; ++nn) {
            const int n = n0 + wave * 16 + nn;
            if (n < N) part[((long)blockIdx.z * W + k) * N + n] = acc[nn][j];
        }
    }
}
__global__ void gather_filter_reduce_kernel(const float* __restrict__ part, float* __restrict__ df, long WN, int nz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WN) return;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += part[(long)z * WN + i];
    df[i] = s;
}

template <int SS>                      // SS rows (the S sources of one mixture share its positions) per workgroup: ONE filter load serves all
__global__ __launch_bounds__(256) void synth_unpool_kernel(const float* __restrict__ vals, const int32_t* __restrict__ pos,
                                                           const float* __restrict__ f2t, float* __restrict__ out, int R, int L, int W,
                                                           int N, int T, int P, int hop, int pl, int S) {
    const int r = blockIdx.y * SS;                                      // (SS > 1 only when S % SS == 0: rows r .. r + SS - 1 share r / S)
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int l_lo = blockIdx.x * blockDim.x, l_hi = min(L, l_lo + (int)blockDim.x) - 1;
    // windows whose positions [t*hop, t*hop+P) can reach any sample of this block: pos in (l+pl-W, l+pl]
    int t0 = (l_lo + pl - W + 1 - (P - 1));
    t0 = t0 <= 0 ? 0 : t0 / hop;
    int t1 = (l_hi + pl) / hop;
    if (t1 > T - 1) t1 = T - 1;
    float s[SS];
#pragma unroll
    for (int q = 0; q < SS; ++q) s[q] = 0.f;
    // the window's position and value are the same for every sample of the block: read through the CONSTANT address space they are
    // scalar loads (both tensors were written by earlier launches), and the vector memory pipe is left with the one load that differs
    // per lane -- as three vector loads per term the kernel was bound by issuing them (round 5: 1.71 -> 1.22 ms; then the filter
    // load shared by the sources of a mixture)
    typedef const __attribute__((address_space(4))) int32_t c_i32;
    typedef const __attribute__((address_space(4))) float c_f32;
    c_i32* am = (c_i32*)(pos + (long)(r / S) * T * N);
    c_f32* vr = (c_f32*)(vals + (long)r * T * N);
    const long rstride = (long)T * N;
    auto rounds = [&](auto EVEN) {                                      // eight windows per round (independent scalar loads)
        constexpr bool even = decltype(EVEN)::value;
        for (int t = t0; t <= t1; ++t)
            for (int n0 = 0; n0 < N; n0 += 8) {
                int pa[8];
                float va[SS][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long i = (long)t * N + (even ? n0 + j : min(n0 + j, N - 1));
                    pa[j] = am[i];
#pragma unroll
                    for (int q = 0; q < SS; ++q) va[q][j] = vr[i + q * rstride];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = l - pa[j] + pl;                       // uniform offset: consecutive l -> consecutive k
                    if ((even || n0 + j < N) && k >= 0 && k < W) {
                        const float fv = f2t[(long)(n0 + j) * W + k];
#pragma unroll
                        for (int q = 0; q < SS; ++q) s[q] += va[q][j] * fv;
                    }
                }
            }
    };
    if ((N & 7) == 0) rounds(std::true_type{}); else rounds(std::false_type{});
    if (l < L) {
#pragma unroll
        for (int q = 0; q < SS; ++q) out[(long)(r + q) * L + l] = s[q];
    }
}

// dvals[r,t,n] = sum_k dout_pad[r, pos + k] * f2[k,n]: one workgroup per (r, t), wave w takes filters n = w, w + 4, ...
// (round 5: a flat index over (r, t, n) per wave cost two 64-bit divisions per item and the lane sum went through six ds_bpermute --
// 1.10 ms at 128 x 80 x 256 items; the position is a scalar load, the lane sum a VALU tree valid in lane 0)
__global__ __launch_bounds__(256) void synth_unpool_bwd_vals_kernel(const float* __restrict__ dout, const int32_t* __restrict__ pos,
                                                                    const float* __restrict__ f2t, float* __restrict__ dvals, int R,
                                                                    int L, int W, int N, int T, int pl, int S) {
    typedef const __attribute__((address_space(4))) int32_t c_i32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt = blockIdx.x, r = rt / T, t = rt - r * T;
    c_i32* pr = (c_i32*)(pos + ((long)(r / S) * T + t) * N);
    const float* dr = dout + (long)r * L;
    float* out = dvals + (long)rt * N;
    // the N windows of a frame start within one pooling window of each other: their union [pmin - pl, pmax - pl + W) of the upstream row
    // is staged in LDS once (zeros outside the row) when it fits, and every filter reads its window from there
    constexpr int SPAN = 4096;
    __shared__ float seg[SPAN];
    __shared__ int mm[2][4];
    int lo = 0x7fffffff, hi = -0x7fffffff;
    for (int n = threadIdx.x; n < N; n += 256) { const int q = pos[((long)(r / S) * T + t) * N + n]; lo = min(lo, q); hi = max(hi, q); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (lane == 0) { mm[0][wave] = lo; mm[1][wave] = hi; }
    __syncthreads();
    const int pmin = min(min(mm[0][0], mm[0][1]), min(mm[0][2], mm[0][3])), pmax = max(max(mm[1][0], mm[1][1]), max(mm[1][2], mm[1][3]));
    const bool staged = pmax - pmin + W <= SPAN;
    if (staged) {
        for (int i = threadIdx.x; i < pmax - pmin + W; i += 256) {
            const int p = pmin - pl + i;
            seg[i] = (p >= 0 && p < L) ? dr[p] : 0.f;
        }
        __syncthreads();
    }
    for (int n = wave; n < N; n += 4) {
        const int p0 = pr[n] - pl;
        const float* fr = f2t + (long)n * W;
        float s = 0.f;
        if (staged) {
            const float* sg = seg + (p0 + pl - pmin);
#pragma unroll 4
            for (int k = lane; k < W; k += 64) s += sg[k] * fr[k];
        } else {
#pragma unroll 4
            for (int k = lane; k < W; k += 64) {
                const int p = p0 + k;
                const float dv = dr[min(max(p, 0), L - 1)], fv = fr[k];     // unconditional loads: four rounds in flight
                if (p >= 0 && p < L) s += dv * fv;
            }
        }
        s = wave_sum_lane0(s);
        if (lane == 0) out[n] = s;
    }
}

// xp[b, pl + l] = x[b, l], zero elsewhere (row pitch Lp): with the SAME padding materialised every frame of the stride-1
// product is a plain in-range window, so its operand fetch needs no per-element validity.
__global__ void pad_rows_kernel(const float* __restrict__ x, float* __restrict__ xp, int Bt, int L, int Lp, int pl) {
    const long total = (long)Bt * Lp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / Lp), p = (int)(i - (long)b * Lp) - pl;
        xp[i] = (p >= 0 && p < L) ? x[(long)b * L + p] : 0.f;
    }
}

inline int maxpool_padded_len(int L, int W) { return (L + W + 3 + 3) / 4 * 4; }

// AMS_MAXPOOL_PS=0 (read once): path B's product cuts its operands inside the kernel (gemm_x6_kernel<A_FRAMES, .., EPI_MAXPOOL>) also where
// the pre-split form applies (A/B runs; tests hold both)
inline bool maxpool_ps() {
    static const bool v = !(getenv("AMS_MAXPOOL_PS") && atoi(getenv("AMS_MAXPOOL_PS")) == 0);
    return v;
}

}  // namespace

extern "C" {

size_t ams_front_maxpool_workspace_bytes(int Bt, int L, int N) {
    return (size_t)Bt * (L / BM + 1) * N * (sizeof(float) + sizeof(int32_t));
}
// with room for the zero-padded copy of the signals (enables the branch-free 16-byte operand fetch of the stride-1 product)
size_t ams_front_maxpool_workspace_bytes_w(int Bt, int L, int N, int W) {
    const size_t base = (ams_front_maxpool_workspace_bytes(Bt, L, N) + 15) / 16 * 16;
    size_t n = base + (size_t)Bt * maxpool_padded_len(L, W) * sizeof(float);
    // the operand images of the pre-split form (csrc/gemm_ps.hip: eight shifted copies of the signals, the filter's image)
    if (maxpool_ps() && ams_detail::conv_maxpool_ps_applies(Bt, L, W, N)) n = (n + 255) / 256 * 256 + ams_detail::conv_maxpool_ps_bytes(Bt, L, W, N);
    return n;
}

// Path B front: y [Bt,T,N], argmax int64 [Bt,T,N], T = (L-P)/hop + 1   (reference models/adapt.py:115-117)
ams_status ams_front_maxpool_fwd(const float* x, const float* f, float* y, long long* argmax, int Bt, int L, int W, int N, int P, int hop,
                                 const float* amax_x, const float* amax_f, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(x && f && y && argmax && Bt > 0 && L >= P && W > 0 && N > 0 && P > 0 && hop > 0);
    hipStream_t st = (hipStream_t)stream;
    const float* const mp_aa = amax_x;                           // operand bounds (both set: fp16x3)
    const float* const mp_ab = amax_f;
    const int T = (L - P) / hop + 1;
    const int pl = (W - 1) / 2;                                  // stride-1 SAME: pad_total = W-1, left = floor
    const long total = (long)Bt * T * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (L % BM == 0 && P % BM == 0 && hop % BM == 0 && ws && ws_bytes >= ams_front_maxpool_workspace_bytes(Bt, L, N)) {
        GemmArgs g{};
        g.A = x; g.B = f; g.bias = nullptr;
        g.M = Bt * L; g.N = N; g.K = W; g.ldb = N; g.ldc = N;
        g.fr_L = L; g.fr_T = L; g.fr_hop = 1; g.fr_pl = pl; g.fr_W = W;
        g.a_vec = 0;                                             // hop 1: frame starts are not 16-byte aligned
        g.b_vec = aligned16(f) && (N % 4 == 0);
        g.splits = 1; g.k_per_split = ceil_div(W, BK) * BK;
        const int tiles_m = g.M / BM;
        g.C = (float*)ws;
        g.pidx = (int32_t*)((float*)ws + (size_t)tiles_m * N);
        dim3 grid(tiles_m * ceil_div(N, BN), 1);
        g.group_m = 1;
        const size_t base = (ams_front_maxpool_workspace_bytes(Bt, L, N) + 15) / 16 * 16;
        const int Lp = maxpool_padded_len(L, W);
        const size_t ps_off = (base + (size_t)Bt * Lp * sizeof(float) + 255) / 256 * 256;
        if (use_x6() && mp_aa && mp_ab && tuning().f16x3 && maxpool_ps() && aligned16(f) && ((uintptr_t)ws & 255) == 0 &&
            ams_detail::conv_maxpool_ps_applies(Bt, L, W, N) && ws_bytes >= ps_off + ams_detail::conv_maxpool_ps_bytes(Bt, L, W, N)) {
            // fp16x3 from operand images cut once per launch: the signals as eight shifted copies (consecutive stride-1 frames are the same
            // samples moved by one: csrc/gemm_ps.hip), the filter as a PS32 image; LDS-DMA main loop, the same fused max-pool partial
            const ams_status r = ams_detail::conv_maxpool_ps(x, f, g.C, g.pidx, Bt, L, W, N, pl, mp_aa, mp_ab, (char*)ws + ps_off, st);
            if (r != AMS_OK) return r;
        } else
        if (g.b_vec && W % 4 == 0 && ws_bytes >= base + (size_t)Bt * Lp * sizeof(float) && !tuning().novec) {
            // stride-1 frames are not 16-byte aligned and would straddle the zero padding: run the product on a padded copy,
            // where every 4-tap fetch is one unconditional (dword-aligned) 16-byte load
            float* xp = (float*)((char*)ws + base);
            int pb = (int)(((long)Bt * Lp + 255) / 256);
            if (pb > 4096) pb = 4096;
            hipLaunchKernelGGL(pad_rows_kernel, dim3(pb), dim3(256), 0, st, x, xp, Bt, L, Lp, pl);
            g.A = xp; g.fr_L = Lp; g.fr_pl = 0; g.a_vec = 1;
            if (use_x6()) {
                g.k_per_split = ceil_div(W, X6_BK) * X6_BK;
                if (mp_aa && mp_ab && tuning().f16x3) {
                    g.amax_a = mp_aa; g.amax_b = mp_ab;
                    // N a multiple of 256: the 128 x 256 tile (8 waves) -- every frame sample is split and staged once per 256 filters
                    // instead of once per 128, and that stage, not the MFMAs, is what this kernel spends its time on (AMS_MAXPOOL_CFG=0: 128 x 128)
                    static const bool wide = [] { const char* e = getenv("AMS_MAXPOOL_CFG"); return !(e && e[0] == '0'); }();
                    if (wide && N % 256 == 0)
                        hipLaunchKernelGGL((gemm_x6_kernel<A_FRAMES, B_ROW, 3, EPI_MAXPOOL, true, true>), dim3(tiles_m * (N / 256), 1), dim3(512), 0, st, g);
                    else
                    hipLaunchKernelGGL((gemm_x6_kernel<A_FRAMES, B_ROW, 0, EPI_MAXPOOL, true, true>), grid, dim3(256), 0, st, g);
                } else
                hipLaunchKernelGGL((gemm_x6_kernel<A_FRAMES, B_ROW, 0, EPI_MAXPOOL, true>), grid, dim3(256), 0, st, g);
            } else
            hipLaunchKernelGGL((gemm_f32_kernel<A_FRAMES, B_ROW, EPI_MAXPOOL, AMS_GEMM_BK, true>), grid, dim3(256), 0, st, g);
        } else
        hipLaunchKernelGGL((gemm_f32_kernel<A_FRAMES, B_ROW, EPI_MAXPOOL>), grid, dim3(256), 0, st, g);
        hipLaunchKernelGGL(maxpool_window_kernel, dim3(blocks), dim3(256), 0, st, (const float*)g.C, (const int32_t*)g.pidx, y, argmax, Bt,
                           L, N, P, hop, T);
    } else {
        hipLaunchKernelGGL(maxpool_naive_kernel, dim3(blocks), dim3(256), 0, st, x, f, y, argmax, Bt, L, W, N, P, hop, T, pl);
    }
    return ams_check_launch();
}

// int32 sample positions from TF's flattened int64 argmax (l*N + n)
ams_status ams_argmax_to_pos(const long long* argmax, int32_t* pos, long count, int N, void* stream) {
    AMS_REQUIRE(argmax && pos && count > 0 && N > 0);
    long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(argmax_pos_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, argmax, pos, count, N);
    return ams_check_launch();
}

// out [cols, rows] = in [rows, cols]^T
ams_status ams_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
    AMS_REQUIRE(in && out && rows > 0 && cols > 0);
    hipLaunchKernelGGL(transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(32, 8), 0, (hipStream_t)stream, in, out, rows,
                       cols);
    return ams_check_launch();
}

static int gather_nz(int R) { int nz = R / 16; if (nz < 1) nz = 1; if (nz > 16) nz = 16; return nz; }
// slices of the (r, t) pairs of the LDS-staged form: ~1024 workgroups in all
static int gather_nz_lds(int R, int W, int N, int T) {
    const long tiles = (long)ceil_div(W, GF_TAPS) * ceil_div(N, GF_FILT);
    long nz = 1024 / tiles;
    if (nz < 1) nz = 1;
    if (nz > 128) nz = 128;
    if (nz > (long)R * T) nz = (long)R * T;
    return (int)nz;
}
static bool gather_use_lds() { static const bool v = getenv("AMS_GATHER_LDS") == nullptr || atoi(getenv("AMS_GATHER_LDS")) != 0; return v; }
size_t ams_gather_filter_grad_workspace_bytes(int R, int W, int N) {
    // the caller does not pass T: size for the larger of the two forms (<= 128 slices)
    const size_t a = (size_t)gather_nz(R) * W * N * sizeof(float), b = (size_t)128 * W * N * sizeof(float);
    return gather_use_lds() ? (a > b ? a : b) : a;
}

// df[k,n] = sum_{r,t} xpad[r, pos[r/rdiv,t,n] + k - pl] * v[r,t,n]   (max-pool front: x = waveforms, v = dy, rdiv = 1;
// sparse synthesis: x = d out, v = pooled values, rdiv = S because the mixture's positions are tiled over speakers)
ams_status ams_gather_filter_grad(const float* x, const float* v, const int32_t* pos, float* df, int R, int L, int W, int N, int T,
                                  int rdiv, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(x && v && pos && df && ws && R > 0 && L > 0 && W > 0 && N > 0 && T > 0 && rdiv > 0);
    if (ws_bytes < ams_gather_filter_grad_workspace_bytes(R, W, N)) return AMS_E_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    if (gather_use_lds()) {
        const int nz = gather_nz_lds(R, W, N, T);
        const long ppz = ((long)R * T + nz - 1) / nz;
        hipLaunchKernelGGL(gather_filter_grad_lds_kernel, dim3(ceil_div(W, GF_TAPS), ceil_div(N, GF_FILT), nz), dim3(256), 0, st, x, v, pos,
                           (float*)ws, R, L, W, N, T, (W - 1) / 2, rdiv, ppz);
        hipLaunchKernelGGL(gather_filter_reduce_kernel, dim3(ceil_div((long)W * N, 256)), dim3(256), 0, st, (const float*)ws, df, (long)W * N, nz);
        return ams_check_launch();
    }
    const int nz = gather_nz(R), rpz = ceil_div(R, nz);
    hipLaunchKernelGGL(gather_filter_grad_kernel, dim3(ceil_div(W, 256), N, nz), dim3(256), 0, st, x, v, pos, (float*)ws, R, L, W, N, T,
                       (W - 1) / 2, rdiv, rpz);
    hipLaunchKernelGGL(gather_filter_reduce_kernel, dim3(ceil_div((long)W * N, 256)), dim3(256), 0, st, (const float*)ws, df, (long)W * N, nz);
    return ams_check_launch();
}

// Path B back: unpool with the mixture's positions (tiled S times) + stride-1 conv2d_transpose SAME, sparse form.  f2t [N, W].
ams_status ams_synth_unpool_fwd(const float* vals, const int32_t* pos, const float* f2t, float* out, int R, int L, int W, int N, int T,
                                int P, int hop, int S, void* stream) {
    AMS_REQUIRE(vals && pos && f2t && out && R > 0 && L > 0 && W > 0 && N > 0 && T > 0 && P > 0 && hop > 0 && S > 0);
    if (S % 2 == 0 && R % 2 == 0)
        hipLaunchKernelGGL(synth_unpool_kernel<2>, dim3(ceil_div(L, 256), R / 2), dim3(256), 0, (hipStream_t)stream, vals, pos, f2t, out, R, L, W,
                           N, T, P, hop, (W - 1) / 2, S);
    else
        hipLaunchKernelGGL(synth_unpool_kernel<1>, dim3(ceil_div(L, 256), R), dim3(256), 0, (hipStream_t)stream, vals, pos, f2t, out, R, L, W,
                           N, T, P, hop, (W - 1) / 2, S);
    return ams_check_launch();
}

ams_status ams_synth_unpool_bwd_vals(const float* dout, const int32_t* pos, const float* f2t, float* dvals, int R, int L, int W, int N,
                                     int T, int S, void* stream) {
    AMS_REQUIRE(dout && pos && f2t && dvals && R > 0 && L > 0 && W > 0 && N > 0 && T > 0 && S > 0);
    AMS_REQUIRE((long)R * T < (1L << 31));
    hipLaunchKernelGGL(synth_unpool_bwd_vals_kernel, dim3((unsigned)(R * T)), dim3(256), 0, (hipStream_t)stream, dout, pos, f2t, dvals, R,
                       L, W, N, T, (W - 1) / 2, S);
    return ams_check_launch();
}

}  // extern "C"

#if AMS_X6_STAMP
extern "C" int ams_dbg_x6_stamps(long long* host_out, int clear) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x6_stamp), sizeof(g_x6_stamp)) != hipSuccess) return -1;
    if (clear) { static long long z[1024 * 8 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_x6_stamp), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
