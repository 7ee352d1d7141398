// f32 products from PRE-SPLIT fp16 operand images: C[M,N] = A[M,K] . B[N,K]^T (+ bias), both operands handed over as "PS32" images.
//
// csrc/gemm.hip runs every dense product of the path as fp16x3 -- each f32 operand scaled by one power of two per tensor and cut into
// two fp16 terms, three fp16 MFMA products per f32 product -- and does the cutting INSIDE the product, once per element and workgroup,
// between staging registers and LDS.  tools/gemm_anatomy.py says what that costs: a k-tile of a K = 600 product takes 2.0 us of which the
// MFMAs are busy 0.72; fetch -> scale -> 2 x cvt_pk -> subtract -> ds_write and the MFMA phase of the 8 waves ADD.  Two of the three
// forward operands have a bound known before they are written (BLSTM outputs: |h| < 1; weights: the optimizer leaves max |p|), so their
// cut can be made ONCE per step by whoever writes them, and the product's main loop shrinks to
//        buffer_load ... lds  (LDS-DMA, no VGPRs, no VALU, no ds_write)  ->  ds_read_b128  ->  v_mfma_f32_32x32x16_f16.
//
// PS32 image of an operand X [R, K] (k contiguous), scale s = 2^(13 - floor(log2 bound)):  row r is Kp * 4 bytes (Kp = K rounded up
// to 32: the SAME bytes as the f32 row), k-tile t of it is one 128-byte line = 32 x fp16 hi (64 B) | 32 x fp16 lo (64 B) with
// hi = fp16(x s), lo = fp16(x s - hi); k >= K is zero in both.  One k-tile of 8 rows = 8 lines = ONE wave-wide 16-byte LDS-DMA.
//
// LDS: three stages of [128 + 256 rows][128 B]; row-major as the DMA writes them, the eight 16-byte pieces of a row XOR-ed with
// (row >> 1) & 7 ON THE SOURCE ADDRESS (a lane fetches piece slot ^ f(row) of its line: the line is still covered by 8 lanes) so that
// the 16 rows a ds_read_b128 lane group reads land in 16 different slots of the two bank rows.  One barrier per k-tile:
//        wait (this wave's pieces of tile t have landed; tile t+1's stay in flight) -> s_barrier -> issue tile t+2 -> 24 MFMAs on tile t
// and the stage tile t+2 lands in is the one tile t-1 was read from, which every wave left before that barrier.
//
// Tile 128 x 256, 8 waves of 64 x 64 (2 x 2 MFMA tiles, two accumulator sets: hi.hi | cross terms, as gemm.hip's uncapped variants),
// persistent walk over the tiles in the XCD-aware band order of gemm.hip; the next tile's first k-tile is in flight before the finished
// tile's stores.  Replaces: the same tf.matmul / conv1d call sites as ams_gemm_f32 (utils/ops.py:366-383, :501-503).
//
// Measured: 0.65-0.7 x the time of the in-product form, matrix-pipe utilisation 0.39 (K = 600 projection) / 0.54 (dense
// forward) by counter.  On some boards that duty makes the clock governor settle ~8 % lower for every kernel of the step, which costs more
// than these products save; the host side therefore lets a captured training step measure both forms and keep the faster one
// (models/network.py::_train_graphed).  -DAMS_PS_SLEEP=N (s_sleep N behind every group of four MFMAs) is the probe that showed the clock
// coming back as the duty goes down (profiles/r06_f_step_clock_and_forward_product_form.txt); it is not a tuning knob.
#include "common.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int PS_BK = 32, PS_BM = 128, PS_BN = 256, PS_NT = 512;
constexpr int PS_A_BYTES = PS_BM * 128, PS_B_BYTES = PS_BN * 128, PS_STAGE = PS_A_BYTES + PS_B_BYTES;     // 16 KB + 32 KB
constexpr int PS_STAGES = 3;
constexpr int PS_BIAS_OFF = PS_STAGES * PS_STAGE;                                                           // two 1 KB bias slices behind the stages
constexpr int PS_LDS = PS_BIAS_OFF + 2 * 1024;                                                              // 146 KB

__device__ __forceinline__ unsigned ps_pk_f16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void ps_split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = ps_pk_f16(a, b);
    const f16x2_t h = __builtin_bit_cast(f16x2_t, hi);
    lo = ps_pk_f16(a - (float)h[0], b - (float)h[1]);
}
// 2^(13 - floor(log2(amax))) for a finite positive amax; 1 for 0, denormals, Inf and NaN (the rule of csrc/gemm.hip: f16_scale)
__device__ __forceinline__ float ps_scale(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    const int se = 127 + 13 - (e - 127);
    return (se >= 1 && se <= 254) ? __uint_as_float((unsigned)se << 23) : 1.0f;
}

// ---- image writers ------------------------------------------------------------------------------------------------------------------
// x [R, K] row-major (row pitch ldx floats) -> image rows of `pitch` bytes.  One thread per (row, group of 8 k): two float4 in, 16 B of
// hi and 16 B of lo out; the 4 threads of a k-tile write its 64 + 64 bytes.
__global__ __launch_bounds__(256) void ps_pack_rows_kernel(const float* __restrict__ x, long ldx, unsigned char* __restrict__ img,
                                                           unsigned pitch, int R, int K, const float* __restrict__ amax) {
    const int groups = (int)(pitch / 32);                       // 8-k groups per image row (Kp / 8)
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)R * groups) return;
    const int r = (int)(id / groups), gq = (int)(id - (long)r * groups), k = gq * 8;
    const float s = ps_scale(amax[0]);
    float v[8];
    const float* row = x + (long)r * ldx;
    if (k + 8 <= K && ((((uintptr_t)(row + k)) & 15) == 0)) {
        const float4 a = *reinterpret_cast<const float4*>(row + k), b = *reinterpret_cast<const float4*>(row + k + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k + j < K) ? row[k + j] : 0.f;
    }
    uint4 hi, lo;
    ps_split2(v[0] * s, v[1] * s, hi.x, lo.x);
    ps_split2(v[2] * s, v[3] * s, hi.y, lo.y);
    ps_split2(v[4] * s, v[5] * s, hi.z, lo.z);
    ps_split2(v[6] * s, v[7] * s, hi.w, lo.w);
    unsigned char* p = img + (long)r * pitch + (k >> 5) * 128 + ((k & 31) >> 3) * 16;
    *reinterpret_cast<uint4*>(p) = hi;
    *reinterpret_cast<uint4*>(p + 64) = lo;
}

// w [K, N] row-major (row pitch ldw floats) -> image of w^T: image row n holds w[:, n] along k.  A workgroup turns 32 (k) x 64 (n) blocks
// round in LDS, PS_PC_KT k-tiles one after another (3040 workgroups of one block each took 18 us for the 600 x 10240 dense kernel):
// coalesced reads along n -- the next block's values are in flight while this one is cut --, then thread (n, group of 8 k) writes 16 bytes
// of hi and of lo: a wave writes 16 image rows x 64 contiguous bytes per plane.
constexpr int PS_PC_KT = 4;
__global__ __launch_bounds__(256) void ps_pack_cols_kernel(const float* __restrict__ w, long ldw, unsigned char* __restrict__ img,
                                                           unsigned pitch, int K, int N, const float* __restrict__ amax) {
    __shared__ float t[32][65];
    const int n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int nl = threadIdx.x >> 2, kg = threadIdx.x & 3;
    const float s = ps_scale(amax[0]);
    const int kt_end = min((int)((K + 31) / 32), (int)(blockIdx.y + 1) * PS_PC_KT);
    float v[8];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kt * 32 + ty * 8 + i, n = n0 + tx;
            v[i] = (k < K && n < N) ? w[(long)k * ldw + n] : 0.f;
        }
    };
    int kt = blockIdx.y * PS_PC_KT;
    if (kt < kt_end) fetch(kt);
    for (; kt < kt_end; ++kt) {
        __syncthreads();                            // the previous block has been read
#pragma unroll
        for (int i = 0; i < 8; ++i) t[ty * 8 + i][tx] = v[i];
        __syncthreads();
        if (kt + 1 < kt_end) fetch(kt + 1);
        if (n0 + nl < N) {
            uint4 hi, lo;
            ps_split2(t[kg * 8 + 0][nl] * s, t[kg * 8 + 1][nl] * s, hi.x, lo.x);
            ps_split2(t[kg * 8 + 2][nl] * s, t[kg * 8 + 3][nl] * s, hi.y, lo.y);
            ps_split2(t[kg * 8 + 4][nl] * s, t[kg * 8 + 5][nl] * s, hi.z, lo.z);
            ps_split2(t[kg * 8 + 6][nl] * s, t[kg * 8 + 7][nl] * s, hi.w, lo.w);
            unsigned char* p = img + (long)(n0 + nl) * pitch + kt * 128 + kg * 16;
            *reinterpret_cast<uint4*>(p) = hi;
            *reinterpret_cast<uint4*>(p + 64) = lo;
        }
    }
}

// Stride-1 frames (path B of the front end, models/adapt.py:115-117): row (s, p) of the product's A operand is the window
// xp[p .. p + W) of the zero-padded signal xp[i] = x[s, i - pl].  Consecutive rows are the same samples moved by ONE, so no image with
// 16-byte pieces serves them all -- but EIGHT do: copy c holds the pieces  piece t = xp[c + 8 t .. c + 8 t + 8)  (a plane of fp16 hi, a
// piece of fp16 lo beside it), and row p is pieces (p >> 3) + 0, 1, 2, ... of copy p & 7: 16-byte aligned, contiguous along k.  A k-tile of a
// row is then 4 consecutive (hi, lo) piece pairs -- the eight pieces of a PS32 line in another order, so the product
// kernel's LDS image, fragment reads and MFMA stream are those of gemm_ps_kernel; only the source address of a piece differs.
// Layout: img [8 copies][R signals][lpp pieces][hi | lo] of 16 bytes (a row's k-tile: 128 contiguous bytes h0 l0 h1 l1 h2 l2 h3 l3); 8 x the signals' bytes (132 MB at 192 x 20480): L2 / MALL traffic
// only, the samples a tile needs from all copies are 9 KB.  One thread per piece pair.
__global__ __launch_bounds__(256) void ps_pack_conv_kernel(const float* __restrict__ x, unsigned char* __restrict__ img, int R, int L, int pl,
                                                           unsigned lpp, const float* __restrict__ amax) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_copy = (long)R * lpp;
    if (id >= 8 * per_copy) return;
    const int c = (int)(id / per_copy);
    const long rem = id - (long)c * per_copy;
    const int sgn = (int)(rem / lpp), t = (int)(rem - (long)sgn * lpp);
    const float s = ps_scale(amax[0]);
    const float* row = x + (long)sgn * L;
    const int i0 = c + 8 * t - pl;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = i0 + j; v[j] = (i >= 0 && i < L) ? row[i] : 0.f; }
    uint4 hi, lo;
    ps_split2(v[0] * s, v[1] * s, hi.x, lo.x);
    ps_split2(v[2] * s, v[3] * s, hi.y, lo.y);
    ps_split2(v[4] * s, v[5] * s, hi.z, lo.z);
    ps_split2(v[6] * s, v[7] * s, hi.w, lo.w);
    unsigned char* p = img + (((long)c * R + sgn) * lpp + t) * 32;
    *reinterpret_cast<uint4*>(p) = hi;
    *reinterpret_cast<uint4*>(p + 16) = lo;
}

// ---- the product --------------------------------------------------------------------------------------------------------------------
struct PsArgs {
    const unsigned char* A; const unsigned char* B; float* C; const float* bias;
    const float* amax_a; const float* amax_b;
    int M, N, K;
    long ldc;
    unsigned pitch_a, pitch_b;
    int tiles_m, tiles_n, group_m;
    // CONV (stride-1 frames of R signals of L positions, csrc/gemm.hip: ams_front_maxpool_fwd): row m = (signal m / L, position m % L) of
    // the operand is the window xp[p .. p + K) of the zero-padded signal; A is the "shifted-copies" image of ps_pack_conv_kernel
    int conv_L, conv_R; unsigned conv_lpp; long a_bytes;
    int32_t* pidx;             // CONV: per (row tile, column) arg-max row, beside the maxima in C (the fused max-pool partial)
};

// position `item` of the flat order -> tile: XCD x (workgroups x, x + 8, ...) owns a contiguous run of the band order (bands of
// group_m tile rows, column-major inside a band), as csrc/gemm.hip's locate_tile
__device__ __forceinline__ void ps_locate(const PsArgs& g, int item, int& tile_m, int& tile_n) {
    const int items = g.tiles_m * g.tiles_n;
    const int q = items / 8, r = items % 8, xcd = item % 8, idx = item / 8;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = g.group_m > 0 ? g.group_m : 1;
    const int band = bid / (gm * g.tiles_n), within = bid - band * (gm * g.tiles_n);
    const int band_rows = min(gm, g.tiles_m - band * gm);
    tile_n = within / band_rows;
    tile_m = band * gm + (within - tile_n * band_rows);
}

#ifndef AMS_PS_SLEEP
#define AMS_PS_SLEEP 0
#endif
#ifndef AMS_PS_STAMP
#define AMS_PS_STAMP 0
#endif
#if AMS_PS_STAMP
__device__ long long g_ps_stamp[1024 * 8 * 8];
#define PS_STAMP(ph) do { if (threadIdx.x == 0 && wi < 8) g_ps_stamp[((int)blockIdx.x * 8 + wi) * 8 + (ph)] = wall_clock64(); } while (0)
#else
#define PS_STAMP(ph) do { } while (0)
#endif

template <bool CONV>
__global__ __launch_bounds__(PS_NT, 1) void gemm_ps_kernel(const PsArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char ps_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lk = lane >> 5;
    const int nk = (g.K + PS_BK - 1) / PS_BK;
    const int n_items = g.tiles_m * g.tiles_n, G = (int)gridDim.x;
    const int n_work = (n_items - (int)blockIdx.x + G - 1) / G;
    if (n_work <= 0) return;
    __builtin_amdgcn_s_setprio(2);

    // raw buffer descriptors {base lo, base hi (stride 0), num_records in bytes, flags} of the two images
    auto rsrc = [](const unsigned char* p, long bytes) {
        const unsigned long long a = (unsigned long long)(uintptr_t)p;
        i32x4_t r = {(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
        r.x = __builtin_amdgcn_readfirstlane(r.x); r.y = __builtin_amdgcn_readfirstlane(r.y);
        r.z = __builtin_amdgcn_readfirstlane(r.z); r.w = __builtin_amdgcn_readfirstlane(r.w);
        return r;
    };
    const i32x4_t rsA = rsrc(g.A, CONV ? g.a_bytes : (long)g.M * g.pitch_a), rsB = rsrc(g.B, (long)g.N * g.pitch_b);
    const i32x4_t rsBias = rsrc(reinterpret_cast<const unsigned char*>(g.bias), g.bias ? (long)g.N * 4 : 0);       // no bias: every load out of range = 0

    // DMA roles: one wave-wide 16-byte piece load = 8 rows x 128 B.  Wave w moves A rows 16 w .. 16 w + 15 (2 loads) and B rows
    // 32 w .. 32 w + 31 (4 loads) of every k-tile.  Lane: row (lane >> 3) of the load, LDS slot lane & 7, source piece slot ^ f(row).
    const int drow = lane >> 3, dslot = lane & 7;
    const int f_even = (drow >> 1) & 7, f_odd = (4 + (drow >> 1)) & 7;          // f(row) = (row >> 1) & 7 for rows 8 q + drow, q even / odd
    const unsigned pa_even = (unsigned)(dslot ^ f_even) * 16u, pa_odd = (unsigned)(dslot ^ f_odd) * 16u;
    unsigned voffA[2], voffB[4];

    // fragment addresses inside a stage (bytes): row * 128 + 16 * ((lk + 2 ks + 4 p) ^ f(row)), f(row) = (l31 >> 1) & 7 for every 32-row tile
    const int ff = (l31 >> 1) & 7;
    unsigned fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (CONV)                         // tile row 64 wm + 32 i + l31 lives in LDS row 64 wm + 8 (l31 & 7) + 4 i + (l31 >> 3), pieces XOR-ed with l31 & 7 (setup)
            fa[i] = (unsigned)((wm * 64 + 8 * (l31 & 7) + 4 * i + (l31 >> 3)) * 128 + ((lk ^ (l31 & 7)) * 16));
        else
            fa[i] = (unsigned)(((wm * 2 + i) * 32 + l31) * 128 + ((lk ^ ff) * 16));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = (unsigned)(PS_A_BYTES + ((wn * 2 + j) * 32 + l31) * 128 + ((lk ^ ff) * 16));

    const float sc_inv = (1.0f / ps_scale(g.amax_a[0])) * (1.0f / ps_scale(g.amax_b[0]));

    int tile_m, tile_n, m0, n0;
    auto setup = [&](int i) {
        ps_locate(g, (int)blockIdx.x + i * G, tile_m, tile_n);
        m0 = tile_m * PS_BM; n0 = tile_n * PS_BN;
        if constexpr (CONV) {
            // The eight LDS rows 8 A + d (d = 0 .. 7) one piece load fills hold the tile rows 64 h + 8 d + a (A = 8 h + a): positions
            // 8 apart, i.e. CONSECUTIVE pieces (p >> 3) of ONE copy (p & 7 = a; L % 128 == 0: a tile's rows are one signal's) -- the load
            // reads ~350 contiguous bytes instead of eight 128-byte runs in eight copies.  LDS piece slot ^ (A & 7) of the row's k-tile
            // (mfma fragments: fa) = pair (that & 3), half (that >> 2) of the copy.  Rows past M: out of range -> zeros
            const int sgn = m0 / g.conv_L, pb = m0 - sgn * g.conv_L;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int A_ = 2 * wave + q, a = A_ & 7, row = (A_ >> 3) * 64 + 8 * drow + a;
                const int p = pb + row;
                const unsigned pc = (unsigned)(dslot ^ a);
                const unsigned off = ((((unsigned)a * (unsigned)g.conv_R + (unsigned)sgn) * g.conv_lpp + (unsigned)(p >> 3) + (pc & 3u)) * 2u + (pc >> 2)) * 16u;
                voffA[q] = (m0 + row < g.M) ? off : 0xfffffff0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) voffA[q] = (unsigned)(m0 + (2 * wave + q) * 8 + drow) * g.pitch_a + (q ? pa_odd : pa_even);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) voffB[q] = (unsigned)(n0 + (4 * wave + q) * 8 + drow) * g.pitch_b + ((q & 1) ? pa_odd : pa_even);
    };
    // rows past M / N: the offset lies beyond num_records and the load returns zeros (raw buffer range check) -- nothing is clamped.
    // The six loads are ONE asm statement: hipcc, which orders every LDS read behind an LDS-DMA it knows of with vmcnt(0), does not see
    // them -- their completion is counted by hand (the s_waitcnt vmcnt(6) of the main loop).  M0 = LDS byte address of the piece row.
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)ps_smem;
    // one 8-row piece load: q = 0, 1 -> A rows 16 w + 8 q, q = 2 .. 5 -> B rows 32 w + 8 (q - 2)
    auto issue1 = [&](int kt, int stage, int q) {
        const bool isA = q < 2;
        const unsigned ld = lds0 + (unsigned)(stage * PS_STAGE + (isA ? (2 * wave + q) * 1024 : PS_A_BYTES + (4 * wave + q - 2) * 1024));
        const unsigned vo = isA ? voffA[q] : voffB[q - 2];
        const int ko = kt * 128;                                    // (CONV: a k-tile of a row is 4 (hi, lo) piece pairs: 128 bytes as well)
        unsigned keep;
        if (isA)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(ld), "v"(vo), "s"(rsA), "s"(ko) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(ld), "v"(vo), "s"(rsB), "s"(ko) : "memory");
    };
    auto issue = [&](int kt, int stage) {
#pragma unroll
        for (int q = 0; q < 6; ++q) issue1(kt, stage, q);
    };

    // the tile's 256 bias values travel the same way (wave 0, one load, slot `par` of two): older than the loads of k-tile 1, so the
    // main loop's first vmcnt(6) covers it; a plain global load here would be the one VMEM operation hipcc counts, and it then waits
    // vmcnt(0) in front of every LDS read of the loop
    auto issue_bias = [&](int par) {
        if (wave == 0) {
            const unsigned lb = lds0 + (unsigned)(PS_BIAS_OFF + par * 1024);
            const unsigned vo = (unsigned)(n0 * 4 + lane * 16);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lb), "v"(vo), "s"(rsBias) : "memory");
        }
    };

    f32x16 acc[2][2], accs[2][2];
    // One k-tile: 2 k-steps x 3 products x 2 x 2 MFMAs, smallest partial products first (lo.hi, hi.lo -> accs; hi.hi -> acc).  The six
    // piece loads of k-tile kt + 2 (`pre`) go out BETWEEN the groups of four MFMAs, and the fragments of k-step 1 are read behind the
    // first group: issued in one block behind the barrier they kept the matrix pipe idle for ~800 of a k-tile's ~1500 cycles
    // (all eight waves leave the barrier together, so nobody's MFMAs covered anybody's issue slots).
    // CONV: rows r + 32 of k-tile j ARE rows r of k-tile j + 1 (the same samples), so the fragments of a wave's second row tile become
    // those of its first row tile one k-tile later: carried in registers, only the second row tile is read from LDS (a quarter of the
    // kernel's LDS reads)
    f16x8_t carry[2][2];
    bool prime = true;
    auto frags = [&](const unsigned char* sb, int ks, f16x8_t (&a)[2][2], f16x8_t (&b)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (CONV && i == 0) a[i][p] = carry[ks][p];
                else a[i][p] = *reinterpret_cast<const f16x8_t*>(sb + (fa[i] ^ (unsigned)(ks * 32 + p * 64)));
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[j][p] = *reinterpret_cast<const f16x8_t*>(sb + (fb[j] ^ (unsigned)(ks * 32 + p * 64)));
    };
    auto four = [&](f32x16 (&c)[2][2], const f16x8_t (&a)[2][2], const f16x8_t (&b)[2][2], int pa, int pb) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][pa], b[j][pb], c[i][j], 0, 0, 0);
#if AMS_PS_SLEEP
        __builtin_amdgcn_s_sleep(AMS_PS_SLEEP);
#endif
    };
    auto mfma_tile = [&](int stage, auto PRE, int kt2, int stage2) {
        constexpr bool pre = decltype(PRE)::value;
        const unsigned char* const sb = ps_smem + stage * PS_STAGE;
        f16x8_t a0[2][2], b0[2][2], a1[2][2], b1[2][2];
        if constexpr (CONV) {
            if (prime) {                                // first k-tile of a tile: the first row tile's fragments come from LDS too
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < 2; ++p) carry[ks][p] = *reinterpret_cast<const f16x8_t*>(sb + (fa[0] ^ (unsigned)(ks * 32 + p * 64)));
                prime = false;
            }
        }
        frags(sb, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        four(accs, a0, b0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pre) issue1(kt2, stage2, 0);
        frags(sb, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        four(accs, a0, b0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pre) issue1(kt2, stage2, 1);
        __builtin_amdgcn_sched_barrier(0);
        four(acc, a0, b0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pre) issue1(kt2, stage2, 2);
        __builtin_amdgcn_sched_barrier(0);
        four(accs, a1, b1, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pre) issue1(kt2, stage2, 3);
        __builtin_amdgcn_sched_barrier(0);
        four(accs, a1, b1, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (pre) { issue1(kt2, stage2, 4); issue1(kt2, stage2, 5); }
        __builtin_amdgcn_sched_barrier(0);
        four(acc, a1, b1, 0, 0);
        if constexpr (CONV) {
#pragma unroll
            for (int p = 0; p < 2; ++p) { carry[0][p] = a0[1][p]; carry[1][p] = a1[1][p]; }
        }
    };

    setup(0);
    issue(0, 0);
    issue_bias(0);
    if (nk > 1) issue(1, 1);
    for (int wi = 0; wi < n_work; ++wi) {
        PS_STAMP(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accs[i][j][r] = 0.f; }
        prime = true;
        int st = 0;
        for (int kt = 0; kt < nk - 2; ++kt) {
            // this wave's six loads of tile kt are older than the six of tile kt + 1: vmcnt(6) retires them (and everything older: the
            // previous tile's stores); the barrier then says every wave's pieces of tile kt are in LDS and tile kt - 1 has been read
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mfma_tile(st, std::true_type{}, kt + 2, st == 0 ? 2 : st - 1);
            st = st == 2 ? 0 : st + 1;
        }
        if (nk >= 2) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mfma_tile(st, std::false_type{}, 0, 0);
            st = st == 2 ? 0 : st + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mfma_tile(st, std::false_type{}, 0, 0);
        PS_STAMP(1);
        // what the stores need, before the per-tile state moves on
        const int em0 = m0, en0 = n0;
        const bool more = wi + 1 < n_work;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // every wave has read its last fragments: all three stages are free
        if (more) {
            setup(wi + 1);
            issue(0, 0);                                // the next tile's first k-tile is in flight before this tile's stores
            issue_bias((wi + 1) & 1);
        }
        if constexpr (CONV) {
            // Fused max-pool partial (csrc/gemm.hip: maxpool_epilogue): the [R, L, N] conv output is never written; the tile emits, per
            // column, its maximum over the 128 rows and the row that holds it (first maximum wins ties)
            float* const sred = reinterpret_cast<float*>(ps_smem + PS_STAGE);          // [4 wn][2 j][32] values, then rows
            int* const srow = reinterpret_cast<int*>(sred + 256);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float best = -3.4e38f;
                int brow = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x16 c = (acc[i][j] + accs[i][j]) * sc_inv;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = em0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                        const float v = c[r];
                        if (row < g.M && (v > best || (v == best && row < brow))) { best = v; brow = row; }
                    }
                }
                const float ob = __shfl_xor(best, 32, 64);
                const int orow = __shfl_xor(brow, 32, 64);
                if (ob > best || (ob == best && orow < brow)) { best = ob; brow = orow; }
                if (wm == 1 && lk == 0) { sred[(wn * 2 + j) * 32 + l31] = best; srow[(wn * 2 + j) * 32 + l31] = brow; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (wm == 0 && lk == 0) {
                    const float ob2 = sred[(wn * 2 + j) * 32 + l31];
                    const int or2 = srow[(wn * 2 + j) * 32 + l31];
                    if (ob2 > best || (ob2 == best && or2 < brow)) { best = ob2; brow = or2; }
                    const int col = en0 + wn * 64 + j * 32 + l31;
                    if (col < g.N) { g.C[(long)(em0 / PS_BM) * g.N + col] = best; g.pidx[(long)(em0 / PS_BM) * g.N + col] = brow; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            PS_STAMP(2);
            if (more && nk > 1) issue(1, 1);            // (the last __syncthreads: the scratch has been read)
            continue;
        }
        // Epilogue (C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): each wave turns its
        // 64-column band round in a private 8 KB patch of stages 1..2, 32 rows at a time, and stores float4 rows (csrc/gemm.hip)
        float* const wl = reinterpret_cast<float*>(ps_smem + PS_STAGE) + wave * 2048;
        const int rr = lane >> 4, c4 = (lane & 15) * 4;
        const int col = en0 + wn * 64 + c4;
        const bool cok = col < g.N;
        const float4 bv = *reinterpret_cast<const float4*>(ps_smem + PS_BIAS_OFF + (wi & 1) * 1024 + (wn * 64 + c4) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = (acc[i][j] + accs[i][j]) * sc_inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * lk) * 64 + j * 32 + l31] = acc[i][j][r];
            }
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) {
                const int rl = p8 * 4 + rr;
                float4 v = *reinterpret_cast<const float4*>(wl + rl * 64 + c4);
                const int row = em0 + (wm * 2 + i) * 32 + rl;
                if (row < g.M && cok) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    *reinterpret_cast<float4*>(g.C + (long)row * g.ldc + col) = v;
                }
            }
        }
        PS_STAMP(2);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // the patches have been read: stage 1 may be filled
            if (nk > 1) issue(1, 1);
        }
    }
}

inline int ps_group_m(int tiles_m, int tiles_n) {
    // band height of the XCD-aware order: an XCD's 32 concurrent tiles should be about 8 tile rows x 4 tile columns
    int gm = 8;
    if (gm > tiles_m) gm = tiles_m;
    (void)tiles_n;
    return gm < 1 ? 1 : gm;
}

}  // namespace

extern "C" {

size_t ams_ps_image_pitch(int K) { return (size_t)((K + 31) / 32) * 128; }
size_t ams_ps_image_bytes(int rows, int K) { return (size_t)rows * ams_ps_image_pitch(K); }

ams_status ams_ps_pack_rows(const float* x, long ldx, void* img, int R, int K, const float* amax, void* stream) {
    AMS_REQUIRE(x && img && amax && R > 0 && K > 0 && ldx >= K && (((uintptr_t)img) & 15) == 0);
    const unsigned pitch = (unsigned)ams_ps_image_pitch(K);
    const long n = (long)R * (pitch / 32);
    hipLaunchKernelGGL(ps_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, (unsigned char*)img, pitch, R, K, amax);
    return ams_check_launch();
}

ams_status ams_ps_pack_cols(const float* w, long ldw, void* img, int K, int N, const float* amax, void* stream) {
    AMS_REQUIRE(w && img && amax && K > 0 && N > 0 && ldw >= N && (((uintptr_t)img) & 15) == 0);
    const unsigned pitch = (unsigned)ams_ps_image_pitch(K);
    hipLaunchKernelGGL(ps_pack_cols_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)(((K + 31) / 32 + PS_PC_KT - 1) / PS_PC_KT)), dim3(256), 0, (hipStream_t)stream, w, ldw,
                       (unsigned char*)img, pitch, K, N, amax);
    return ams_check_launch();
}

ams_status ams_gemm_ps(int M, int N, int K, const void* A_img, const void* B_img, float* C, long ldc, const float* bias, const float* amax_a,
                       const float* amax_b, void* stream) {
    AMS_REQUIRE(A_img && B_img && C && amax_a && amax_b && M > 0 && N > 0 && K > 0);
    AMS_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && ldc >= N && ((((uintptr_t)C) | ((uintptr_t)bias) | ((uintptr_t)A_img) | ((uintptr_t)B_img)) & 15) == 0);
    PsArgs g{};
    g.A = (const unsigned char*)A_img; g.B = (const unsigned char*)B_img; g.C = C; g.bias = bias; g.amax_a = amax_a; g.amax_b = amax_b;
    g.M = M; g.N = N; g.K = K; g.ldc = ldc;
    g.pitch_a = g.pitch_b = (unsigned)ams_ps_image_pitch(K);
    AMS_REQUIRE((long)M * g.pitch_a < (1L << 31) && (long)N * g.pitch_b < (1L << 31));          // 32-bit buffer offsets
    g.tiles_m = (M + PS_BM - 1) / PS_BM; g.tiles_n = (N + PS_BN - 1) / PS_BN;
    g.group_m = ps_group_m(g.tiles_m, g.tiles_n);
    static const int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
    static const bool raised = [] { return hipFuncSetAttribute((const void*)gemm_ps_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS) == hipSuccess; }();
    if (!raised) return AMS_E_LAUNCH_FAILED;
    const int items = g.tiles_m * g.tiles_n;
    int grid = items < cus ? items : cus;
    if (grid >= 8) grid -= grid % 8;                    // a workgroup's items stay on one XCD (ps_locate)
    hipLaunchKernelGGL(gemm_ps_kernel<false>, dim3((unsigned)grid), dim3(PS_NT), PS_LDS, (hipStream_t)stream, g);
    return ams_check_launch();
}

}  // extern "C"

// ---- path B of the front end on pre-split images (called by csrc/gemm.hip: ams_front_maxpool_fwd; not an entry point of its own) -----------
namespace ams_detail {

// pieces of 16 bytes per signal and plane of the shifted-copies image: the last row's last k-tile ends at piece (L - 1) / 8 + 4 nk - 1
static inline unsigned conv_lpp(int L, int W) { return (unsigned)((L - 1) / 8 + 4 * ((W + 31) / 32) + 1); }

__attribute__((visibility("hidden"))) size_t conv_maxpool_ps_bytes(int Bt, int L, int W, int N) {
    const size_t a = (size_t)8 * 2 * Bt * conv_lpp(L, W) * 16, b = ams_ps_image_bytes(N, W);
    return (a + 255) / 256 * 256 + (b + 255) / 256 * 256;
}
// the shapes the kernel takes: whole 128-row tiles inside one signal, 32-bit offsets into the images
__attribute__((visibility("hidden"))) bool conv_maxpool_ps_applies(int Bt, int L, int W, int N) {
    return L % PS_BM == 0 && N % 4 == 0 && (size_t)8 * 2 * Bt * conv_lpp(L, W) * 16 < ((size_t)1 << 31) - 64 && ams_ps_image_bytes(N, W) < ((size_t)1 << 31);
}

// tile maxima / rows of  x (*) f  (stride 1, SAME) into pmax / pidx [Bt L / 128, N]; img: conv_maxpool_ps_bytes of scratch, 256-byte aligned
__attribute__((visibility("hidden"))) ams_status conv_maxpool_ps(const float* x, const float* f, float* pmax, int32_t* pidx, int Bt, int L, int W,
                                                                 int N, int pl, const float* amax_x, const float* amax_f, void* img,
                                                                 hipStream_t st) {
    const unsigned lpp = conv_lpp(L, W);
    const size_t a_bytes = (size_t)8 * 2 * Bt * lpp * 16;
    unsigned char* const ia = (unsigned char*)img;
    unsigned char* const ib = ia + (a_bytes + 255) / 256 * 256;
    const long pieces = (long)8 * Bt * lpp;
    hipLaunchKernelGGL(ps_pack_conv_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, x, ia, Bt, L, pl, lpp, amax_x);
    ams_status r = ams_ps_pack_cols(f, N, ib, W, N, amax_f, st);
    if (r != AMS_OK) return r;
    PsArgs g{};
    g.A = ia; g.B = ib; g.C = pmax; g.pidx = pidx; g.bias = nullptr; g.amax_a = amax_x; g.amax_b = amax_f;
    g.M = Bt * L; g.N = N; g.K = W; g.ldc = N;
    g.pitch_a = 0; g.pitch_b = (unsigned)ams_ps_image_pitch(W);
    g.conv_L = L; g.conv_R = Bt; g.conv_lpp = lpp; g.a_bytes = (long)a_bytes;
    g.tiles_m = g.M / PS_BM; g.tiles_n = (N + PS_BN - 1) / PS_BN;
    g.group_m = ps_group_m(g.tiles_m, g.tiles_n);
    static const int cus = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
    static const bool raised = [] { return hipFuncSetAttribute((const void*)gemm_ps_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS) == hipSuccess; }();
    if (!raised) return AMS_E_LAUNCH_FAILED;
    const int items = g.tiles_m * g.tiles_n;
    int grid = items < cus ? items : cus;
    if (grid >= 8) grid -= grid % 8;
    hipLaunchKernelGGL(gemm_ps_kernel<true>, dim3((unsigned)grid), dim3(PS_NT), PS_LDS, st, g);
    return ams_check_launch();
}

}  // namespace ams_detail
