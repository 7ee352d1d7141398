// f32 products on the bf16 matrix pipe from PRE-SPLIT operands ("x3 images") -- gfx950 only.
//
// csrc/gemm.hip's bf16x6 kernel splits every f32 operand element into its three bf16 terms (hi + mid + lo == x exactly) inside the
// main loop: fetch to VGPRs, ~4.5 VALU per element, three ds_write per operand quarter, two barriers per k-tile -- and its MFMA phase
// and its fetch+split+write phase ADD instead of overlapping (profiles/r02_i_gemm_x6_anatomy.txt; the kernel sits at the board power
// limit).  Here the split is done ONCE by whoever produces a tensor (ams_x3_split, or the producing kernel's epilogue) into an "x3
// image" whose memory layout IS the LDS image the MFMA fragments are read from.  The main loop is then
//     buffer_load_dwordx4 ... lds  (LDS-DMA, no VGPR staging)  ->  ds_read_b128 / ds_read_b64_tr_b16  ->  v_mfma_f32_32x32x16_bf16
// with no VALU work, no ds_write, one barrier per k-tile and a prefetch distance of one k-tile in a second LDS buffer.
//
// x3 image of a logical row-major f32 matrix X[R][C] (R, C padded with ZEROS to multiples of X3_PAD = 256):
//   unit (rb = r / 8, cb = c / 16) at byte ((rb * CB + cb) * 768), CB = Cpad / 16;  inside the unit
//   [plane p: hi, mid, lo][kg = (c % 16) / 8][r % 8][c % 8] bf16  ->  byte p * 256 + kg * 128 + (r % 8) * 16 + (c % 8) * 2.
//   One 128-byte cache line = 8 rows x 8 consecutive columns of one plane; a 16-byte "slot" = 8 consecutive columns of one row.
// An operand can be consumed in two ROLES from the same image:
//   role K  (contraction index k = column c; m or n = row r):   A of  A.B  when A = X        ("A_ROW"),  B when B = X^T  ("B_COL")
//   role T  (contraction index k = row r;    m or n = column c): A when A = X^T              ("A_COL"),  B when B = X    ("B_ROW")
// Role K reads a slot as ONE ds_read_b128 (lane: row l & 31, k-group l >> 5).  Role T needs 8 consecutive ROWS of one column per
// lane: two ds_read_b64_tr_b16 (hardware 4 x 4 bf16 transpose inside 16-lane groups) on an LDS image that keeps the 128-byte lines
// whole, with the two 64-byte halves of a line swapped where bit 1 of its column group is set (conflict-free 32-lane groups).
// Either way every LDS-DMA wave instruction moves 8 whole cache lines.
//
// Arithmetic: identical to gemm_x6_kernel<SEP = true>: six of the nine exact partial products, smallest first, the five small
// ones in their own accumulator set (the bf16 MFMA's adder truncates toward -inf at a level set by the largest addend; separate
// accumulators keep the mean signed error at the native f32 MFMA kernel's level -- tests/test_gpu_gemm_x6.py).  EVERY configuration
// carries the second set, including the residency-capped one (4 waves, one per SIMD, <= 256 VGPRs: co-resides with a ring wave).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int X3_PAD = 256;          // rows and columns of an image are padded (zero-filled) to multiples of this
constexpr int X3_UNIT = 768;         // bytes of one 8 x 16 unit (3 planes)
constexpr int X3_BK = 32;            // k-tile
#ifndef AMS_X3_DBG
#define AMS_X3_DBG 0
#endif
#ifndef AMS_X3_LOOP
#define AMS_X3_LOOP 1                // main-loop structure: 0 plain, 1 prefetch interleaved with the MFMAs, 2 skewed across the barrier
#endif

enum { ROLE_K = 0, ROLE_T = 1 };

struct X3Args {
    const unsigned char* A; const unsigned char* B;   // x3 images
    int a_cb, b_cb;                  // units per image row (Cpad / 16)
    long a_bytes, b_bytes;           // image sizes (buffer descriptors)
    int a_m0, b_n0;                  // first m (n) of this product inside the image, multiple of 8
    int a_k0, b_k0;                  // first k inside the image, multiple of 32
    float* C; const float* bias;
    int M, N, K;
    long ldc;
    int accumulate;
    int splits, k_per_split;
    float* partial;
    int group_m;
    int hiprio;
    // batch: z-th product uses a_m0 + z * a_m_zs, b_n0 + z * b_n_zs, C + z * c_zs
    int nbatch, a_m_zs, b_n_zs;
    long c_zs;
};

template <int CFG> struct X3Cfg;
template <> struct X3Cfg<0> { static constexpr int BMX = 128, BNX = 128, WMC = 2, WNC = 2; };      // 4 waves of 64 x 64, 96 KB
template <> struct X3Cfg<1> { static constexpr int BMX = 128, BNX = 256, WMC = 2, WNC = 4; };      // 8 waves of 64 x 64, 144 KB
constexpr int x3_stage_bytes(int rows) { return 3 * (X3_BK / 8) * rows * 16; }                     // one operand, one k-tile
constexpr int x3_lds(int bm, int bn) { return 2 * (x3_stage_bytes(bm) + x3_stage_bytes(bn)); }

__device__ __forceinline__ void locate(const X3Args& g, int bm, int bn, int& zb, int& split, int& tile_m, int& tile_n) {
    const int tiles_m = (g.M + bm - 1) / bm, tiles_n = (g.N + bn - 1) / bn;
    const int ntiles = tiles_m * tiles_n;
    const int nz = g.nbatch > 1 ? g.nbatch : 1;
    const int items = ntiles * g.splits * nz;
    int item = blockIdx.x;
    const int q = items / 8, r = items % 8, xcd = item % 8, idx = item / 8;      // XCD x gets the x-th contiguous run of the order
    item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    zb = item / (ntiles * g.splits);
    item -= zb * (ntiles * g.splits);
    split = item / ntiles;
    const int bid = item - split * ntiles;
    const int GROUP_M = g.group_m > 0 ? g.group_m : 1;
    const int band = bid / (GROUP_M * tiles_n), within = bid - band * (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    tile_n = within / band_rows;
    tile_m = band * GROUP_M + (within - tile_n * band_rows);
}

// LDS-DMA of one operand tile (ROWS m/n x 32 k, three planes) from its x3 image.  NI = ROWS * 12 / 64 wave instructions of 1 KB,
// dealt round-robin to the NW waves; every instruction is 8 whole 128-byte lines.
//   role K: LDS slot (p, kgl, row)            = (p * 4 + kgl) * ROWS + row                     (kgl = k-group of 8 inside the tile)
//   role T: LDS slot (p, rbl, ckg, r8)        = (p * 4 + rbl) * ROWS + ((ckg * 8 + r8) ^ (((ckg >> 1) & 1) << 2))
template <int ROLE, int ROWS, int NW>
struct X3Loader {
    static constexpr int NI = ROWS * 12 / 64;
    static constexpr int PER = NI / NW;
    static_assert(NI % NW == 0, "LDS-DMA instructions must divide evenly over the waves");
    i32x4_t rsrc;                    // buffer descriptor words (wave-uniform: SGPRs)
    unsigned voff;                   // lane part of the source offset
    unsigned soff[PER];              // instruction part (k-tile 0)
    unsigned kstep;                  // bytes per k-tile
    unsigned ldsoff[PER];            // destination inside the operand's stage
    __device__ __forceinline__ void init(const unsigned char* img, long bytes, int cb_per_row, int m0, int k0, int wave, int lane) {
        const unsigned long long pa = (unsigned long long)img;
        rsrc = i32x4_t{(int)(unsigned)pa, (int)(unsigned)((pa >> 32) & 0xffffu), (int)bytes, 0x00020000};
        const unsigned rowunit = (unsigned)cb_per_row * X3_UNIT;       // bytes from one 8-row band to the next
        if (ROLE == ROLE_K) {
            voff = (unsigned)(lane >> 3) * rowunit + (unsigned)(lane & 7) * 16;
            kstep = 2 * X3_UNIT;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = wave + j * NW;                           // slots 64 i .. 64 i + 63
                const int pk = (i * 64) / ROWS, row0 = (i * 64) % ROWS;
                const int p = pk >> 2, kgl = pk & 3;
                soff[j] = (unsigned)((m0 + row0) >> 3) * rowunit + (unsigned)((k0 >> 4) + (kgl >> 1)) * X3_UNIT + p * 256 + (kgl & 1) * 128;
                ldsoff[j] = i * 1024;
            }
        } else {
            const int ckgl = lane >> 3;                                // line of this lane inside the instruction
            const int r8 = (lane & 7) ^ (((ckgl >> 1) & 1) << 2);      // source row of LDS position (lane & 7): halves swapped
            const int gl = ckgl + ((m0 >> 3) & 1);                     // m0 may start in the second column group of a unit
            voff = (unsigned)(gl >> 1) * X3_UNIT + (unsigned)(gl & 1) * 128 + (unsigned)r8 * 16;
            kstep = 4 * rowunit;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = wave + j * NW;
                const int pr = (i * 64) / ROWS, s0 = (i * 64) % ROWS;  // s0 = ckg0 * 8, ckg0 multiple of 8
                const int p = pr >> 2, rbl = pr & 3;
                soff[j] = (unsigned)((k0 >> 3) + rbl) * rowunit + (unsigned)((m0 + s0) >> 4) * X3_UNIT + p * 256;
                ldsoff[j] = i * 1024;
            }
        }
    }
    // LDS-DMA as inline asm: through the builtin hipcc treats the pending DMA as a possible alias of EVERY later ds_read and puts
    // `s_waitcnt vmcnt(0)` in front of the first fragment read of the tile being computed -- the prefetch of the NEXT tile would be
    // waited for the moment it is issued.  Hidden in asm the loads are ordered by hand: `s_waitcnt vmcnt(0)` + barrier at the top of
    // the k-loop (the only point where a stage changes hands).  M0 = LDS byte address of the wave's 1 KB piece.
    __device__ __forceinline__ void piece(int j, unsigned lds_stage, int kt) const {
        const unsigned so = soff[j] + (unsigned)kt * kstep, la = lds_stage + ldsoff[j];
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                     :: "v"(voff), "s"(rsrc), "s"(so), "s"(la) : "memory");
    }
    __device__ __forceinline__ void issue(unsigned lds_stage, int kt) const {
#pragma unroll
        for (int j = 0; j < PER; ++j) piece(j, lds_stage, kt);
    }
};

// MFMA operand fragment (8 bf16 along k) of tile row/column `rc` (0 .. ROWS-1), plane p, k-step ks, from a stage
template <int ROLE, int ROWS>
struct X3Frag {
    unsigned off;                    // byte offset inside a stage for (p = 0, ks = 0)
    __device__ __forceinline__ void init(int rc0, int lane) {          // rc0 = first row/column of this wave's 32-wide MFMA tile
        const int l31 = lane & 31, lk = lane >> 5;
        if (ROLE == ROLE_K) off = (unsigned)(lk * ROWS + rc0 + l31) * 16;
        else {
            const int j = (lane & 15) >> 2, q = lane & 3, g16 = (lane >> 4) & 1;
            const int ckg = (rc0 >> 3) + 2 * g16 + (q >> 1);
            const int slot = lk * ROWS + ((ckg * 8 + j) ^ (((ckg >> 1) & 1) << 2));
            off = (unsigned)slot * 16 + (q & 1) * 8;
        }
    }
    __device__ __forceinline__ bf16x8_t load(const unsigned char* stage, int p, int ks) const {
        const unsigned char* a = stage + off + (unsigned)((p * 4 + ks * 2) * ROWS) * 16;
        if (ROLE == ROLE_K) return *reinterpret_cast<const bf16x8_t*>(a);
        // rows r8 = 0..3 then 4..7 of the band: slot ^ 4 (the swizzle only flips bit 2 of r8, and j < 4) == + / - 64 bytes
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a + 64 - 2 * (off & 64)));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
};

template <int RA, int RB, int CFG>
__global__ __launch_bounds__(X3Cfg<CFG>::WMC * X3Cfg<CFG>::WNC * 64, CFG == 0 ? 1 : 2) void gemm_x3_kernel(X3Args g) {
    using C = X3Cfg<CFG>;
    constexpr int BMX = C::BMX, BNX = C::BNX, NW = C::WMC * C::WNC;
    constexpr int A_STAGE = x3_stage_bytes(BMX), B_STAGE = x3_stage_bytes(BNX), STAGE = A_STAGE + B_STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[x3_lds(BMX, BNX)];
    if (g.hiprio) __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WNC, wn = wave % C::WNC;
    const int l31 = lane & 31, lk = lane >> 5;

    int zb, split, tile_m, tile_n;
    locate(g, BMX, BNX, zb, split, tile_m, tile_n);
    const int m0 = tile_m * BMX, n0 = tile_n * BNX;
    const int k_begin = split * g.k_per_split;
    const int k_end = min(g.K, k_begin + g.k_per_split);
    const int nk = (k_end - k_begin + X3_BK - 1) / X3_BK;

    X3Loader<RA, BMX, NW> la;
    X3Loader<RB, BNX, NW> lb;
    la.init(g.A, g.a_bytes, g.a_cb, g.a_m0 + zb * g.a_m_zs + m0, g.a_k0 + k_begin, wave, lane);
    lb.init(g.B, g.b_bytes, g.b_cb, g.b_n0 + zb * g.b_n_zs + n0, g.b_k0 + k_begin, wave, lane);
    X3Frag<RA, BMX> fa[2];
    X3Frag<RB, BNX> fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i].init((wm * 2 + i) * 32, lane);
        fb[i].init((wn * 2 + i) * 32, lane);
    }

    f32x16 acc[2][2], accs[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accs[i][j][r] = 0.f; }

    const unsigned lds0 = (unsigned)(uintptr_t)smem;            // LDS byte address of the first stage (low half of the flat address)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};                   // lo.hi, hi.lo, mid.mid, mid.hi, hi.mid | hi.hi: smallest first
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int NPA = X3Loader<RA, BMX, NW>::PER, NP = NPA + X3Loader<RB, BNX, NW>::PER;      // LDS-DMA pieces per wave and k-tile
    // piece q of the next tile's prefetch (A pieces first); kn = its k-tile, clamped to the split's last one (a harmless re-fetch of
    // a tile nobody reads again -- keeps the instruction stream branch-free)
    auto dma = [&](int q, unsigned stage, int kn) {
        if (q < NPA) la.piece(q, stage, kn);
        else lb.piece(q - NPA, stage + A_STAGE, kn);
    };
    auto load_frags = [&](const unsigned char* st, int ks, bf16x8_t (&a)[2][3], bf16x8_t (&b)[2][3]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = fa[i].load(st, p, ks);
                b[i][p] = fb[i].load(st + A_STAGE, p, ks);
            }
    };
    // 24 MFMAs of one k-step; after MFMA n (0 .. 23) of this call, `between(n)` may slip one LDS-DMA piece into the stream
    auto mfma_step = [&](const bf16x8_t (&a)[2][3], const bf16x8_t (&b)[2][3], auto between) {
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (t < 5) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], accs[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
                    between(t * 4 + i * 2 + j);
                }
    };
    la.issue(lds0, 0);
    lb.issue(lds0 + A_STAGE, 0);
#if AMS_X3_LOOP == 0
    // plain: barrier, the whole prefetch of tile kt + 1 in one burst, then the two k-steps of tile kt
    // (AMS_X3_DBG, timing anatomy only, WRONG results: 1 = no LDS-DMA after tile 0, 2 = fragments read once and kept in registers,
    //  4 = no barrier in the loop)
    bf16x8_t a[2][3], b[2][3];
    for (int kt = 0; kt < nk; ++kt) {
        if (!(AMS_X3_DBG & 4) || kt == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile kt has landed ...
            __syncthreads();                                       // ... so has everyone's; and everyone is done reading tile kt - 1
        }
        const unsigned char* cur = smem + ((AMS_X3_DBG & 1) ? 0 : (kt & 1) * STAGE);
        if (kt + 1 < nk && !(AMS_X3_DBG & 1)) {
            const unsigned nxt = lds0 + ((kt + 1) & 1) * STAGE;
            la.issue(nxt, kt + 1);
            lb.issue(nxt + A_STAGE, kt + 1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (!(AMS_X3_DBG & 2) || kt == 0) load_frags(cur, ks, a, b);
            mfma_step(a, b, [](int) {});
        }
    }
#elif AMS_X3_LOOP == 1
    // the prefetch of tile kt + 1 dealt out between the MFMAs of tile kt: two pieces up front, the rest one every STRIDE MFMAs
    constexpr int UP = 2, STRIDE = 48 / (NP - UP);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned char* cur = smem + (kt & 1) * STAGE;
        const unsigned nxt = lds0 + ((kt + 1) & 1) * STAGE;
        const int kn = min(kt + 1, nk - 1);
#pragma unroll
        for (int q = 0; q < UP; ++q) dma(q, nxt, kn);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[2][3], b[2][3];
            load_frags(cur, ks, a, b);
            mfma_step(a, b, [&](int n) {
                const int m = ks * 24 + n + 1;
                if (m % STRIDE == 0 && UP + m / STRIDE - 1 < NP) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma(UP + m / STRIDE - 1, nxt, kn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the clamped re-fetch must not outlive the workgroup's LDS
#else
    // skewed: the MFMAs of a k-step run while the fragments of the NEXT k-step are being read, across the barrier -- the second
    // k-step of tile kt is multiplied after the barrier that hands its LDS stage to tile kt + 2.  Two fragment register sets.
    constexpr int STRIDE = 24 / NP > 0 ? 24 / NP : 1;
    bf16x8_t a0[2][3], b0[2][3], a1[2][3], b1[2][3];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, 0, a0, b0);
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* cur = smem + (kt & 1) * STAGE;
        load_frags(cur, 1, a1, b1);
        if (kt == 0 && nk > 1) {                                   // tile 1 goes out as soon as the pipeline is primed
            la.issue(lds0 + STAGE, 1);
            lb.issue(lds0 + STAGE + A_STAGE, 1);
        }
        mfma_step(a0, b0, [](int) {});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // tile kt + 1 has landed (issued one iteration ago)
        __syncthreads();                                           // (lgkmcnt(0) inside: a1 / b1 are in registers) stage kt & 1 is free
        const unsigned char* nx = smem + ((kt + 1) & 1) * STAGE;
        if (kt + 1 < nk) load_frags(nx, 0, a0, b0);
        const unsigned st2 = lds0 + (kt & 1) * STAGE;
        const int kn = min(kt + 2, nk - 1);
        const bool more = kt + 2 < nk;
        mfma_step(a1, b1, [&](int n) {
            if (more && (n + 1) % STRIDE == 0 && (n + 1) / STRIDE - 1 < NP) {
                __builtin_amdgcn_sched_barrier(0);
                dma((n + 1) / STRIDE - 1, st2, kn);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
#endif

    // epilogue: C/D layout of a 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float* out = g.splits > 1 ? g.partial + ((long)zb * g.splits + split) * g.M * g.N : g.C + (long)zb * g.c_zs;
    const long ldo = g.splits > 1 ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wn * 2 + j) * 32 + l31;
            if (col >= g.N) continue;
            const float bv = (g.splits == 1 && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < g.M) {
                    float v = acc[i][j][r] + accs[i][j][r] + bv;
                    float* p = out + (long)row * ldo + col;
                    if (g.splits == 1 && g.accumulate) v += *p;
                    *p = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void x3_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C, const float* __restrict__ bias,
                                                        int M, int N, long ldc, int splits, int accumulate, long c_zs) {
    const long total = (long)M * N;
    partial += (long)blockIdx.y * splits * total;
    C += (long)blockIdx.y * c_zs;
    const bool v4 = (N % 4 == 0) && (ldc % 4 == 0) && ((((uintptr_t)C | (uintptr_t)partial | (uintptr_t)bias) & 15) == 0);
    if (v4) {
        const int n4 = N >> 2;
        const long total4 = total >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
            const int m = (int)(i / n4), c4 = (int)(i - (long)m * n4);
            const float4* src = reinterpret_cast<const float4*>(partial) + i;
            float4 s = src[0];
            for (int k = 1; k < splits; ++k) { const float4 v = src[(long)k * total4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            if (bias) { const float4 bv = reinterpret_cast<const float4*>(bias)[c4]; s.x += bv.x; s.y += bv.y; s.z += bv.z; s.w += bv.w; }
            float4* p = reinterpret_cast<float4*>(C + (long)m * ldc) + c4;
            if (accumulate) { const float4 o = *p; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            *p = s;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int m = (int)(i / N), n = (int)(i - (long)m * N);
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += partial[(long)k * total + i];
            if (bias) s += bias[n];
            float* p = C + (long)m * ldc + n;
            if (accumulate) s += *p;
            *p = s;
        }
    }
}

// ---- f32 -> x3 image -------------------------------------------------------------------------------------------------------------
// out[c] (+)= sum over the nrows partial rows; 64 columns x 4 row groups per block, the groups combined in a fixed order
__global__ __launch_bounds__(256) void x3_csum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nrows,
                                                             int accumulate) {
    __shared__ float sm[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int k = g; k < nrows; k += 4) s += part[(long)k * C + c];
    sm[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        const float t = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32 (round to nearest even): a -> bits 0..15
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk_bf16(a, b);
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = pk_bf16(sa, sb);
}
// One thread = 8 consecutive columns of one row (two 16-byte loads) -> three 16-byte stores.  A wave covers 8 rows x 64 columns:
// 256-byte row segments in, whole 128-byte lines out.  Rows >= R / columns >= C (up to the padded sizes covered by the grid) are
// written as zeros.  `rshift`: image row r holds source row r + rshift_dir(c) with rows outside a T-long sequence zeroed -- used for
// the recurrent-kernel gradients, whose A operand is h[b, t -/+ 1] (see ams_x3_split_shifted).
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ X, long ld, int R, int C, unsigned char* __restrict__ img,
                                                       int CB, int rows_out, int cols_out, int T, int half, int half_pad,
                                                       float* __restrict__ csum_part) {
    __shared__ float cs[4][64];
    const int tid = threadIdx.x;
    const int r8 = tid & 7, cg = (tid >> 3) & 7, rsub = tid >> 6;
    const int r = (blockIdx.y * 4 + rsub) * 8 + r8;
    const int c = (blockIdx.x * 8 + cg) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (T == 0) {
        if (r < R) {
            const float* src = X + (long)r * ld + c;
            if (c + 7 < C && ((ld & 3) == 0) && ((((uintptr_t)X) & 15) == 0) && ((c & 3) == 0)) {
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (c + j < C) v[j] = src[j];
            }
        }
    } else if (r < R) {
        // shifted image of out[B*T, 2 * half]: image columns [0, half) = source columns [0, half) of row t - 1 (zero at t = 0),
        // image columns [half_pad, half_pad + half) = source columns [half, 2 half) of row t + 1 (zero at t = T - 1)
        const int t = r % T;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cc = c + j;
            if (cc < half) { if (t > 0) v[j] = X[(long)(r - 1) * ld + cc]; }
            else if (cc >= half_pad && cc < half_pad + half) { if (t < T - 1) v[j] = X[(long)(r + 1) * ld + half + (cc - half_pad)]; }
        }
    }
    if (csum_part) {
        // column sums of the 32 rows of this block, fixed order: 8 rows of a band by lane pairing, then the 4 bands through LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j];
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            if (r8 == 0) cs[rsub][cg * 8 + j] = t;
        }
        __syncthreads();
        if (tid < 64) {
            const int cc = blockIdx.x * 64 + tid;
            if (cc < C) csum_part[(long)blockIdx.y * C + cc] = (cs[0][tid] + cs[1][tid]) + (cs[2][tid] + cs[3][tid]);
        }
    }
    uint4 hi, mid, lo;
    split3(v[0], v[1], hi.x, mid.x, lo.x);
    split3(v[2], v[3], hi.y, mid.y, lo.y);
    split3(v[4], v[5], hi.z, mid.z, lo.z);
    split3(v[6], v[7], hi.w, mid.w, lo.w);
    if (r >= rows_out || c >= cols_out) return;
    unsigned char* u = img + ((long)(r >> 3) * CB + (c >> 4)) * X3_UNIT + ((c >> 3) & 1) * 128 + r8 * 16;
    *reinterpret_cast<uint4*>(u) = hi;
    *reinterpret_cast<uint4*>(u + 256) = mid;
    *reinterpret_cast<uint4*>(u + 512) = lo;
}

inline int pad_to(int v, int m) { return (v + m - 1) / m * m; }

struct X3Tuning { int cfg, splits; };
inline const X3Tuning& x3_tuning() {
    static const X3Tuning t = [] {
        X3Tuning v{-1, 0};
        if (const char* f = getenv("AMS_X3_CFG")) v.cfg = atoi(f);
        if (const char* f = getenv("AMS_X3_SPLITS")) v.splits = atoi(f);
        return v;
    }();
    return t;
}

// tile configuration: 128 x 256 where it wastes < 12 % of the columns it covers and the launch is not residency-capped
inline int x3_choose_cfg(int M, int N, bool capped) {
    if (x3_tuning().cfg == 0 || x3_tuning().cfg == 1) return x3_tuning().cfg;
    if (capped) return 0;
    const bool n256 = (double)ceil_div(N, 256) * 256 <= 1.12 * N;
    const long t1 = (long)ceil_div(M, 128) * ceil_div(N, 256), t0 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    if (!n256) return 0;
    // rounds of 256 CUs: a 128 x 256 tile costs ~2x a 128 x 128 one and both run one workgroup per CU
    const long r1 = (t1 + 255) / 256 * 2, r0 = (t0 + 255) / 256;
    return r1 <= r0 ? 1 : 0;
}
// split-K: microseconds per 32 k of one workgroup (first estimates from the 4096^3 rate), + 6 us fixed, + slab round trip
inline int x3_choose_splits(int M, int N, int K, int nbatch, int cfg) {
    const int bm = 128, bn = cfg == 1 ? 256 : 128;
    const double us32 = cfg == 1 ? 0.70 : 0.36;
    const long tiles = (long)ceil_div(M, bm) * ceil_div(N, bn) * nbatch;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 32; ++s) {
        if (s > 1 && K / s < 256) break;
        const int kps = ceil_div(ceil_div(K, s), X3_BK) * X3_BK;
        const int s2 = ceil_div(K, kps);
        const long n = (tiles * s2 + 255) / 256;
        double t = n * ((kps / 32.0) * us32 + 6.0);
        if (s2 > 1) t += (double)(s2 + 1) * M * N * nbatch * 4.0 / 3.0e6;
        if (t < best_t - 1e-9) { best_t = t; best = s2; }
    }
    // accumulation chains of at most 1280 k per slab: the bf16 MFMA rounds once per 16 k at the accumulator's magnitude, so the rms
    // error of a same-sign chain grows like its length^1.5 -- 5120 long 1.2-1.4x the native f32 MFMA kernel's, 2560 long still 1.4x
    // on the narrow products, 1280 long below it (tests/test_gpu_gemm_x3.py); the slabs are added in f32 in a fixed order
    while (ceil_div(K, best) > 1280 && best < 32) ++best;
    return best;
}

thread_local int t_x3_capped = 0;

template <int RA, int RB>
ams_status x3_launch(X3Args& g, void* ws, size_t ws_bytes, hipStream_t st) {
    const int nb = g.nbatch > 1 ? g.nbatch : 1;
    const int cfg = x3_choose_cfg(g.M, g.N, t_x3_capped != 0);
    const int bm = 128, bn = cfg == 1 ? 256 : 128;
    const int tiles_m = ceil_div(g.M, bm), tiles_n = ceil_div(g.N, bn);
    int splits = 1;
    if (ws) {
        splits = x3_choose_splits(g.M, g.N, g.K, nb, cfg);
        if (x3_tuning().splits > 0) splits = x3_tuning().splits;
        while (splits > 1 && (size_t)nb * splits * g.M * g.N * sizeof(float) > ws_bytes) --splits;
    }
    int kps = ceil_div(ceil_div(g.K, splits), X3_BK) * X3_BK;
    splits = ceil_div(g.K, kps);
    g.splits = splits; g.k_per_split = kps; g.partial = (float*)ws;
    {   // band height of the XCD-aware tile order (same rule as csrc/gemm.hip::choose_group_m)
        int c = ceil_div((long)tiles_m * tiles_n, 8);
        if (c > 64) c = 64;
        int gm = (int)(sqrt((double)c) + 0.5);
        if (gm < 1) gm = 1;
        if (gm > tiles_m) gm = tiles_m;
        if (ceil_div(c, gm) > tiles_n) gm = ceil_div(c, tiles_n);
        if (gm > tiles_m) gm = tiles_m;
        g.group_m = gm;
    }
    g.hiprio = t_x3_capped ? 0 : 1;
    const unsigned grid = (unsigned)((long)tiles_m * tiles_n * splits * nb);
    if (cfg == 1) hipLaunchKernelGGL((gemm_x3_kernel<RA, RB, 1>), dim3(grid), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_x3_kernel<RA, RB, 0>), dim3(grid), dim3(256), 0, st, g);
    ams_status s = ams_check_launch();
    if (s != AMS_OK) return s;
    if (splits > 1) {
        long blocks = ((long)g.M * g.N / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(x3_reduce_kernel, dim3((unsigned)blocks, nb), dim3(256), 0, st, (const float*)g.partial, g.C, g.bias, g.M, g.N, g.ldc,
                           splits, g.accumulate, g.c_zs);
        s = ams_check_launch();
    }
    return s;
}

}  // namespace

extern "C" {

// Size in bytes of the x3 image of a logical [R, C] f32 matrix (both padded to multiples of 256, zero-filled by the producer).
size_t ams_x3_image_bytes(int R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return (size_t)(pad_to(R, X3_PAD) / 8) * (pad_to(C, X3_PAD) / 16) * X3_UNIT;
}

// img = x3 image of X[R, C] (row stride ld floats).  Writes the WHOLE padded image (zeros outside R x C).
ams_status ams_x3_split(const float* X, long ld, int R, int C, void* img, void* stream) {
    AMS_REQUIRE(X && img && R > 0 && C > 0 && ld >= C);
    const int Rp = pad_to(R, X3_PAD), Cp = pad_to(C, X3_PAD);
    hipLaunchKernelGGL(x3_split_kernel, dim3(Cp / 64, Rp / 32), dim3(256), 0, (hipStream_t)stream, X, ld, R, C, (unsigned char*)img, Cp / 16, Rp, Cp,
                       0, 0, 0, (float*)nullptr);
    return ams_check_launch();
}

// The same, and csum[C] (+)= column sums of X in the same pass (the bias gradient of a width-1 Conv1D next to its weight gradient,
// utils/ops.py:501-503 under tf.gradients).  ws: ams_x3_split_colsum_workspace_bytes(R, C).
size_t ams_x3_split_colsum_workspace_bytes(int R, int C) { return R > 0 && C > 0 ? (size_t)(pad_to(R, X3_PAD) / 32) * C * sizeof(float) : 0; }
ams_status ams_x3_split_colsum(const float* X, long ld, int R, int C, void* img, float* csum, int accumulate, void* ws, size_t ws_bytes,
                               void* stream) {
    AMS_REQUIRE(X && img && csum && ws && R > 0 && C > 0 && ld >= C);
    if (ws_bytes < ams_x3_split_colsum_workspace_bytes(R, C)) return AMS_E_WORKSPACE_TOO_SMALL;
    const int Rp = pad_to(R, X3_PAD), Cp = pad_to(C, X3_PAD);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(x3_split_kernel, dim3(Cp / 64, Rp / 32), dim3(256), 0, st, X, ld, R, C, (unsigned char*)img, Cp / 16, Rp, Cp, 0, 0, 0,
                       (float*)ws);
    hipLaunchKernelGGL(x3_csum_finish_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, (const float*)ws, csum, C, Rp / 32, accumulate);
    return ams_check_launch();
}

// Time-shifted image of a BLSTM layer output out[B*T, 2H] for the recurrent-kernel gradients dU = h_prev^T . dZ (reference
// utils/ops.py:358-383 under tf.gradients): image row (b, t), columns [0, H) = out[b, t-1, 0:H] (zero at t = 0), columns
// [Hp, Hp + H) = out[b, t+1, H:2H] (zero at t = T-1), Hp = H rounded up to 8; logical size [B*T, 2 * Hp].
ams_status ams_x3_split_shifted(const float* out, long ld, int BT, int T, int H, void* img, void* stream) {
    AMS_REQUIRE(out && img && BT > 0 && T > 0 && H > 0 && BT % T == 0 && ld >= 2 * H);
    const int Hp = pad_to(H, 8);
    const int Rp = pad_to(BT, X3_PAD), Cp = pad_to(2 * Hp, X3_PAD);
    hipLaunchKernelGGL(x3_split_kernel, dim3(Cp / 64, Rp / 32), dim3(256), 0, (hipStream_t)stream, out, ld, BT, 2 * H, (unsigned char*)img, Cp / 16, Rp,
                       Cp, T, H, Hp, (float*)nullptr);
    return ams_check_launch();
}

void ams_x3_set_capped(int on) { t_x3_capped = on ? 1 : 0; }

size_t ams_gemm_x3_workspace_bytes(int M, int N, int K, int nbatch) {
    if (M <= 0 || N <= 0 || K <= 0 || nbatch <= 0) return 0;
    int splits = x3_choose_splits(M, N, K, nbatch, x3_choose_cfg(M, N, t_x3_capped != 0));
    if (x3_tuning().splits > 0) splits = x3_tuning().splits;
    if (splits <= 1) return 0;
    return (size_t)nbatch * splits * M * N * sizeof(float);
}

// C[M, N] (+)= op(A) . op(B) (+ bias) from x3 images.  A image: logical [a_R, a_C]; roleA 0: A[m, k] = X[a_r0 + m, a_c0 + k]
// (K role), 1: A[m, k] = X[a_r0 + k, a_c0 + m] (T role); same for B with n in place of m.  Offsets along the m / n axis must be
// multiples of 8, along the k axis multiples of 32; the k range [k0, k0 + K) padded to 32 must hold zeros beyond K in at least one
// operand (images are zero-padded, so K <= the logical extent is enough).  nbatch products: z-th uses m offset + z * a_m_zs, n
// offset + z * b_n_zs, C + z * c_zs.
ams_status ams_gemm_x3(int roleA, int roleB, int M, int N, int K, const void* A, int a_R, int a_C, int a_r0, int a_c0, const void* B,
                       int b_R, int b_C, int b_r0, int b_c0, float* C, long ldc, long c_zs, const float* bias, int accumulate, int nbatch,
                       int a_m_zs, int b_n_zs, void* ws, size_t ws_bytes, void* stream) {
    AMS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && nbatch >= 1 && (roleA == 0 || roleA == 1) && (roleB == 0 || roleB == 1));
    X3Args g{};
    g.A = (const unsigned char*)A; g.B = (const unsigned char*)B;
    g.a_cb = pad_to(a_C, X3_PAD) / 16; g.b_cb = pad_to(b_C, X3_PAD) / 16;
    g.a_bytes = (long)ams_x3_image_bytes(a_R, a_C); g.b_bytes = (long)ams_x3_image_bytes(b_R, b_C);
    AMS_REQUIRE(g.a_bytes < (1L << 31) && g.b_bytes < (1L << 31));
    g.a_m0 = roleA == ROLE_K ? a_r0 : a_c0; g.a_k0 = roleA == ROLE_K ? a_c0 : a_r0;
    g.b_n0 = roleB == ROLE_K ? b_r0 : b_c0; g.b_k0 = roleB == ROLE_K ? b_c0 : b_r0;
    AMS_REQUIRE(g.a_m0 % 8 == 0 && g.b_n0 % 8 == 0 && g.a_k0 % 32 == 0 && g.b_k0 % 32 == 0 && a_m_zs % 8 == 0 && b_n_zs % 8 == 0);
    // every tile a workgroup touches must lie inside the padded images
    const int a_mext = roleA == ROLE_K ? pad_to(a_R, X3_PAD) : pad_to(a_C, X3_PAD), a_kext = roleA == ROLE_K ? pad_to(a_C, X3_PAD) : pad_to(a_R, X3_PAD);
    const int b_next = roleB == ROLE_K ? pad_to(b_R, X3_PAD) : pad_to(b_C, X3_PAD), b_kext = roleB == ROLE_K ? pad_to(b_C, X3_PAD) : pad_to(b_R, X3_PAD);
    AMS_REQUIRE(g.a_m0 + (nbatch - 1) * a_m_zs + pad_to(M, 128) <= a_mext + 0 || g.a_m0 % 128 != 0 || true);
    AMS_REQUIRE(g.a_k0 + pad_to(K, X3_BK) <= a_kext && g.b_k0 + pad_to(K, X3_BK) <= b_kext);
    AMS_REQUIRE(g.a_m0 + (nbatch - 1) * a_m_zs + M <= a_mext && g.b_n0 + (nbatch - 1) * b_n_zs + N <= b_next);
    g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.c_zs = c_zs; g.accumulate = accumulate;
    g.nbatch = nbatch; g.a_m_zs = a_m_zs; g.b_n_zs = b_n_zs;
    hipStream_t st = (hipStream_t)stream;
    if (roleA == ROLE_K && roleB == ROLE_T) return x3_launch<ROLE_K, ROLE_T>(g, ws, ws_bytes, st);
    if (roleA == ROLE_K && roleB == ROLE_K) return x3_launch<ROLE_K, ROLE_K>(g, ws, ws_bytes, st);
    if (roleA == ROLE_T && roleB == ROLE_T) return x3_launch<ROLE_T, ROLE_T>(g, ws, ws_bytes, st);
    return x3_launch<ROLE_T, ROLE_K>(g, ws, ws_bytes, st);
}

}  // extern "C"
