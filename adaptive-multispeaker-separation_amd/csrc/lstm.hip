// BLSTM recurrence for gfx950 (reference utils/ops.py:358-383; TF-1.4 BasicLSTMCell).
//
// The input projection x.Wx + b for all T steps is hoisted into one MFMA GEMM (gemm.hip).  What is
// left is strictly sequential: per step  a = z_t + h_{t-1}.U,  gates,  c/h update.  One launch per time
// step handles BOTH directions (forward at t = s, backward at t = T-1-s) so the 2-way concurrency the
// model offers is used, and each workgroup owns a 16(batch) x 16(hidden unit) output tile for which it
// needs all four gate columns -- the gate non-linearity and the state update are then workgroup-local.
//
// Work split inside a workgroup: NWF (forward) / NWB (backward) waves split K (=H for forward, =4H for backward) and each runs
// v_mfma_f32_16x16x4_f32 chains on fragments fetched as one float4 per lane:
//   * the recurrent matrix is re-packed once per layer call into MFMA fragment order (pack kernels
//     below) so a lane reads 16 contiguous bytes per 4 k-steps (L2-resident: 2 x 1.46 MB per layer);
//   * h_{t-1} (or da_{t+1}) is read straight from the layer's output / gate buffers.
// Partial accumulators meet in LDS, then each of the 256 threads finishes one (batch,unit) element.
//
// Buffers (all fp32, row-major):
//   G    [B,T,2,4H]  in: pre-activations z (+bias); fwd overwrites with activated gates i,g,f,o;
//                    bwd overwrites with d(pre-activation) = da  (feeds the hoisted dWx/dU/dx GEMMs)
//   out  [B,T,2H]    h, forward dir in cols [0,H), backward dir in [H,2H)   (utils/ops.py:383 concat)
//   cst  [B,T,2,H]   cell states
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TU = 16;   // hidden units per workgroup
constexpr int TB = 16;   // batch rows per workgroup
// Waves per workgroup splitting K (the first 4 also run the epilogue).  Measured on MI355X, B=64 T=80 H=300,
// per layer: forward 4/8/16 waves = 0.648/0.627/0.662 ms; backward 4 (2 chunks)/4 (1 chunk)/8/16 waves =
// 0.589/0.559/0.603/0.533 ms -- every wave issues ALL its operand loads before its first MFMA.
#ifndef AMS_LSTM_FWD_WAVES
#define AMS_LSTM_FWD_WAVES 8
#endif
#ifndef AMS_LSTM_BWD_WAVES
#define AMS_LSTM_BWD_WAVES 16
#endif
constexpr int NWF = AMS_LSTM_FWD_WAVES, NWB = AMS_LSTM_BWD_WAVES;

// Upk[dir][ut][g][q][lane][4]: lane l, j -> U[k = g*16 + (l>>4)*4 + j][q*H + ut*16 + (l&15)]
__global__ void pack_u_fwd_kernel(const float* __restrict__ Uf, const float* __restrict__ Ub, long ldu, float* __restrict__ pk,
                                  int H, int n_ut, int n_g) {
    const long total = (long)2 * n_ut * n_g * 4 * 64 * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int j = r & 3; r >>= 2;
        const int l = r & 63; r >>= 6;
        const int q = r & 3; r >>= 2;
        const int g = (int)(r % n_g); r /= n_g;
        const int ut = (int)(r % n_ut); r /= n_ut;
        const int dir = (int)r;
        const int k = g * 16 + (l >> 4) * 4 + j;
        const int u = ut * TU + (l & 15);
        const float* U = dir ? Ub : Uf;
        pk[i] = (k < H && u < H) ? U[(long)k * ldu + q * H + u] : 0.f;
    }
}

// UTpk[dir][ut][g][lane][4]: lane l, j -> U[unit = ut*16 + (l&15)][col = g*16 + (l>>4)*4 + j]
__global__ void pack_u_bwd_kernel(const float* __restrict__ Uf, const float* __restrict__ Ub, long ldu, float* __restrict__ pk,
                                  int H, int n_ut, int n_g) {
    const long total = (long)2 * n_ut * n_g * 64 * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int j = r & 3; r >>= 2;
        const int l = r & 63; r >>= 6;
        const int g = (int)(r % n_g); r /= n_g;
        const int ut = (int)(r % n_ut); r /= n_ut;
        const int dir = (int)r;
        const int col = g * 16 + (l >> 4) * 4 + j;
        const int u = ut * TU + (l & 15);
        const float* U = dir ? Ub : Uf;
        pk[i] = (col < 4 * H && u < H) ? U[(long)u * ldu + col] : 0.f;
    }
}

struct StepArgs {
    float* G; float* out; float* cst;
    const float* pk;
    int B, T, H, n_ut, n_g, s;     // s = step index
    int xcd_map;                   // 1: 1-D grid, workgroups that read the same packed weights share an XCD (see block_coords)
    // backward only
    const float* dout;             // [B,T,2H] gradient w.r.t. layer output
    float* dc;                     // [B,2,H] running dc
    // state dropout (DropoutWrapper(state_keep_prob), utils/ops.py:373,379): the state handed to the NEXT step is (c.mc, h.mh) while
    // the cell output stays h.  mh / mc [B,T,2,H]: masks already scaled by 1/keep; hs [B,T,2H] / cs [B,T,2,H]: the masked states
    // (written by the forward kernel, read by it one step later, by the backward kernel and by the recurrent-kernel gradients).
    const float* mh; const float* mc;
    float* hs; float* cs;
};

// Workgroup -> (unit tile, batch tile, direction).  The n_bt batch tiles of one (direction, unit tile) read the SAME 76 KB slice
// of the packed recurrent matrix every step.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with a private
// 4 MB L2; in the plain 3-D grid the tiles sharing a slice sit 19 ids apart, i.e. on different XCDs, so every L2 has to hold
// nearly the whole 2.9 MB matrix next to the streaming gate/state traffic.  The 1-D form gives the sharers ids that are equal
// modulo 8: each L2 then serves 1/8 of the matrix to all its readers.  (Measured slower -- see step_grid.)
__device__ __forceinline__ bool block_coords(const StepArgs& a, int& ut, int& bt, int& dir) {
    if (!a.xcd_map) { ut = blockIdx.x; bt = blockIdx.y; dir = blockIdx.z; return true; }
    const int n_bt = (a.B + TB - 1) / TB;
    if (a.xcd_map == 2) {
        // chain-per-XCD: all unit tiles of one (batch tile, direction) chain get ids equal modulo (2 * n_bt); with 8 chains
        // (B = 64) chain c lives on XCD c, so h_t / da_t is produced and consumed through ONE L2.  Measured: 0.47 -> 0.43 ms
        // (forward recurrence) and 0.528 -> 0.463 ms (BPTT) per layer alone; BPTT steps beside a capped GEMM 7.7-8.0 -> 7.1 us.
        const int nc = 2 * n_bt, chain = blockIdx.x % nc;
        ut = blockIdx.x / nc;
        bt = chain % n_bt;
        dir = chain / n_bt;
        return true;
    }
    const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
    bt = r % n_bt;
    const int j = (r / n_bt) * 8 + xcd;
    if (j >= 2 * a.n_ut) return false;
    dir = j / a.n_ut;
    ut = j - dir * a.n_ut;
    return true;
}

__device__ __forceinline__ float4 ld4_guard(const float* p, int k, int kmax, bool row_ok, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (vec && k + 3 < kmax) return *reinterpret_cast<const float4*>(p + k);
    if (k + 0 < kmax) v.x = p[k + 0];
    if (k + 1 < kmax) v.y = p[k + 1];
    if (k + 2 < kmax) v.z = p[k + 2];
    if (k + 3 < kmax) v.w = p[k + 3];
    return v;
}

__global__ __launch_bounds__(NWF * 64) void lstm_step_fwd_kernel(StepArgs a) {
    __builtin_amdgcn_s_setprio(3);      // latency-bound: win issue arbitration over the weight-gradient GEMM waves sharing the CU
    __shared__ __attribute__((aligned(16))) float red[NWF][4][64][4];   // [wave][gate][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ut, bt, dir;
    if (!block_coords(a, ut, bt, dir)) return;
    const int H = a.H, T = a.T;
    const int t = dir ? (T - 1 - a.s) : a.s;
    const int tp = dir ? t + 1 : t - 1;                 // time index holding h_{prev}, c_{prev}
    const bool has_prev = a.s > 0;
    const int b_row = bt * TB + (lane & 15);
    const bool vec = (H % 4 == 0);

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue operands (this thread's (batch,unit) element) are fetched FIRST so their latency hides under
    // the recurrent product.  thread -> (batch row, unit); C/D layout: row = (lane>>4)*4 + reg, col = lane&15
    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    const bool live = (tid < 256 && b < a.B && u < H);
    float* grow = a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H);
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) zq[q] = grow[q * H + u];
        if (has_prev) c_prev = (a.cs ? a.cs : a.cst)[(((long)b * T + tp) * 2 + dir) * H + u];
    }

    if (has_prev) {
        const float* hrow = (a.hs ? a.hs : a.out) + ((long)b_row * T + tp) * (2 * H) + dir * H;
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (4 * 64 * 4) + lane * 4;
        // Issue EVERY operand load of a chunk before the first MFMA: the h/U fetches are L2 round trips
        // (~1 us) and a load->MFMA->load loop would pay that latency once per k-group.
        constexpr int CH = (19 + NWF - 1) / NWF;         // k-groups per wave per chunk (H = 300: 19 groups / NWF waves)
        for (int g0 = wave; g0 < a.n_g; g0 += NWF * CH) {
            float4 av[CH], bv[CH][4];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int g = g0 + NWF * i;
                if (g < a.n_g) {
                    av[i] = ld4_guard(hrow, g * 16 + (lane >> 4) * 4, H, b_row < a.B, vec);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const float4*>(pk + ((long)g * 4 + q) * 256);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int g = g0 + NWF * i;
                if (g < a.n_g) {
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
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&red[wave][q][lane][0]) = acc[q];
    __syncthreads();

    if (!live) return;
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = zq[q];
#pragma unroll
        for (int w = 0; w < NWF; ++w) s += red[w][q][src_lane][src_reg];
        pre[q] = s;
    }
    const float ig = 1.0f / (1.0f + expf(-pre[0]));
    const float gg = tanhf(pre[1]);
    const float fg = 1.0f / (1.0f + expf(-(pre[2] + 1.0f)));     // forget_bias = 1.0
    const float og = 1.0f / (1.0f + expf(-pre[3]));
    const float c = c_prev * fg + ig * gg;
    const float h = tanhf(c) * og;
    grow[0 * H + u] = ig;
    grow[1 * H + u] = gg;
    grow[2 * H + u] = fg;
    grow[3 * H + u] = og;
    a.cst[(((long)b * T + t) * 2 + dir) * H + u] = c;
    a.out[((long)b * T + t) * (2 * H) + dir * H + u] = h;
    if (a.hs) {
        const long si = (((long)b * T + t) * 2 + dir) * H + u;
        a.cs[si] = c * a.mc[si];
        a.hs[((long)b * T + t) * (2 * H) + dir * H + u] = h * a.mh[si];
    }
}

// Pipelined-fetch form of the forward step (H % 4 == 0 and NWF < n_g <= 3 * NWF, i.e. 132 <= H <= 384).
__global__ __launch_bounds__(NWF * 64) void lstm_step_fwd_pipe_kernel(StepArgs a) {
    __builtin_amdgcn_s_setprio(3);      // latency-bound: win issue arbitration over the weight-gradient GEMM waves sharing the CU
    __shared__ __attribute__((aligned(16))) float red[NWF][4][64][4];   // [wave][gate][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ut, bt, dir;
    if (!block_coords(a, ut, bt, dir)) return;
    const int H = a.H, T = a.T;
    const int t = dir ? (T - 1 - a.s) : a.s;
    const int tp = dir ? t + 1 : t - 1;                 // time index holding h_{prev}, c_{prev}
    const bool has_prev = a.s > 0;
    const int b_row = bt * TB + (lane & 15);

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Epilogue operands (this thread's (batch,unit) element) are fetched FIRST so their latency hides under
    // the recurrent product.  thread -> (batch row, unit); C/D layout: row = (lane>>4)*4 + reg, col = lane&15
    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    const bool live = (tid < 256 && b < a.B && u < H);
    float* grow = a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H);
    float zq[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) zq[q] = grow[q * H + u];
        if (has_prev) c_prev = a.cst[(((long)b * T + tp) * 2 + dir) * H + u];
    }

    if (has_prev) {
        // H % 4 == 0, n_g <= 3 * NWF.  Software-pipelined fetch: groups 0 and 1 are requested together, group 0's MFMAs run
        // while group 1 (and 2) are still in flight.  Unconditional 16-byte loads on clamped addresses (columns >= H meet zero
        // rows of the packed matrix, rows >= B are never stored), the optional third group in its own branch with its own
        // copy of the MFMAs, so every s_waitcnt counts exactly.
        const float* hrow = a.out + ((long)min(b_row, a.B - 1) * T + tp) * (2 * H) + dir * H;
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (4 * 64 * 4) + lane * 4;
        const int kq = (lane >> 4) * 4;
        const int g0 = wave, g1 = wave + NWF, g2 = wave + 2 * NWF;
        auto ldh = [&](int g) { return *reinterpret_cast<const float4*>(hrow + min(g * 16 + kq, H - 4)); };
        auto ldw = [&](int g, int q) { return *reinterpret_cast<const float4*>(pk + ((long)g * 4 + q) * 256); };
        auto mm = [&](const float4& av, const float4 (&bv)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[q].x, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[q].y, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[q].z, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[q].w, acc[q], 0, 0, 0);
        };
        const int g1c = min(g1, a.n_g - 1);                 // n_g > NWF in every shape this kernel is chosen for
        float4 a0 = ldh(g0), b0[4], a1 = ldh(g1c), b1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b0[q] = ldw(g0, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) b1[q] = ldw(g1c, q);
        if (g2 < a.n_g) {
            float4 a2 = ldh(g2), b2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b2[q] = ldw(g2, q);
            __builtin_amdgcn_sched_barrier(0);              // keep the third group's loads AHEAD of the MFMAs (hipcc sinks them)
            mm(a0, b0);
            mm(a1, b1);
            mm(a2, b2);
        } else {
            mm(a0, b0);
            if (g1 < a.n_g) mm(a1, b1);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&red[wave][q][lane][0]) = acc[q];
    __syncthreads();

    if (!live) return;
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = zq[q];
#pragma unroll
        for (int w = 0; w < NWF; ++w) s += red[w][q][src_lane][src_reg];
        pre[q] = s;
    }
    const float ig = 1.0f / (1.0f + expf(-pre[0]));
    const float gg = tanhf(pre[1]);
    const float fg = 1.0f / (1.0f + expf(-(pre[2] + 1.0f)));     // forget_bias = 1.0
    const float og = 1.0f / (1.0f + expf(-pre[3]));
    const float c = c_prev * fg + ig * gg;
    const float h = tanhf(c) * og;
    grow[0 * H + u] = ig;
    grow[1 * H + u] = gg;
    grow[2 * H + u] = fg;
    grow[3 * H + u] = og;
    a.cst[(((long)b * T + t) * 2 + dir) * H + u] = c;
    a.out[((long)b * T + t) * (2 * H) + dir * H + u] = h;
}


// Backward step s: forward direction handles t = T-1-s, backward direction t = s.
// VEC4 (H % 4 == 0) is a COMPILE-TIME switch here, and the operand fetch is ONE unconditional 16-byte load on a row clamped to
// the batch (rows >= B only feed values that are never stored).  With the guarded fetch (runtime flag, vector / scalar paths
// merged per operand) every merge is a use of the loaded value and hipcc placed `s_waitcnt vmcnt(0)` in front of each k-group's
// load: 5 dependent memory round trips per step instead of one.  Layer BPTT 0.525 -> 0.512 ms alone, step +1 %.
// (The forward kernel keeps the guarded form: the same change measured 1-2 % SLOWER there -- its 95 KB per workgroup arrive
// in three bursts instead of one and the per-CU load path is the bottleneck)
template <bool VEC4>
__global__ __launch_bounds__(NWB * 64) void lstm_step_bwd_kernel(StepArgs a) {
    __builtin_amdgcn_s_setprio(3);      // latency-bound: win issue arbitration over the weight-gradient GEMM waves sharing the CU
    __shared__ __attribute__((aligned(16))) float red[NWB][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ut, bt, dir;
    if (!block_coords(a, ut, bt, dir)) return;
    const int H = a.H, T = a.T;
    const int t = dir ? a.s : (T - 1 - a.s);
    const int tn = dir ? t - 1 : t + 1;                 // step processed just before in BPTT order (its da feeds dh)
    const int tp = dir ? t + 1 : t - 1;                 // time holding c_{prev} of step t
    const bool has_next = a.s > 0;
    const bool has_prev = (tp >= 0 && tp < T);
    const int b_row = bt * TB + (lane & 15);
    constexpr bool vec = VEC4;

    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    const bool live = (tid < 256 && b < a.B && u < H);
    float* grow = a.G + (((long)(live ? b : 0) * T + t) * 2 + dir) * (4 * H);
    float* dcp = a.dc + ((long)(live ? b : 0) * 2 + dir) * H + (live ? u : 0);
    float dh = 0.f, ig = 0.f, gg = 0.f, fg = 0.f, og = 0.f, c = 0.f, c_prev = 0.f, dc_next = 0.f;
    if (live) {
        dh = a.dout[((long)b * T + t) * (2 * H) + dir * H + u];
        ig = grow[0 * H + u]; gg = grow[1 * H + u]; fg = grow[2 * H + u]; og = grow[3 * H + u];
        c = a.cst[(((long)b * T + t) * 2 + dir) * H + u];
        if (has_prev) c_prev = (a.cs ? a.cs : a.cst)[(((long)b * T + tp) * 2 + dir) * H + u];
        if (has_next) dc_next = *dcp;
    }
    float mh = 1.f;
    if (live && a.mh) {                                  // the state that left step t was (c_t.mc_t, h_t.mh_t)
        const long si = (((long)b * T + t) * 2 + dir) * H + u;
        mh = a.mh[si];
        dc_next *= a.mc[si];
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (has_next) {
        const float* darow = a.G + (((long)min(b_row, a.B - 1) * T + tn) * 2 + dir) * (4 * H);
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (64 * 4) + lane * 4;
        constexpr int CH = (75 + NWB - 1) / NWB;            // k-groups per wave per chunk (4H = 1200: 75 groups / NWB waves)
        for (int g0 = wave; g0 < a.n_g; g0 += NWB * CH) {
            float4 av[CH], bv[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int g = g0 + NWB * i;
                if (g < a.n_g) {
                    if (vec) av[i] = *reinterpret_cast<const float4*>(darow + g * 16 + (lane >> 4) * 4);
                    else av[i] = ld4_guard(darow, g * 16 + (lane >> 4) * 4, 4 * H, true, false);
                    bv[i] = *reinterpret_cast<const float4*>(pk + (long)g * 256);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; i += 2) {
                const int g = g0 + NWB * i;
                const bool ok0 = g < a.n_g, ok1 = (i + 1 < CH) && (g + NWB < a.n_g);
                if (ok0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[i].x, acc0, 0, 0, 0);
                if (ok1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].x, bv[i + 1].x, acc1, 0, 0, 0);
                if (ok0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[i].y, acc0, 0, 0, 0);
                if (ok1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].y, bv[i + 1].y, acc1, 0, 0, 0);
                if (ok0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[i].z, acc0, 0, 0, 0);
                if (ok1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].z, bv[i + 1].z, acc1, 0, 0, 0);
                if (ok0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[i].w, acc0, 0, 0, 0);
                if (ok1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i + 1].w, bv[i + 1].w, acc1, 0, 0, 0);
            }
        }
    }
    acc0 += acc1;
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc0;
    __syncthreads();

    if (!live) return;
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    float dh_rec = 0.f;
#pragma unroll
    for (int w = 0; w < NWB; ++w) dh_rec += red[w][src_lane][src_reg];
    dh += dh_rec * mh;
    const float tc = tanhf(c);
    const float d_o = dh * tc;
    const float dcv = dc_next + dh * og * (1.0f - tc * tc);
    grow[0 * H + u] = dcv * gg * ig * (1.0f - ig);
    grow[1 * H + u] = dcv * ig * (1.0f - gg * gg);
    grow[2 * H + u] = dcv * c_prev * fg * (1.0f - fg);
    grow[3 * H + u] = d_o * og * (1.0f - og);
    *dcp = dcv * fg;
}

inline dim3 step_grid(StepArgs& a) {
    // MEASURED, default off (AMS_LSTM_XCD=1 turns it on): 0.645 / 0.536 ms per forward / backward layer vs 0.632 / 0.524 for the
    // plain 3-D grid, 9.86 k vs 10.03 k mixtures/s -- four sharers hammering one L2 lose to four L2s serving one reader each.
    // 2 (default) = chain-per-XCD ids, 0 = plain 3-D grid, 1 = weight sharers per XCD (measured slower, see block_coords)
    static const int mode = getenv("AMS_LSTM_XCD") ? atoi(getenv("AMS_LSTM_XCD")) : 2;                       // tuning aid
    const bool off = mode == 0;
    const int n_bt = ceil_div(a.B, TB);
    a.xcd_map = mode;
    if (off) return dim3(a.n_ut, n_bt, 2);
    if (mode == 2) return dim3(2 * n_bt * a.n_ut);
    return dim3(8 * n_bt * ceil_div(2 * a.n_ut, 8));
}

inline bool fwd_pipe_ok(int H, int n_g) {
    // MEASURED, default off: 0.646 vs 0.625 ms per forward layer, 9.82 k vs 10.02 k mixtures/s.  Requesting a workgroup's 95 KB
    // earlier does not help the forward step -- the guarded kernel's three dependent bursts are faster than one big one.
    static const bool on = getenv("AMS_LSTM_FWD_PIPE") && atoi(getenv("AMS_LSTM_FWD_PIPE")) == 1;      // tuning aid
    return on && H % 4 == 0 && n_g > NWF && n_g <= 3 * NWF;
}

}  // namespace

extern "C" {

size_t ams_blstm_pack_floats(int H, int backward) {
    const int n_ut = ceil_div(H, TU);
    if (!backward) return (size_t)2 * n_ut * ceil_div(H, 16) * 4 * 64 * 4;
    return (size_t)2 * n_ut * ceil_div(4 * H, 16) * 64 * 4;
}

// Re-pack both directions' recurrent matrices into MFMA fragment order (forward: U, backward: U^T).
ams_status ams_blstm_pack(const float* Uf, const float* Ub, long ldu, float* pack, int H, int backward, void* stream) {
    AMS_REQUIRE(Uf && Ub && pack && H > 0);
    hipStream_t st = (hipStream_t)stream;
    const int n_ut = ceil_div(H, TU);
    if (!backward) {
        const int n_g = ceil_div(H, 16);
        const long total = (long)2 * n_ut * n_g * 4 * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_fwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    } else {
        const int n_g = ceil_div(4 * H, 16);
        const long total = (long)2 * n_ut * n_g * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_bwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    }
    return ams_check_launch();
}

// Recurrence, forward.  Uf/Ub: recurrent part of each direction's TF kernel (rows D.. of [D+H,4H]), ldu = 4H.
ams_status ams_blstm_recurrent_fwd(float* G, float* out, float* cst, const float* Uf, const float* Ub, long ldu, float* pack,
                                   int B, int T, int H, void* stream) {
    AMS_REQUIRE(G && out && cst && Uf && Ub && pack && B > 0 && T > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    const int n_ut = ceil_div(H, TU), n_g = ceil_div(H, 16);
    {
        const long total = (long)2 * n_ut * n_g * 4 * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_fwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    }
    StepArgs a{};
    a.G = G; a.out = out; a.cst = cst; a.pk = pack;
    a.B = B; a.T = T; a.H = H; a.n_ut = n_ut; a.n_g = n_g;
    dim3 grid = step_grid(a);
    const bool use_pipe = fwd_pipe_ok(H, n_g);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        if (use_pipe) hipLaunchKernelGGL(lstm_step_fwd_pipe_kernel, grid, dim3(NWF * 64), 0, st, a);
        else hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(NWF * 64), 0, st, a);
    }
    return ams_check_launch();
}

// The same recurrence with DropoutWrapper's STATE dropout (utils/ops.py:373,379 at --recurrent_dropout != 0; TF 1.4 masks both parts
// of the LSTMStateTuple): out / cst keep the cell's own h_t / c_t, hs [B,T,2H] / cs [B,T,2,H] receive the masked states h_t.mh_t /
// c_t.mc_t that step t+1 (t-1 for the backward direction) starts from.  mh, mc [B,T,2,H]: keep-masks scaled by 1/keep, drawn by the
// caller.  Input and output dropout are elementwise on x and on out and stay with the caller.  Per-step kernels only.
ams_status ams_blstm_recurrent_fwd_dropout(float* G, float* out, float* cst, float* hs, float* cs, const float* mh, const float* mc,
                                           const float* Uf, const float* Ub, long ldu, float* pack, int B, int T, int H, void* stream) {
    AMS_REQUIRE(G && out && cst && hs && cs && mh && mc && Uf && Ub && pack && B > 0 && T > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    const int n_ut = ceil_div(H, TU), n_g = ceil_div(H, 16);
    {
        const long total = (long)2 * n_ut * n_g * 4 * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_fwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    }
    StepArgs a{};
    a.G = G; a.out = out; a.cst = cst; a.pk = pack; a.hs = hs; a.cs = cs; a.mh = mh; a.mc = mc;
    a.B = B; a.T = T; a.H = H; a.n_ut = n_ut; a.n_g = n_g;
    dim3 grid = step_grid(a);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(NWF * 64), 0, st, a);
    }
    return ams_check_launch();
}

// BPTT of the recurrence above: dh_t = dout_t + mh_t . (da_{t+1} U^T), dc_t = mc_t . (dc_{t+1} f_{t+1}) + ..., c_prev = the MASKED state.
ams_status ams_blstm_recurrent_bwd_dropout(float* G, const float* cst, const float* cs, const float* dout, float* dc, const float* mh,
                                           const float* mc, const float* Uf, const float* Ub, long ldu, float* pack, int B, int T, int H,
                                           void* stream) {
    AMS_REQUIRE(G && cst && cs && dout && dc && mh && mc && Uf && Ub && pack && B > 0 && T > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    const int n_ut = ceil_div(H, TU), n_g = ceil_div(4 * H, 16);
    {
        const long total = (long)2 * n_ut * n_g * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_bwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    }
    StepArgs a{};
    a.G = G; a.cst = const_cast<float*>(cst); a.cs = const_cast<float*>(cs); a.pk = pack; a.dout = dout; a.dc = dc; a.mh = mh; a.mc = mc;
    a.B = B; a.T = T; a.H = H; a.n_ut = n_ut; a.n_g = n_g;
    dim3 grid = step_grid(a);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        if (H % 4 == 0) hipLaunchKernelGGL(lstm_step_bwd_kernel<true>, grid, dim3(NWB * 64), 0, st, a);
        else hipLaunchKernelGGL(lstm_step_bwd_kernel<false>, grid, dim3(NWB * 64), 0, st, a);
    }
    return ams_check_launch();
}

// Recurrence, backward (BPTT).  On return G holds da (gradient w.r.t. the pre-activations).
ams_status ams_blstm_recurrent_bwd(float* G, const float* cst, const float* dout, float* dc, const float* Uf, const float* Ub,
                                   long ldu, float* pack, int B, int T, int H, void* stream) {
    AMS_REQUIRE(G && cst && dout && dc && Uf && Ub && pack && B > 0 && T > 0 && H > 0);
    hipStream_t st = (hipStream_t)stream;
    const int n_ut = ceil_div(H, TU), n_g = ceil_div(4 * H, 16);
    {
        const long total = (long)2 * n_ut * n_g * 64 * 4;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_u_bwd_kernel, dim3(blocks), dim3(256), 0, st, Uf, Ub, ldu, pack, H, n_ut, n_g);
    }
    StepArgs a{};
    a.G = G; a.cst = const_cast<float*>(cst); a.pk = pack; a.dout = dout; a.dc = dc;
    a.B = B; a.T = T; a.H = H; a.n_ut = n_ut; a.n_g = n_g;
    dim3 grid = step_grid(a);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        if (H % 4 == 0) hipLaunchKernelGGL(lstm_step_bwd_kernel<true>, grid, dim3(NWB * 64), 0, st, a);
        else hipLaunchKernelGGL(lstm_step_bwd_kernel<false>, grid, dim3(NWB * 64), 0, st, a);
    }
    return ams_check_launch();
}

}  // extern "C"
