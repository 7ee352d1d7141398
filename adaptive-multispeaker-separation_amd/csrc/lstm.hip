// BLSTM recurrence for gfx950 (reference utils/ops.py:358-383; TF-1.4 BasicLSTMCell).
//
// The input projection x.Wx + b for all T steps is hoisted into one MFMA GEMM (gemm.hip).  What is
// left is strictly sequential: per step  a = z_t + h_{t-1}.U,  gates,  c/h update.  One launch per time
// step handles BOTH directions (forward at t = s, backward at t = T-1-s) so the 2-way concurrency the
// model offers is used, and each workgroup owns a 16(batch) x 16(hidden unit) output tile for which it
// needs all four gate columns -- the gate non-linearity and the state update are then workgroup-local.
//
// Work split inside a workgroup: 4 waves split K (=H for forward, =4H for backward) and each runs
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

namespace {

constexpr int TU = 16;   // hidden units per workgroup
constexpr int TB = 16;   // batch rows per workgroup

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
    // backward only
    const float* dout;             // [B,T,2H] gradient w.r.t. layer output
    float* dc;                     // [B,2,H] running dc
};

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

__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(StepArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][4][64][4];   // [wave][gate][lane][reg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ut = blockIdx.x, bt = blockIdx.y, dir = blockIdx.z;
    const int H = a.H, T = a.T;
    const int t = dir ? (T - 1 - a.s) : a.s;
    const int tp = dir ? t + 1 : t - 1;                 // time index holding h_{prev}, c_{prev}
    const bool has_prev = a.s > 0;
    const int b_row = bt * TB + (lane & 15);
    const bool vec = (H % 4 == 0);

    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (has_prev) {
        const float* hrow = a.out + ((long)b_row * T + tp) * (2 * H) + dir * H;
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (4 * 64 * 4) + lane * 4;
        for (int g = wave; g < a.n_g; g += 4) {
            const float4 av = ld4_guard(hrow, g * 16 + (lane >> 4) * 4, H, b_row < a.B, vec);
            float4 bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const float4*>(pk + ((long)g * 4 + q) * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[q].x, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[q].y, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[q].z, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[q].w, acc[q], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&red[wave][q][lane][0]) = acc[q];
    __syncthreads();

    // thread -> (batch row, unit); C/D layout: row = (lane>>4)*4 + reg, col = lane&15
    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    if (b >= a.B || u >= H) return;
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    float* grow = a.G + (((long)b * T + t) * 2 + dir) * (4 * H);
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = grow[q * H + u];
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[w][q][src_lane][src_reg];
        pre[q] = s;
    }
    const float ig = 1.0f / (1.0f + expf(-pre[0]));
    const float gg = tanhf(pre[1]);
    const float fg = 1.0f / (1.0f + expf(-(pre[2] + 1.0f)));     // forget_bias = 1.0
    const float og = 1.0f / (1.0f + expf(-pre[3]));
    const float c_prev = has_prev ? a.cst[(((long)b * T + tp) * 2 + dir) * H + u] : 0.f;
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
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(StepArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ut = blockIdx.x, bt = blockIdx.y, dir = blockIdx.z;
    const int H = a.H, T = a.T;
    const int t = dir ? a.s : (T - 1 - a.s);
    const int tn = dir ? t - 1 : t + 1;                 // step processed just before in BPTT order (its da feeds dh)
    const int tp = dir ? t + 1 : t - 1;                 // time holding c_{prev} of step t
    const bool has_next = a.s > 0;
    const bool has_prev = (tp >= 0 && tp < T);
    const int b_row = bt * TB + (lane & 15);
    const bool vec = (H % 4 == 0);

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (has_next) {
        const float* darow = a.G + (((long)b_row * T + tn) * 2 + dir) * (4 * H);
        const float* pk = a.pk + (((long)dir * a.n_ut + ut) * a.n_g) * (64 * 4) + lane * 4;
        int g = wave;
        for (; g + 4 < a.n_g; g += 8) {
            const float4 a0 = ld4_guard(darow, g * 16 + (lane >> 4) * 4, 4 * H, b_row < a.B, vec);
            const float4 a1 = ld4_guard(darow, (g + 4) * 16 + (lane >> 4) * 4, 4 * H, b_row < a.B, vec);
            const float4 b0 = *reinterpret_cast<const float4*>(pk + (long)g * 256);
            const float4 b1 = *reinterpret_cast<const float4*>(pk + (long)(g + 4) * 256);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
        }
        for (; g < a.n_g; g += 4) {
            const float4 a0 = ld4_guard(darow, g * 16 + (lane >> 4) * 4, 4 * H, b_row < a.B, vec);
            const float4 b0 = *reinterpret_cast<const float4*>(pk + (long)g * 256);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
        }
    }
    acc0 += acc1;
    *reinterpret_cast<f32x4*>(&red[wave][lane][0]) = acc0;
    __syncthreads();

    const int bl = tid >> 4, ul = tid & 15;
    const int b = bt * TB + bl, u = ut * TU + ul;
    if (b >= a.B || u >= H) return;
    const int src_lane = (bl >> 2) * 16 + ul, src_reg = bl & 3;
    float dh = a.dout[((long)b * T + t) * (2 * H) + dir * H + u];
#pragma unroll
    for (int w = 0; w < 4; ++w) dh += red[w][src_lane][src_reg];

    float* grow = a.G + (((long)b * T + t) * 2 + dir) * (4 * H);
    const float ig = grow[0 * H + u], gg = grow[1 * H + u], fg = grow[2 * H + u], og = grow[3 * H + u];
    const float c = a.cst[(((long)b * T + t) * 2 + dir) * H + u];
    const float c_prev = has_prev ? a.cst[(((long)b * T + tp) * 2 + dir) * H + u] : 0.f;
    float* dcp = a.dc + ((long)b * 2 + dir) * H + u;
    const float dc_next = has_next ? *dcp : 0.f;
    const float tc = tanhf(c);
    const float d_o = dh * tc;
    const float dcv = dc_next + dh * og * (1.0f - tc * tc);
    grow[0 * H + u] = dcv * gg * ig * (1.0f - ig);
    grow[1 * H + u] = dcv * ig * (1.0f - gg * gg);
    grow[2 * H + u] = dcv * c_prev * fg * (1.0f - fg);
    grow[3 * H + u] = d_o * og * (1.0f - og);
    *dcp = dcv * fg;
}

}  // namespace

extern "C" {

size_t ams_blstm_pack_floats(int H, int backward) {
    const int n_ut = ceil_div(H, TU);
    if (!backward) return (size_t)2 * n_ut * ceil_div(H, 16) * 4 * 64 * 4;
    return (size_t)2 * n_ut * ceil_div(4 * H, 16) * 64 * 4;
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
    dim3 grid(n_ut, ceil_div(B, TB), 2);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(256), 0, st, a);
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
    dim3 grid(n_ut, ceil_div(B, TB), 2);
    for (int s = 0; s < T; ++s) {
        a.s = s;
        hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, dim3(256), 0, st, a);
    }
    return ams_check_launch();
}

}  // extern "C"
