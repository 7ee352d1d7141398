"""Path B of the adaptive front/back end (--with_max_pool): fused conv + max-pool-with-argmax analysis, sparse
(unpool-free) synthesis, gather-form gradients.  Reference models/adapt.py:114-117, 210-243; utils/ops.py:94-120."""
import torch
from torch.autograd import Function

from . import ops
from ._lib import load, check

_p, _s = ops._p, ops._s


def _ll(shape, dev):
    return torch.empty(shape, dtype=torch.int64, device=dev)


def front_maxpool_fwd(x, f, P, hop):
    ops._chk(x, f)
    lib = load()
    Bt, L = x.shape
    W, N = f.shape
    T = (L - P) // hop + 1
    y = torch.empty((Bt, T, N), dtype=torch.float32, device=x.device)
    am = _ll((Bt, T, N), x.device)
    nb = lib.ams_front_maxpool_workspace_bytes_w(Bt, L, N, W)
    ws = ops._ws(nb, x)
    # 2 TFLOP of brute-force stride-1 conv: worth two small passes for the operand bounds that let it run as fp16x3
    bounds = (ops.absmax(x), ops.absmax(f)) if ops.F16X3 else None
    ev = ops.PROFILE.begin() if ops.PROFILE.enabled else None
    pa, pb, gt = ops._bounds(bounds)
    check(lib.ams_front_maxpool_fwd(_p(x), _p(f), _p(y), _p(am), Bt, L, W, N, P, hop, pa, pb, _p(ws), nb, _s()), 'ams_front_maxpool_fwd')
    if ev is not None:
        ops.PROFILE.end(ev, 2.0 * Bt * L * N * W, 4.0 * (Bt * L + W * N + Bt * T * N), gt)
    return y, am


def argmax_to_pos(argmax, N):
    """int64 flattened argmax (l*N + n, TF layout) -> int32 sample positions l; converted once per use site."""
    pos = torch.empty(argmax.shape, dtype=torch.int32, device=argmax.device)
    check(load().ams_argmax_to_pos(_p(argmax), _p(pos), argmax.numel(), N, _s()), 'ams_argmax_to_pos')
    return pos


def transpose(m):
    ops._chk(m)
    out = torch.empty((m.shape[1], m.shape[0]), dtype=torch.float32, device=m.device)
    check(load().ams_transpose_f32(_p(m), _p(out), m.shape[0], m.shape[1], _s()), 'ams_transpose_f32')
    return out


def gather_filter_grad(x, v, pos, W, rdiv):
    ops._chk(x, v)
    lib = load()
    R, L = x.shape
    T, N = v.shape[1:]
    df = torch.empty((W, N), dtype=torch.float32, device=x.device)
    nb = lib.ams_gather_filter_grad_workspace_bytes(R, W, N)
    ws = ops._ws(nb, x)
    check(lib.ams_gather_filter_grad(_p(x), _p(v), _p(pos), _p(df), R, L, W, N, T, rdiv, _p(ws), nb, _s()), 'ams_gather_filter_grad')
    return df


class FrontMaxPool(Function):
    """y, argmax = max_pool_with_argmax(conv2d(x, f, stride 1, SAME), P, hop)   (adapt.py:115-117)."""

    @staticmethod
    def forward(ctx, x, f, P, hop):
        y, am = front_maxpool_fwd(x, f, P, hop)
        ctx.save_for_backward(x, argmax_to_pos(am, f.shape[1]))
        ctx.W = f.shape[0]
        ctx.mark_non_differentiable(am)
        return y, am

    @staticmethod
    def backward(ctx, dy, _dam):
        x, pos = ctx.saved_tensors
        df = gather_filter_grad(x, dy.contiguous(), pos, ctx.W, 1) if ctx.needs_input_grad[1] else None
        return None, df, None, None


class SynthUnpool(Function):
    """unpool(vals, mixture argmax tiled S times) + conv2d_transpose stride 1 SAME, sparse form (adapt.py:210-243)."""

    @staticmethod
    def forward(ctx, vals, argmax_mix, f2, L, P, hop, S):
        ops._chk(vals, f2)
        R, T, N = vals.shape
        W = f2.shape[0]
        out = torch.empty((R, L), dtype=torch.float32, device=vals.device)
        pos = argmax_to_pos(argmax_mix, N)
        f2t = transpose(f2.detach().contiguous())
        check(load().ams_synth_unpool_fwd(_p(vals), _p(pos), _p(f2t), _p(out), R, L, W, N, T, P, hop, S, _s()), 'ams_synth_unpool_fwd')
        ctx.save_for_backward(vals, pos, f2t)
        ctx.cfg = (R, T, N, W, L, S)
        return out

    @staticmethod
    def backward(ctx, dout):
        vals, am, f2t = ctx.saved_tensors
        R, T, N, W, L, S = ctx.cfg
        dout = dout.contiguous()
        dvals = df2 = None
        if ctx.needs_input_grad[0]:
            dvals = torch.empty_like(vals)
            check(load().ams_synth_unpool_bwd_vals(_p(dout), _p(am), _p(f2t), _p(dvals), R, L, W, N, T, S, _s()), 'ams_synth_unpool_bwd_vals')
        if ctx.needs_input_grad[2]:
            df2 = gather_filter_grad(dout, vals, am, W, S)
        return dvals, None, df2, None, None, None, None


def front_maxpool(x, f, P, hop):
    return FrontMaxPool.apply(x.contiguous(), f, P, hop)


def synth_unpool(vals, argmax_mix, f2, L, P, hop, S):
    return SynthUnpool.apply(vals.contiguous(), argmax_mix.contiguous(), f2, L, P, hop, S)


# ---------------------------------------------------------------------------------------------------------------
# Path C (--with_average_pool): mean over P stride-1 conv outputs == ONE strided conv with the box-filtered filter
#   y[b,t,n] = (1/P) sum_{i<P} X[b,tP+i,n] = sum_k' xpad[b, tP + k' - pl] g[k',n],  g = box_P * f  (length W+P-1)
# and the back end (nearest up-sampling x P + stride-1 transposed conv) is the matching strided synthesis with
# g2 = P * box_P * f2.  Neither [Bt,L,N] tensor is ever built.  Reference models/adapt.py:118-120, 225-228.
# ---------------------------------------------------------------------------------------------------------------
def box_filter(f, P, scale):
    """g[k'] = scale * sum_{i=0..P-1} f[k'-i]  (tiny [W,N] tensor: torch glue, differentiable)."""
    W, N = f.shape
    z = torch.zeros((P, N), dtype=f.dtype, device=f.device)
    cs = torch.cumsum(torch.cat([z, f, z[:P - 1]], dim=0), dim=0)             # cs[j] = sum_{q<=j} padded[q]
    return (cs[P:] - cs[:-P]) * scale                                            # [W+P-1, N]


class FramesConv(Function):
    """out[(r,t),n] = sum_k xpad[r, t*hop + k - pad_left] * Bm[k,n] with gradient w.r.t. Bm."""

    @staticmethod
    def forward(ctx, x, Bm, hop, T, pad_left):
        ctx.save_for_backward(x)
        ctx.cfg = (Bm.shape[0], hop, T, pad_left)
        return ops.frames_matmul(x, Bm.contiguous(), hop, T, pad_left).view(x.shape[0], T, Bm.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        Wg, hop, T, pl = ctx.cfg
        lib = load()
        R, L = x.shape
        N = dy.shape[2]
        dy = dy.contiguous()
        nb = lib.ams_frames_matmul_bwd_filter_workspace_bytes(R, Wg, N, T)
        ws = ops._ws(nb, x)
        dB = torch.empty((Wg, N), dtype=torch.float32, device=x.device)
        check(lib.ams_frames_matmul_bwd_filter(_p(x), _p(dy), _p(dB), R, L, Wg, N, hop, T, pl, _p(ws), nb, _s()), 'ams_frames_matmul_bwd_filter')
        return None, dB, None, None, None


class SynthFrames(Function):
    """out[r,l] = sum_{t,n} z[r,t,n] * Bm[l + pad_left - t*hop, n]   (transposed framed product, explicit geometry)."""

    @staticmethod
    def forward(ctx, z, Bm, hop, L, pad_left):
        R, T, N = z.shape
        Wg = Bm.shape[0]
        Bm = Bm.contiguous()
        frames = ops.gemm(z.view(R * T, N), Bm, transB=True)
        out = ops.overlap_add(frames, R, T, Wg, L, hop, pad_left)
        ctx.save_for_backward(z, Bm)
        ctx.cfg = (hop, L, pad_left)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, Bm = ctx.saved_tensors
        hop, L, pl = ctx.cfg
        R, T, N = z.shape
        dout = dout.contiguous()
        dz = ops.frames_matmul(dout, Bm, hop, T, pl).view(R, T, N) if ctx.needs_input_grad[0] else None
        dB = None
        if ctx.needs_input_grad[1]:
            lib = load()
            Wg = Bm.shape[0]
            nb = lib.ams_frames_matmul_bwd_filter_workspace_bytes(R, Wg, N, T)
            ws = ops._ws(nb, z)
            dB = torch.empty((Wg, N), dtype=torch.float32, device=z.device)
            check(lib.ams_frames_matmul_bwd_filter(_p(dout), _p(z), _p(dB), R, L, Wg, N, hop, T, pl, _p(ws), nb, _s()), 'ams_frames_matmul_bwd_filter')
        return dz, dB, None, None, None


def front_avgpool(x, f, P):
    W = f.shape[0]
    T = x.shape[1] // P
    return FramesConv.apply(x.contiguous(), box_filter(f, P, 1.0 / P), P, T, (W - 1) // 2)


def synth_avgpool(z, f2, P, L):
    W = f2.shape[0]
    return SynthFrames.apply(z.contiguous(), box_filter(f2, P, 1.0), P, L, (W - 1) // 2)
