"""torch.autograd.Function wrappers around ams_hip.ops.

Autograd is used only to sequence the hand-written backward kernels; every forward and backward body is
a call into libams_hip.so.  Citations are to the reference call sites each op replaces.
"""
import torch
from torch.autograd import Function

from . import ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class FrontFilter(Function):
    """f = |w| * bases   (models/adapt.py:106, :234)."""

    @staticmethod
    def forward(ctx, w, bases):
        ctx.save_for_backward(w, bases)
        return ops.front_filter(w, bases)

    @staticmethod
    def backward(ctx, df):
        w, bases = ctx.saved_tensors
        return ops.front_filter_bwd(w, bases, _c(df))


class FrontConv(Function):
    """Strided analysis conv, SAME padding (models/adapt.py:122).  x carries no gradient."""

    @staticmethod
    def forward(ctx, x, f, hop):
        ctx.save_for_backward(x)
        ctx.W, ctx.hop = f.shape[0], hop
        return ops.front_conv(x, f, hop)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        df = ops.front_conv_bwd_filter(x, _c(dy), ctx.W, ctx.hop) if ctx.needs_input_grad[1] else None
        return None, df, None


class BLSTMLayer(Function):
    """utils/ops.py:358-383 (BasicLSTMCell x 2 directions, concat)."""

    @staticmethod
    def forward(ctx, x, Kf, bf, Kb, bb):
        out, G, cst = ops.blstm_fwd(x, Kf, bf, Kb, bb)
        ctx.save_for_backward(x, Kf, Kb, out, G, cst)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, Kf, Kb, out, G, cst = ctx.saved_tensors
        dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(x, Kf, Kb, out, G, cst, _c(dout), need_dx=ctx.needs_input_grad[0])
        return dx, dKf, dbf, dKb, dbb


class Dense(Function):
    """Conv1D with kernel width 1: u = x.W + b  (utils/ops.py:486-503).  x [..., Din], W [Din, Dout]."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        x2 = x.reshape(-1, x.shape[-1])
        u = ops.gemm(x2, W, bias=b)
        return u.view(x.shape[:-1] + (W.shape[1],))

    @staticmethod
    def backward(ctx, du):
        x, W = ctx.saved_tensors
        du2 = _c(du).view(-1, W.shape[1])
        x2 = x.reshape(-1, x.shape[-1])
        dx = ops.gemm(du2, W, transB=True).view(x.shape) if ctx.needs_input_grad[0] else None
        dW = ops.gemm(x2, du2, transA=True) if ctx.needs_input_grad[1] else None
        db = ops.colsum(du2) if ctx.needs_input_grad[2] else None
        return dx, dW, db


class L2Norm(Function):
    """tf.nn.l2_normalize over the trailing E (utils/ops.py:323-324).  Input [..., F*E] -> output [..., F, E]."""

    @staticmethod
    def forward(ctx, u, E):
        v, inv = ops.l2norm_fwd(u, E)
        ctx.save_for_backward(v, inv)
        ctx.E = E
        return v.view(u.shape[:-1] + (u.shape[-1] // E, E))

    @staticmethod
    def backward(ctx, dv):
        v, inv = ctx.saved_tensors
        return ops.l2norm_bwd(v, inv, _c(dv).view(v.shape), ctx.E), None


class DPCLLoss(Function):
    """models/dpcl.py:41-87.  V [B,TF,E] (unit norm), Y [B,TF,S] -> tensor[4] = cost and the three summaries."""

    @staticmethod
    def forward(ctx, V, Y):
        out, ws = ops.dpcl_loss_fwd(V, Y)
        ctx.save_for_backward(V, Y, ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        V, Y, ws = ctx.saved_tensors
        # out[0] is the cost; the summaries out[1..3] carry no gradient (tf.summary only)
        return ops.dpcl_loss_bwd(V, Y, ws, upstream=_c(dout)), None


class L2NormKeep(Function):
    """L2Norm that also hands back the inverse norms, so a loss can differentiate w.r.t. the PRE-normalised
    tensor in one fused kernel (see DPCLLossFromU).  Returns (v [..., F, E], inv [rows])."""

    @staticmethod
    def forward(ctx, u, E):
        v, inv = ops.l2norm_fwd(u, E)
        ctx.save_for_backward(v, inv)
        ctx.E = E
        ctx.mark_non_differentiable(inv)
        return v.view(u.shape[:-1] + (u.shape[-1] // E, E)), inv

    @staticmethod
    def backward(ctx, dv, _dinv):
        v, inv = ctx.saved_tensors
        return ops.l2norm_bwd(v, inv, _c(dv).view(v.shape), ctx.E), None


class DPCLLossFromU(Function):
    """DPCL loss whose autograd input is u (dense output BEFORE l2-normalise): the backward kernel applies
    d loss/dV and the l2norm Jacobian in one pass over V (models/dpcl.py:41-87 + utils/ops.py:323-324)."""

    @staticmethod
    def forward(ctx, u, V, inv, Y):
        out, ws = ops.dpcl_loss_fwd(V, Y)
        ctx.save_for_backward(V, inv, Y, ws)
        ctx.ushape = u.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        V, inv, Y, ws = ctx.saved_tensors
        du = ops.dpcl_loss_bwd(V, Y, ws, inv=inv, upstream=_c(dout))
        return du.view(ctx.ushape), None, None, None


def l2norm_keep(u, E):
    return L2NormKeep.apply(_c(u), E)


def dpcl_loss_from_u(u, V, inv, Y):
    B = V.shape[0]
    E = V.shape[-1]
    return DPCLLossFromU.apply(u, _c(V.detach()).reshape(B, -1, E), inv, _c(Y))


def front_filter(w, bases):
    return FrontFilter.apply(w, bases)


def front_conv(x, f, hop):
    return FrontConv.apply(x, f, hop)


def blstm(x, Kf, bf, Kb, bb):
    return BLSTMLayer.apply(_c(x), Kf, bf, Kb, bb)


def dense(x, W, b):
    return Dense.apply(_c(x), W, b)


def l2norm(u, E):
    return L2Norm.apply(_c(u), E)


def dpcl_loss(V, Y):
    return DPCLLoss.apply(_c(V), _c(Y))
