"""torch.autograd.Function wrappers around ams_hip.ops.

Autograd is used only to sequence the hand-written backward kernels; every forward and backward body is
a call into libams_hip.so.  Citations are to the reference call sites each op replaces.
"""
import torch
from torch.autograd import Function

from . import ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Overlap(object):
    """Weight-gradient products are off the backward critical path (only dx feeds the next layer's BPTT), while the
    recurrent step kernels are latency-bound and leave most CUs idle.  When enabled, dW / dU / db products are launched on
    a SIDE HIP stream and accumulate straight into the parameters' pre-assigned gradient views (the flat all-reduce buffer),
    so they overlap the next layer's recurrence; `join()` is called before the optimizer / all-reduce."""

    def __init__(self):
        self.enabled = False
        self.stream = None
        self.n_cap = 0
        self.deferred = []
        self.on_ready = None                     # FlatOptimizer.bucket_ready under AMS_DP_OVERLAP=1 (models/network.py::optimize)

    def side(self):
        if self.stream is None:
            import os
            self.stream = torch.cuda.Stream(priority=int(os.environ.get('AMS_SIDE_PRIORITY', '0')))
        return self.stream

    def usable(self, *params):
        return self.enabled and all(p.grad is not None and p.grad.stride(-1) == 1 for p in params)

    def fork(self, *tensors):
        s = self.side()
        s.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            t.record_stream(s)
        return s

    def cap(self, on, kind='dense'):
        import os
        # residency cap of side-stream products via a dynamic-LDS pad: 1 workgroup per CU for all of them, with the step kernels
        # at s_setprio 3.  (Measured history: before the GEMM's operand fetch was made branch-free a capped product was
        # latency-bound at 54 TFLOP/s and the dense dW product paid off at 2 per CU, 9.62 k vs 9.32 k mixtures/s; with the
        # branch-free fetch 1 per CU reaches 72 TFLOP/s alone and wins, 9.85 k vs 9.74 k; without the priority 9.52 k.)
        # With the ring recurrence (csrc/lstm_ring.hip, default) the BPTT is ONE resident, latency-bound launch whose hand-offs queue in
        # each CU's memory pipeline behind whatever streams there (MI355X_MICROARCH.md, handoff-1to1 by streaming waves): beside an
        # UNCAPPED dense weight-gradient product the top layer's BPTT took 900 us instead of 240 alone.  2 workgroups per CU (50 KB
        # pad) for every side product is the measured optimum with the PF=2 GEMM: pad 0 / 20 / 35 / 50 / 70 KB = 13.54 / 13.54 /
        # 13.29 / 14.02 / 13.66 k mixtures/s (same box, 100 steps); no overlap at all 12.99 k.
        ring = ops.LSTM_RING != '0'
        pad = int(os.environ.get('AMS_SIDE_LDS_PAD', '50000' if ring else '70000'))
        if kind == 'lstm':
            pad = int(os.environ.get('AMS_SIDE_LDS_PAD_LSTM', '50000' if ring else '70000'))
        if kind == 'lstm_last':
            pad = int(os.environ.get('AMS_SIDE_LDS_PAD_LAST', '50000' if ring else '40000'))
        tab = os.environ.get('AMS_SIDE_PADS')           # tuning aid: one pad per capped launch group, in issue order
        if on and tab:
            tab = [int(v) for v in tab.split(',')]
            pad = tab[min(self.n_cap, len(tab) - 1)]
            self.n_cap += 1
        ops.LDS_PAD[0] = pad if on else 0               # handed to every product launched until the next cap() (include/ams.h: lds_pad)

    def capped(self, kind='dense', on=True):
        """with OVERLAP.capped('lstm'): ...  -- the products inside carry the residency cap, and the cap is lifted again whatever
        happens inside (a product that raises must not leave every later critical-path product capped)."""
        ov = self

        class _Cap(object):
            def __enter__(self_):
                ov.cap(on, kind)

            def __exit__(self_, *exc):
                ov.cap(False)
        return _Cap()

    def ready(self, *params):
        """The side stream now holds the last writer of these parameters' gradients: a data-parallel optimizer may start exchanging them."""
        if self.on_ready is not None:
            self.on_ready(params, self.side())

    def defer(self, fn):
        """A piece of side-stream work that need not run in the window it was produced in (a column block of the dense weight
        gradient): run by flush() in a LATER window, behind that window's own products."""
        self.deferred.append(fn)

    def flush(self, n=1):
        """Run up to n deferred pieces on the side stream (the caller has forked it)."""
        s = self.side()
        with torch.cuda.stream(s):
            for _ in range(min(n, len(self.deferred))):
                self.deferred.pop(0)()

    def join(self):
        if self.deferred:
            s = self.side()
            s.wait_stream(torch.cuda.current_stream())
            self.flush(len(self.deferred))
        self.n_cap = 0
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


OVERLAP = _Overlap()


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
        # fp16x3 (bounds of the waveforms and of the filter: tagged by the staging launch / the frozen-filter cache, measured otherwise);
        # the launch leaves max |y| for the product that reads y
        am = (ops.amax_of(x), ops.amax_of(f)) if (ops.F16X3 and x.is_cuda) else None
        return ops.front_conv(x, f, hop, amax=am, measure=True)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        df = ops.front_conv_bwd_filter(x, _c(dy), ctx.W, ctx.hop) if ctx.needs_input_grad[1] else None
        return None, df, None


import os as _os
_ORDER = int(_os.environ.get('AMS_OVERLAP_ORDER', '2'))
# dense layer backward: 0 = dX first, dW capped beside the next recurrence (9.30 k mixtures/s); 1 = dW || dX uncapped (9.05 k);
# 2 = dW alone, then dX (8.97 k) -- measured on the B=64 step, kept as a tuning aid
_DENSE_MODE = int(_os.environ.get('AMS_DENSE_MODE', '0'))
_L1_TAIL = int(_os.environ.get('AMS_L1_TAIL', '0'))
# column blocks (in 256-column tiles) of the dense weight gradient: the first runs beside the top BPTT ring, the others one per later window
_DW_SPLIT = [int(v) for v in _os.environ.get('AMS_DW_SPLIT', '24,12,4').split(',') if v]


def _dw_cuts(N):
    tiles = N // 256
    if N % 256 or len(_DW_SPLIT) < 2 or tiles < sum(_DW_SPLIT):
        return [(0, N)]
    tot = sum(_DW_SPLIT)
    cuts, n0 = [], 0
    for i, f in enumerate(_DW_SPLIT):
        n1 = N if i == len(_DW_SPLIT) - 1 else n0 + (tiles * f // tot) * 256
        if n1 > n0:
            cuts.append((n0, n1))
        n0 = n1
    return cuts

class BLSTMLayer(Function):
    """utils/ops.py:358-383 (BasicLSTMCell x 2 directions, concat)."""

    @staticmethod
    def forward(ctx, x, Kf, bf, Kb, bb, last_capped=False):
        # fp16x3 products (amax_a / amax_b of the product entry points): bounds of the input and of the kernels (one bound over the optimizer's flat buffer; two
        # kernels measured separately have no common bound at hand and keep bf16x6)
        aw = ops.param_amax(Kf) if ops.F16X3 and x.is_cuda else None
        if aw is not None and aw is not ops.param_amax(Kb):
            aw = None
        ax = ops.amax_of(x) if aw is not None else None
        ctx.amax = (ax, aw) if aw is not None else None
        out, G, cst = ops.blstm_fwd(x, Kf, bf, Kb, bb, amax=ctx.amax)
        if x.is_cuda:
            ops.tag_amax(out, ops.amax_one(out.device))          # |tanh(c) sigmoid(o)| < 1
        ctx.last_capped = bool(last_capped)
        ctx.save_for_backward(x, Kf, Kb, out, G, cst)
        ctx.biases = (bf, bb)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, Kf, Kb, out, G, cst = ctx.saved_tensors
        bf, bb = ctx.biases
        need_dx = ctx.needs_input_grad[0]
        if OVERLAP.usable(Kf, Kb, bf, bb) and all(ctx.needs_input_grad[1:5]):
            B, T, D = x.shape
            dbpart = ops.blstm_bwd_recurrent(x, Kf, Kb, G, cst, _c(dout), amax_u=ctx.amax[1] if ctx.amax is not None else None)
            am_dx = am_w = None
            if ctx.amax is not None:
                az = ops.amax_of(G)                              # G now holds dZ; the ring left its bound, the step kernels do not
                am_dx = (az, ctx.amax[1])
                am_w = (ctx.amax[0], ops.amax_one(G.device), az)
            dx = None
            if _ORDER >= 2 and need_dx:
                dx = ops.blstm_bwd_dx(G, Kf, Kb, B, T, D, amax=am_dx)
            s = OVERLAP.fork(x, out, G, *([dbpart] if dbpart is not None else []))
            if not need_dx:
                # first layer: no recurrence follows -- both streams work on the weight gradients, uncapped
                if _L1_TAIL == 0:
                    with torch.cuda.stream(s):
                        ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='u', amax=am_w)
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='wx', dbpart=dbpart, amax=am_w)
                elif _L1_TAIL == 1:
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='wx', dbpart=dbpart, amax=am_w)
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='u', amax=am_w)
                else:
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='u', amax=am_w)
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='wx', dbpart=dbpart, amax=am_w)
                return None, None, None, None, None, None
            with torch.cuda.stream(s):
                with OVERLAP.capped('lstm'):
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='wx', dbpart=dbpart, amax=am_w)
                # the LAST capped product of the backward pass (recurrent-kernel gradient of the layer above the first one) ends
                # after the BPTT it hides behind: 2 workgroups per CU there (+0.6 %)
                with OVERLAP.capped('lstm_last' if ctx.last_capped else 'lstm'):
                    ops.blstm_bwd_weights(x, out, G, Kf.grad, bf.grad, Kb.grad, bb.grad, True, part='u', amax=am_w)
                OVERLAP.ready(Kf, bf, Kb, bb)
                OVERLAP.flush(1)                    # one deferred column block of the dense weight gradient per window
            if dx is None and need_dx:
                dx = ops.blstm_bwd_dx(G, Kf, Kb, B, T, D, amax=am_dx)
            return dx, None, None, None, None, None
        dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(x, Kf, Kb, out, G, cst, _c(dout), need_dx=need_dx,
                                                amax_u=ctx.amax[1] if ctx.amax is not None else None)
        return dx, dKf, dbf, dKb, dbb, None


class BLSTMLayerDropout(Function):
    """utils/ops.py:358-383 with --recurrent_dropout != 0 while training: each direction's BasicLSTMCell sits in a
    DropoutWrapper(cell, keep, keep, keep) -- independent masks per time step on the cell input, on the carried state (TF 1.4: both c
    and h) and on the cell output.  Per-step recurrence kernels (csrc/lstm.hip); the masks come in as tensors (ops.blstm_dropout_masks)."""

    @staticmethod
    def forward(ctx, x, Kf, bf, Kb, bb, m_in, m_h, m_c, m_out):
        masks = {'in': m_in, 'h': m_h, 'c': m_c, 'out': m_out}
        y, saved = ops.blstm_fwd_dropout(x, Kf, bf, Kb, bb, masks)
        ctx.save_for_backward(x, Kf, Kb, m_in, m_h, m_c, m_out, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Kf, Kb, m_in, m_h, m_c, m_out = ctx.saved_tensors[:7]
        saved = ctx.saved_tensors[7:]
        masks = {'in': m_in, 'h': m_h, 'c': m_c, 'out': m_out}
        dx, dKf, dbf, dKb, dbb = ops.blstm_bwd_dropout(_c(dy), x, Kf, Kb, saved, masks, need_dx=ctx.needs_input_grad[0])
        return dx, dKf, dbf, dKb, dbb, None, None, None, None


def blstm_dropout(x, Kf, bf, Kb, bb, keep, masks=None):
    """One BLSTM layer under the reference's dropout wrappers; masks default to a fresh draw (ops.blstm_dropout_masks)."""
    if masks is None:
        B, T, D = x.shape
        masks = ops.blstm_dropout_masks(B, T, D, Kf.shape[1] // 4, keep, x.device)
    return BLSTMLayerDropout.apply(_c(x), Kf, bf, Kb, bb, masks['in'], masks['h'], masks['c'], masks['out'])


class Dense(Function):
    """Conv1D with kernel width 1: u = x.W + b  (utils/ops.py:486-503).  x [..., Din], W [Din, Dout]."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.bias = b
        aw = ops.param_amax(W) if ops.F16X3 and x.is_cuda else None
        ctx.amax = (ops.amax_of(x), aw) if aw is not None else None         # fp16x3 products (amax_a / amax_b of the product entry points)
        return ops.dense_fwd(x, W, b, amax=ctx.amax)

    @staticmethod
    def backward(ctx, du):
        x, W = ctx.saved_tensors
        du2 = _c(du).view(-1, W.shape[1])
        x2 = x.reshape(-1, x.shape[-1])
        b = ctx.bias
        am_dx = am_dw = None
        if ctx.amax is not None:
            adu = ops.amax_of(du2)                               # the loss kernel that wrote dU left its bound (ops.dpcl_loss_bwd_u)
            am_dx, am_dw = (adu, ctx.amax[1]), (ctx.amax[0], adu)
        if OVERLAP.usable(W, b) and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dx = None
            if _DENSE_MODE == 0 and _ORDER >= 1 and ctx.needs_input_grad[0]:
                dx = ops.gemm(du2, W, transB=True, amax=am_dx).view(x.shape)
            s = OVERLAP.fork(x2, du2)
            Nw = W.shape[1]
            cuts = _dw_cuts(Nw) if (_DENSE_MODE == 0 and ctx.needs_input_grad[0] and W.grad.stride(0) == Nw) else [(0, Nw)]

            def piece(n0, n1):
                # dW[:, n0:n1] = x^T dU[:, n0:n1] and db[n0:n1] = colsum from ONE pass over those columns of dU
                Wg = W.grad if (n0, n1) == (0, Nw) else W.grad[:, n0:n1]
                dus = du2 if (n0, n1) == (0, Nw) else du2[:, n0:n1]
                with OVERLAP.capped('dense', _DENSE_MODE == 0):
                    fused = ops.gemm_at_b_colsum(x2, dus, Wg, b.grad[n0:n1], accumulate=True, amax=am_dw, ldc=W.grad.stride(0))
                    if not fused:
                        ops.gemm(x2, dus, transA=True, out=Wg, accumulate=True, amax=am_dw, M=x2.shape[1], N=n1 - n0, K=x2.shape[0],
                                 lda=x2.stride(0), ldb=du2.stride(0), ldc=W.grad.stride(0))
                if not fused:
                    ops.colsum_into(_c(dus), b.grad[n0:n1], True)
            with torch.cuda.stream(s):
                piece(*cuts[0])
            # The product (394 us beside the top layer's 300-us BPTT ring) used to run 130 us into the next layer's dX, which then
            # took 151 us instead of 73: its trailing column blocks run in the LATER BPTT windows instead, behind those windows' own
            # weight-gradient products (the windows of the two lower layers have ~55 us of slack each).
            for i, c in enumerate(cuts[1:]):
                last = i == len(cuts) - 2
                OVERLAP.defer(lambda c=c, last=last: (piece(*c), OVERLAP.ready(W, b) if last else None))
            if len(cuts) == 1:
                OVERLAP.ready(W, b)
            if _DENSE_MODE == 2 and ctx.needs_input_grad[0]:
                # the weight-gradient product runs FIRST and alone (uncapped), dX after it: nothing of the dense layer is left
                # on the side stream when the recurrence below starts
                torch.cuda.current_stream().wait_stream(s)
            if dx is None and ctx.needs_input_grad[0]:
                dx = ops.gemm(du2, W, transB=True, amax=am_dx).view(x.shape)
            return dx, None, None
        dx = ops.gemm(du2, W, transB=True, amax=am_dx).view(x.shape) if ctx.needs_input_grad[0] else None
        dW = ops.gemm(x2, du2, transA=True, amax=am_dw) if ctx.needs_input_grad[1] else None
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


class DPCLLossU(Function):
    """l2-normalise + DPCL loss in ONE pass over u (dense output before Normalize): V is never materialised in a
    training step; backward recomputes v = u/|u| (models/dpcl.py:41-87 + utils/ops.py:323-324).
    Two outputs -- the cost as a 1-element tensor and all 4 terms (cost + the three summaries) -- so that the training step
    differentiates the cost output directly: slicing terms[0:1] and selecting [0] put five fill/copy launches between the
    loss kernels of every step."""

    @staticmethod
    def forward(ctx, u, Y):
        out, inv, _, ws = ops.dpcl_loss_fwd_u(u, Y)
        ctx.save_for_backward(u, inv, Y, ws)
        ctx.set_materialize_grads(False)
        return out.narrow(0, 0, 1), out

    @staticmethod
    def backward(ctx, dcost, dterms):
        u, inv, Y, ws = ctx.saved_tensors
        if dterms is not None:                       # someone differentiated a summary term: only term 0 carries a gradient
            up = _c(dterms).clone()
            if dcost is not None:
                up[0:1] += dcost
        elif dcost is None:
            return None, None
        else:
            up = _c(dcost)
        return ops.dpcl_loss_bwd_u(u, Y, inv, ws, upstream=up), None


def dpcl_loss_u(u, Y, E):
    """u [B, ..., F*E] (column = f*E + e) -> (cost [1], the 4 loss terms [4]); Y [B, TF, S]."""
    B = u.shape[0]
    return DPCLLossU.apply(_c(u).reshape(B, -1, E), _c(Y))


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


def blstm(x, Kf, bf, Kb, bb, last_capped=False):
    return BLSTMLayer.apply(_c(x), Kf, bf, Kb, bb, last_capped)


def dense(x, W, b):
    return Dense.apply(_c(x), W, b)


def l2norm(u, E):
    return L2Norm.apply(_c(u), E)


def dpcl_loss(V, Y):
    return DPCLLoss.apply(_c(V), _c(Y))


# =====================================================================================================
# Synthesis, waveform costs, masks, STFT, L41, k-means
# =====================================================================================================
class SynthStrided(Function):
    """tf.nn.conv2d_transpose stride=hop SAME (models/adapt.py:236-243): z [R,T,N], f2 [W,N] -> [R,L].
    MFMA GEMM (z . f2^T) + gather-form overlap-add; backward = analysis conv of dout (+ its filter gradient)."""

    @staticmethod
    def forward(ctx, z, f2, hop, L):
        R, T, N = z.shape
        W = f2.shape[0]
        Tn = -(-L // hop)
        pl = max((Tn - 1) * hop + W - L, 0) // 2
        frames = ops.gemm(z.view(R * T, N), f2, transB=True)              # [R*T, W]
        out = ops.overlap_add(frames, R, T, W, L, hop, pl)
        ctx.save_for_backward(z, f2)
        ctx.hop, ctx.L = hop, L
        return out

    @staticmethod
    def backward(ctx, dout):
        z, f2 = ctx.saved_tensors
        dout = _c(dout)
        dz = ops.front_conv(dout, f2, ctx.hop)[:, :z.shape[1]] if ctx.needs_input_grad[0] else None
        df2 = ops.front_conv_bwd_filter(dout, _c(z), f2.shape[0], ctx.hop) if ctx.needs_input_grad[1] else None
        if dz is not None and not dz.is_contiguous():
            dz = dz.contiguous()
        return dz, df2, None, None


def synth_strided(z, f2, hop, L):
    return SynthStrided.apply(_c(z), f2, hop, L)


def upsample_nearest(z, P):
    """avg-pool back path: tf.keras UpSampling2D((1,P)) (adapt.py:226-228) -- index glue."""
    return z.repeat_interleave(P, dim=1)


class PairStats(Function):
    """One pass over the waveforms -> [B, 2S^2+3S+1] table (D | Q | Na | Nt | Tm | Nm), see csrc/synth.hip."""

    @staticmethod
    def forward(ctx, target, est, mix):
        ctx.save_for_backward(target, est)
        return ops.pair_stats_fwd(target, est, mix)

    @staticmethod
    def backward(ctx, g):
        target, est = ctx.saved_tensors
        return None, ops.pair_stats_bwd(target, est, _c(g)), None


def pair_stats(target, est, mix):
    """Returns dict of views D,Q [B,S,S], Na,Nt,Tm [B,S], Nm [B]; 'table' is the [B, 2S^2+3S+1] tensor they are views of."""
    B, S, L = est.shape
    st = PairStats.apply(_c(target), _c(est), _c(mix) if mix is not None else None)
    SS = S * S
    return {'D': st[:, :SS].reshape(B, S, S), 'Q': st[:, SS:2 * SS].reshape(B, S, S), 'Na': st[:, 2 * SS:2 * SS + S],
            'Nt': st[:, 2 * SS + S:2 * SS + 2 * S], 'Tm': st[:, 2 * SS + 2 * S:2 * SS + 3 * S], 'Nm': st[:, 2 * SS + 3 * S],
            'table': st, 'S': S, 'L': L}


class PairCombine(Function):
    """The [B,S,S] arithmetic between the pair table and the scalar costs in ONE launch each way (csrc/synth.hip
    pair_combine_*): mode 0 pre-training (adapt.py:321-330), 1 PIT squared error (adapt.py:404-431, network.py:662-724),
    2 Adapt.cost non-pretraining branch with the cross-batch SDR table D2 [S,B,B] (adapt.py:339-365).  As torch glue these
    were 30-80 launches of ~5 us on 64 x 2 x 2 numbers: 12 % of the fine-tuning step, 11 % of the pre-training step."""

    @staticmethod
    def forward(ctx, table, D2, S, mode, cl, cs):
        perms = _perm_table32(S, table.device) if mode != 0 else None
        out, pbest, jbest = ops.pair_combine_fwd(table, D2, perms, S, mode, cl, cs)
        ctx.save_for_backward(table, D2, perms, pbest, jbest)
        ctx.cfg = (S, mode, cl, cs)
        return out

    @staticmethod
    def backward(ctx, g):
        table, D2, perms, pbest, jbest = ctx.saved_tensors
        S, mode, cl, cs = ctx.cfg
        gst, gD2 = ops.pair_combine_bwd(table, D2, perms, _c(g), pbest, jbest, S, mode, cl, cs)
        return gst, gD2, None, None, None, None


def _log10(x):
    return torch.log(x) / 2.302585092994046


def _diag(t):
    return torch.diagonal(t, dim1=1, dim2=2)


_PERM_CACHE = {}


def _perm_table(S, device):
    """[S!, S] permutations in lexicographic order (App. A-14); cached per device so a hipGraph capture never sees the upload."""
    key = (S, str(device))
    if key not in _PERM_CACHE:
        from itertools import permutations
        _PERM_CACHE[key] = torch.tensor(list(permutations(range(S))), dtype=torch.long, device=device)
    return _PERM_CACHE[key]


def _perm_table32(S, device):
    key = (S, str(device), 'i32')
    if key not in _PERM_CACHE:
        _PERM_CACHE[key] = _perm_table(S, device).to(torch.int32).contiguous()
    return _PERM_CACHE[key]


def sdr_improvement(x_mix, s_target, s_approx, with_perm=False):
    """Network.sdr_improvement (network.py:196-221) for [B,S,L] operands (identity pairing)."""
    st = pair_stats(s_target, s_approx, x_mix)
    D, Na, Nt, Tm, Nm = _diag(st['D']), st['Na'], st['Nt'], st['Tm'], st['Nm'].unsqueeze(1)
    sep = 10.0 * _log10(1.0 / ((Nt * Na) / (D * D) - 1.0))
    non = 10.0 * _log10(1.0 / ((Nt * Nm) / (Tm * Tm) - 1.0))
    val = (sep - non).mean(dim=-1)
    val = val.mean(dim=-1) if not with_perm else val.mean(dim=0)
    return val, (Nt * Na) / (D * D + 1e-12)


def pretrain_cost(x_mix, x_non_mix, back, want_imp=True, st=None):
    """Adapt.cost pretraining branch (adapt.py:321-330) -> tensor [3] = (l2, sdr, sdr_improvement); [2] without the (summary-only,
    gradient-free) improvement term.  `st`: a pair_stats table computed earlier in the same run."""
    st = pair_stats(x_non_mix, back, x_mix) if st is None else st
    out = PairCombine.apply(st['table'], None, st['S'], 0, 1.0, 1.0)            # l2 = mean_b sum_s Q_ss, sdr = mean Nt Na / (D_ss^2 + 1e-12)
    if not want_imp:
        return out
    return torch.cat([out, pretrain_improvement(st).reshape(1)])


def pretrain_improvement(st):
    """SDR improvement of the identity pairing (network.py:200-219) from a pair_stats table; no gradient."""
    with torch.no_grad():
        D, Na, Nt, Tm, Nm = _diag(st['D']), st['Na'], st['Nt'], st['Tm'], st['Nm'].unsqueeze(1)
        sep = 10.0 * _log10(1.0 / ((Nt * Na) / (D * D) - 1.0))
        non = 10.0 * _log10(1.0 / ((Nt * Nm) / (Tm * Tm) - 1.0))
        return (sep - non).mean()


class CrossDots(Function):
    """D2[i,j,s] = <t[i,s,:], a[j,s,:]>: the cross-batch table the reference's fine-tune SDR term builds by
    broadcasting [B,1,S,L] against [B,S,L] (quirk C-3, adapt.py:361-365).  One MFMA GEMM per speaker."""

    @staticmethod
    def forward(ctx, t, a):
        B, S, L = a.shape
        out = torch.empty((S, B, B), dtype=torch.float32, device=a.device)
        # speaker s: rows t[:, s, :] . a[:, s, :]^T -- S products of one shape, one launch (operand s starts s*L floats further)
        ops.gemm_batched(t, a, out, S, L, L, B * B, False, True, B, B, L, S * L, S * L, B)
        ctx.save_for_backward(t)
        ctx.shape = (B, S, L)
        return out.permute(1, 2, 0)

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        B, S, L = ctx.shape
        gs = _c(g.permute(2, 0, 1))                                        # [S, i, j]
        da = torch.empty((B, S, L), dtype=torch.float32, device=t.device)
        # da[j, s, :] = sum_i g[i,j,s] t[i,s,:]
        ops.gemm_batched(gs, t, da, S, B * B, L, L, True, False, B, L, B, B, S * L, S * L)
        return None, da


def pit_adapt_tables(x_mix, x_non_mix, back):
    """What both the cost and its summary need: the pair table and the cross-batch dots D2[i,j,s] = <t[i,s], a[j,s]> (quirk C-3)."""
    return pair_stats(x_non_mix, back, x_mix), CrossDots.apply(_c(x_non_mix), _c(back))


def pit_cost_adapt(x_mix, x_non_mix, back, want_imp=True, tables=None):
    """Adapt.cost non-pretraining branch (adapt.py:339-365) -> tensor [3] = (l2, sdr, sdr_improvement); [2] without the
    (summary-only, gradient-free) improvement term."""
    B, S, L = back.shape
    st, D2 = pit_adapt_tables(x_mix, x_non_mix, back) if tables is None else tables
    # l2 = mean_b min_p sum_s Q[b,s,p(s)] / L;  sdr = mean_i sum_s min_j Nt[i,s] Na[j,s] / (D2[i,j,s]^2 + 1e-12)
    out = PairCombine.apply(st['table'], D2.permute(2, 0, 1), S, 2, 1.0 / L, 1.0)   # D2 is a [i,j,s] view of a dense [S,B,B] tensor
    if not want_imp:
        return out
    return torch.cat([out, pit_adapt_improvement(x_mix, x_non_mix, (st, D2)).reshape(1)])


def pit_adapt_improvement(x_mix, x_non_mix, tables):
    """The SDR-improvement summary of the same branch (network.py:200-219 on the broadcast operands); no gradient."""
    st, D2 = tables
    B, S = st['Nt'].shape
    L = x_mix.shape[-1]
    with torch.no_grad():
        Nm = st['Nm']
        D2 = D2.detach()
        sep = 10.0 * _log10(1.0 / ((st['Nt'].unsqueeze(1) * st['Na'].unsqueeze(0)) / (D2 * D2) - 1.0))
        # mix term broadcasts the same way: target i against mix j
        Tm2 = CrossDots.apply(_c(x_non_mix), _c(x_mix.unsqueeze(1).expand(B, S, L).contiguous()))
        non = 10.0 * _log10(1.0 / ((st['Nt'].unsqueeze(1) * Nm.view(1, B, 1)) / (Tm2 * Tm2) - 1.0))
        return (sep - non).mean(dim=-1).mean(dim=0).max()


def pit_l2(x_non_mix, est, reduce_l, reduce_s, scale=1.0):
    """Generic PIT squared error from the pair table (cost_finetuning / enhance_cost)."""
    B, S, L = est.shape
    st = pair_stats(x_non_mix, est, None)
    cl = float(scale) / L if reduce_l == 'mean' else float(scale)
    return PairCombine.apply(st['table'], None, S, 1, cl, 1.0 if reduce_s == 'sum' else 1.0 / S)[0:1]


class OverlapMetric(Function):
    @staticmethod
    def forward(ctx, y, B, S):
        ctx.save_for_backward(y)
        ctx.B, ctx.S = B, S
        return ops.overlap_metric_fwd(y, B, S)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.overlap_metric_bwd(y, _c(g), ctx.B, ctx.S), None, None


def overlap_metric(y, B, S):
    return OverlapMetric.apply(_c(y), B, S)


class PretrainSeparator(Function):
    """Adapt.separator pretraining branch (adapt.py:173-196)."""

    @staticmethod
    def forward(ctx, y, B, S, separation):
        ctx.cfg = (B, S, separation)
        return ops.pretrain_separator_fwd(y, B, S, separation)

    @staticmethod
    def backward(ctx, dout):
        B, S, separation = ctx.cfg
        return ops.pretrain_separator_bwd(_c(dout), B, S, separation), None, None, None


def pretrain_separator(y, B, S, separation):
    """y [B(1+S), T, N] -> [B*S, T, N]."""
    return PretrainSeparator.apply(_c(y), B, S, separation)


class SumSq(Function):
    """sum x^2 -> [1]  (tf.nn.l2_loss = sum x^2 / 2, adapt.py:312); backward 2 x g on the device (ams_sumsq_bwd)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.sumsq(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.sumsq_bwd(x, _c(g).reshape(1), 1.0, 0)


def sumsq(x):
    return SumSq.apply(x) if x.requires_grad else ops.sumsq(_c(x))


class NegativeEnergy(Function):
    """mean_b sum_{t,n} min(y, 0)^2 -> [1]  (tf.square(tf.where(front < 0, front, 0)), adapt.py:314-316)."""

    @staticmethod
    def forward(ctx, y):
        y = _c(y)
        ctx.save_for_backward(y)
        return ops.negative_energy_fwd(y.reshape(y.shape[0], -1))

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return ops.sumsq_bwd(y, _c(g).reshape(1), 1.0 / y.shape[0], 1)


def negative_energy(y):
    return NegativeEnergy.apply(y).reshape(())


class SparseKL(Function):
    """sum kl_div(p, p_hat), p_hat = sum_b |y|  (adapt.py:130-132, utils/ops.py:46-54).  Under data parallelism p_hat is a batch
    SUM that enters a non-linear term: it is all-reduced before the KL, and the backward carries `world` because the gradient
    exchange averages per-rank gradients (tests/test_dist_gloo.py::test_sparsity_term_gradient_matches_single_process)."""

    @staticmethod
    def forward(ctx, y2, p, dist):
        y2 = _c(y2)
        p_hat = ops.abs_colsum(y2)
        world = 1.0
        if dist is not None and getattr(dist, 'enabled', False):
            dist.all_reduce_sum(p_hat)
            world = float(dist.world_size)
        ctx.save_for_backward(y2, p_hat)
        ctx.p, ctx.world = float(p), world
        return ops.kl_sparsity_fwd(p_hat, p)

    @staticmethod
    def backward(ctx, g):
        y2, p_hat = ctx.saved_tensors
        return ops.kl_sparsity_bwd(y2, p_hat, _c(g).reshape(1), ctx.p, ctx.world), None, None


def sparse_kl(y2, p, dist=None):
    return SparseKL.apply(y2, p, dist).reshape(())


class ApplyMasks(Function):
    """separated = X_input * masks, rows (b,s)  (models/network.py:577-581)."""

    @staticmethod
    def forward(ctx, X, masks):
        ctx.save_for_backward(X)
        ctx.S = masks.shape[2]
        return ops.apply_masks_fwd(X, masks)

    @staticmethod
    def backward(ctx, dsep):
        (X,) = ctx.saved_tensors
        return None, ops.apply_masks_bwd(X, _c(dsep), ctx.S)


def apply_masks(X_input, masks):
    """X_input [B,T,F], masks [B,TF,S] -> [B*S, T, F]."""
    B, T, Fq = X_input.shape
    sep = ApplyMasks.apply(_c(X_input).view(B, T * Fq), _c(masks))
    return sep.view(-1, T, Fq)


class L41Loss(Function):
    """models/L41.py:150-178 on already gathered speaker vectors.  from_u: emb is the network output before Normalize(3) -- normalised
    inside the loss pass, gradient returned w.r.t. it (include/ams.h: emb_is_u)."""

    @staticmethod
    def forward(ctx, emb, y, vspk, from_u=False):
        ctx.save_for_backward(emb, y, vspk)
        ctx.from_u = bool(from_u)
        return ops.l41_loss_fwd(emb, y, vspk, from_u)

    @staticmethod
    def backward(ctx, g):
        emb, y, vspk = ctx.saved_tensors
        demb, dvs = ops.l41_loss_bwd(emb, y, vspk, _c(g), ctx.from_u)
        return demb, None, dvs, None


class L41LossNS(Function):
    """models/L41.py:150-178 plus the negative-sampling term (:143-147,165-166) on gathered vectors."""

    @staticmethod
    def forward(ctx, emb, y, vspk, negs, ns_rate, from_u=False):
        ctx.save_for_backward(emb, y, vspk, negs)
        ctx.ns_rate, ctx.from_u = float(ns_rate), bool(from_u)
        return ops.l41_loss_ns_fwd(emb, y, vspk, negs, ns_rate, from_u)

    @staticmethod
    def backward(ctx, g):
        emb, y, vspk, negs = ctx.saved_tensors
        demb, dvs, dnegs = ops.l41_loss_ns_bwd(emb, y, vspk, negs, _c(g), ctx.ns_rate, ctx.from_u)
        return demb, None, dvs, dnegs, None, None


class L41Speakers(Function):
    """tf.nn.l2_normalize(speaker_centroids, 1) + gather_nd by I (models/L41.py:60-68); backward scatters through the
    normalise Jacobian in a fixed order."""

    @staticmethod
    def forward(ctx, table, I, normalize):
        I32 = I.to(torch.int32).contiguous()
        ctx.save_for_backward(table, I32)
        ctx.normalize = normalize
        return ops.l41_speaker_fwd(table, I32, normalize)

    @staticmethod
    def backward(ctx, d_vs):
        table, I32 = ctx.saved_tensors
        return ops.l41_speaker_bwd(table, I32, _c(d_vs), ctx.normalize), None, None


def l41_knearest(speaker_vectors, I, K, normalize):
    """Indices [B,S,K] of the K rows of the (normalised) speaker table nearest to each mixture speaker (L41.py:95-104: tf.nn.top_k
    of the dot products; the speaker itself is among them).  Index selection only -- no gradient (TF's top_k indices carry none)."""
    with torch.no_grad():
        table = speaker_vectors.detach()
        if normalize:
            table = table * torch.rsqrt(torch.clamp((table * table).sum(1, keepdim=True), min=1e-12))
        prod = table[I.long()] @ table.t()                                            # [B,S,|S|]
        return torch.topk(prod, int(K), dim=2, sorted=False).indices.to(torch.int32).contiguous()


def l41_random_negatives(I, tot_speakers, K):
    """Indices [B,1,K]: K speakers drawn without replacement among those NOT in the mixture, one set per utterance (L41.py:123-134,
    tf.random_shuffle of the available indices, first K).  Drawn on the device from torch's generator (graph-capturable, advances
    on every replay); TensorFlow's own stream is not reproducible -- the distribution (a uniform K-subset) is."""
    B = I.shape[0]
    score = torch.rand(B, int(tot_speakers), device=I.device)
    score.scatter_(1, I.long(), 2.0)                                                   # mixture speakers sort last
    return torch.topk(score, int(K), dim=1, largest=False).indices.to(torch.int32).reshape(B, 1, int(K)).contiguous()


def l41_loss(emb, y, speaker_vectors, I, normalize, neg_idx=None, ns_rate=0.1, from_u=False):
    """emb [B,T,F,E], y [B,T,F,S]; speaker_vectors [251,E], I [B,S].  neg_idx (optional, int [B,NSEL,K], NSEL = 1 or S): rows of the
    speaker table used as negatives (--sampling, L41.py:69-147): the cost gains ns_rate * mean_k -log(sigmoid(-<neg_k, emb>)).
    from_u: emb is the network output BEFORE its Normalize(3) layer (any shape [B, ..., k*E]); the loss normalises it in its own pass."""
    B, E = emb.shape[0], speaker_vectors.shape[1]
    S = y.shape[-1]
    table = _c(speaker_vectors)
    if neg_idx is None:
        vs = L41Speakers.apply(table, I, bool(normalize))                             # [B,S,E]
        return L41Loss.apply(_c(emb).reshape(B, -1, E), _c(y).reshape(B, -1, S), vs, bool(from_u))
    NSEL, K = neg_idx.shape[1], neg_idx.shape[2]
    # ONE gather through the normalise Jacobian for the mixture speakers and the negatives: [B, S + NSEL*K, E]
    allidx = torch.cat([I.to(torch.int32).reshape(B, S), neg_idx.to(torch.int32).reshape(B, NSEL * K)], dim=1).contiguous()
    allv = L41Speakers.apply(table, allidx, bool(normalize))
    vs = allv[:, :S].contiguous()
    negs = allv[:, S:].reshape(B, NSEL, K, E).contiguous()
    return L41LossNS.apply(_c(emb).reshape(B, -1, E), _c(y).reshape(B, -1, S), vs, negs, float(ns_rate), bool(from_u))


class EnhanceOutput(Function):
    """Separator.enhance output stage (network.py:640-660): act over the speaker axis, times X; two layouts out."""

    @staticmethod
    def forward(ctx, u, X, S, nonlinearity):
        cost_in, sep = ops.enhance_output_fwd(u, X, S, nonlinearity)
        ctx.save_for_backward(u, X)
        ctx.S, ctx.nl = S, nonlinearity
        return cost_in, sep

    @staticmethod
    def backward(ctx, d_cost_in, d_sep):
        u, X = ctx.saved_tensors
        dc = _c(d_cost_in) if d_cost_in is not None else None
        ds = _c(d_sep) if d_sep is not None else None
        return ops.enhance_output_bwd(u, X, ctx.S, ctx.nl, dc, ds), None, None, None


def enhance_output(u, X, S, nonlinearity):
    """u [B*S,T,F], X [B,T,F] -> (cost_in [B,TF,S], separated [B,S,TF])."""
    return EnhanceOutput.apply(_c(u), _c(X), S, nonlinearity)


def one_hot_masks(labels, S):
    """tf.one_hot(labels, S, 1.0, 0.0) (network.py:569) -- integer glue, no gradient."""
    return (labels.long().unsqueeze(-1) == torch.arange(S, device=labels.device)).float().contiguous()


class KMeansSoft(Function):
    """Soft k-means (beta set) with gradient to the embeddings (needed by the front_*_finetuning recipes)."""

    @staticmethod
    def forward(ctx, X, init_idx, C, tries, iterations, beta, w, assign_at_end, normalize_input, faithful_tile, from_u=False):
        # from_u: X is the embedding network's output BEFORE its Normalize layer (models/dpcl.py:32) and normalize_input is on: both
        # normalisations in one pass here, both Jacobians in the pass that writes dX (the unit-norm tensor in between is never stored)
        b, L, E = X.shape
        inv0 = None
        if normalize_input and from_u:
            xn, inv0, inv = ops.l2norm2_fwd(X.view(b, L * E), E)
            xn = xn.view(b, L, E)
        elif normalize_input:
            xn, inv = ops.l2norm_fwd(X.view(b, L * E), E)
            xn = xn.view(b, L, E)
        else:
            xn, inv = X, None
        sel, out, best, trace = ops.kmeans_run(xn, init_idx, C, tries, iterations, beta, w, assign_at_end, faithful_tile)
        extra = [t for t in (inv, inv0, w) if t is not None]
        ctx.save_for_backward(xn, init_idx, best, *extra, *trace)
        ctx.cfg = (C, tries, iterations, beta, inv is not None, w is not None, assign_at_end, faithful_tile, inv0 is not None)
        ctx.mark_non_differentiable(best)
        return sel, out, best

    @staticmethod
    def backward(ctx, dsel, dout, _dbest):
        from .kmeans_bwd import soft_bwd
        C, tries, iterations, beta, has_inv, has_w, assign_at_end, faithful_tile, has_inv0 = ctx.cfg
        saved = list(ctx.saved_tensors)
        xn, init_idx, best = saved[:3]
        k = 3
        inv = saved[k] if has_inv else None
        k += int(has_inv)
        inv0 = saved[k] if has_inv0 else None
        k += int(has_inv0)
        w = saved[k] if has_w else None
        k += int(has_w)
        dX = soft_bwd(xn, inv, init_idx, best, w, saved[k:], dsel, dout, C, tries, iterations, beta, assign_at_end, faithful_tile, inv0=inv0)
        return (dX,) + (None,) * 10


def kmeans(X, init_idx, C, tries, iterations, beta, w, assign_at_end, normalize_input=True, faithful_tile=True, pre_norm=None):
    """KMeans.network (Kmeans_2.py:86-111).  Returns (centroids [b,C,E], labels, best_try).
    pre_norm: a callable returning u with X = l2-normalise(u) over E (same shape) -- a soft k-means under gradient that normalises its
    input takes u instead of X (see KMeansSoft, from_u); X may then be a callable too and is not evaluated."""
    if pre_norm is not None and beta is not None and normalize_input and torch.is_grad_enabled():
        u = _c(pre_norm())
        if u.requires_grad:
            return KMeansSoft.apply(u, init_idx, C, tries, iterations, beta, w, assign_at_end, True, faithful_tile, True)
    if pre_norm is not None and normalize_input and callable(X) and (beta is None or not torch.is_grad_enabled()):
        # no gradient wanted: the Normalize layer and the k-means' own normalisation in ONE pass over u (the bits of the two passes);
        # the once-normalised tensor is not evaluated
        with torch.no_grad():
            u = _c(pre_norm())
            xn = ops.l2norm_kmeans_normalize(u, u.shape[-1])
            sel, out, best, _ = ops.kmeans_run(xn, init_idx, C, tries, iterations, beta, w, assign_at_end, faithful_tile)
        return sel, out, best
    if callable(X):
        X = X()
    X = _c(X)
    if beta is None or not (X.requires_grad and torch.is_grad_enabled()):
        with torch.no_grad():
            xn = ops.kmeans_normalize(X) if normalize_input else X
            sel, out, best, _ = ops.kmeans_run(xn, init_idx, C, tries, iterations, beta, w, assign_at_end, faithful_tile)
        return sel, out, best
    return KMeansSoft.apply(X, init_idx, C, tries, iterations, beta, w, assign_at_end, normalize_input, faithful_tile)


def stft_mag_phase(x, W, hop, want_phase=True):
    """tf.contrib.signal.stft(frame_length=W, frame_step=hop, fft_length=W) -> (|.| [R,T,F], unit phasor [R*T,2F]).
    No gradient: the waveforms carry none on any reference path."""
    from .stft_host import dft_matrices
    x = _c(x)
    R, L = x.shape
    T = 1 + (L - W) // hop
    F_ = W // 2 + 1
    D, _ = dft_matrices(W, hop, x.device)
    with torch.no_grad():
        ri = ops.frames_matmul(x, D, hop, T, 0)                          # [R*T, 2F] = [Re | Im]
        mag, ph = ops.cplx_mag_phase(ri, F_, want_phase)
    return mag.view(R, T, F_), ph


class ISTFT(Function):
    """inverse_stft with the mixture phase re-attached (network.py:589-603): cplx_apply -> DFT^-1 GEMM -> overlap-add."""

    @staticmethod
    def forward(ctx, sep, phasor, Dinv, W, hop, S):
        BS, T, F_ = sep.shape
        z = ops.cplx_apply_fwd(sep.view(BS * T, F_), phasor, S, T)       # [BS*T, 2F]
        frames = ops.gemm(z, Dinv)                                       # [BS*T, W]
        L = (T - 1) * hop + W
        out = ops.overlap_add(frames, BS, T, W, L, hop, 0)
        ctx.save_for_backward(phasor, Dinv)
        ctx.cfg = (BS, T, F_, W, hop, S, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        phasor, Dinv = ctx.saved_tensors
        BS, T, F_, W, hop, S, L = ctx.cfg
        # d frames[r,t,n] = dout[r, t*hop + n]  ->  a framed product with Dinv^T gives dz directly
        dz = ops.frames_matmul(_c(dout), _dinv_t(Dinv), hop, T, 0)       # [BS*T, 2F]
        dsep = ops.cplx_apply_bwd(dz, phasor, S, T)
        return dsep.view(BS, T, F_), None, None, None, None, None


_DINV_T = {}


def _dinv_t(Dinv):
    k = Dinv.data_ptr()
    if k not in _DINV_T:
        _DINV_T[k] = Dinv.t().contiguous()
    return _DINV_T[k]


def istft(sep, phasor, W, hop, S):
    from .stft_host import dft_matrices
    _, Dinv = dft_matrices(W, hop, sep.device)
    return ISTFT.apply(_c(sep), phasor, Dinv, W, hop, S)


def front_maxpool(x, f, P, hop):
    from .pooling import front_maxpool as _f
    return _f(x, f, P, hop)


def synth_unpool(vals, argmax_mix, f2, L, S, P=None, hop=None):
    from .pooling import synth_unpool as _f
    return _f(vals, argmax_mix, f2, L, P, hop, S)


def front_avgpool(x, f, P):
    from .pooling import front_avgpool as _f
    return _f(x, f, P)


def synth_avgpool(z, f2, P, L):
    from .pooling import synth_avgpool as _f
    return _f(z, f2, P, L)
