"""Oracle: adaptive analysis / synthesis filterbank (reference models/adapt.py).

Test infrastructure only -- see oracle/__init__.py.  Layouts:
  x        [Bt, L]          raw waveforms (rows 0..B-1 mixtures, then (b,s) row-major;
                            reference models/adapt.py:43,47)
  w        [W]              'front/window/w'          (adapt.py:104)
  bases    [W, N]           'front/bases/bases'       (adapt.py:105)
  y        [Bt, T', N]      front output (adapt.py:126 transposes to [Bt,T',N,1])
"""
import numpy as np


def same_pads(L, W, stride):
    """TF 'SAME' padding arithmetic (tf.nn.conv2d at adapt.py:115,122)."""
    out = -(-L // stride)
    pad_total = max((out - 1) * stride + W - L, 0)
    pl = pad_total // 2
    return out, pl, pad_total - pl


def front_filter(w, bases):
    """f[k,n] = |w[k]| * bases[k,n]   (adapt.py:106, :234)."""
    return np.abs(w)[:, None] * bases


def front_filter_bwd(w, bases, df):
    """Gradients of front_filter."""
    dbases = np.abs(w)[:, None] * df
    dw = np.sign(w) * np.sum(bases * df, axis=1)
    return dw, dbases


def _frames(x, W, stride, T, pl):
    """[Bt, T, W] view of zero padded x: frame t starts at sample t*stride - pl."""
    Bt, L = x.shape
    need = (T - 1) * stride + W
    xp = np.zeros((Bt, max(need, pl + L)), dtype=x.dtype)
    xp[:, pl:pl + L] = x
    s0, s1 = xp.strides
    return np.lib.stride_tricks.as_strided(xp, shape=(Bt, T, W), strides=(s0, stride * s1, s1), writeable=False)


def conv_strided(x, f, hop):
    """Path A (default): tf.nn.conv2d stride=hop SAME (adapt.py:122).

    y[b,t,n] = sum_k xpad[b, t*hop + k - pl] * f[k,n]  -- cross-correlation, no flip.
    """
    W, N = f.shape
    T, pl, _ = same_pads(x.shape[1], W, hop)
    return _frames(x, W, hop, T, pl) @ f


def conv_strided_bwd_filter(x, dy, W, hop):
    """df[k,n] = sum_{b,t} xpad[b,t*hop+k-pl] * dy[b,t,n]  (SURVEY Appendix D-1)."""
    T, pl, _ = same_pads(x.shape[1], W, hop)
    fr = _frames(x, W, hop, T, pl)
    return np.einsum('btk,btn->kn', fr, dy)


def conv_dense(x, f):
    """Stride-1 SAME conv (adapt.py:115,119): X[b,l,n], l in [0,L)."""
    return conv_strided(x, f, 1)


def maxpool_with_argmax(X, P, hop):
    """tf.nn.max_pool_with_argmax ksize [1,1,P,1] strides [1,1,hop,1] VALID (adapt.py:116-117).

    X [Bt, L, N] -> y [Bt, T', N], argmax int64 [Bt, T', N] with the TF-1.x GPU
    convention: flattened index WITHOUT the batch term, l*N + n (SURVEY Appendix A-3;
    the reference's unpool prepends the batch index itself, utils/ops.py:111-116).
    Signed max (no abs); first maximum wins on ties.
    """
    Bt, L, N = X.shape
    T = (L - P) // hop + 1
    s0, s1, s2 = X.strides
    win = np.lib.stride_tricks.as_strided(X, shape=(Bt, T, P, N), strides=(s0, hop * s1, s1, s2), writeable=False)
    am = np.argmax(win, axis=2)                       # first max on ties
    y = np.take_along_axis(win, am[:, :, None, :], axis=2)[:, :, 0, :]
    pos = am + (np.arange(T) * hop)[None, :, None]    # sample index l
    argmax = pos.astype(np.int64) * N + np.arange(N, dtype=np.int64)[None, None, :]
    return np.ascontiguousarray(y), argmax


def front_maxpool(x, f, P, hop):
    """Path B (--with_max_pool): stride-1 conv then max-pool with argmax (adapt.py:115-117)."""
    return maxpool_with_argmax(conv_dense(x, f), P, hop)


def front_avgpool(x, f, P):
    """Path C (--with_average_pool): stride-1 conv then average pool pool=stride=P (adapt.py:119-120)."""
    X = conv_dense(x, f)
    Bt, L, N = X.shape
    T = L // P
    return X[:, :T * P].reshape(Bt, T, P, N).mean(axis=2)


def front_maxpool_bwd_filter(x, dy, argmax, W):
    """df for path B: dX is non-zero only at argmax positions."""
    Bt, L = x.shape
    N = dy.shape[2]
    _, pl, pr = same_pads(L, W, 1)
    xp = np.zeros((Bt, L + pl + pr), dtype=x.dtype)
    xp[:, pl:pl + L] = x
    pos = (argmax // N).astype(np.int64)              # [Bt,T,N]
    df = np.zeros((W, N), dtype=x.dtype)
    k = np.arange(W)
    for b in range(Bt):
        for t in range(dy.shape[1]):
            # gather xpad[b, pos+k] for all n -> [W,N]
            seg = xp[b][pos[b, t][None, :] + k[:, None]]
            df += seg * dy[b, t][None, :]
    return df


def sparsity_terms(y, p):
    """p_hat = sum_b |y| ; sparse = sum kl_div(p, p_hat)  (adapt.py:130-132, utils/ops.py:46-54)."""
    Bt = y.shape[0]
    p_hat = np.abs(y.reshape(Bt, -1)).sum(axis=0)

    def logfunc(a, b):
        ca = np.clip(a, 1e-10, 1.0)
        cb = np.clip(b, 1e-10, 1.0)
        return a * np.log(ca / cb)

    kl = logfunc(p, p_hat) + logfunc(1 - p, 1 - p_hat)
    return p_hat, kl.sum()


def sparsity_terms_bwd(y, p, p_hat=None):
    """d [sum kl_div(p, p_hat)] / d y with p_hat = sum_b |y| (adapt.py:130-132): tf.abs' = sign; tf.clip_by_value passes the
    gradient where 1e-10 <= value <= 1 and blocks it outside (utils/ops.py:46-49)."""
    Bt = y.shape[0]
    if p_hat is None:
        p_hat = np.abs(y.reshape(Bt, -1)).sum(axis=0)
    q, qh = 1.0 - p, 1.0 - p_hat
    g = np.zeros_like(p_hat)
    inside = (p_hat >= 1e-10) & (p_hat <= 1.0)
    g[inside] -= p / p_hat[inside]
    inside_q = (qh >= 1e-10) & (qh <= 1.0)
    g[inside_q] += q / qh[inside_q]
    return (np.sign(y.reshape(Bt, -1)) * g[None, :]).reshape(y.shape)


# ----------------------------------------------------------------------------------------
# Synthesis (reference Adapt.back, adapt.py:205-252)
# ----------------------------------------------------------------------------------------

def synth_strided(z, f2, hop, L):
    """Path A back: tf.nn.conv2d_transpose stride hop SAME to length L (adapt.py:236-243).

    z [R, T', N], f2 [W, N] -> out [R, L]; exact adjoint of conv_strided w.r.t. its input:
    out[r,l] = sum_{t,n,k: t*hop + k - pl = l} z[r,t,n] * f2[k,n].
    """
    W, N = f2.shape
    R, T, _ = z.shape
    _, pl, _ = same_pads(L, W, hop)
    fr = z @ f2.T                                      # [R, T, W]
    buf = np.zeros((R, (T - 1) * hop + W + pl + L), dtype=z.dtype)
    for t in range(T):
        buf[:, t * hop:t * hop + W] += fr[:, t]
    return np.ascontiguousarray(buf[:, pl:pl + L])


def synth_strided_bwd(z, f2, hop, dout):
    """Backward of synth_strided: dz = analysis conv of dout with f2; df2 = frames(dout)^T z
    (SURVEY Appendix D-2)."""
    W, N = f2.shape
    dz = conv_strided(dout, f2, hop)[:, :z.shape[1]]
    df2 = conv_strided_bwd_filter(dout, z, W, hop)
    return dz, df2


def unpool(vals, argmax, L, N):
    """utils/ops.py:94-120 (tf.scatter_nd; duplicates add).  vals/argmax [R,T',N] -> dense [R,L,N]."""
    R = vals.shape[0]
    out = np.zeros((R, L * N), dtype=vals.dtype)
    for r in range(R):
        np.add.at(out[r], argmax[r].reshape(-1), vals[r].reshape(-1))
    return out.reshape(R, L, N)


def synth_unpool(vals, argmax, f2, L):
    """Path B back: unpool with the (mixture's, tiled) argmax then stride-1 conv2d_transpose SAME
    (adapt.py:210-223, 236-243), computed sparsely -- never builds [R,L,N].

    out[r,l] = sum_{t,n} vals[r,t,n] * f2[l - pos + pl, n],  pos = argmax // N.
    """
    W, N = f2.shape
    R, T, _ = vals.shape
    _, pl, pr = same_pads(L, W, 1)
    buf = np.zeros((R, L + pl + pr), dtype=vals.dtype)
    pos = (argmax // N).astype(np.int64)
    k = np.arange(W)
    for r in range(R):
        for t in range(T):
            idx = pos[r, t][None, :] + k[:, None]      # [W,N] positions in padded coords
            np.add.at(buf[r], idx, f2 * vals[r, t][None, :])
    return np.ascontiguousarray(buf[:, pl:pl + L])


def synth_unpool_bwd(vals, argmax, f2, dout):
    """Backward of synth_unpool: gather form."""
    W, N = f2.shape
    R, T, _ = vals.shape
    L = dout.shape[1]
    _, pl, pr = same_pads(L, W, 1)
    dp = np.zeros((R, L + pl + pr), dtype=dout.dtype)
    dp[:, pl:pl + L] = dout
    pos = (argmax // N).astype(np.int64)
    k = np.arange(W)
    dvals = np.zeros_like(vals)
    df2 = np.zeros_like(f2)
    for r in range(R):
        for t in range(T):
            seg = dp[r][pos[r, t][None, :] + k[:, None]]   # [W,N]
            dvals[r, t] = (seg * f2).sum(axis=0)
            df2 += seg * vals[r, t][None, :]
    return dvals, df2


def upsample_nearest(z, P):
    """avg-pool back path: tf.keras UpSampling2D((1,P)) (adapt.py:226-228)."""
    return np.repeat(z, P, axis=1)


# ----------------------------------------------------------------------------------------
# Adapt.separator for pretraining (adapt.py:136-203)
# ----------------------------------------------------------------------------------------

def overlap_metric(y, B, S):
    """adapt.py:141-160: mean over batch/pairs/bins of 1 - |a-b| / (max(a,b) + 1e-8) on |non-mix rep|."""
    from itertools import combinations
    nm = np.abs(y[B:].reshape(B, S, -1))
    vals = []
    for (i, j) in combinations(range(S), 2):
        a, b = nm[:, i], nm[:, j]
        vals.append((1.0 - np.abs(a - b) / (np.maximum(a, b) + 1e-8)).mean(axis=-1))
    return np.mean(np.stack(vals, axis=1), axis=1).mean()


def pretrain_separator(y, B, S, separation):
    """adapt.py:173-196.  y [B(1+S), T', N] -> [B*S, T', N].

    'mask'    : mix * (non_mix / mix)   (NaN where mix == 0; quirk C-4)
    'perfect' : mix - sum_{others} non_mix
    """
    T, N = y.shape[1:]
    mix = y[:B][:, None]                               # [B,1,T,N]
    nm = y[B:].reshape(B, S, T, N)
    if separation == 'mask':
        with np.errstate(divide='ignore', invalid='ignore'):
            out = mix * (nm / mix)
    else:
        out = mix - (nm.sum(axis=1, keepdims=True) - nm)
    return out.reshape(B * S, T, N)


def pretrain_separator_bwd(y, B, S, separation, dout):
    """Gradient of pretrain_separator w.r.t. y (all rows)."""
    T, N = y.shape[1:]
    d = dout.reshape(B, S, T, N)
    dy = np.zeros_like(y)
    if separation == 'mask':
        # out = mix * (nm / mix): d/dnm = 1 ; d/dmix = nm/mix - mix*nm/mix^2 = 0 (TF autodiff gives
        # exactly (nm/mix)*d + mix * (-nm/mix^2) * d, which cancels up to rounding)
        dy[B:] = d.reshape(B * S, T, N)
    else:
        dy[:B] = d.sum(axis=1)
        tot = d.sum(axis=1, keepdims=True)
        dy[B:] = (-(tot - d)).reshape(B * S, T, N)
    return dy


def overlap_metric_bwd(y, B, S):
    """d overlap / d y (all rows; mixture rows get 0).  m = 1 - |a-c|/(max(a,c)+1e-8) with a=|y_s|, c=|y_s'|."""
    from itertools import combinations
    T, N = y.shape[1:]
    nm = y[B:].reshape(B, S, T * N)
    a_abs = np.abs(nm)
    g = np.zeros_like(nm)
    npairs = S * (S - 1) // 2
    for (i, j) in combinations(range(S), 2):
        a, c = a_abs[:, i], a_abs[:, j]
        mx = np.maximum(a, c) + 1e-8
        diff = a - c
        sg = np.sign(diff)
        da = -sg / mx + np.where(a >= c, np.abs(diff) / mx ** 2, 0.0)
        dc = sg / mx + np.where(c > a, np.abs(diff) / mx ** 2, 0.0)
        g[:, i] += da * np.sign(nm[:, i])
        g[:, j] += dc * np.sign(nm[:, j])
    dy = np.zeros_like(y)
    dy[B:] = (g / (B * npairs * T * N)).reshape(B * S, T, N)
    return dy
