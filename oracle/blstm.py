"""Oracle: BLSTM stack (reference utils/ops.py:358-383; TF-1.4 BasicLSTMCell semantics).

Test infrastructure only -- see oracle/__init__.py.

Per direction one kernel K [D+H, 4H] applied to concat([x, h]), bias [4H]; gate split order
i, j (candidate), f, o;  c' = c*sigmoid(f + 1.0) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o);
zero initial state; all T steps.  Backward direction runs on the time-reversed input and its
outputs are reversed back before the concat (utils/ops.py:368,383).
"""
import numpy as np

FORGET_BIAS = 1.0


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_dir_fwd(x, K, b, reverse=False, masks=None):
    """One direction.  x [B,T,D] -> h [B,T,H] (in *original* time order) and a cache for bwd.

    masks (utils/ops.py:363,373,379: DropoutWrapper(cell, keep, keep, keep) when --recurrent_dropout != 0 and training; TF 1.4
    rnn_cell_impl.DropoutWrapper.__call__, variational_recurrent=False): dict of keep-masks ALREADY scaled by 1/keep, in original
    time order -- 'in' [B,T,D] on the cell input, 'h' / 'c' [B,T,H] on the two parts of the LSTMStateTuple handed to the next step
    (TF 1.4 maps the dropout over the whole state structure), 'out' [B,T,H] on the cell output.  The cell output is the un-masked
    h_t; only the carried state is masked."""
    B, T, D = x.shape
    H = K.shape[1] // 4
    Wx, U = K[:D], K[D:]
    if masks is not None:
        return _lstm_dir_fwd_dropout(x, K, b, reverse, masks)
    z = x.reshape(B * T, D) @ Wx + b                    # hoisted input projection
    z = z.reshape(B, T, 4 * H)
    h = np.zeros((B, H), dtype=x.dtype)
    c = np.zeros((B, H), dtype=x.dtype)
    hs = np.zeros((B, T, H), dtype=x.dtype)
    cs = np.zeros((B, T, H), dtype=x.dtype)
    gates = np.zeros((B, T, 4 * H), dtype=x.dtype)      # activated i, g, f, o
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        a = z[:, t] + h @ U
        i = _sigmoid(a[:, :H])
        g = np.tanh(a[:, H:2 * H])
        f = _sigmoid(a[:, 2 * H:3 * H] + FORGET_BIAS)
        o = _sigmoid(a[:, 3 * H:])
        c = c * f + i * g
        h = np.tanh(c) * o
        hs[:, t] = h
        cs[:, t] = c
        gates[:, t] = np.concatenate([i, g, f, o], axis=1)
    return hs, (x, K, hs, cs, gates, reverse)


def _lstm_dir_fwd_dropout(x, K, b, reverse, masks):
    B, T, D = x.shape
    H = K.shape[1] // 4
    Wx, U = K[:D], K[D:]
    xd = x * masks['in']
    z = (xd.reshape(B * T, D) @ Wx + b).reshape(B, T, 4 * H)
    h = np.zeros((B, H), dtype=x.dtype)                 # the (masked) state
    c = np.zeros((B, H), dtype=x.dtype)
    hs = np.zeros((B, T, H), dtype=x.dtype)
    cs = np.zeros((B, T, H), dtype=x.dtype)
    hst = np.zeros((B, T, H), dtype=x.dtype)            # state AFTER the state dropout
    cst = np.zeros((B, T, H), dtype=x.dtype)
    gates = np.zeros((B, T, 4 * H), dtype=x.dtype)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        a = z[:, t] + h @ U
        i = _sigmoid(a[:, :H])
        g = np.tanh(a[:, H:2 * H])
        f = _sigmoid(a[:, 2 * H:3 * H] + FORGET_BIAS)
        o = _sigmoid(a[:, 3 * H:])
        cn = c * f + i * g
        hn = np.tanh(cn) * o
        hs[:, t], cs[:, t] = hn, cn
        gates[:, t] = np.concatenate([i, g, f, o], axis=1)
        h, c = hn * masks['h'][:, t], cn * masks['c'][:, t]
        hst[:, t], cst[:, t] = h, c
    return hs * masks['out'], (x, K, hs, cs, gates, reverse, masks, xd, hst, cst)


def _lstm_dir_bwd_dropout(dy, cache, need_dx):
    x, K, hs, cs, gates, reverse, masks, xd, hst, cst = cache
    B, T, D = x.shape
    H = K.shape[1] // 4
    Wx, U = K[:D], K[D:]
    dh_out = dy * masks['out']
    da_all = np.zeros((B, T, 4 * H), dtype=x.dtype)
    dh_rec = np.zeros((B, H), dtype=x.dtype)            # gradient w.r.t. the masked state that LEFT the step being processed
    dc_next = np.zeros((B, H), dtype=x.dtype)
    order = list(range(T)) if reverse else list(range(T - 1, -1, -1))
    step = 1 if reverse else -1
    for t in order:
        tp = t + step
        has_prev = 0 <= tp < T
        c_prev = cst[:, tp] if has_prev else np.zeros((B, H), dtype=x.dtype)
        i, g, f, o = (gates[:, t, :H], gates[:, t, H:2 * H], gates[:, t, 2 * H:3 * H], gates[:, t, 3 * H:])
        tc = np.tanh(cs[:, t])
        dh = dh_out[:, t] + dh_rec * masks['h'][:, t]
        do = dh * tc
        dc = dc_next * masks['c'][:, t] + dh * o * (1.0 - tc * tc)
        da = np.concatenate([dc * g * i * (1.0 - i), dc * i * (1.0 - g * g), dc * c_prev * f * (1.0 - f), do * o * (1.0 - o)], axis=1)
        da_all[:, t] = da
        dc_next = dc * f
        dh_rec = da @ U.T
    da2 = da_all.reshape(B * T, 4 * H)
    dWx = xd.reshape(B * T, D).T @ da2
    h_prev = np.zeros_like(hst)
    if reverse:
        h_prev[:, :-1] = hst[:, 1:]
    else:
        h_prev[:, 1:] = hst[:, :-1]
    dU = h_prev.reshape(B * T, H).T @ da2
    db = da2.sum(axis=0)
    dx = ((da2 @ Wx.T).reshape(B, T, D) * masks['in']) if need_dx else None
    return dx, np.concatenate([dWx, dU], axis=0), db


def lstm_dir_bwd(dh_out, cache, need_dx=True):
    """BPTT for one direction.  dh_out [B,T,H] -> dx [B,T,D], dK, db."""
    if len(cache) > 6:
        return _lstm_dir_bwd_dropout(dh_out, cache, need_dx)
    x, K, hs, cs, gates, reverse = cache
    B, T, D = x.shape
    H = K.shape[1] // 4
    Wx, U = K[:D], K[D:]
    da_all = np.zeros((B, T, 4 * H), dtype=x.dtype)
    dh_rec = np.zeros((B, H), dtype=x.dtype)
    dc_next = np.zeros((B, H), dtype=x.dtype)
    # process in the reverse of the forward processing order
    order = list(range(T)) if reverse else list(range(T - 1, -1, -1))
    step = 1 if reverse else -1                          # index of the *previous processed* step = t + step
    for t in order:
        tp = t + step                                    # state that fed step t came from time tp
        has_prev = 0 <= tp < T
        c_prev = cs[:, tp] if has_prev else np.zeros((B, H), dtype=x.dtype)
        i, g, f, o = (gates[:, t, :H], gates[:, t, H:2 * H], gates[:, t, 2 * H:3 * H], gates[:, t, 3 * H:])
        tc = np.tanh(cs[:, t])
        dh = dh_out[:, t] + dh_rec
        do = dh * tc
        dc = dc_next + dh * o * (1.0 - tc * tc)
        da = np.concatenate([dc * g * i * (1.0 - i),
                             dc * i * (1.0 - g * g),
                             dc * c_prev * f * (1.0 - f),
                             do * o * (1.0 - o)], axis=1)
        da_all[:, t] = da
        dc_next = dc * f
        dh_rec = da @ U.T
    # hoisted GEMMs
    da2 = da_all.reshape(B * T, 4 * H)
    dWx = x.reshape(B * T, D).T @ da2
    h_prev = np.zeros_like(hs)
    if reverse:
        h_prev[:, :-1] = hs[:, 1:]
    else:
        h_prev[:, 1:] = hs[:, :-1]
    dU = h_prev.reshape(B * T, H).T @ da2
    db = da2.sum(axis=0)
    dK = np.concatenate([dWx, dU], axis=0)
    dx = (da2 @ Wx.T).reshape(B, T, D) if need_dx else None
    return dx, dK, db


def blstm_fwd(x, Kf, bf, Kb, bb, masks=None):
    """BLSTM.f_prop (utils/ops.py:366-383): concat([forward_out, backward_out[:, ::-1]], 2).
    masks: None, or (masks of the forward wrapper, masks of the backward wrapper) -- see lstm_dir_fwd; each wrapper draws its own."""
    hf, cf = lstm_dir_fwd(x, Kf, bf, reverse=False, masks=None if masks is None else masks[0])
    hb, cb = lstm_dir_fwd(x, Kb, bb, reverse=True, masks=None if masks is None else masks[1])
    return np.concatenate([hf, hb], axis=2), (cf, cb)


def blstm_bwd(dout, cache, need_dx=True):
    cf, cb = cache
    H = cf[1].shape[1] // 4
    dxf, dKf, dbf = lstm_dir_bwd(dout[:, :, :H], cf, need_dx)
    dxb, dKb, dbb = lstm_dir_bwd(dout[:, :, H:], cb, need_dx)
    dx = dxf + dxb if need_dx else None
    return dx, (dKf, dbf, dKb, dbb)


def blstm_stack_fwd(x, params):
    """params: list of (Kf, bf, Kb, bb) per layer  (dpcl.py:26-27, L41.py:31)."""
    caches = []
    for p in params:
        x, c = blstm_fwd(x, *p)
        caches.append(c)
    return x, caches


def blstm_stack_bwd(dout, caches, need_dx=False):
    grads = []
    n = len(caches)
    for li in range(n - 1, -1, -1):
        dout, g = blstm_bwd(dout, caches[li], need_dx=(li > 0 or need_dx))
        grads.append(g)
    return dout, grads[::-1]
