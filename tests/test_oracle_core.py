"""Oracle self-consistency (CPU): each numpy restatement vs an independently written torch-CPU /
scipy formulation, and each hand-derived backward vs torch autograd (float64)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import front, stft, blstm, dense, dpcl, l41, losses, optim, step

RNG = np.random.RandomState(0)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64 if np.asarray(x).dtype.kind == 'f' else None))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---------------------------------------------------------------- front / back
@pytest.mark.parametrize('L,W,hop', [(2048, 256, 64), (1000, 128, 48), (20480, 1024, 256)])
def test_same_pads(L, W, hop):
    T, pl, pr = front.same_pads(L, W, hop)
    assert T == -(-L // hop)
    assert pl + pr == max((T - 1) * hop + W - L, 0) and pl <= pr <= pl + 1
    if (L, W, hop) == (20480, 1024, 256):
        assert (T, pl, pr) == (80, 384, 384)
    assert front.same_pads(20480, 1024, 1)[1:] == (511, 512)


def test_conv_strided_vs_torch():
    Bt, L, W, N, hop = 3, 1000, 128, 8, 48
    x = RNG.randn(Bt, L)
    f = RNG.randn(W, N)
    y = front.conv_strided(x, f, hop)
    T, pl, pr = front.same_pads(L, W, hop)
    yt = F.conv1d(F.pad(t(x)[:, None], (pl, pr)), t(f.T.copy())[:, None], stride=hop)
    assert rel(y, yt.permute(0, 2, 1).numpy()) < 1e-12
    # np.correlate on one filter
    xp = np.pad(x[0], (pl, pr))
    ref = np.correlate(xp, f[:, 3], mode='valid')[::hop][:T]
    assert rel(y[0, :, 3], ref) < 1e-12


def test_conv_filter_grads_vs_autograd():
    Bt, L, W, N, hop = 2, 600, 64, 5, 16
    x = RNG.randn(Bt, L)
    w, bases = RNG.randn(W), RNG.randn(W, N)
    dy = RNG.randn(Bt, -(-L // hop), N)
    wt, bt = t(w).requires_grad_(), t(bases).requires_grad_()
    T, pl, pr = front.same_pads(L, W, hop)
    ft = wt.abs()[:, None] * bt
    yt = F.conv1d(F.pad(t(x)[:, None], (pl, pr)), ft.t()[:, None], stride=hop).permute(0, 2, 1)
    (yt * t(dy)).sum().backward()
    df = front.conv_strided_bwd_filter(x, dy, W, hop)
    dw, db = front.front_filter_bwd(w, bases, df)
    assert rel(dw, wt.grad.numpy()) < 1e-12 and rel(db, bt.grad.numpy()) < 1e-12


def test_synth_is_adjoint_and_grads():
    R, L, W, N, hop = 3, 640, 64, 6, 16
    T = -(-L // hop)
    z, f2, x = RNG.randn(R, T, N), RNG.randn(W, N), RNG.randn(R, L)
    out = front.synth_strided(z, f2, hop, L)
    assert abs(np.sum(front.conv_strided(x, f2, hop) * z) - np.sum(x * out)) < 1e-9
    _, pl, pr = front.same_pads(L, W, hop)
    zt, ft = t(z).requires_grad_(), t(f2).requires_grad_()
    ot = F.conv_transpose1d(zt.permute(0, 2, 1), ft.t()[:, None], stride=hop)[:, 0, pl:pl + L]
    assert rel(out, ot.detach().numpy()) < 1e-12
    dout = RNG.randn(R, L)
    (ot * t(dout)).sum().backward()
    dz, df2 = front.synth_strided_bwd(z, f2, hop, dout)
    assert rel(dz, zt.grad.numpy()) < 1e-12 and rel(df2, ft.grad.numpy()) < 1e-12


def test_maxpool_argmax_and_unpool():
    Bt, L, W, N, P, hop = 2, 256, 32, 4, 32, 16
    x, f = RNG.randn(Bt, L), RNG.randn(W, N)
    y, am = front.front_maxpool(x, f, P, hop)
    X = front.conv_dense(x, f)
    T = (L - P) // hop + 1
    assert y.shape == (Bt, T, N) and am.dtype == np.int64
    for b in range(Bt):
        for tt in range(T):
            for n in range(N):
                seg = X[b, tt * hop:tt * hop + P, n]
                l = tt * hop + int(np.argmax(seg))
                assert y[b, tt, n] == seg.max() and am[b, tt, n] == l * N + n
    # sparse synthesis == dense unpool + stride-1 transposed conv
    f2 = RNG.randn(W, N)
    dense_u = front.unpool(y, am, L, N)
    ref = front.synth_strided(dense_u, f2, 1, L)
    out = front.synth_unpool(y, am, f2, L)
    assert rel(out, ref) < 1e-12
    # gradients of the sparse form vs autograd through the dense form
    ut, ft = t(dense_u).requires_grad_(), t(f2).requires_grad_()
    _, pl, pr = front.same_pads(L, W, 1)
    ot = F.conv_transpose1d(ut.permute(0, 2, 1), ft.t()[:, None])[:, 0, pl:pl + L]
    dout = RNG.randn(Bt, L)
    (ot * t(dout)).sum().backward()
    dv, df2 = front.synth_unpool_bwd(y, am, f2, dout)
    g = ut.grad.numpy().reshape(Bt, L * N)
    gv = np.stack([g[b][am[b].reshape(-1)].reshape(T, N) for b in range(Bt)])
    assert rel(dv, gv) < 1e-12 and rel(df2, ft.grad.numpy()) < 1e-12
    # filter gradient of the max-pool front (gather form) vs autograd through conv + max_pool
    wt = t(f).requires_grad_()
    Xt = F.conv1d(F.pad(t(x)[:, None], front.same_pads(L, W, 1)[1:]), wt.t()[:, None]).permute(0, 2, 1)
    yt = F.max_pool1d(Xt.permute(0, 2, 1), P, hop).permute(0, 2, 1)
    dy = RNG.randn(Bt, T, N)
    (yt * t(dy)).sum().backward()
    assert rel(front.front_maxpool_bwd_filter(x, dy, am, W), wt.grad.numpy()) < 1e-12


def test_pretrain_separator_grads():
    B, S, T, N = 2, 3, 4, 5
    y = RNG.randn(B * (S + 1), T, N)
    d = RNG.randn(B * S, T, N)
    yt = t(y).requires_grad_()
    mix, nm = yt[:B][:, None], yt[B:].reshape(B, S, T, N)
    out = (mix - (nm.sum(1, keepdim=True) - nm)).reshape(B * S, T, N)
    assert rel(front.pretrain_separator(y, B, S, 'perfect'), out.detach().numpy()) < 1e-12
    (out * t(d)).sum().backward()
    assert rel(front.pretrain_separator_bwd(y, B, S, 'perfect', d), yt.grad.numpy()) < 1e-12
    m = front.pretrain_separator(y, B, S, 'mask')
    assert rel(m, y[B:]) < 1e-12


# ---------------------------------------------------------------- STFT
def test_stft_vs_torch_and_istft():
    R, L, W, hop = 2, 2048, 256, 128
    x = RNG.randn(R, L)
    s = stft.stft(x, W, hop)
    ref = torch.stft(t(x), W, hop, W, window=torch.hann_window(W, periodic=True, dtype=torch.float64), center=False, return_complex=True)
    assert rel(s.real, ref.permute(0, 2, 1).real.numpy()) < 1e-12
    assert rel(s.imag, ref.permute(0, 2, 1).imag.numpy()) < 1e-10
    rec = stft.istft(np.abs(s), np.angle(s), W, hop)
    assert rec.shape[1] == L
    assert np.abs(rec[:, W:-W] - x[:, W:-W]).max() < 1e-10          # interior identity (App. A-6)
    assert np.abs(rec[:, :hop] - x[:, :hop]).max() > 1e-3            # single-coverage edges are not
    # backward w.r.t. magnitude via finite differences on a linear map
    ang = np.angle(s)
    dout = RNG.randn(R, L)
    g = stft.istft_bwd(ang, W, hop, dout)
    mag = np.abs(s)
    probe = RNG.randn(*mag.shape)
    lhs = np.sum(stft.istft(probe, ang, W, hop) * dout)
    assert abs(lhs - np.sum(g * probe)) < 1e-8 * max(1.0, abs(lhs))


# ---------------------------------------------------------------- BLSTM
def _torch_lstm_from_tf(K, b, D, H):
    """TF kernel [D+H,4H] gates (i,j,f,o) -> torch nn.LSTM weights gates (i,f,g,o); +1 forget bias."""
    lstm = torch.nn.LSTM(D, H, batch_first=True).double()
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    Kp, bp = K[:, perm], b[perm].copy()
    bp[H:2 * H] += 1.0
    with torch.no_grad():
        lstm.weight_ih_l0.copy_(t(Kp[:D].T.copy()))
        lstm.weight_hh_l0.copy_(t(Kp[D:].T.copy()))
        lstm.bias_ih_l0.copy_(t(bp))
        lstm.bias_hh_l0.zero_()
    return lstm


def test_blstm_vs_torch_lstm_and_grads():
    B, T, D, H = 3, 7, 5, 4
    x = RNG.randn(B, T, D)
    Kf, Kb = RNG.randn(D + H, 4 * H) * 0.5, RNG.randn(D + H, 4 * H) * 0.5
    bf, bb = RNG.randn(4 * H) * 0.1, RNG.randn(4 * H) * 0.1
    out, cache = blstm.blstm_fwd(x, Kf, bf, Kb, bb)
    lf, lb = _torch_lstm_from_tf(Kf, bf, D, H), _torch_lstm_from_tf(Kb, bb, D, H)
    xt = t(x).requires_grad_()
    of, _ = lf(xt)
    ob, _ = lb(torch.flip(xt, [1]))
    ot = torch.cat([of, torch.flip(ob, [1])], 2)
    assert rel(out, ot.detach().numpy()) < 1e-12
    dout = RNG.randn(B, T, 2 * H)
    (ot * t(dout)).sum().backward()
    dx, (dKf, dbf, dKb, dbb) = blstm.blstm_bwd(dout, cache)
    assert rel(dx, xt.grad.numpy()) < 1e-11
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    for dK, db, l in ((dKf, dbf, lf), (dKb, dbb, lb)):
        assert rel(dK[:D][:, perm], l.weight_ih_l0.grad.numpy().T) < 1e-11
        assert rel(dK[D:][:, perm], l.weight_hh_l0.grad.numpy().T) < 1e-11
        assert rel(db[perm], l.bias_ih_l0.grad.numpy()) < 1e-11


def _dropout_masks(rng, B, T, D, H, keep):
    def m(*s):
        return (rng.rand(*s) < keep) / keep
    return {'in': m(B, T, D), 'h': m(B, T, H), 'c': m(B, T, H), 'out': m(B, T, H)}


def test_blstm_with_dropout_wrappers_vs_autograd():
    """--recurrent_dropout != 0 (utils/ops.py:363,373,379): DropoutWrapper(cell, keep, keep, keep) per direction, restated step by
    step in torch (input mask -> cell -> state mask on c AND h as TF 1.4 does -> output mask) and differentiated by autograd, against
    the oracle's forward and hand-written BPTT with the same masks."""
    B, T, D, H, keep = 3, 6, 5, 4, 0.7
    rng = np.random.RandomState(5)
    x = rng.randn(B, T, D)
    Kf, Kb = rng.randn(D + H, 4 * H) * 0.5, rng.randn(D + H, 4 * H) * 0.5
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    masks = (_dropout_masks(rng, B, T, D, H, keep), _dropout_masks(rng, B, T, D, H, keep))
    out, cache = blstm.blstm_fwd(x, Kf, bf, Kb, bb, masks)
    dout = rng.randn(B, T, 2 * H)
    dx, (dKf, dbf, dKb, dbb) = blstm.blstm_bwd(dout, cache)
    tx, tKf, tbf, tKb, tbb = [t(a).requires_grad_() for a in (x, Kf, bf, Kb, bb)]

    def run_dir(K, b, m, rev):
        h = torch.zeros(B, H, dtype=torch.float64)
        c = torch.zeros(B, H, dtype=torch.float64)
        outs = [None] * T
        for tt in (range(T - 1, -1, -1) if rev else range(T)):
            a = torch.cat([tx[:, tt] * t(m['in'][:, tt]), h], 1) @ K + b
            i, j, f, o = a[:, :H], a[:, H:2 * H], a[:, 2 * H:3 * H], a[:, 3 * H:]
            cn = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            hn = torch.tanh(cn) * torch.sigmoid(o)
            outs[tt] = hn * t(m['out'][:, tt])
            h, c = hn * t(m['h'][:, tt]), cn * t(m['c'][:, tt])
        return torch.stack(outs, 1)
    ot = torch.cat([run_dir(tKf, tbf, masks[0], False), run_dir(tKb, tbb, masks[1], True)], 2)
    assert rel(out, ot.detach().numpy()) < 1e-12
    (ot * t(dout)).sum().backward()
    for mine, ref in ((dx, tx), (dKf, tKf), (dbf, tbf), (dKb, tKb), (dbb, tbb)):
        assert rel(mine, ref.grad.numpy()) < 1e-11
    # keep = 1: the wrapper is the identity
    ones = tuple({k: np.ones_like(v) for k, v in m.items()} for m in masks)
    assert rel(blstm.blstm_fwd(x, Kf, bf, Kb, bb, ones)[0], blstm.blstm_fwd(x, Kf, bf, Kb, bb)[0]) < 1e-14


# ---------------------------------------------------------------- dense + l2norm + DPCL + L41
def test_dense_l2norm_dpcl_grads():
    B, T, Fq, E, S, Din = 2, 3, 4, 5, 2, 6
    x, W, b = RNG.randn(B, T, Din), RNG.randn(Din, Fq * E), RNG.randn(Fq * E)
    lab = RNG.randint(0, S, (B, T * Fq))
    Y = np.eye(S)[lab]
    xt, Wt, bt = t(x).requires_grad_(), t(W).requires_grad_(), t(b).requires_grad_()
    u = xt @ Wt + bt
    v = F.normalize(u.reshape(B, T, Fq, E), dim=3, eps=1e-6)       # eps on the norm == 1e-12 on the sum of squares
    Vt, Yt = v.reshape(B, T * Fq, E), t(Y)
    cnt = Yt.transpose(1, 2) @ torch.ones(B, T * Fq, 1, dtype=torch.float64)
    D = 1 / torch.sqrt(Yt @ cnt)
    DV, DY = D * Vt, D * Yt
    cost = (torch.linalg.matrix_norm(Vt.transpose(1, 2) @ DV) - 2 * torch.linalg.matrix_norm(Vt.transpose(1, 2) @ DY)
            + torch.linalg.matrix_norm(Yt.transpose(1, 2) @ DY)).mean()
    cost.backward()
    u_o = dense.dense_fwd(x, W, b)
    V_o, inv = dense.l2norm_fwd(u_o, E)
    c_o, _ = dpcl.dpcl_cost(V_o.reshape(B, -1, E), Y)
    assert abs(c_o - cost.item()) < 1e-12
    dV = dpcl.dpcl_cost_bwd(V_o.reshape(B, -1, E), Y).reshape(V_o.shape)
    du = dense.l2norm_bwd(V_o, inv, dV).reshape(B, T, -1)
    dx, dW, db = dense.dense_bwd(x, W, du)
    assert rel(dx, xt.grad.numpy()) < 1e-11 and rel(dW, Wt.grad.numpy()) < 1e-11 and rel(db, bt.grad.numpy()) < 1e-11


@pytest.mark.parametrize('normalize', [True, False])
def test_l41_grads(normalize):
    B, T, Fq, E, S, NS = 2, 3, 4, 5, 2, 7
    emb, spk = RNG.randn(B, T, Fq, E), RNG.randn(NS, E)
    I = np.array([[0, 3], [3, 5]])
    y = np.where(RNG.rand(B, T, Fq, S) > 0.5, 1.0, -1.0)
    et, st = t(emb).requires_grad_(), t(spk).requires_grad_()
    sv = F.normalize(st, dim=1, eps=1e-6) if normalize else st
    dot = torch.einsum('btfe,bse->btfs', et, sv[t(I)])
    cost = (-torch.log(torch.sigmoid(t(y) * dot))).mean(3).mean(0).mean()
    cost.backward()
    assert abs(l41.l41_cost(emb, y, spk, I, normalize) - cost.item()) < 1e-12
    de, ds = l41.l41_cost_bwd(emb, y, spk, I, normalize)
    assert rel(de, et.grad.numpy()) < 1e-11 and rel(ds, st.grad.numpy()) < 1e-11


@pytest.mark.parametrize('normalize', [True, False])
@pytest.mark.parametrize('method', ['k-nearest', 'random'])
def test_l41_negative_sampling_grads(normalize, method):
    """--sampling K (models/L41.py:69-147,165-166) restated with torch-CPU autograd: gather of the neighbour / random sets,
    -log(sigmoid(-dot)) averaged over K, scaled by ns_rate, added per bin before the batch mean."""
    B, T, Fq, E, S, NS, K, rate = 2, 3, 4, 5, 2, 9, 3, 0.3
    emb, spk = RNG.randn(B, T, Fq, E), RNG.randn(NS, E)
    I = np.array([[0, 3], [5, 1]])
    lab = RNG.randint(0, S, (B, T, Fq))
    y = np.where(np.eye(S)[lab] > 0, 1.0, -1.0)
    et, st = t(emb).requires_grad_(), t(spk).requires_grad_()
    sv = F.normalize(st, dim=1, eps=1e-6) if normalize else st
    vs = sv[t(I)]
    if method == 'k-nearest':
        idx = l41.knearest_indices(spk, I, K, normalize)
        top = torch.topk(torch.einsum('bse,ne->bsn', vs, sv), K, dim=2).indices.numpy()
        assert np.array_equal(np.sort(top, 2), np.sort(idx, 2)) and idx.shape == (B, S, K)
        vec = sv[t(idx)][torch.arange(B)[:, None, None], t(np.argmax(y, -1))]          # [B,T,F,K,E]: set of the dominant speaker
    else:
        idx = l41.random_indices(I, NS, K, np.random.RandomState(4))
        assert idx.shape == (B, 1, K) and all(not (set(idx[b, 0]) & set(I[b])) and len(set(idx[b, 0])) == K for b in range(B))
        vec = sv[t(idx)][:, 0][:, None, None].expand(B, T, Fq, K, E)
    dot = torch.einsum('btfe,bse->btfs', et, vs)
    doto = (vec * et[:, :, :, None, :]).sum(-1)
    cost = (-torch.log(torch.sigmoid(t(y) * dot))).mean(3) + rate * (-torch.log(torch.sigmoid(-doto))).mean(-1)
    cost = cost.mean(0).mean()
    cost.backward()
    assert abs(l41.l41_cost(emb, y, spk, I, normalize, idx, rate) - cost.item()) < 1e-12
    de, ds = l41.l41_cost_bwd(emb, y, spk, I, normalize, idx, rate)
    assert rel(de, et.grad.numpy()) < 1e-11 and rel(ds, st.grad.numpy()) < 1e-11


# ---------------------------------------------------------------- losses
def test_pretrain_and_pit_costs_grads():
    B, S, L = 3, 2, 50
    xm, xn, bk = RNG.randn(B, L), RNG.randn(B, S, L), RNG.randn(B, S, L)
    for kind in ('l2', 'sdr', 'l2+sdr'):
        bt = t(bk).requires_grad_()
        l2 = ((t(xn) - bt) ** 2).sum(-1).sum(-1).mean()
        tn, an, ta = (t(xn) ** 2).sum(-1), (bt ** 2).sum(-1), (t(xn) * bt).sum(-1)
        sdr = ((tn * an) / (ta ** 2 + 1e-12)).mean()
        loss = l2 if kind == 'l2' else sdr if kind == 'sdr' else l2 + sdr
        loss.backward()
        lo, _, _ = losses.pretrain_cost(xm, xn, bk, kind)
        assert abs(lo - loss.item()) < 1e-10 * max(1, abs(lo))
        assert rel(losses.pretrain_cost_bwd(xn, bk, kind), bt.grad.numpy()) < 1e-11
    # PIT: brute force over permutations
    c, best = losses.cost_finetuning(xn, bk)
    ref = np.mean([min(0.5 * np.mean([np.sum((xn[b, s] - bk[b, p[s]]) ** 2) for s in range(S)])
                       for p in losses.perms(S)) for b in range(B)])
    assert abs(c - ref) < 1e-12
    bt = t(bk).requires_grad_()
    P = losses.perms(S)
    d = (0.5 * ((t(xn)[:, None] - bt[:, t(P)]) ** 2).sum(-1)).mean(-1).min(-1)[0].mean()
    d.backward()
    assert rel(losses.pit_l2_bwd(xn, bk, best, 'sum', 'mean', 0.5), bt.grad.numpy()) < 1e-11
    # quirk C-3 shapes
    lo, l2, sdr = losses.pit_cost_adapt(xm, xn, bk, 'sdr+l2')
    assert np.isfinite(lo)


# ---------------------------------------------------------------- optimizers
def test_amsgrad_matches_formula():
    p0, g = RNG.randn(5), RNG.randn(3, 5)
    p = p0.copy()
    opt = optim.AMSGrad(0.01)
    m = v = vh = np.zeros(5)
    q = p0.copy()
    for k in range(3):
        opt.apply([p], [g[k]])
        tt = k + 1
        lr_t = 0.01 * np.sqrt(1 - 0.99 ** tt) / (1 - 0.9 ** tt)
        m = 0.9 * m + 0.1 * g[k]
        v = 0.99 * v + 0.01 * g[k] ** 2
        vh = np.maximum(vh, v)
        q = q - lr_t * m / (np.sqrt(vh) + 1e-3)
    assert rel(p, q) < 1e-14
    tp = torch.nn.Parameter(t(p0.copy()))
    o = torch.optim.RMSprop([tp], lr=0.01, alpha=0.9, eps=0.0)
    o.state[tp]['step'] = torch.tensor(0., dtype=torch.float64)
    o.state[tp]['square_avg'] = torch.ones(5, dtype=torch.float64)
    pr = p0.copy()
    r = optim.RMSProp(0.01, eps=0.0)
    for k in range(3):
        tp.grad = t(g[k].copy())
        o.step()
        r.apply([pr], [g[k]])
    assert rel(pr, tp.detach().numpy()) < 1e-12


# ---------------------------------------------------------------- assembled step vs autograd
def test_front_dpcl_step_grads_finite_difference():
    rng = np.random.RandomState(3)
    B, S, L, W, N, hop, LS, E = 2, 2, 256, 32, 6, 16, 8, 3
    P = step.init_params(rng, np.float64, front_W=W, N=N, D_in=N, layer_size=LS, nb_layers=2, E=E, F=N, conv1d_scale=0.3)
    xn = rng.randn(B, S, L) * 0.1
    xm = xn.sum(1)
    cost, grads, V, Y = step.front_dpcl_loss(xm, xn, P, hop, 2, E)
    for name in ('prediction/W', 'prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel',
                 'prediction/backward_BLSTM_1/rnn/basic_lstm_cell/bias'):
        idx = tuple(rng.randint(0, s) for s in P[name].shape)
        h = 1e-6
        P[name][idx] += h
        cp = step.front_dpcl_loss(xm, xn, P, hop, 2, E, want_grads=False)[0]
        P[name][idx] -= 2 * h
        cm = step.front_dpcl_loss(xm, xn, P, hop, 2, E, want_grads=False)[0]
        P[name][idx] += h
        fd = (cp - cm) / (2 * h)
        assert abs(fd - grads[name][idx]) < 1e-6 * max(1.0, abs(fd)), name


def test_overlap_metric_grad():
    B, S, T, N = 2, 3, 4, 5
    y = RNG.randn(B * (S + 1), T, N)
    yt = t(y).requires_grad_()
    nm = yt[B:].reshape(B, S, -1).abs()
    from itertools import combinations
    vals = [(1.0 - (nm[:, i] - nm[:, j]).abs() / (torch.maximum(nm[:, i], nm[:, j]) + 1e-8)).mean(-1) for i, j in combinations(range(S), 2)]
    ov = torch.stack(vals, 1).mean(1).mean()
    ov.backward()
    assert abs(front.overlap_metric(y, B, S) - ov.item()) < 1e-12
    assert rel(front.overlap_metric_bwd(y, B, S), yt.grad.numpy()) < 1e-11


def test_default_on_pretraining_terms_vs_autograd():
    """The terms the reference's CLI turns on by default (utils/trainer.py:151-161) -- sum kl_div(p, p_hat) with p_hat = sum_b |y|
    and both clip_by_value gates (models/adapt.py:130-132, utils/ops.py:46-54), the twice-applied filter l2 and non-negativity
    terms (adapt.py:312-316, 377-384) -- inside the whole pretraining objective: oracle cost and gradients vs torch autograd of an
    independently written graph (torch.clamp has TensorFlow's clip_by_value gradient: 1 inside the closed interval, 0 outside)."""
    from oracle import recipes
    rng = np.random.RandomState(12)
    B, S, L, W, N, hop = 2, 2, 256, 32, 6, 8
    P = {'front/window/w': rng.randn(W) * 0.4, 'front/bases/bases': rng.randn(W, N) * 0.9,
         'back/window/value': rng.randn(W) * 0.4, 'back/bases/value': rng.randn(W, N) * 0.5}
    xn = rng.randn(B, S, L) * 0.2
    xm = xn.sum(1)
    beta, p, lam, nn, ov = 0.3, 0.01, 0.2, 0.6, 0.05
    c, g, _ = recipes.pretrain_loss(xm, xn, P, hop, 'sdr+l2', 'perfect', ov, beta=beta, sparsity=p, regularization=lam, non_negativity=nn)
    T, pl, _ = front.same_pads(L, W, hop)
    Pt = {k: t(v).requires_grad_() for k, v in P.items()}
    x = torch.cat([t(xm), t(xn).reshape(B * S, L)], 0)
    f = Pt['front/window/w'].abs()[:, None] * Pt['front/bases/bases']
    f2 = Pt['back/window/value'].abs()[:, None] * Pt['back/bases/value']
    pad_total = max((T - 1) * hop + W - L, 0)
    y = F.conv1d(F.pad(x[:, None, :], (pl, pad_total - pl)), f.t()[:, None, :], stride=hop).transpose(1, 2)      # [Bt, T, N]
    Bt = y.shape[0]
    mix, nm = y[:B][:, None], y[B:].reshape(B, S, T, N)
    z = (mix - (nm.sum(1, keepdim=True) - nm)).reshape(B * S, T, N)
    full = F.conv_transpose1d(z.transpose(1, 2), f2.t()[:, None, :], stride=hop)                                  # [R, 1, (T-1)hop+W]
    back = full[:, 0, pl:pl + L].reshape(B, S, L)
    tgt = t(xn)
    l2 = ((tgt - back) ** 2).sum(-1).sum(-1).mean()
    sdr = (((tgt ** 2).sum(-1) * (back ** 2).sum(-1)) / ((tgt * back).sum(-1) ** 2 + 1e-12)).mean()
    p_hat = y.reshape(Bt, -1).abs().sum(0)
    pt = torch.tensor(p, dtype=torch.float64)

    def logfunc(a, b):
        return a * torch.log(torch.clamp(a, 1e-10, 1.0) / torch.clamp(b, 1e-10, 1.0))
    kl = (logfunc(pt, p_hat) + logfunc(1 - pt, 1 - p_hat)).sum()
    reg = lam * (0.5 * (f2 ** 2).sum() + 0.5 * (f ** 2).sum())
    neg = torch.where(y < 0, y, torch.zeros_like(y)) ** 2
    nnv = nn * neg.reshape(Bt, -1).sum(1).mean()
    from itertools import combinations
    a = y[B:].reshape(B, S, -1).abs()
    ovv = torch.stack([(1.0 - (a[:, i] - a[:, j]).abs() / (torch.maximum(a[:, i], a[:, j]) + 1e-8)).mean(-1)
                       for i, j in combinations(range(S), 2)], 1).mean(1).mean()
    cost = l2 + sdr + beta * kl + lam * reg + ov * ovv + nn * nnv
    cost.backward()
    assert float((p_hat > 1).sum()) > 0 and float((p_hat < 1).sum()) > 0          # both sides of the upper clip bound are exercised
    assert abs(c - cost.item()) < 1e-10 * abs(cost.item())
    for k in P:
        assert rel(g[k], Pt[k].grad.numpy()) < 1e-9, k


def test_kmeans_oracle_matches_sklearn_lloyd():
    """Independent cross-check of the hard k-means restatement (Kmeans_2.py:86-188): without silence weights and with one try it
    is plain Lloyd's algorithm for a fixed number of iterations from given seeds -- compare labels/centroids with scikit-learn."""
    sk = pytest.importorskip('sklearn.cluster')
    from oracle import kmeans as okm
    rng = np.random.RandomState(3)
    b, L, E, C, iters = 2, 400, 6, 3, 4
    means = rng.randn(b, C, E) * 3.0
    lab_true = rng.randint(0, C, (b, L))
    X = means[np.arange(b)[:, None], lab_true] + 0.3 * rng.randn(b, L, E)
    Xn = X / np.linalg.norm(X, axis=-1, keepdims=True)                      # the reference l2-normalises its input (Kmeans_2.py:61)
    init = np.stack([np.array([np.flatnonzero(lab_true[i] == c)[0] for c in range(C)]) for i in range(b)]).astype(np.int32)
    cent, labels, best = okm.kmeans(X, init, C, 1, iters, assign_at_end=True)
    for i in range(b):
        km = sk.KMeans(n_clusters=C, init=Xn[i][init[i]], n_init=1, max_iter=iters, tol=0.0, algorithm='lloyd').fit(Xn[i])
        assert np.allclose(km.cluster_centers_, cent[i], atol=1e-10)
        assert np.array_equal(km.labels_, labels[i])


def test_torch_cpu_step_matches_the_numpy_oracle():
    """oracle/torch_step.py (the timed CPU baseline of bench.py: torch-CPU / oneDNN formulation, fused nn.LSTM kernel with the TF cell
    re-laid out) against oracle/step.py::front_dpcl_loss in float64: cost, every gradient, and one AMSGrad update."""
    from oracle import torch_step
    rng = np.random.RandomState(1)
    B, S, L, W, N, hop, LS, NL, E = 3, 2, 1024, 64, 16, 16, 16, 2, 8
    P = step.init_params(rng, np.float64, front_W=W, N=N, D_in=N, layer_size=LS, nb_layers=NL, E=E, F=N, conv1d_scale=0.05)
    xn = rng.randn(B, S, L) * 0.1
    xm = xn.sum(1)
    c_ref, g_ref, _, _ = step.front_dpcl_loss(xm, xn, P, hop, NL, E)
    ts = torch_step.FrontDPCLStep(P, hop, NL, E, lr=1e-3, dtype=torch.float64)
    c = ts.step(torch.tensor(xm), torch.tensor(xn))
    assert abs(c - c_ref) < 1e-12 * abs(c_ref)
    names = sorted(g_ref)
    assert names == ts.names
    for n in names:
        assert rel(ts.P[n].grad.numpy(), g_ref[n]) < 1e-11, n
    opt = optim.AMSGrad(1e-3)
    plist = [P[n].copy() for n in names]
    opt.apply(plist, [g_ref[n] for n in names])
    for n, p in zip(names, plist):
        assert rel(ts.P[n].detach().numpy(), p) < 1e-11, n
