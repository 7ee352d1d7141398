"""Oracle: waveform-domain costs -- SDR improvement, pretraining cost, PIT costs
(reference models/network.py:196-221, 662-724; models/adapt.py:307-431).

Test infrastructure only -- see oracle/__init__.py.
"""
from itertools import permutations
import numpy as np
from .separate import log10


def sdr_improvement(x_mix, s_target, s_approx, with_perm=False):
    """Network.sdr_improvement (network.py:196-221).  s_target/s_approx broadcastable [B,S,L] or [B,P,S,L].
    Returns (improvement scalar, loss tensor = (|t|^2 |a|^2) / (<t,a>^2 + 1e-12))."""
    S = s_target.shape[-2]
    mix = np.broadcast_to(x_mix[:, None, :], (x_mix.shape[0], S, x_mix.shape[1]))
    tn = np.sum(s_target ** 2, axis=-1)
    an = np.sum(s_approx ** 2, axis=-1)
    mn = np.sum(mix ** 2, axis=-1)
    ts2 = np.sum(s_target * s_approx, axis=-1) ** 2
    tm2 = np.sum(s_target * mix, axis=-1) ** 2
    with np.errstate(divide='ignore', invalid='ignore'):
        separated = 10.0 * log10(1.0 / ((tn * an) / ts2 - 1.0))
        non_sep = 10.0 * log10(1.0 / ((tn * mn) / tm2 - 1.0))
    loss = (tn * an) / (ts2 + 1e-12)
    val = (separated - non_sep).mean(axis=-1)
    val = val.mean(axis=-1) if not with_perm else val.mean(axis=0).max(axis=-1)
    return val, loss


def pretrain_cost(x_mix, x_non_mix, back, loss_kind):
    """Adapt.cost, pretraining branch (adapt.py:321-337): returns (loss, l2, sdr)."""
    l2 = np.sum((x_non_mix - back) ** 2, axis=-1).sum(axis=-1).mean()
    _, sdr_t = sdr_improvement(x_mix, x_non_mix, back)
    sdr = sdr_t.mean()
    loss = l2 if loss_kind == 'l2' else sdr if loss_kind == 'sdr' else l2 + sdr
    return loss, l2, sdr


def pretrain_cost_bwd(x_non_mix, back, loss_kind):
    """d loss / d back for the pretraining branch."""
    B, S, L = back.shape
    g = np.zeros_like(back)
    if loss_kind != 'sdr':
        g += -2.0 * (x_non_mix - back) / B
    if loss_kind != 'l2':
        tn = np.sum(x_non_mix ** 2, axis=-1, keepdims=True)
        an = np.sum(back ** 2, axis=-1, keepdims=True)
        ta = np.sum(x_non_mix * back, axis=-1, keepdims=True)
        den = ta ** 2 + 1e-12
        # d/da [ tn*an/den ] = tn*2a/den - tn*an*2*ta*t/den^2
        g += (tn * 2.0 * back / den - tn * an * 2.0 * ta * x_non_mix / den ** 2) / (B * S)
    return g


def adapt_regularizers(f, f2, front_y, lam, nonneg):
    """adapt.py:312-316,379-384: lam * (lam * (l2_loss(f2)+l2_loss(f)))  and  nn * (nn * mean_b sum neg^2)
    (coefficients applied twice -- quirk C-2)."""
    reg = lam * (0.5 * np.sum(f2 ** 2) + 0.5 * np.sum(f ** 2))
    neg = np.where(front_y < 0, front_y, 0.0) ** 2
    nn = nonneg * neg.reshape(neg.shape[0], -1).sum(axis=1).mean()
    return reg, nn


def perms(S):
    """itertools.permutations(range(S)) in lexicographic order."""
    return np.array(list(permutations(range(S))), dtype=np.int64)


def pit_cost_adapt(x_mix, x_non_mix, back, loss_kind):
    """Adapt.cost non-pretraining branch (adapt.py:339-372), including quirk C-3: the sdr term
    uses the UN-permuted back [B,S,L] against the [B,1,S,L] target => broadcast [B,B,S,L]."""
    B, S, L = back.shape
    P = perms(S)
    pb = back[:, P]                                        # [B,P,S,L]
    X = x_non_mix[:, None]
    l2 = np.mean((X - pb) ** 2, axis=-1).sum(axis=-1).min(axis=-1).mean()
    _, sdr_t = sdr_improvement(x_mix, X, back, True)       # loss tensor [B,B,S]
    sdr = sdr_t.min(axis=1).sum(axis=-1).mean()
    loss = l2 if loss_kind == 'l2' else sdr if loss_kind == 'sdr' else 1e-3 * l2 + sdr
    return loss, l2, sdr


def pit_l2_best(x_non_mix, est, reduce_l, reduce_s, scale=1.0):
    """Generic PIT squared error: returns (cost, best permutation index [B]).
    reduce_l in {'sum','mean'} over samples, reduce_s in {'sum','mean'} over speakers."""
    S = est.shape[1]
    P = perms(S)
    d = (x_non_mix[:, None] - est[:, P]) ** 2
    d = d.sum(axis=-1) if reduce_l == 'sum' else d.mean(axis=-1)
    d = scale * d
    d = d.sum(axis=-1) if reduce_s == 'sum' else d.mean(axis=-1)
    best = np.argmin(d, axis=1)
    return d.min(axis=1).mean(), best


def cost_finetuning(x_non_mix, est):
    """adapt.py:404-431 / network.py:697-724: 0.5*sum_l, mean_s, min_perm, mean_b."""
    return pit_l2_best(x_non_mix, est, 'sum', 'mean', 0.5)


def enhance_cost(X_non_mix, cost_in):
    """Separator.enhance_cost (network.py:662-693).  X_non_mix [B,T,F,S], cost_in [B,TF,S]."""
    B = cost_in.shape[0]
    S = cost_in.shape[2]
    est = cost_in.transpose(0, 2, 1)                       # [B,S,TF]
    tgt = X_non_mix.reshape(B, -1, S).transpose(0, 2, 1)   # [B,S,TF]
    return pit_l2_best(tgt, est, 'sum', 'sum', 1.0)


def pit_l2_bwd(x_non_mix, est, best, reduce_l, reduce_s, scale=1.0):
    """Sub-gradient through the selected permutation only."""
    B, S, L = est.shape
    P = perms(S)
    g = np.zeros_like(est)
    coef = scale * 2.0 / B
    if reduce_l == 'mean':
        coef /= L
    if reduce_s == 'mean':
        coef /= S
    for b in range(B):
        p = P[best[b]]
        for s in range(S):
            g[b, p[s]] += -coef * (x_non_mix[b, s] - est[b, p[s]])
    return g
