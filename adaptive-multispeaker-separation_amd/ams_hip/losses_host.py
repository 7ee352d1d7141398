"""Node builders for Adapt.separator / Adapt.back / Adapt.cost and Network.sdr_improvement
(reference models/adapt.py:136-252, 307-402; models/network.py:196-221).

Kept out of models/adapt.py so that file reads like the reference's class; every heavy op is a HIP kernel via
ams_hip.functional.
"""

from . import functional as F
from .graph import Node, get_default_graph, get_scope_variable


def sparse_constraint(y, p, dist=None):
    """adapt.py:130-132 + utils/ops.py:46-54.  p_hat is a batch SUM fed to a non-linear KL, so under data
    parallelism it is all-reduced before the KL.  |.| column sum, KL and their backward are HIP kernels
    (csrc/preproc.hip: ams_abs_colsum_fwd, ams_kl_sparsity_*)."""
    Bt = y.shape[0]
    return F.sparse_kl(y.reshape(Bt, -1), p, dist)


def build_adapt_separator(m):
    """Adapt.separator (adapt.py:136-203)."""
    front = m.front
    S = m.S
    x_mix = m.x_mix

    def _overlap(run):
        y = front.value(run)
        B = x_mix.value(run).shape[0]
        return F.overlap_metric(y, B, S)
    m.overlapping = Node('overlapping', _overlap)
    m.overlapping_constraint = m.overlapping

    if m.pretraining:
        def _sep(run):
            y = front.value(run)
            B = x_mix.value(run).shape[0]
            return F.pretrain_separator(y, B, S, m.separation)            # [B*S, T', N]
        return Node('output', _sep)

    sep = m.sepNet
    m.prediction = sep.prediction
    m.true_masks = getattr(sep, 'true_masks', None)
    m.X_non_mix = sep.X_non_mix
    m.X = sep.X
    return Node('output', lambda run: m.sepNet.output.value(run))


def build_adapt_back(m):
    """Adapt.back (adapt.py:205-252): (unpool with the mixture's argmax |) transposed conv to waveforms [B,S,L]."""
    from utils.ops import xavier_uniform
    m.window_filter_2 = get_scope_variable('window', 'value', shape=(m.window,), initializer=xavier_uniform)
    m.bases_2 = get_scope_variable('bases', 'value', shape=(m.window, m.N), initializer=xavier_uniform)
    w2, b2 = m.window_filter_2, m.bases_2
    m.conv_filter_2 = Node('filters', lambda run: F.front_filter(w2, b2))
    f2 = m.conv_filter_2
    x_mix = m.x_mix
    S, hop, P, N = m.S, m.hop_size, m.max_pool_value, m.N
    sep_holder = m          # m.separator is resolved lazily so recipes can build `back` before wiring sepNet.output

    def _back(run):
        z = sep_holder.separator.value(run)
        xm = x_mix.value(run)
        B, L = xm.shape
        z = z.reshape(B * S, -1, N)
        if m.with_max_pool:
            am = m.argmax.value(run)[:B]                                   # mixture argmax, tiled S times (adapt.py:212-218)
            out = F.synth_unpool(z, am, f2.value(run), L, S, P, hop)
        elif m.with_average_pool:
            out = F.synth_avgpool(z, f2.value(run), P, L)
        else:
            out = F.synth_strided(z, f2.value(run), hop, L)
        return out.reshape(B, S, L)
    back = Node('back_output', _back)
    m.output = back
    x_non_mix = m.x_non_mix
    m.sdr_imp = Node('sdr_imp', lambda run: F.sdr_improvement(x_mix.value(run), x_non_mix.value(run), back.value(run))[0])
    return back


def build_adapt_cost(m):
    """Adapt.cost (adapt.py:307-402)."""
    g = get_default_graph()
    back, front = m.back, m.front
    x_mix, x_non_mix = m.x_mix, m.x_non_mix
    f1, f2 = m.conv_filter, m.conv_filter_2

    # the pair table (one pass over the waveforms) is shared by the cost and by the SDR-improvement summary; the summary's own
    # arithmetic (gradient-free, for pit_cost_adapt two more cross-batch products) runs only in a step that fetches it
    def _tables(run):
        xm, xn, bk = x_mix.value(run), x_non_mix.value(run), back.value(run)
        return F.pair_stats(xn, bk, xm) if m.pretraining else F.pit_adapt_tables(xm, xn, bk)
    tables = Node('tables', _tables, register=False)

    def _parts(run):
        xm, xn, bk = x_mix.value(run), x_non_mix.value(run), back.value(run)
        if m.pretraining:
            return F.pretrain_cost(xm, xn, bk, want_imp=False, st=tables.value(run))        # tensor [2] = l2, sdr
        return F.pit_cost_adapt(xm, xn, bk, want_imp=False, tables=tables.value(run))      # l2, sdr (quirk C-3)
    parts = Node('parts', _parts)

    def _imp(run):
        if m.pretraining:
            return F.pretrain_improvement(tables.value(run))
        return F.pit_adapt_improvement(x_mix.value(run), x_non_mix.value(run), tables.value(run))

    def _cost(run):
        p = parts.value(run)
        l2, sdr = p[0:1], p[1:2]
        if m.loss == 'l2':
            loss = l2
        elif m.loss == 'sdr':
            loss = sdr
        else:
            loss = (l2 if m.pretraining else 1e-3 * l2) + sdr
        cost_value = loss
        if m.beta != 0.0:
            cost_value = cost_value + m.beta * m.sparse_constraint.value(run)
        if m.l != 0.0:
            # coefficient applied twice (quirk C-2): l * (l * (l2_loss(f2) + l2_loss(f)))
            reg = m.l * 0.5 * (F.sumsq(f2.value(run)) + F.sumsq(f1.value(run)))
            cost_value = cost_value + m.l * reg
        if m.overlap_coef != 0.0:
            cost_value = cost_value + m.overlap_coef * m.overlapping_constraint.value(run)
        if m.non_negativity is not None and m.non_negativity != 0.0:
            nn = m.non_negativity * F.negative_energy(front.value(run))
            cost_value = cost_value + m.non_negativity * nn
        return cost_value
    cost = Node('cost_value', _cost)
    g.summaries['cost/loss_values/l2_loss'] = Node('l2_loss', lambda run: parts.value(run)[0])
    g.summaries['cost/loss_values/SDR'] = Node('SDR', lambda run: parts.value(run)[1])
    g.summaries['cost/loss_values/SDR_improvement'] = Node('SDR_improvement', _imp)
    g.summaries['cost/loss_values/loss'] = cost
    m.sdr_imp = g.summaries['cost/loss_values/SDR_improvement']
    return cost


def sdr_improvement(m, s_target, s_approx, with_perm=False):
    """Network.sdr_improvement (network.py:196-221) as nodes."""
    x_mix = m.x_mix
    node = Node('sdr_improvement', lambda run: F.sdr_improvement(x_mix.value(run), s_target.value(run), s_approx.value(run),
                                                                 with_perm), register=False)
    return (Node('sdr_imp_val', lambda run: node.value(run)[0], register=False),
            Node('sdr_loss', lambda run: node.value(run)[1], register=False))
