"""Node builders for Separator.separate / enhance / enhance_cost / cost_finetuning
(reference models/network.py:554-582, 610-724; models/adapt.py:404-431)."""
import torch

from . import functional as F
from .graph import Node, get_default_graph


def build_separate(sep, KMeans):
    """Separator.separate (network.py:554-582): KMeans on the embeddings -> masks -> masked mixture representation."""
    pred, X_input = sep.prediction, sep.X_input
    E, S = sep.embedding_size, sep.S
    emb = Node('embeddings', lambda run: pred.value(run).reshape(pred.value(run).shape[0], -1, E))
    sep.embeddings = emb
    lat = Node('latent', lambda run: X_input.value(run).abs(), register=False) if sep.with_silence else None
    km = KMeans(nb_clusters=S, nb_tries=sep.nb_tries, nb_iterations=sep.nb_steps, input_tensor=emb, beta=sep.beta,
                latent_space_tensor=lat, threshold=sep.threshold, assign_at_end=sep.args['end_assign'],
                init_indices=sep.args.get('kmeans_init_indices'), seeding=sep.args.get('kmeans_seeding') or 'reference',
                pre_norm=(sep._embed, E) if getattr(sep, '_embed_normalized', False) else None, dist=getattr(sep, 'dist', None))
    sep.kmeans = km
    _, labels = km.network

    def _masks(run):
        lab = labels.value(run)
        if sep.beta is None:
            return F.one_hot_masks(lab, S)                     # [B, TF, S] hard assignments (no gradient)
        return lab                                             # soft assignments
    sep.masks = Node('masks', _masks)

    def _sep(run):
        return F.apply_masks(X_input.value(run), sep.masks.value(run)).unsqueeze(-1)      # [B*S, T, F, 1]
    out = Node('separated', _sep)
    sep.separated = out
    return out


def build_enhance(sep, BLSTM, Conv1D, f_props):
    """Separator.enhance (network.py:610-660)."""
    S, Fq = sep.S, sep.F
    separate, X_input = sep.separate, sep.X_input
    a = sep.args
    LS = a['layer_size_enhance']
    layers = [BLSTM(LS, drop_val=a['recurrent_dropout_enhance'], name='BLSTM_' + str(i), in_dim=(2 * Fq if i == 0 else LS))
              for i in range(a['nb_layers_enhance'])]
    conv = Conv1D([1, LS, Fq])

    def _net(run):
        Xin = X_input.value(run)
        B, T, _ = Xin.shape
        sepv = separate.value(run).reshape(B, S, T, Fq)
        z = torch.cat([sepv, Xin.unsqueeze(1).expand(B, S, T, Fq)], dim=3).reshape(B * S, T, 2 * Fq)
        if a['normalize_enhance']:
            m = z.mean(dim=(1, 2), keepdim=True)
            v = ((z - m) ** 2).mean(dim=(1, 2), keepdim=True)
            z = (z - m) / torch.sqrt(v)
        y = conv.f_prop(f_props(layers, z.contiguous()))              # [B*S, T, F]
        return F.enhance_output(y, Xin, S, a['nonlinearity'])          # (cost_in [B, TF, S], separated [B, S, TF])
    both = Node('enhance_out', _net)
    sep.cost_in = Node('cost_in', lambda run: both.value(run)[0])
    sep.enhanced_masks = sep.cost_in

    def _out(run):
        sp = both.value(run)[1]
        B = sp.shape[0]
        T = X_input.value(run).shape[1]
        return sp.reshape(B * S, T, Fq, 1)
    out = Node('enhanced', _out)
    sep.separated = out
    return out


def build_enhance_cost(sep):
    """Separator.enhance_cost (network.py:662-693): PIT sum of squared errors against |X_non_mix|."""
    sep.enhance
    cost_in, X_nm = sep.cost_in, sep.X_non_mix

    def _cost(run):
        c = cost_in.value(run)                                         # [B, TF, S]
        B, TF, S = c.shape
        est = c.transpose(1, 2).contiguous()                           # [B, S, TF]
        tgt = X_nm.value(run).reshape(B, TF, S).transpose(1, 2).contiguous()
        return F.pit_l2(tgt, est, 'sum', 'sum', 1.0)
    cost = Node('cost_value', _cost)
    get_default_graph().summaries['enhance_cost/cost'] = cost
    return cost


def build_cost_finetuning(model, est_node):
    """cost_finetuning (adapt.py:404-431 / network.py:697-724): 0.5*sum_l, mean_s, min over permutations, mean_b."""
    x_non_mix = model.x_non_mix

    def _cost(run):
        est = est_node.value(run)
        return F.pit_l2(x_non_mix.value(run), est, 'sum', 'mean', 0.5)
    cost = Node('cost_value', _cost)
    get_default_graph().summaries['cost_finetuning/cost'] = cost
    return cost
