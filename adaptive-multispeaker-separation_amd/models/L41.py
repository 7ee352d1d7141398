# -*- coding: utf-8 -*-
"""Lab41 source-contrastive separator (reference models/L41.py), host mirror over the HIP kernels."""
import numpy as np

from ams_hip import functional as F
from ams_hip.graph import Node, get_default_graph, scope
from models.network import Separator
from utils.ops import BLSTM, Conv1D, f_props, _graph_rng


class L41Model(Separator):

    def __init__(self, graph=None, **kwargs):
        kwargs['mask_a'] = 1.0
        kwargs['mask_b'] = -1.0

        super(L41Model, self).__init__(graph, **kwargs)

        if self.sampling is not None:
            # negative sampling (L41.py:69-147): limits of the loss kernel (csrc/l41.hip)
            K, S = int(self.sampling), int(self.S)
            if K < 1 or K > 16 or (self.ns_method == 'k-nearest' and S * K > 32):
                raise ValueError('--sampling %d: the L41 loss kernel takes 1..16 negatives per set and at most 32 per utterance '
                                 '(nb_speakers * sampling for k-nearest)' % K)
            if K > self.num_speakers - (0 if self.ns_method == 'k-nearest' else S):
                raise ValueError('--sampling %d exceeds the speakers available (%d)' % (K, self.num_speakers))

        # Define the speaker vectors to use during training (L41.py:16-18): truncated normal, stddev sqrt(2/E)
        E = self.embedding_size

        def _trunc_normal(shape):
            sd = np.sqrt(2.0 / float(E))
            r = _graph_rng()
            v = r.standard_normal(shape) * sd
            bad = np.abs(v) > 2 * sd
            while bad.any():                                   # tf.truncated_normal re-draws beyond 2 sigma
                v[bad] = r.standard_normal(int(bad.sum())) * sd
                bad = np.abs(v) > 2 * sd
            return v.astype('float32')
        self.speaker_vectors = get_default_graph().get_variable('speaker_centroids', (self.num_speakers, E), _trunc_normal)
        self.init_separator()

    @scope
    def prediction(self):
        # L41 network (L41.py:21-45): as DPCL, Normalize(3) only when self.normalize
        E, Fq = self.embedding_size, self.F
        y = self.y
        self.true_masks = Node('true_masks', lambda run: 1.0 + y.value(run), register=False)
        layers = [BLSTM(self.layer_size, name='BLSTM_' + str(i), drop_val=self.rdropout,
                        in_dim=(Fq if i == 0 else self.layer_size)) for i in range(self.nb_layers)]
        conv = Conv1D([1, self.layer_size, E * Fq])
        x_node, normalize = self.X, self.normalize

        # the dense output before Normalize(3): a training step hands THIS to the loss, which normalises inside its own pass (cost below)
        self._embed = Node('embed', lambda run: conv.f_prop(f_props(layers, x_node.value(run), then=conv)), register=False)
        self._embed_normalized = bool(normalize)            # prediction = l2-normalise(_embed) only then (separate_host: k-means from _embed)
        embed = self._embed

        def _pred(run):
            u = embed.value(run)
            if normalize:
                return F.l2norm(u, E)
            return u.reshape(u.shape[:-1] + (Fq, E))
        return Node('prediction', _pred, register=False)

    @scope
    def cost(self):
        # L41.py:47-186; --sampling K adds ns_rate * mean_k -log(sigmoid(-<neg_k, emb>)) per bin (:69-147,165-166)
        pred, y, I, spk, normalize = self.prediction, self.y, self.I, self.speaker_vectors, self.normalize
        sampling, ns_rate, ns_method, tot = self.sampling, self.ns_rate, self.ns_method, self.num_speakers

        def _cost(run):
            Iv = I.value(run)
            neg = None
            if sampling is not None:
                if ns_method == 'k-nearest':
                    neg = F.l41_knearest(spk, Iv, sampling, normalize)                 # [B,S,K], set of the bin's dominant speaker
                else:
                    neg = F.l41_random_negatives(Iv, tot, sampling)                    # [B,1,K], one set per utterance
            if normalize and run.training and id(pred) not in run.cache:
                # K13 fused into K15: the un-normalised dense output goes to the loss kernels, V is never written in a training step
                return F.l41_loss(self._embed.value(run), y.value(run), spk, Iv, normalize, neg_idx=neg, ns_rate=ns_rate, from_u=True)
            return F.l41_loss(pred.value(run), y.value(run), spk, Iv, normalize, neg_idx=neg, ns_rate=ns_rate)
        cost = Node('cost_value', _cost)
        get_default_graph().summaries['cost/cost'] = cost
        return cost
