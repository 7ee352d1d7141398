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
            raise NotImplementedError('--sampling (negative sampling, L41.py:69-147) is off by default and not on the HIP path')

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

        def _pred(run):
            x = x_node.value(run)
            u = conv.f_prop(f_props(layers, x, then=conv))
            if normalize:
                return F.l2norm(u, E)
            return u.reshape(u.shape[:-1] + (Fq, E))
        return Node('prediction', _pred, register=False)

    @scope
    def cost(self):
        # L41.py:47-186 (sampling=None)
        pred, y, I, spk, normalize = self.prediction, self.y, self.I, self.speaker_vectors, self.normalize

        def _cost(run):
            return F.l41_loss(pred.value(run), y.value(run), spk, I.value(run), normalize)
        cost = Node('cost_value', _cost)
        get_default_graph().summaries['cost/cost'] = cost
        return cost
