# -*- coding: utf-8 -*-
"""Deep Clustering separator (reference models/dpcl.py), host mirror over the HIP kernels."""
from ams_hip import functional as F
from ams_hip.graph import Node, get_default_graph, scope
from models.network import Separator
from utils.ops import BLSTM, Conv1D, f_props


class DPCL(Separator):

    def __init__(self, graph=None, **kwargs):
        kwargs['mask_a'] = 1.0
        kwargs['mask_b'] = 0.0

        super(DPCL, self).__init__(graph, **kwargs)
        self.init_separator()

    @scope
    def prediction(self):
        # DPCL network (dpcl.py:19-39): BLSTM x nb_layers -> Conv1D -> Reshape [B,T,F,E] -> Normalize(3)
        self.true_masks = self.y
        self.count_labels_for = self.embedding_size          # network.py create_masks: labels are counted as they are made (fused loss)
        E, Fq = self.embedding_size, self.F
        layers = [BLSTM(self.layer_size, name='BLSTM_' + str(i), drop_val=self.rdropout,
                        in_dim=(Fq if i == 0 else self.layer_size)) for i in range(self.nb_layers)]
        conv = Conv1D([1, self.layer_size, E * Fq])
        x_node = self.X

        def _embed(run):
            x = x_node.value(run)
            return conv.f_prop(f_props(layers, x, then=conv))             # [B, T, F*E]  (column = f*E + e)
        self._embed = Node('embed', _embed, register=False)
        self._embed_normalized = True                        # prediction = l2-normalise(_embed): separate_host hands _embed to the k-means
        # Reshape + Normalize(3); only evaluated when the embeddings themselves are fetched (inference / k-means):
        # a training step goes u -> fused normalise+loss kernel and never writes V.
        return Node('prediction', lambda run: F.l2norm_keep(self._embed.value(run), E)[0], register=False)

    @scope
    def cost(self):
        # dpcl.py:41-87
        self.prediction
        embed, y, E = self._embed, self.y, self.embedding_size
        g = get_default_graph()

        def _terms(run):
            u = embed.value(run)
            Y = y.value(run)
            return F.dpcl_loss_u(u, Y.reshape(u.shape[0], -1, Y.shape[-1]), E)
        both = Node('loss', _terms)
        terms = Node('terms', lambda run: both.value(run)[1])
        cost = Node('cost_value', lambda run: both.value(run)[0])
        g.summaries['cost/cost'] = cost
        for k, name in ((1, '1'), (2, '2'), (3, '3')):           # dpcl.py:83-85
            g.summaries['cost/' + name] = Node(name, lambda run, k=k: terms.value(run)[k])
        return cost
