"""Host mirror of the reference's op library subset used on the hot path (reference utils/ops.py).

Same names and call shapes -- ``scope``, ``get_scope_variable``, ``f_props``, ``BLSTM``, ``Conv1D``, ``Reshape``,
``Normalize``, ``log10``, ``kl_div`` -- but every ``f_prop`` launches hand-written HIP kernels through
ams_hip.functional instead of building TF graph ops.  Layers create their variables at construction time
under the active variable scope, so names match the reference's checkpoints
('prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel', 'prediction/W', ...).
"""
import numpy as np
import torch

from ams_hip import functional as F
from ams_hip.graph import scope, get_scope_variable, get_default_graph  # noqa: F401  (re-exported API)

rng = np.random.RandomState(42)          # reference utils/ops.py:5 (module-level, shared by Conv1D inits)


def f_props(layers, x, then=None):
    """utils/ops.py:82-85."""
    for i, layer in enumerate(layers):
        if isinstance(layer, BLSTM):                 # the layer whose backward runs just before the first layer's (functional.py)
            layer._last_capped = (i == 1 and isinstance(layers[0], BLSTM))
        x = layer.f_prop(x)
    return x


def log10(x):
    """utils/ops.py:56-59."""
    return torch.log(x) / np.log(10.0)


def _graph_rng():
    g = get_default_graph()
    if not hasattr(g, '_init_rng'):
        g._init_rng = np.random.RandomState(g.seed)
    return g._init_rng


def xavier_uniform(shape):
    """tf.contrib.layers.xavier_initializer_conv2d for the shapes the reference uses:
    [W] -> +-sqrt(3/W);  [W,N] -> +-sqrt(6/(W+N))."""
    if len(shape) == 1:
        lim = np.sqrt(3.0 / shape[0])
    else:
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return _graph_rng().uniform(-lim, lim, size=shape).astype('float32')


def glorot_uniform(shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return _graph_rng().uniform(-lim, lim, size=shape).astype('float32')


class Reshape:
    def __init__(self, shape, name='Reshape'):
        self.shape = shape
        self.name = name

    def f_prop(self, x):
        shape = [s() if callable(s) else s for s in self.shape]
        return x.reshape(shape)


class Normalize:
    """tf.nn.l2_normalize over the last axis of a [B,T,F,E] tensor (utils/ops.py:317-324)."""

    def __init__(self, axis, name='Normalize'):
        self.axis = axis
        self.name = name

    def f_prop(self, x):
        E = x.shape[-1]
        return F.l2norm(x.reshape(x.shape[:-2] + (x.shape[-2] * E,)), E)


class BLSTM:
    """utils/ops.py:358-383.  `in_dim` is an addition: TF infers it at graph-build time from the static shape;
    here variables are created eagerly so the caller states it."""

    def __init__(self, hid_dim, name, drop_val=0.0, in_dim=None):
        self.hid_dim = hid_dim
        self.name = name
        self.drop_val = float(drop_val or 0.0)
        if not 0.0 <= self.drop_val < 1.0:
            raise ValueError('recurrent_dropout must be in [0, 1), got %r' % (drop_val,))
        H = hid_dim // 2
        g = get_default_graph()

        def mk(direction):
            with g.variable_scope(direction + '_' + name):
                with g.variable_scope('rnn'):
                    with g.variable_scope('basic_lstm_cell'):
                        k = g.get_variable('kernel', (in_dim + H, 4 * H), glorot_uniform)
                        b = g.get_variable('bias', (4 * H,), lambda s: np.zeros(s, 'float32'))
            return k, b

        self.Kf, self.bf = mk('forward')
        self.Kb, self.bb = mk('backward')
        self.Kf._ams_twin, self.bf._ams_twin = self.Kb, self.bb      # storage hint for FlatOptimizer (interleaved rows)

    def f_prop(self, x):
        # utils/ops.py:362-363: keep = cond(is_training, 1 - drop_val, 1.0); with keep = 1 tf.nn.dropout is the identity, so only a
        # training pass with drop_val != 0 takes the wrapper form (per-step kernels, masks drawn on the device)
        if self.drop_val != 0.0:
            from ams_hip.graph import current_run
            run = current_run()
            if run is not None and run.training and torch.is_grad_enabled():
                return F.blstm_dropout(x, self.Kf, self.bf, self.Kb, self.bb, 1.0 - self.drop_val)
        return F.blstm(x, self.Kf, self.bf, self.Kb, self.bb, getattr(self, '_last_capped', False))


class Conv1D:
    """utils/ops.py:486-503: kernel width 1 => dense layer.  filter_shape = [1, Din, Dout]."""

    def __init__(self, filter_shape, function=lambda x: x, stride=1, padding='SAME', name='Conv1D'):
        fan_in = np.sqrt(2 / (float(filter_shape[1] + filter_shape[2])))
        lim = np.sqrt(2 / fan_in)                       # the reference's nested-sqrt range
        g = get_default_graph()
        self.W = g.get_variable('W', tuple(filter_shape[1:]),
                                lambda s: rng.uniform(low=-lim, high=lim, size=s).astype('float32'))
        self.b = g.get_variable('b', (filter_shape[-1],), lambda s: np.zeros(s, 'float32'))
        self.function = function
        self.name = name
        if filter_shape[0] != 1 or stride != 1:
            raise NotImplementedError('only the width-1 / stride-1 Conv1D the reference builds is supported')

    def f_prop(self, x):
        return self.function(F.dense(x, self.W, self.b))
