"""Oracle: optimizers (reference models/network.py:167-194, utils/ops.py:639-792).

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np


def lr_decay(lr, epoch, decay_epoch):
    """tf.train.exponential_decay(lr, global_epoch, decay_epoch, 0.5, staircase=True) (network.py:175-177)."""
    return lr * 0.5 ** (epoch // decay_epoch)


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm (network.py:191-192): g * clip / max(||g||, clip)."""
    gn = np.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads))
    s = clip / max(gn, clip)
    return [g * s for g in grads], gn


class AMSGrad:
    """utils/ops.py:648-792; the reference instantiates AMSGrad(lr, beta1=.9, beta2=.99, epsilon=1e-3)
    for --optimizer Adam with the UN-decayed learning rate (network.py:181-182, quirk C-7)."""

    def __init__(self, lr, beta1=0.9, beta2=0.99, eps=1e-3):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.b1p, self.b2p = beta1, beta2                  # beta*_power start at beta* (ops.py:677-678)
        self.slots = {}

    def apply(self, params, grads):
        lr_t = self.lr * np.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        for k, (p, g) in enumerate(zip(params, grads)):
            if k not in self.slots:
                self.slots[k] = [np.zeros_like(p), np.zeros_like(p), np.zeros_like(p)]
            m, v, vh = self.slots[k]
            m[...] = self.b1 * m + (1.0 - self.b1) * g
            v[...] = self.b2 * v + (1.0 - self.b2) * g * g
            vh[...] = np.maximum(v, vh)
            p -= (lr_t * m / (np.sqrt(vh) + self.eps)).astype(p.dtype)
        self.b1p *= self.b1                               # powers updated after the step (ops.py:781-792)
        self.b2p *= self.b2


class RMSProp:
    """tf.train.RMSPropOptimizer(lr): decay .9, momentum 0, eps 1e-10, ms initialised to ones (App. A-13)."""

    def __init__(self, lr, decay=0.9, eps=1e-10):
        self.lr, self.decay, self.eps = lr, decay, eps
        self.slots = {}

    def apply(self, params, grads):
        for k, (p, g) in enumerate(zip(params, grads)):
            if k not in self.slots:
                self.slots[k] = np.ones_like(p)
            ms = self.slots[k]
            ms[...] = self.decay * ms + (1.0 - self.decay) * g * g
            p -= (self.lr * g / np.sqrt(ms + self.eps)).astype(p.dtype)


class Momentum:
    """tf.train.MomentumOptimizer(lr, 0.9): accum = .9*accum + g; p -= lr*accum (App. A-13)."""

    def __init__(self, lr, momentum=0.9):
        self.lr, self.mom = lr, momentum
        self.slots = {}

    def apply(self, params, grads):
        for k, (p, g) in enumerate(zip(params, grads)):
            if k not in self.slots:
                self.slots[k] = np.zeros_like(p)
            a = self.slots[k]
            a[...] = self.mom * a + g
            p -= (self.lr * a).astype(p.dtype)
