"""Oracle: Conv1D(k=1) dense layer + Reshape + l2-normalise (reference utils/ops.py:486-503, 310-324).

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np

L2_EPS = 1e-12   # tf.nn.l2_normalize epsilon


def dense_fwd(x, W, b):
    """Conv1D.f_prop with kernel width 1: u = x.W + b.  x [..., Din], W [Din, Dout] (the reference
    stores [1, Din, Dout], ops.py:489-494)."""
    return x @ W + b


def dense_bwd(x, W, du):
    x2 = x.reshape(-1, x.shape[-1])
    du2 = du.reshape(-1, du.shape[-1])
    return (du2 @ W.T).reshape(x.shape), x2.T @ du2, du2.sum(axis=0)


def l2norm_fwd(u, E):
    """Reshape([B,T,F,E]) then Normalize(3) (dpcl.py:31-32): column index = f*E + e.
    v = u * rsqrt(max(sum_e u^2, 1e-12))."""
    shp = u.shape
    g = u.reshape(shp[:-1] + (shp[-1] // E, E))
    ss = np.sum(g * g, axis=-1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ss, L2_EPS))
    return g * inv, inv


def l2norm_bwd(v, inv, dv):
    """du = (dv - v <v,dv>) * inv  (eps clamp inactive unless sum u^2 < 1e-12,
    in which case v = u*inv with constant inv => du = dv*inv)."""
    dot = np.sum(v * dv, axis=-1, keepdims=True)
    active = inv < 1.0 / np.sqrt(L2_EPS) * (1 - 1e-12)
    return np.where(active, (dv - v * dot) * inv, dv * inv)
