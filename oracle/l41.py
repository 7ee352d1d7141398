"""Oracle: L41 (Lab41 source-contrastive) loss (reference models/L41.py:47-186, sampling=None path).

Test infrastructure only -- see oracle/__init__.py.

  emb  [B,T,F,E]   (l2-normalised iff `normalize`; L41.py:38-39)
  y    [B,T,F,S]   +1 / -1 masks (L41.py:9-10 with network.py:378/502)
  spk  [tot_speakers, E] 'speaker_centroids' (L41.py:16-18), I [B,S] indices
  cost = mean_{t,f} mean_b mean_s -log(sigmoid(y * <Vspk[b,s], emb[b,t,f]>))   (L41.py:150-178)
"""
import numpy as np
from .dense import L2_EPS


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def spk_normalize(spk):
    ss = np.sum(spk * spk, axis=1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ss, L2_EPS))
    return spk * inv, inv


def l41_cost(emb, y, spk, I, normalize=True):
    sv, _ = spk_normalize(spk) if normalize else (spk, None)
    Vs = sv[I]                                            # [B,S,E]  (tf.gather_nd, L41.py:66-68)
    dot = np.einsum('btfe,bse->btfs', emb, Vs)
    cost = -np.log(_sig(y * dot))
    return cost.mean(axis=3).mean(axis=0).mean()


def l41_cost_bwd(emb, y, spk, I, normalize=True):
    """SURVEY Appendix D-6.  Returns d/d emb and d/d speaker_centroids."""
    B, T, F, E = emb.shape
    S = y.shape[3]
    sv, inv = spk_normalize(spk) if normalize else (spk, None)
    Vs = sv[I]
    dot = np.einsum('btfe,bse->btfs', emb, Vs)
    z = y * dot
    dz = -_sig(-z) * y / (S * B * T * F)                  # d cost / d dot
    demb = np.einsum('btfs,bse->btfe', dz, Vs)
    dVs = np.einsum('btfs,btfe->bse', dz, emb)
    dsv = np.zeros_like(spk)
    np.add.at(dsv, I.reshape(-1), dVs.reshape(-1, E))     # gather_nd backward = scatter-add
    if normalize:
        dotn = np.sum(sv * dsv, axis=1, keepdims=True)
        dspk = (dsv - sv * dotn) * inv
    else:
        dspk = dsv
    return demb, dspk
