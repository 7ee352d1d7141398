"""Oracle: deep-clustering affinity loss (reference models/dpcl.py:41-87).

Test infrastructure only -- see oracle/__init__.py.

  Y [B, TF, S] targets, V [B, TF, E] unit-norm embeddings
  D_i = 1/sqrt( (Y (Y^T 1))_i )                                   (dpcl.py:58-62)
  cost = mean_b( ||V^T D V||_F - 2 ||V^T D Y||_F + ||Y^T D Y||_F )  (dpcl.py:67-80; UN-squared norms)
"""
import numpy as np


def dpcl_terms(V, Y):
    cnt = Y.sum(axis=1, keepdims=True)                   # Y^T 1  -> [B,1,S]
    diag = np.sum(Y * cnt, axis=2)                       # [B,TF]
    with np.errstate(divide='ignore'):
        D = 1.0 / np.sqrt(diag)
    DV = V * D[:, :, None]
    DY = Y * D[:, :, None]
    G = np.einsum('bie,bif->bef', V, DV)                 # V^T D V   [B,E,E]
    A = np.einsum('bie,bis->bes', V, DY)                 # V^T D Y   [B,E,S]
    C = np.einsum('bis,bir->bsr', Y, DY)                 # Y^T D Y   [B,S,S]
    return D, G, A, C


def dpcl_cost(V, Y):
    D, G, A, C = dpcl_terms(V, Y)
    nG = np.sqrt(np.sum(G * G, axis=(1, 2)))
    nA = np.sqrt(np.sum(A * A, axis=(1, 2)))
    nC = np.sqrt(np.sum(C * C, axis=(1, 2)))
    per_utt = nG - 2.0 * nA + nC
    return per_utt.mean(), (nG.mean(), (-2.0 * nA).mean(), nC.mean())


def dpcl_cost_bwd(V, Y):
    """dcost/dV: (1/B) * ( 2 D V G/||G|| - 2 D Y A^T/||A|| )."""
    B = V.shape[0]
    D, G, A, C = dpcl_terms(V, Y)
    nG = np.sqrt(np.sum(G * G, axis=(1, 2)))[:, None, None]
    nA = np.sqrt(np.sum(A * A, axis=(1, 2)))[:, None, None]
    t1 = 2.0 * np.einsum('bie,bef->bif', V, G / nG)
    t2 = 2.0 * np.einsum('bis,bes->bie', Y, A / nA)
    return (t1 - t2) * D[:, :, None] / B
