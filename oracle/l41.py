"""Oracle: L41 (Lab41 source-contrastive) loss (reference models/L41.py:47-186, sampling=None path).

Test infrastructure only -- see oracle/__init__.py.

  emb  [B,T,F,E]   (l2-normalised iff `normalize`; L41.py:38-39)
  y    [B,T,F,S]   +1 / -1 masks (L41.py:9-10 with network.py:378/502)
  spk  [tot_speakers, E] 'speaker_centroids' (L41.py:16-18), I [B,S] indices
  cost = mean_{t,f} mean_b mean_s -log(sigmoid(y * <Vspk[b,s], emb[b,t,f]>))   (L41.py:150-178)

Negative sampling (`--sampling K`, L41.py:69-147,165-166): K further speaker vectors per bin enter as negatives,
  cost[b,t,f] += ns_rate * mean_k -log(sigmoid(-<neg[b,t,f,k], emb[b,t,f]>))
  'k-nearest' (L41.py:91-116): for each (b, s) the K rows of the (normalised) table with the largest dot product with Vspk[b,s]
              (tf.nn.top_k -- the speaker itself is among them); a bin uses the set of its DOMINANT speaker argmax_s y;
  'random'    (L41.py:117-139): K speakers not in I[b], one set per utterance (tf.random_shuffle: the draw itself is not
              reproducible outside TensorFlow -- the indices are an input here).
  neg_idx [B, NSEL, K]: NSEL = S (set chosen by the dominant speaker) or 1 (one set per utterance).
"""
import numpy as np
from .dense import L2_EPS


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def spk_normalize(spk):
    ss = np.sum(spk * spk, axis=1, keepdims=True)
    inv = 1.0 / np.sqrt(np.maximum(ss, L2_EPS))
    return spk * inv, inv


def knearest_indices(spk, I, K, normalize=True):
    """L41.py:95-104: top_k of <Vspk[b,s], table[n]> over n; ties -> lower index first (tf.nn.top_k)."""
    sv, _ = spk_normalize(spk) if normalize else (spk, None)
    prod = np.einsum('bse,ne->bsn', sv[I], sv)
    return np.argsort(-prod, axis=2, kind='stable')[:, :, :K]


def random_indices(I, tot_speakers, K, rng):
    """L41.py:123-134: K of the tot_speakers - S speakers that are not in the mixture, per utterance -> [B, 1, K]."""
    out = []
    for row in I:
        avail = np.array([n for n in range(tot_speakers) if n not in set(int(v) for v in row)])
        out.append(rng.permutation(avail)[:K])
    return np.stack(out)[:, None, :]


def _neg_vectors(sv, y, neg_idx):
    """[B,T,F,K,E] negatives of every bin (L41.py:108-116 / :139) and the set index each bin used."""
    B = y.shape[0]
    negs = sv[neg_idx]                                    # [B,NSEL,K,E]
    if neg_idx.shape[1] == 1:
        sel = np.zeros(y.shape[:3], dtype=np.int64)
    else:
        sel = np.argmax(y, axis=-1)                       # dominant speaker, first maximum (tf.argmax, L41.py:75)
    return negs[np.arange(B)[:, None, None], sel], sel


def l41_cost(emb, y, spk, I, normalize=True, neg_idx=None, ns_rate=0.1):
    sv, _ = spk_normalize(spk) if normalize else (spk, None)
    Vs = sv[I]                                            # [B,S,E]  (tf.gather_nd, L41.py:66-68)
    dot = np.einsum('btfe,bse->btfs', emb, Vs)
    cost = -np.log(_sig(y * dot)).mean(axis=3)
    if neg_idx is not None:
        vec, _ = _neg_vectors(sv, y, neg_idx)
        doto = np.einsum('btfke,btfe->btfk', vec, emb)
        cost = cost + ns_rate * (-np.log(_sig(-doto))).mean(axis=3)      # L41.py:143-147,165-166
    return cost.mean(axis=0).mean()


def l41_cost_bwd(emb, y, spk, I, normalize=True, neg_idx=None, ns_rate=0.1):
    """SURVEY Appendix D-6.  Returns d/d emb and d/d speaker_centroids (the top_k / shuffle indices carry no gradient)."""
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
    if neg_idx is not None:
        K = neg_idx.shape[2]
        vec, sel = _neg_vectors(sv, y, neg_idx)
        doto = np.einsum('btfke,btfe->btfk', vec, emb)
        dd = _sig(doto) * ns_rate / (K * B * T * F)       # d/d doto of -log sigmoid(-doto) = sigmoid(doto)
        demb = demb + np.einsum('btfk,btfke->btfe', dd, vec)
        dvec = dd[..., None] * emb[:, :, :, None, :]      # [B,T,F,K,E]
        dneg = np.zeros((B,) + neg_idx.shape[1:] + (E,))
        for b in range(B):
            for j in range(neg_idx.shape[1]):
                m = sel[b] == j
                dneg[b, j] = dvec[b][m].sum(axis=0)
        np.add.at(dsv, neg_idx.reshape(-1), dneg.reshape(-1, E))
    if normalize:
        dotn = np.sum(sv * dsv, axis=1, keepdims=True)
        dspk = (dsv - sv * dotn) * inv
    else:
        dspk = dsv
    return demb, dspk
