"""Oracle: batched in-graph k-means (reference models/Kmeans_2.py:14-188).

Test infrastructure only -- see oracle/__init__.py.

Restated behaviour (quirks kept):
  * input [b, L, E] is l2-normalised (Kmeans_2.py:40-41), tiled nb_tries times so that row
    r = b_idx*nb_tries + try (Kmeans_2.py:47-54);
  * initial centroids = rows of X picked by host RNG indices (Kmeans_2.py:61-71) -- the indices are an
    explicit input here (`init_idx` [R, C]);
  * silence weights `notsilent` [b, L] zero the centroid numerator but silent bins are still counted
    in the hard-assignment denominator, and all get label 0 (distance 0 to every centroid);
  * QUIRK (not in SURVEY): the weights are tiled with tf.tile(w, [nb_tries,1,1]) (Kmeans_2.py:80), i.e.
    weight row r = w[r mod b], whereas data row r = x[r // nb_tries].  They agree only when
    nb_tries == 1 or b == 1.  `faithful_tile=True` reproduces this; False uses the matching row.
  * hard: labels = argmin_c sqrt(sum_e (x-c)^2 * w) (first index on ties); centroids = segment_sum(x*w) /
    segment_count; soft (beta): labels = softmax_c(-beta * sum_e (x-c)^2 * w), centroids =
    sum_l x*w*lab / sum_l lab;
  * best try = argmin inertia (Kmeans_2.py:97-104,114-143); `assign_at_end` recomputes labels from the
    un-tiled input with all-ones weights (Kmeans_2.py:106-107,171-173).

Summation order.  TensorFlow's unsorted_segment_sum is atomics-based and order-nondeterministic on
GPU, so any fixed order is a valid restatement.  For bit-exact label parity the oracle and the HIP
kernels share ONE order (`ordered_sum`): points are cut into chunks of 8192; inside a chunk lane
j (0..255) adds its 32 points j, j+256, ... sequentially; each group of 64 consecutive lanes (one
wavefront) combines its lane partials by a halving tree (v[j] += v[j+s], s = 32..1); the group totals
are added sequentially in (chunk, group) order.  (Rounds 1-4: chunks of 2048 and one 256-lane tree;
round 5 moved the cross-wavefront part of the tree into the sequential tail so that a 64-lane column
of a chunk is a unit of work that needs no other column -- csrc/kmeans.hip, kmeans_hard_tries_kernel.)
HARD distances: d = sum over e, left to right, of round(round((x_e - c_e)^2) * w) -- `sqdist`.  tf.square, the multiply by notsilent
and reduce_sum are three ops of the TF-1.4 graph (Kmeans_2.py:187, no XLA): every square is rounded to float32 before it is weighted
and added.  Only the ORDER of reduce_sum over e is the library's (Eigen packets on the CPU, a tree on the GPU) and is restated here as
left to right.  Round 5 restated the chain as d <- fma((x_e - c_e) * w, x_e - c_e, d) (`sqdist_fused`, kept below with its correctly
rounded `fma32` only for the record): that is NOT an evaluation of the reference's graph under any reduction order, and although the
two forms give identical labels, centroids and best tries on the golden fixture and on the benchmark-shape inputs, they disagree on
about one point-pass in 10^7 on structureless data (tests/test_kmeans_distance_forms.py measures both) -- so round 6 went back to
rounded squares in the oracle and in csrc/kmeans.hip.
"""
import numpy as np
from .dense import L2_EPS

CHUNK = 8192
LANES = 256
WAVE = 64


def ordered_sum(a):
    """Sum over axis 0 of a [L, ...] in the order shared with the HIP kernel (see module doc)."""
    L = a.shape[0]
    G = -(-L // CHUNK)
    pad = G * CHUNK - L
    if pad:
        a = np.concatenate([a, np.zeros((pad,) + a.shape[1:], dtype=a.dtype)], axis=0)
    a = a.reshape((G, CHUNK // LANES, LANES) + a.shape[1:])
    tot = None
    for g in range(G):
        v = a[g, 0].copy()
        for j in range(1, CHUNK // LANES):
            v = v + a[g, j]
        for k in range(LANES // WAVE):
            u = v[k * WAVE:(k + 1) * WAVE]
            s = WAVE // 2
            while s >= 1:
                u = u[:s] + u[s:2 * s]
                s //= 2
            tot = u[0] if tot is None else tot + u[0]
    return tot


def l2_normalize_rows(x):
    """tf.nn.l2_normalize(x, axis=-1) with left-to-right accumulation over E."""
    ss = np.zeros(x.shape[:-1], dtype=x.dtype)
    for e in range(x.shape[-1]):
        ss = ss + x[..., e] * x[..., e]
    inv = (np.asarray(1.0, x.dtype) / np.sqrt(np.maximum(ss, np.asarray(L2_EPS, x.dtype))))
    return x * inv[..., None]


def fma32(a, b, c):
    """Correctly rounded fused multiply-add for float32 arrays (what v_fma_f32 / v_pk_fma_f32 compute); plain a*b+c for float64."""
    if np.result_type(a, b, c) != np.float32:
        return a * b + c
    p = a.astype(np.float64) * b.astype(np.float64)                # exact: 24 + 24 significant bits
    c64 = np.broadcast_to(c.astype(np.float64), p.shape)
    s = p + c64                                                     # round to nearest in float64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                               # TwoSum: s + err == p + c exactly
    odd = (s.view(np.int64) & 1) == 1
    nudge = (err != 0) & ~odd & np.isfinite(s)                      # round to odd: an inexact even sum moves to its odd neighbour
    s = np.where(nudge, np.nextafter(s, np.where(err > 0, np.inf, -np.inf)), s)
    return s.astype(np.float32)


def sqdist_fused(x, cent, w):
    """d2[l,c] = fused chain over e of ((x[l,e]-cent[c,e]) * w[l]) * (x[l,e]-cent[c,e]) (module doc).  x [L,E], cent [C,E], w [L]."""
    L, E = x.shape
    C = cent.shape[0]
    d = np.zeros((L, C), dtype=x.dtype)
    for e in range(E):
        diff = x[:, e:e + 1] - cent[None, :, e]
        d = fma32(diff * w[:, None], diff, d)
    return d


def sqdist(x, cent, w):
    """d2[l,c] = sum_e ((x[l,e]-cent[c,e])^2 * w[l]), sequential over e.  x [L,E], cent [C,E], w [L]."""
    L, E = x.shape
    C = cent.shape[0]
    d = np.zeros((L, C), dtype=x.dtype)
    for e in range(E):
        diff = x[:, e:e + 1] - cent[None, :, e]
        d = d + (diff * diff) * w[:, None]
    return d


HARD_DIST = sqdist                # the hard distance in force (module doc); tests/test_kmeans_distance_forms.py holds it there


def labels_hard(x, cent, w, dist=None):
    return np.argmin(np.sqrt((dist or HARD_DIST)(x, cent, w)), axis=1).astype(np.int32)


def labels_soft(x, cent, w, beta):
    d = sqdist(x, cent, w)
    e = np.exp(-np.asarray(beta, x.dtype) * d)
    return e / e.sum(axis=1, keepdims=True)


def update_hard(x, w, lab, C):
    """Kmeans_2.py:158-165: total = segment_sum(x*w), count = segment_sum(1) (silent bins counted)."""
    xw = x * w[:, None]
    cent = np.zeros((C, x.shape[1]), dtype=x.dtype)
    for c in range(C):
        m = (lab == c).astype(x.dtype)
        tot = ordered_sum(xw * m[:, None])
        cnt = ordered_sum(m)
        with np.errstate(divide='ignore', invalid='ignore'):
            cent[c] = tot / cnt
    return cent


def update_soft(x, w, lab):
    """Kmeans_2.py:152-155."""
    xw = x * w[:, None]
    num = np.einsum('le,lc->ce', xw, lab)
    den = lab.sum(axis=0)
    return num / den[:, None]


def inertia_hard(x, cent, w):
    """Kmeans_2.py:116,126-141: labels with weights, distances on the UNMASKED x."""
    lab = labels_hard(x, cent, w)
    C = cent.shape[0]
    diff = x - cent[lab]
    dist = np.zeros(x.shape[0], dtype=x.dtype)
    for e in range(x.shape[1]):
        dist = dist + diff[:, e] * diff[:, e]
    tot = np.zeros(C, dtype=x.dtype)
    for c in range(C):
        m = (lab == c).astype(x.dtype)
        with np.errstate(divide='ignore', invalid='ignore'):
            tot[c] = ordered_sum(dist * m) / ordered_sum(m)
    return tot.sum()


def inertia_soft(x, cent, w, beta):
    """Kmeans_2.py:117-124."""
    lab = labels_soft(x, cent, w, beta)
    d = sqdist(x, cent, np.ones_like(w))
    return ((d * lab).sum(axis=0) / lab.sum(axis=0)).sum()


def kmeans(X_in, init_idx, nb_clusters, nb_tries, nb_iterations, beta=None, notsilent=None,
           assign_at_end=True, normalize_input=True, faithful_tile=True):
    """Full KMeans.network (Kmeans_2.py:86-111).

    X_in [b, L, E]; init_idx [b*nb_tries, C] int; notsilent [b, L] or None.
    Returns (centroids [b, C, E], labels [b, L] int32 (hard) or [b, L, C] (soft), best_try [b]).
    """
    b, L, E = X_in.shape
    C = nb_clusters
    x0 = l2_normalize_rows(X_in) if normalize_input else X_in
    ones = np.ones((L,), dtype=X_in.dtype)
    R = b * nb_tries
    cents = np.zeros((R, C, E), dtype=X_in.dtype)
    labs = []
    inert = np.zeros(R, dtype=X_in.dtype)
    for r in range(R):
        x = x0[r // nb_tries]
        if notsilent is None:
            w = ones
        else:
            w = notsilent[r % b] if faithful_tile else notsilent[r // nb_tries]
        cent = x[init_idx[r]]
        lab = labels_hard(x, cent, w) if beta is None else labels_soft(x, cent, w, beta)
        for _ in range(nb_iterations):
            cent = update_hard(x, w, lab, C) if beta is None else update_soft(x, w, lab)
            lab = labels_hard(x, cent, w) if beta is None else labels_soft(x, cent, w, beta)
        cents[r] = cent
        labs.append(lab)
        inert[r] = inertia_hard(x, cent, w) if beta is None else inertia_soft(x, cent, w, beta)
    best = np.argmin(inert.reshape(b, nb_tries), axis=1).astype(np.int32)
    index = best + np.arange(b) * nb_tries
    cent_sel = cents[index]
    if assign_at_end:
        out = [labels_hard(x0[i], cent_sel[i], ones) if beta is None else labels_soft(x0[i], cent_sel[i], ones, beta)
               for i in range(b)]
    else:
        out = [labs[i] for i in index]
    return cent_sel, np.stack(out, axis=0), best


def masks_from_labels(labels, S, beta):
    """Separator.separate (network.py:567-572): one_hot(labels, S, 1, 0) for hard, identity for soft."""
    if beta is None:
        return (labels[..., None] == np.arange(S)).astype(np.float32)
    return labels
