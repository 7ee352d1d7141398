"""CPU: which arithmetic the hard k-means distance is restated in, and what the alternative would have cost.

The reference computes `tf.sqrt(tf.reduce_sum(tf.square(X_ - centroids_) * notsilent, axis=3))` (models/Kmeans_2.py:187): in a TF-1.4
graph without XLA the square is an op of its own, rounded to float32 before it is weighted and summed.  `oracle.kmeans.sqdist` follows
that; `sqdist_fused` (one FMA chain per cluster, round 5's restatement, chosen because it halves the device's vector instructions) does
not.  This file (1) pins the oracle to the rounded-square form, (2) shows both forms agree on every k-means fixture the repo holds and on
the inputs of tests/test_gpu_benchshape.py::test_kmeans_hard_at_benchmark_shape (the reason round 5 did not notice), and (3) measures the
disagreement on data without cluster structure, where every pass has points next to the bisector -- non-zero, which is why the fused form
was dropped from oracle and kernels in round 6.
"""
import os

import numpy as np
import pytest

from oracle import kmeans as okm

HERE = os.path.dirname(os.path.abspath(__file__))


def both_forms(X, idx, C, tries, iters, w):
    out = {}
    saved = okm.HARD_DIST
    try:
        for name, f in (('rounded', okm.sqdist), ('fused', okm.sqdist_fused)):
            okm.HARD_DIST = f
            out[name] = okm.kmeans(X, idx, C, tries, iters, beta=None, notsilent=w, assign_at_end=True)
    finally:
        okm.HARD_DIST = saved
    return out['rounded'], out['fused']


def test_the_oracle_rounds_every_square():
    assert okm.HARD_DIST is okm.sqdist
    # the two forms are different functions: on random points most distances differ in the last bit, and `sqdist` is, operation by
    # operation, float32 multiply -> float32 multiply by w -> float32 add
    rng = np.random.RandomState(0)
    x = rng.randn(512, 40).astype(np.float32)
    c = rng.randn(2, 40).astype(np.float32)
    w = np.ones(512, dtype=np.float32)
    r, f = okm.sqdist(x, c, w), okm.sqdist_fused(x, c, w)
    assert (r != f).mean() > 0.2 and np.allclose(r, f, rtol=1e-5)
    d = np.zeros(512, dtype=np.float32)
    for e in range(40):
        diff = np.float32(x[:, e] - c[1, e])
        d = np.float32(d + np.float32(np.float32(diff * diff) * w))
    assert np.array_equal(d, r[:, 1])


def test_fma32_is_the_correctly_rounded_fma():
    """`sqdist_fused` is only a fair comparison if its FMA is the device's: checked against exact rational arithmetic."""
    from fractions import Fraction
    rng = np.random.RandomState(3)
    a = rng.randn(2000).astype(np.float32)
    b = rng.randn(2000).astype(np.float32)
    c = (-(a.astype(np.float64) * b.astype(np.float64)) + rng.randn(2000) * 1e-7).astype(np.float32)      # heavy cancellation
    got = okm.fma32(a, b, c)
    for i in range(0, 2000, 7):
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        lo, hi = np.nextafter(got[i], np.float32(-np.inf)), np.nextafter(got[i], np.float32(np.inf))
        assert abs(Fraction(float(got[i])) - exact) <= min(abs(Fraction(float(lo)) - exact), abs(Fraction(float(hi)) - exact))


def test_both_forms_agree_on_the_golden_fixture():
    k = np.load(os.path.join(HERE, 'golden', 'kmeans_hard.npz'))
    C, tries, iters = [int(v) for v in k['cfg']]
    r, f = both_forms(k['X'], k['idx'], C, tries, iters, k['w'])
    for a, b, g in zip(r, f, (k['centroids'], k['labels'], k['best'])):
        assert np.array_equal(a, b) and np.array_equal(a, g)


def test_both_forms_agree_on_the_benchmark_shape_inputs():
    """The inputs of test_kmeans_hard_at_benchmark_shape (L = 20480, E = 40, 10 restarts x 10 iterations, silence weights), first
    utterance: labels, centroids and the chosen restart are identical under both forms (measured once for both utterances: 0 of 40960
    labels, centroids bit-equal) -- well-separated blobs have no points within an ulp of the bisector."""
    E, T, N = 40, 80, 256
    b, C, tries, iters = 2, 2, 10, 10
    rng = np.random.RandomState(31)
    centers = rng.randn(C, E).astype(np.float32) * 1.5
    lab_true = rng.randint(0, C, (b, T * N))
    X = (centers[lab_true] + rng.randn(b, T * N, E).astype(np.float32) * 0.9).astype(np.float32)
    w = (rng.rand(b, T * N) > 0.2).astype(np.float32)
    idx = np.stack([rng.choice(T * N, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    # one utterance, four of its restarts, keeps this under half a minute; b = 1 also makes weight row r % b the matching one
    r, f = both_forms(X[:1], idx[:4], C, 4, iters, w[:1])
    for a, c in zip(r, f):
        assert np.array_equal(a, c)


def test_the_forms_disagree_where_points_sit_on_the_bisector(capsys):
    """Structureless data (normalised Gaussian noise): per label pass, along the rounded-square trajectory, count the points whose
    label differs between the two forms.  Recorded at L = 20480, 4 utterances x 10 restarts x 11 passes: 1 of 9.0 M point-passes
    (1.1e-7); here a smaller sweep -- the assertion is only that the count is what a rate of that order gives (0..few);
    test_the_oracle_rounds_every_square holds a constructed value where the two forms differ."""
    E, L, C = 40, 4096, 2
    rng = np.random.RandomState(5)
    X = okm.l2_normalize_rows(rng.randn(3, L, E).astype(np.float32))
    ones = np.ones(L, dtype=np.float32)
    differ = total = 0
    for u in range(3):
        cent = X[u][rng.choice(L, C, replace=False)]
        for _ in range(6):
            lr = okm.labels_hard(X[u], cent, ones, dist=okm.sqdist)
            lf = okm.labels_hard(X[u], cent, ones, dist=okm.sqdist_fused)
            differ += int((lr != lf).sum())
            total += L
            cent = okm.update_hard(X[u], ones, lr, C)
    with capsys.disabled():
        print('\n[kmeans distance forms] structureless data: %d of %d point-passes labelled differently by the fused chain' % (differ, total))
    assert differ <= 3
