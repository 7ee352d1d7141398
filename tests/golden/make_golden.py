#!/usr/bin/env python
"""Generates the golden vectors in this directory from the float64 numpy oracle (seeded inputs, injected weights and
k-means init indices).  PARITY UNPINNED: the reference (Python-2 / TensorFlow 1.4) cannot run in the build container and
ships no fixtures of its own, so these vectors pin the ORACLE (and through it the HIP kernels), not TensorFlow.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import step, recipes, kmeans, front, stft, losses  # noqa: E402


def synth(rng, B, S, L):
    t = np.arange(L) / 8000.0
    xn = np.zeros((B, S, L))
    for b in range(B):
        for s in range(S):
            f0 = rng.uniform(90, 250)
            xn[b, s] = sum(np.sin(2 * np.pi * k * f0 * t + rng.uniform(0, 6.28)) / k for k in range(1, 6)) * 0.02 + 0.005 * rng.randn(L)
    return xn.sum(1), xn


def main():
    rng = np.random.RandomState(2024)
    # ---- front_DPCL training step (cfg3(i)) at B=2, L=2048, W=128, hop=32, N=16, 2xBLSTM(16), E=8
    B, S, L, W, N, hop, LS, NL, E = 2, 2, 2048, 128, 16, 32, 16, 2, 8
    P = step.init_params(rng, np.float64, front_W=W, N=N, D_in=N, layer_size=LS, nb_layers=NL, E=E, F=N, conv1d_scale=0.4)
    xm, xn = synth(rng, B, S, L)
    cost, grads, V, Y = step.front_dpcl_loss(xm, xn, P, hop, NL, E)
    out = {'x_mix': xm, 'x_non_mix': xn, 'cost': cost, 'V': V, 'Y': Y, 'cfg': np.array([B, S, L, W, N, hop, LS, NL, E])}
    out.update({'P/' + k: v for k, v in P.items()})
    out.update({'G/' + k: v for k, v in grads.items()})
    np.savez_compressed(os.path.join(HERE, 'front_dpcl_step.npz'), **out)

    # ---- pretraining step (cfg2, path A)
    c, g, back = recipes.pretrain_loss(xm, xn, P, hop, 'sdr+l2', 'mask', 1.0)
    o = {'cost': c, 'back': back}
    o.update({'G/' + k: v for k, v in g.items()})
    np.savez_compressed(os.path.join(HERE, 'pretraining_step.npz'), **o)

    # ---- hard k-means (bit-exact labels, float32) incl. silence weights and the reference's weight-tile order
    b, Lk, Ek, C, tries = 2, 3000, 40, 2, 3
    cen = rng.randn(C, Ek).astype(np.float32) * 2
    X = (cen[rng.randint(0, C, (b, Lk))] + rng.randn(b, Lk, Ek).astype(np.float32) * 0.7).astype(np.float32)
    w = (rng.rand(b, Lk) > 0.25).astype(np.float32)
    idx = np.stack([rng.choice(Lk, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    cent, lab, best = kmeans.kmeans(X, idx, C, tries, 5, beta=None, notsilent=w, assign_at_end=True)
    assert np.isfinite(cent).all()
    np.savez_compressed(os.path.join(HERE, 'kmeans_hard.npz'), X=X, w=w, idx=idx, centroids=cent, labels=lab, best=best,
                        cfg=np.array([C, tries, 5]))

    # ---- STFT / iSTFT
    x = rng.randn(3, 2048)
    s = stft.stft(x, 256, 128)
    rec = stft.istft(np.abs(s), np.angle(s), 256, 128)
    np.savez_compressed(os.path.join(HERE, 'stft.npz'), x=x, mag=np.abs(s), cos=np.cos(np.angle(s)), sin=np.sin(np.angle(s)), rec=rec)


if __name__ == '__main__':
    main()
    print('golden vectors written to', HERE)
