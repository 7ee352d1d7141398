"""GPU: the hard k-means accumulation pass that serves FIVE tries of an utterance from one read of its points (csrc/kmeans.hip,
kmeans_hard_tries_kernel; reference models/Kmeans_2.py:145-188 with nb_tries restarts, :47-54).

It applies at E = 40, C = 2, tries a multiple of 5, no silence weights -- the inference / enhance geometry.  What must hold:
  * labels, centroids and the best restart are IDENTICAL to the float32 oracle (oracle/kmeans.py: one summation order), at ragged
    lengths, lengths below one 64-point slab, several 8192-point chunks;
  * it is bit-identical to the one-workgroup-per-try kernel (AMS_KM_TRIES=0, a process of its own) at the full benchmark size, where the
    oracle would take minutes."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import kmeans as okm

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _data(seed, b, L, E=40, C=2, tries=5, spread=0.9):
    rng = np.random.RandomState(seed)
    centers = rng.randn(C, E).astype(np.float32) * 1.5
    lab_true = rng.randint(0, C, (b, L))
    X = (centers[lab_true] + rng.randn(b, L, E).astype(np.float32) * spread).astype(np.float32)
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    return X, idx


@pytest.mark.parametrize('b,L,tries,iters,end', [(2, 3000, 5, 4, True), (1, 9000, 10, 3, True), (3, 197, 5, 3, False), (2, 64, 5, 2, True),
                                                (1, 20480, 5, 3, False), (9, 8192 + 257, 5, 2, True), (2, 8192 + 130, 10, 2, False)])
def test_five_tries_per_read_is_bit_exact(b, L, tries, iters, end):
    """end = False returns the labels of the chosen restart as the final pass wrote them (kmeans_hard_tries_final_kernel's label output);
    True re-assigns at the end (labels-only pass).  Either way the restart is chosen by the inertia the final pass adds up."""
    from ams_hip import functional as F
    X, idx = _data(L + tries, b, L, tries=tries, spread=2.5)
    cent_ref, lab_ref, best_ref = okm.kmeans(X, idx, 2, tries, iters, beta=None, notsilent=None, assign_at_end=end)
    cent, lab, best = F.kmeans(torch.from_numpy(X).cuda(), torch.from_numpy(idx).cuda(), 2, tries, iters, None, None, end)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert np.array_equal(cent.cpu().numpy(), cent_ref)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)


@pytest.mark.parametrize('b,L,tries,iters,kind,end', [(2, 3000, 5, 3, 'mask', True), (3, 8192 + 300, 10, 2, 'mask', False), (2, 2500, 5, 3, 'real', True),
                                                     (1, 20480, 5, 2, 'mixed', True), (4, 700, 5, 2, 'mask', False)])
def test_five_tries_per_read_with_silence_weights(b, L, tries, iters, kind, end):
    """Silence weights (Kmeans_2.py:76-82, 175-181): the 0/1 mask the reference builds, arbitrary positive weights, a mix -- and the
    reference's weight-tile quirk (row r of the tiled problem takes weight row r % b: with b > 1 every try of an utterance may see another
    utterance's weights), which is why the sums role takes its weights per try."""
    from ams_hip import functional as F
    X, idx = _data(L * 3 + tries, b, L, tries=tries, spread=2.0)
    rng = np.random.RandomState(b + L)
    w = (rng.rand(b, L) > 0.25).astype(np.float32)
    if kind == 'real':
        w = rng.uniform(0.05, 1.7, (b, L)).astype(np.float32)
    elif kind == 'mixed':
        w[:, ::3] = rng.uniform(0.05, 1.7, (b, L))[:, ::3].astype(np.float32)
    cent_ref, lab_ref, best_ref = okm.kmeans(X, idx, 2, tries, iters, beta=None, notsilent=w, assign_at_end=end)
    cent, lab, best = F.kmeans(torch.from_numpy(X).cuda(), torch.from_numpy(idx).cuda(), 2, tries, iters, None, torch.from_numpy(w).cuda(), end)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert np.array_equal(cent.cpu().numpy(), cent_ref, equal_nan=True)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)


_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, sys.argv[4])
from ams_hip import ops
d = np.load(sys.argv[1])
xn = ops.kmeans_normalize(torch.from_numpy(d['X']).cuda())
w = torch.from_numpy(d['w']).cuda() if 'w' in d.files else None
beta = float(d['beta']) if 'beta' in d.files else None
cent, lab, best, _ = ops.kmeans_run(xn, torch.from_numpy(d['idx']).cuda(), 2, int(d['tries']), int(d['iters']), beta=beta, w=w)
torch.cuda.synchronize()
np.savez(sys.argv[2], cent=cent.cpu().numpy(), lab=lab.cpu().numpy(), best=best.cpu().numpy())
'''


def _in_child(env_var, X, idx, tries, iters, w=None, beta=None):
    """The same k-means in a process of its own with `env_var`=0 (the switches are read once per process)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory(prefix='ams_km_') as tmp:
        extra = {}
        if w is not None:
            extra['w'] = w
        if beta is not None:
            extra['beta'] = beta
        np.savez(os.path.join(tmp, 'in.npz'), X=X, idx=idx, tries=tries, iters=iters, **extra)
        env = dict(os.environ, **{env_var: '0'})
        r = subprocess.run([sys.executable, '-c', _CHILD, os.path.join(tmp, 'in.npz'), os.path.join(tmp, 'out.npz'),
                            os.path.join(root, 'adaptive-multispeaker-separation_amd'), root], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        ref = np.load(os.path.join(tmp, 'out.npz'))
        return {k: ref[k] for k in ref.files}


@pytest.mark.parametrize('weights', [False, True])
def test_same_bits_as_one_workgroup_per_try_at_benchmark_size(weights):
    """b = 64 utterances x 10 restarts x 10 iterations at TF = 20480 (cfg3 inference), without and with silence weights: every centroid
    bit, label and chosen restart equal to what kmeans_pass_kernel gives (AMS_KM_TRIES=0 is read once per process, hence the child)."""
    from ams_hip import ops
    if os.environ.get('AMS_KM_TRIES') == '0':
        pytest.skip('this process runs the per-try kernel itself')
    b, L, tries, iters = 64, 20480, 10, 10
    X, idx = _data(77, b, L, tries=tries, spread=1.3)
    w = (np.random.RandomState(5).rand(b, L) > 0.3).astype(np.float32) if weights else None
    ref = _in_child('AMS_KM_TRIES', X, idx, tries, iters, w=w)
    xn = ops.kmeans_normalize(torch.from_numpy(X).cuda())
    cent, lab, best, _ = ops.kmeans_run(xn, torch.from_numpy(idx).cuda(), 2, tries, iters, w=None if w is None else torch.from_numpy(w).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), ref['best'])
    assert np.array_equal(cent.cpu().numpy(), ref['cent'])
    assert np.array_equal(lab.cpu().numpy(), ref['lab'])


def test_soft_accumulation_kernel_agrees_with_the_pass_kernel_at_benchmark_size():
    """kmeans_soft_acc_kernel (scalar centroid operands, |x|^2 - 2 <x, c> + |c|^2) against the soft branch of kmeans_pass_kernel
    (AMS_KM_SOFT=0, a process of its own) at the fine-tuning geometry: 64 utterances, TF = 20480, beta = 10, silence weights, 10 iterations:
    centroids and soft labels to 2e-5 (the soft modes are tolerance-checked everywhere: tests/test_gpu_kmeans_soft.py vs float64)."""
    from ams_hip import ops
    if os.environ.get('AMS_KM_SOFT') == '0':
        pytest.skip('this process runs the pass kernel itself')
    b, L, tries, iters, beta = 64, 20480, 1, 10, 10.0
    X, idx = _data(78, b, L, tries=tries, spread=1.3)
    w = (np.random.RandomState(6).rand(b, L) > 0.3).astype(np.float32)
    ref = _in_child('AMS_KM_SOFT', X, idx, tries, iters, w=w, beta=beta)
    xn = ops.kmeans_normalize(torch.from_numpy(X).cuda())
    cent, lab, best, _ = ops.kmeans_run(xn, torch.from_numpy(idx).cuda(), 2, tries, iters, beta=beta, w=torch.from_numpy(w).cuda())
    torch.cuda.synchronize()
    assert np.abs(cent.cpu().numpy() - ref['cent']).max() < 2e-5 * max(1.0, np.abs(ref['cent']).max())
    assert np.abs(lab.cpu().numpy() - ref['lab']).max() < 2e-4


@pytest.mark.parametrize('E', [40, 32, 20, 8])
def test_fused_normalisations_carry_the_bits_of_the_two_passes(E):
    """ams_l2norm_kmeans_normalize (the Normalize layer, models/dpcl.py:32, and the k-means' own normalisation, Kmeans_2.py:40-41, in one
    pass over the dense output -- what the inference / enhance paths run) against ams_l2norm_fwd followed by ams_kmeans_normalize, and
    against the float32 oracle's normalisation of the once-normalised rows: zero rows, tiny rows, a ragged last slab."""
    from ams_hip import ops
    rng = np.random.RandomState(E)
    b, L = 3, 1000 + E
    u = (rng.randn(b, L, E) * rng.uniform(1e-3, 30.0, (b, L, 1))).astype(np.float32)
    u[0, 5] = 0.0
    u[1, 7] = 1e-20
    ud = torch.from_numpy(u).cuda()
    v, _ = ops.l2norm_fwd(ud.view(b, L * E), E)
    ref = ops.kmeans_normalize(v.view(b, L, E))
    xn = ops.l2norm_kmeans_normalize(ud.view(b, L * E), E).view(b, L, E)
    torch.cuda.synchronize()
    assert torch.equal(xn, ref)
    assert np.array_equal(xn.cpu().numpy(), okm.l2_normalize_rows(v.view(b, L, E).cpu().numpy()))
