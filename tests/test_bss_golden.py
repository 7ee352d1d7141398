"""BSS-eval pinned against the REFERENCE's own numpy implementation (utils/bss_eval.py:74-371, executed in the build container
by tests/golden/make_bss_golden.py -> tests/golden/bss_eval.npz).  CPU: oracle/bss_eval.py vs the vectors.  GPU: libams_bss.so
(through utils/bss_eval.py, the reference's `bss_eval_sources_cupy` entry point) vs the same vectors.

dB tolerance: 1e-6 dB where the criterion is below 100 dB; criteria above 100 dB are ratios against an energy at the float64
rounding floor (e.g. SAR of "mixture as estimate": the estimate lies exactly in the span of the references) and are only
required to stay above 100 dB on both sides.
"""
import os

import numpy as np
import pytest

from oracle import bss_eval as obss

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bss_eval.npz'))
CASES = sorted({k.split('/')[0] for k in G.files if k.startswith('n')})
TOL_DB = 1e-6


def _close_db(got, ref, tol=TOL_DB):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    small = ref < 100.0
    assert np.abs(got[small] - ref[small]).max(initial=0.0) < tol, (got, ref)
    assert (got[~small] > 100.0).all(), (got, ref)


def _cupy_db(name):
    """The reference's GPU-path dB convention (_safe_db_cupy, utils/bss_eval.py:742-748) from the stored reference energies."""
    e = G[name + '/energies']
    ea = G[name + '/e_artif_energy']
    db = lambda num, den: 10.0 * np.log10(num / (den + 1e-12))      # noqa: E731
    return np.stack([db(e[..., 0], e[..., 1]), db(e[..., 0], e[..., 2]), db(e[..., 3], ea)])


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_numpy(name):
    s, est = G[name + '/ref'].astype(np.float64), G[name + '/est'].astype(np.float64)
    sdr, sir, sar, perm, mats = obss.bss_eval_sources(s, est, mir_eval_db=True, return_matrices=True)
    assert np.array_equal(perm, G[name + '/perm'])
    for got, key in ((sdr, 'sdr'), (sir, 'sir'), (sar, 'sar')):
        _close_db(got, G[name + '/' + key])
    for k in range(3):
        _close_db(mats[k], G[name + '/pair_matrices'][k])
    out = obss.bss_eval_sources(s, est, compute_permutation=False, mir_eval_db=True)
    for got, key in zip(out[:3], ('sdr_noperm', 'sir_noperm', 'sar_noperm')):
        _close_db(got, G[name + '/' + key])
    # GPU-path convention of the oracle (den + 1e-12) against the same reference energies
    mats_c = obss.bss_eval_sources(s, est, return_matrices=True)[4]
    ref_c = _cupy_db(name)
    for k in range(3):
        _close_db(mats_c[k], ref_c[k])


def test_oracle_projection_matches_reference_numpy():
    s, e = G['project/ref'].astype(np.float64), G['project/est'].astype(np.float64)
    for got, key in ((obss.project(s, e), 'project/sproj_all'), (obss.project(s[1:2], e), 'project/sproj_single')):
        ref = G[key]
        assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()


def test_reference_rejects_silent_sources():
    assert G['silent/raises_valueerror'].tolist() == [1, 1]        # validate(), utils/bss_eval.py:101-117


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_hip_matches_reference_numpy(name):
    from utils import bss_eval as hb
    s, est = G[name + '/ref'], G[name + '/est']
    nsrc = s.shape[0]
    ref_c = _cupy_db(name)
    mats = hb.bss_eval_pairs(s, est)
    for k in range(3):
        _close_db(mats[k], ref_c[k])
    sdr, sir, sar, perm = hb.bss_eval_sources_cupy(s, est, nsrc=nsrc)
    assert np.array_equal(perm, G[name + '/perm'])
    dum = np.arange(nsrc)
    for got, k in ((sdr, 0), (sir, 1), (sar, 2)):
        _close_db(got, ref_c[k][perm, dum])
