#!/usr/bin/env python
"""Pins the BSS-eval oracle against the REFERENCE ITSELF (build container only).

/root/reference/utils/bss_eval.py cannot be imported as a module (it imports tensorflow and cupy at :81,83 and has Python-2
`print` statements from :382 on), but its numpy implementation -- validate / bss_eval_sources / _bss_decomp_mtifilt /
_project / _bss_source_crit / _safe_db, utils/bss_eval.py:74-371, the mir_eval original the cupy path was transcribed from --
is plain numpy/scipy and compiles under Python 3.  This script executes exactly that span of the reference file *where it
lies* (nothing is copied into the repo), minus the two unavailable imports, on seeded inputs and stores inputs + outputs as
`bss_eval.npz`.  tests/test_bss_golden.py then checks oracle/bss_eval.py (CPU) and libams_bss.so (GPU) against the vectors.

    python tests/golden/make_bss_golden.py      # needs /root/reference; rewrites tests/golden/bss_eval.npz

The reference's GPU entry point (bss_eval_sources_cupy, :586-748) differs from the numpy original in ONE arithmetic place:
_safe_db_cupy divides by (den + 1e-12) instead of returning +inf at den == 0 (:742-748 vs :361-368).  (For nsrc != 2 its
filter reshape (:718-720) is also C-order where the original is F-order; eval.py only ever calls it with nsrc = 2.)  The
fixture stores the raw energy sums as well, so both dB conventions are pinned from the same reference run.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/utils/bss_eval.py'


def load_reference_numpy_path():
    with open(REF) as f:
        lines = f.read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('import numpy as np'))
    stop = next(i for i, l in enumerate(lines) if l.startswith('def bss_eval_sources_tf'))
    body = [l for l in lines[start:stop] if l.strip() not in ('import tensorflow as tf', 'import cupy as cp')]
    if not hasattr(np, 'Inf'):            # numpy >= 2 dropped the alias the 2014 code uses at :367
        np.Inf = np.inf
    mod = types.ModuleType('ref_bss_eval_numpy')
    exec(compile('\n' * start + '\n'.join(body), REF, 'exec'), mod.__dict__)
    return mod


def coloured(rng, nsrc, L):
    s = rng.randn(nsrc, L)
    for i in range(nsrc):
        s[i] = np.convolve(s[i], rng.randn(8 + 3 * i), mode='same')
    return s


def cases():
    """name -> (references, estimates).  Inputs are float32-representable so the GPU host mirror (which takes fp32 waveforms
    from the separator) sees the very same numbers."""
    out = {}
    for name, nsrc, L, seed in (('n2_L3000', 2, 3000, 11), ('n3_L3000', 3, 3000, 12), ('n2_L20480', 2, 20480, 13),
                                ('n3_L20480', 3, 20480, 14)):
        rng = np.random.RandomState(seed)
        s = coloured(rng, nsrc, L)
        a = rng.randn(nsrc, nsrc) * 0.3 + np.eye(nsrc)
        est = (a.dot(s) + 0.05 * rng.randn(nsrc, L))[::-1]
        out[name] = (s, est)
    # the evaluation loop's "no separation" call: both estimates are the mixture (eval.py:50-52)
    rng = np.random.RandomState(15)
    s = coloured(rng, 2, 20480)
    out['n2_mixture_as_estimate'] = (s, np.stack([s.sum(0), s.sum(0)]))
    # one estimate nearly silent (1e-6 of the signal level) -- stays valid for validate(), stresses the dB tails
    rng = np.random.RandomState(16)
    s = coloured(rng, 2, 3000)
    est = np.stack([s[0] + 0.1 * s[1], 1e-6 * rng.randn(3000)])
    out['n2_near_silent_estimate'] = (s, est)
    return {k: tuple(np.asarray(v, np.float32).astype(np.float64) for v in pair) for k, pair in out.items()}


def main():
    ref = load_reference_numpy_path()
    store = {}
    for name, (s, est) in cases().items():
        nsrc = s.shape[0]
        sdr, sir, sar, perm = ref.bss_eval_sources(s, est)                       # :154-250
        sdr_n, sir_n, sar_n, _ = ref.bss_eval_sources(s, est, compute_permutation=False)
        mats = np.empty((3, nsrc, nsrc))
        energies = np.empty((nsrc, nsrc, 4))     # |s_filt|^2, |e_interf + e_artif|^2, |e_interf|^2, (|s_filt+e_interf|^2, |e_artif|^2 below)
        energies2 = np.empty((nsrc, nsrc, 2))
        for je in range(nsrc):
            for jt in range(nsrc):
                st, es, ei, ea = ref._bss_decomp_mtifilt(s, est[je], jt, 512)    # :252-274
                mats[:, je, jt] = ref._bss_source_crit(st, es, ei, ea)           # :349-358
                sf = st + es
                energies[je, jt] = [np.sum(sf ** 2), np.sum((ei + ea) ** 2), np.sum(ei ** 2), np.sum((sf + ei) ** 2)]
                energies2[je, jt] = [np.sum(ea ** 2), 0.0]
        store.update({name + '/ref': s.astype(np.float32), name + '/est': est.astype(np.float32), name + '/sdr': sdr, name + '/sir': sir, name + '/sar': sar,
                      name + '/perm': perm, name + '/sdr_noperm': sdr_n, name + '/sir_noperm': sir_n,
                      name + '/sar_noperm': sar_n, name + '/pair_matrices': mats, name + '/energies': energies,
                      name + '/e_artif_energy': energies2[..., 0]})
        print(name, 'sdr', sdr, 'sir', sir, 'sar', sar, 'perm', perm)
    # projection of a single estimate (the least-squares core, :276-347) at a small size, full vector stored
    rng = np.random.RandomState(21)
    s = np.asarray(coloured(rng, 2, 1200), np.float32).astype(np.float64)
    e = np.asarray(0.7 * s[0] - 0.2 * s[1] + 0.05 * rng.randn(1200), np.float32).astype(np.float64)
    store.update({'project/ref': s.astype(np.float32), 'project/est': e.astype(np.float32), 'project/sproj_all': ref._project(s, e, 512),
                  'project/sproj_single': ref._project(s[1:2], e, 512)})
    # validate(): an all-zero estimate or reference is rejected (:101-117)
    raised = []
    for which in ('reference', 'estimate'):
        z = s.copy()
        z[1] = 0.0
        try:
            ref.bss_eval_sources(z, s) if which == 'reference' else ref.bss_eval_sources(s, z)
            raised.append(0)
        except ValueError:
            raised.append(1)
    store['silent/raises_valueerror'] = np.array(raised)
    np.savez_compressed(os.path.join(HERE, 'bss_eval.npz'), **store)
    print('wrote', os.path.join(HERE, 'bss_eval.npz'))


if __name__ == '__main__':
    if not os.path.exists(REF):
        sys.exit('needs the reference tree at /root/reference (build container only)')
    main()
