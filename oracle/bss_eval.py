"""Oracle: BSS-eval SDR / SIR / SAR.

Test infrastructure only -- see oracle/__init__.py.  PARITY PINNED (the only oracle in this repo that is): the reference file
(utils/bss_eval.py, a copy of mir_eval.separation with TensorFlow and cupy variants added) cannot be imported whole (Python-2
prints, tensorflow/cupy imports), but its numpy implementation (:74-371) runs under Python 3; tests/golden/make_bss_golden.py
executes exactly those reference lines in the build container and tests/test_bss_golden.py holds this restatement (and the
HIP library) to the resulting vectors at 1e-6 dB -- nsrc 2 and 3, L 3000 / 20480, "mixture as estimate", a near-silent estimate,
the raw projection, and validate()'s rejection of silent sources.  The cupy variant below differs from that numpy original only
in _safe_db (den + 1e-12); its dB values are pinned from the same reference run through the stored energies.

Follows the reference's GPU path, the one experiments/evaluation/eval.py:48-73 calls:
  bss_eval_sources_cupy   utils/bss_eval.py:586-637   (pair matrix + permutation by mean SIR)
  _bss_decomp_mtifilt_cupy              :641-663      (s_true / e_spat / e_interf / e_artif)
  _project_cupy                         :674-730      (least-squares projection on delayed references, flen = 512)
  _bss_source_crit_cupy, _safe_db_cupy  :732-748      (10 log10(num / (den + 1e-12)))
`mir_eval_db=True` switches to the numpy original's _safe_db (:361-368: +inf when the denominator is exactly 0).
"""
import itertools

import numpy as np

FLEN = 512          # utils/bss_eval.py:608 / :237


def _nfft(nsampl, flen):
    return int(2 ** np.ceil(np.log2(nsampl + flen - 1.0)))        # :689


def project(refs, est, flen=FLEN):
    """Least-squares projection of `est` [nsampl] on the span of the delayed references refs [nsrc, nsampl] (:674-730).
    Returns sproj [nsampl + flen - 1]."""
    nsrc, nsampl = refs.shape
    n = _nfft(nsampl, flen)
    refs_p = np.concatenate([refs, np.zeros((nsrc, flen - 1))], axis=1)
    est_p = np.concatenate([est, np.zeros(flen - 1)])
    sf = np.fft.fft(refs_p, n=n, axis=1)
    sef = np.fft.fft(est_p, n=n)
    G = np.empty((nsrc * flen, nsrc * flen))
    k = np.arange(flen)
    lag = (k[None, :] - k[:, None]) % n                            # toeplitz(c=[s0, s[-1], ...], r=s[:flen])[a,b] = s[(b-a) mod n]
    for i in range(nsrc):                                          # same write order as :696-702 (later writes win)
        for j in range(nsrc):
            ss = np.real(np.fft.ifft(sf[i] * np.conj(sf[j])))
            blk = ss[lag]
            G[i * flen:(i + 1) * flen, j * flen:(j + 1) * flen] = blk
            G[j * flen:(j + 1) * flen, i * flen:(i + 1) * flen] = blk.T
    D = np.empty(nsrc * flen)
    for i in range(nsrc):
        s = np.real(np.fft.ifft(sf[i] * np.conj(sef)))
        D[i * flen:(i + 1) * flen] = s[(-k) % n]                   # [s0, s[-1], ..., s[-flen+1]]  (:710-711)
    C = np.linalg.solve(G, D).reshape(nsrc, flen)                  # block i = filter applied to reference i (:716-720)
    out_len = nsampl + flen - 1
    fshape = flen + out_len - 1
    sproj = np.zeros(out_len)
    for i in range(nsrc):
        conv = np.fft.irfft(np.fft.rfft(C[i], fshape) * np.fft.rfft(refs_p[i], fshape), fshape)
        sproj += conv[:out_len]
    return sproj


def decompose(refs, est, j, flen=FLEN):
    """:641-663 -> (s_true, e_spat, e_interf, e_artif), each [nsampl + flen - 1]."""
    pad = np.zeros(flen - 1)
    s_true = np.concatenate([refs[j], pad])
    e_spat = project(refs[j:j + 1], est, flen) - s_true
    e_interf = project(refs, est, flen) - s_true - e_spat
    e_artif = -s_true - e_spat - e_interf + np.concatenate([est, pad])
    return s_true, e_spat, e_interf, e_artif


def _db(num, den, mir_eval_db):
    if mir_eval_db:
        return np.inf if den == 0 else 10.0 * np.log10(num / den)
    return 10.0 * np.log10(num / (den + 1e-12))


def criteria(s_true, e_spat, e_interf, e_artif, mir_eval_db=False):
    s_filt = s_true + e_spat
    sdr = _db(np.sum(s_filt ** 2), np.sum((e_interf + e_artif) ** 2), mir_eval_db)
    sir = _db(np.sum(s_filt ** 2), np.sum(e_interf ** 2), mir_eval_db)
    sar = _db(np.sum((s_filt + e_interf) ** 2), np.sum(e_artif ** 2), mir_eval_db)
    return sdr, sir, sar


def bss_eval_sources(reference_sources, estimated_sources, compute_permutation=True, flen=FLEN, mir_eval_db=False,
                     return_matrices=False):
    """reference_sources, estimated_sources [nsrc, nsampl] -> (sdr, sir, sar [nsrc], perm [nsrc]); estimated source perm[j]
    corresponds to true source j (:586-637)."""
    refs = np.asarray(reference_sources, np.float64)
    ests = np.asarray(estimated_sources, np.float64)
    nsrc = refs.shape[0]
    if not compute_permutation:
        out = [criteria(*decompose(refs, ests[j], j, flen), mir_eval_db=mir_eval_db) for j in range(nsrc)]
        sdr, sir, sar = (np.array(v) for v in zip(*out))
        return sdr, sir, sar, np.arange(nsrc)
    sdr, sir, sar = (np.empty((nsrc, nsrc)) for _ in range(3))
    for je in range(nsrc):
        for jt in range(nsrc):
            sdr[je, jt], sir[je, jt], sar[je, jt] = criteria(*decompose(refs, ests[je], jt, flen), mir_eval_db=mir_eval_db)
    perms = list(itertools.permutations(range(nsrc)))
    dum = np.arange(nsrc)
    mean_sir = np.array([np.mean(sir[list(p), dum]) for p in perms])
    popt = np.array(perms[int(np.argmax(mean_sir))])
    out = (sdr[popt, dum], sir[popt, dum], sar[popt, dum], popt)
    return out + ((sdr, sir, sar),) if return_matrices else out
