"""GPU: BSS-eval -- libams_bss.so through utils/bss_eval.py vs the float64 numpy oracle (oracle/bss_eval.py,
restating utils/bss_eval.py:586-748), plus defining properties of the metric."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import bss_eval as obss


def _mix(rng, nsrc, L):
    s = rng.randn(nsrc, L)
    for i in range(nsrc):                                     # colour the sources so the Gram matrices are not near-identity
        s[i] = np.convolve(s[i], rng.randn(8 + 3 * i), mode='same')
    a = rng.randn(nsrc, nsrc) * 0.3 + np.eye(nsrc)
    est = a.dot(s) + 0.05 * rng.randn(nsrc, L)
    return s, est


@pytest.mark.parametrize('nsrc,L', [(2, 3000), (3, 2500), (2, 20480)])
def test_pair_matrices_match_oracle(nsrc, L):
    from utils import bss_eval as hb
    rng = np.random.RandomState(nsrc * 7 + L)
    s, est = _mix(rng, nsrc, L)
    est = est[::-1].copy()                                    # estimates come out permuted
    sdr, sir, sar = hb.bss_eval_pairs(s, est)
    r = obss.bss_eval_sources(s, est, return_matrices=True)
    for got, ref in zip((sdr, sir, sar), r[4]):
        assert np.abs(got - ref).max() < 1e-6, np.abs(got - ref).max()          # dB, float64; Cholesky vs LU, FFT orders
    out = hb.bss_eval_sources_cupy(s, est, nsrc=nsrc)
    assert np.array_equal(out[3], r[3])
    for k in range(3):
        assert np.abs(out[k] - r[k]).max() < 1e-6
    out_np = hb.bss_eval_sources_cupy(s, est, compute_permutation=False, nsrc=nsrc)
    r_np = obss.bss_eval_sources(s, est, compute_permutation=False)
    for k in range(3):
        assert np.abs(out_np[k] - r_np[k]).max() < 1e-6


def test_filtered_reference_is_not_distortion():
    """An estimate that is a short FIR filtering of its own reference (an allowed distortion, flen = 512) has (numerically)
    infinite SDR/SIR/SAR; adding a scaled copy of the OTHER reference lowers SIR to the predicted level and leaves SAR high."""
    from utils import bss_eval as hb
    rng = np.random.RandomState(5)
    L, tail = 6000, 40
    s = rng.randn(2, L)
    s[:, -tail:] = 0.0                                        # room for the filter tail inside the signal length
    h0, h1 = rng.randn(tail) * 0.2, rng.randn(tail) * 0.2
    h0[0] = h1[0] = 1.0
    f0 = np.convolve(s[0], h0)[:L]
    f1 = np.convolve(s[1], h1)[:L]
    sdr, sir, sar, perm = hb.bss_eval_sources_cupy(s, np.stack([f0, f1]), nsrc=2)
    assert np.array_equal(perm, [0, 1]) and sdr.min() > 80 and sir.min() > 80 and sar.min() > 80
    g = 0.1
    sdr2, sir2, sar2, perm2 = hb.bss_eval_sources_cupy(s, np.stack([f0 + g * s[1], f1]), nsrc=2)
    expect = 10 * np.log10(np.sum(f0 ** 2) / np.sum((g * s[1]) ** 2))
    assert abs(sir2[0] - expect) < 0.5 and sar2[0] > 80 and abs(sdr2[0] - sir2[0]) < 1e-3


def test_eval_loop_accumulates_improvement():
    """experiments.evaluation.eval.evaluate on device tensors: separated = references (+ small noise) must improve on the mixture."""
    from experiments.evaluation.eval import evaluate
    rng = np.random.RandomState(9)
    B, S, L = 3, 2, 4096
    nm = rng.randn(B, S, L)
    for b in range(B):
        for k in range(S):
            nm[b, k] = np.convolve(nm[b, k], rng.randn(12), mode='same')
    mix = nm.sum(1)
    sep = nm + 0.05 * rng.randn(B, S, L)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')     # noqa: E731
    means, arr = evaluate([(t(mix), t(nm), t(sep))], nsrc=S, verbose=False)
    assert arr.shape == (B, 2, S) and means[0] > 10 and means[1] > 10
    # oracle on the first utterance
    m32, n32, s32 = (np.asarray(x, np.float32).astype(np.float64) for x in (mix[0], nm[0], sep[0]))
    ref0 = obss.bss_eval_sources(n32, np.stack([m32, m32]))
    ref1 = obss.bss_eval_sources(n32, s32)
    assert np.abs(arr[0, 0] - ref0[0]).max() < 1e-5 and np.abs(arr[0, 1] - ref1[0]).max() < 1e-5
