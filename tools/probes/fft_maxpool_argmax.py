"""Would an FFT (overlap-save) form of path B keep the arg-max of tf.nn.max_pool_with_argmax?  (reference models/adapt.py:115-117.)

Path B = stride-1 SAME conv of the waveform with the W x N filter, then max over windows of P samples with its flat arg-max; the unpool of
the back end scatters by that index, so the index is part of the result (tests/golden/pretraining_maxpool_step.npz holds it `array_equal`).
This probe computes y = conv(x, f) three ways on the CPU at the cfg2 geometry (L = 20480, W = 1024, N = 256, P = hop = 256):
  exact     float64 direct form (the oracle's arithmetic);
  direct32  float32 matrix product of the frames with the filter (what the MFMA kernel's f32-level arithmetic amounts to);
  fft32     float32 real FFT of 2048-sample blocks, spectral product, inverse FFT (scipy.fft in single precision) -- the arithmetic an
            overlap-save kernel would have, whose error is ~log2(n) roundings of the LARGEST output of the block, not of each output;
and counts the pooling windows whose arg-max differs from `exact`.

    python tools/probes/fft_maxpool_argmax.py [rows]
"""
import sys

import numpy as np
import scipy.fft as sfft

ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
sys.path[:0] = [ROOT, __import__('os').path.join(ROOT, 'adaptive-multispeaker-separation_amd')]


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    L, W, N, P = 20480, 1024, 256, 256
    rng = np.random.RandomState(1)
    # SURVEY 8(d) synthetic speech-like sources (harmonic stacks under a syllable gate) + the front's initialisation
    t = np.arange(L) / 8000.0
    x = np.zeros((rows, L))
    for r in range(rows):
        f0 = rng.uniform(90, 250)
        s = sum((1.0 / k) * rng.uniform(0.5, 1) * np.sin(2 * np.pi * k * f0 * t + rng.uniform(0, 2 * np.pi)) for k in range(1, 21))
        gate = 0.5 * (1 + np.cos(2 * np.pi * rng.uniform(4, 6) * t + rng.uniform(0, 2 * np.pi))) > 0.35
        s = s * gate + 0.005 * rng.randn(L)
        x[r] = s * 0.05 / np.sqrt((s ** 2).mean())
    w = rng.uniform(-np.sqrt(3.0 / W), np.sqrt(3.0 / W), W)
    bases = rng.uniform(-np.sqrt(6.0 / (W + N)), np.sqrt(6.0 / (W + N)), (W, N))
    f = np.abs(w)[:, None] * bases
    x32, f32 = x.astype(np.float32), f.astype(np.float32)
    pl = (W - 1) // 2                                            # SAME, stride 1: 511 left, 512 right
    T = (L - P) // P + 1
    tot = dif_d = dif_f = 0
    margins = []
    for r in range(rows):
        xp = np.concatenate([np.zeros(pl), x32[r].astype(np.float64), np.zeros(W - 1 - pl)])
        xp32 = xp.astype(np.float32)
        # exact and direct32, window by window (frames of one window: [P, W])
        idx = np.arange(P)[:, None] + np.arange(W)[None, :]
        # fft32 of the whole row in 2048-blocks (overlap-save): valid outputs per block = 2048 - W + 1 = 1025
        nfft, step = 2048, 2048 - W + 1
        F = sfft.rfft(np.concatenate([f32[::-1], np.zeros((nfft - W, N), np.float32)]), axis=0)     # correlation = convolution with the reversed filter
        yf = np.zeros((L, N), np.float32)
        for s0 in range(0, L, step):
            blk = xp32[s0:s0 + nfft]
            if blk.shape[0] < nfft:
                blk = np.concatenate([blk, np.zeros(nfft - blk.shape[0], np.float32)])
            Y = sfft.irfft(sfft.rfft(blk)[:, None] * F, n=nfft, axis=0)
            n_ok = min(step, L - s0)
            yf[s0:s0 + n_ok] = Y[W - 1:W - 1 + n_ok]
        for tt in range(T):
            fr = xp[tt * P + idx]                                # [P, W] float64
            ye = fr @ f32.astype(np.float64)
            yd = xp32[tt * P + idx] @ f32
            ae, ad, af = ye.argmax(0), yd.argmax(0), yf[tt * P:(tt + 1) * P].argmax(0)
            tot += N
            dif_d += int((ae != ad).sum())
            dif_f += int((ae != af).sum())
            srt = np.sort(ye, axis=0)
            margins.append((srt[-1] - srt[-2]) / np.abs(ye).max())
        print('row %d: windows x filters %d   direct32 differs %d   fft32 differs %d' % (r, tot, dif_d, dif_f), flush=True)
    m = np.concatenate(margins)
    print('relative margin between the two largest values of a window (per filter, / max |y| of the window): quantiles 1e-4 %.2e  1e-3 %.2e  1e-2 %.2e'
          % tuple(np.quantile(m, [1e-4, 1e-3, 1e-2])))
    print('arg-max differs from the float64 direct form:  direct32 %d / %d (%.2e)   fft32 %d / %d (%.2e)' % (dif_d, tot, dif_d / tot, dif_f, tot, dif_f / tot))


if __name__ == '__main__':
    main()
