"""Oracle: STFT front / iSTFT back (reference models/network.py:480-502, 584-607).

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np


def hann_periodic(W, dtype=np.float64):
    """tf.contrib.signal.hann_window(periodic=True): 0.5 - 0.5 cos(2 pi n / W)  (SURVEY App. A-5)."""
    n = np.arange(W)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / W)).astype(dtype)


def stft(x, W, hop):
    """tf.contrib.signal.stft(frame_length=W, frame_step=hop, fft_length=W), pad_end=False
    (network.py:482-492).  x [R, L] -> complex [R, T, W//2+1], T = 1 + (L - W)//hop."""
    R, L = x.shape
    T = 1 + (L - W) // hop
    s0, s1 = x.strides
    fr = np.lib.stride_tricks.as_strided(x, shape=(R, T, W), strides=(s0, hop * s1, s1), writeable=False)
    win = hann_periodic(W, x.dtype)
    out = np.fft.rfft(fr * win, n=W, axis=-1)
    return out.astype(np.complex128 if x.dtype == np.float64 else np.complex64)


def stft_preprocessing(x_mix, x_non_mix, W, hop):
    """Separator.preprocessing (network.py:480-502): returns X [B,T,F], X_non_mix [B,T,F,S], angle [B,T,F]."""
    B, S, L = x_non_mix.shape
    st = stft(x_mix, W, hop)
    st_nm = stft(x_non_mix.reshape(B * S, L), W, hop)
    X = np.abs(st)
    T, F = X.shape[1:]
    X_nm = np.abs(st_nm).reshape(B, S, T, F).transpose(0, 2, 3, 1)
    return X, np.ascontiguousarray(X_nm), np.angle(st)


def inverse_window(W, hop, dtype=np.float64):
    """tf.contrib.signal.inverse_stft_window_fn(hop) applied to the periodic Hann:
    w_inv[n] = w[n] / sum_k w[(n mod hop) + k*hop]^2."""
    w = hann_periodic(W, np.float64)
    denom = np.zeros(W)
    for n in range(W):
        denom[n] = np.sum(w[(n % hop)::hop] ** 2)
    return (w / denom).astype(dtype)


def istft(mag, angle, W, hop):
    """Separator.postprocessing (network.py:584-607): re-attach phase, irfft, inverse window,
    overlap-and-add.  mag/angle [R, T, F] -> [R, (T-1)*hop + W]."""
    R, T, F = mag.shape
    spec = mag * np.exp(1j * angle)
    fr = np.fft.irfft(spec, n=W, axis=-1)[..., :W] * inverse_window(W, hop, np.float64)
    out = np.zeros((R, (T - 1) * hop + W), dtype=np.float64)
    for t in range(T):
        out[:, t * hop:t * hop + W] += fr[:, t]
    return out.astype(mag.dtype)


def istft_bwd(angle, W, hop, dout):
    """d(istft)/d(mag): linear in mag.  dmag[r,t,f] = Re( conj(e^{j angle}) * c_f * rfft-like(frame(dout)*w_inv) ).

    irfft(n=W) of a half spectrum Z: x[n] = (1/W) * ( Re Z_0 + (-1)^n Re Z_{W/2}
    + 2 sum_{f=1}^{W/2-1} Re(Z_f e^{j 2 pi f n / W}) ); imaginary parts of the DC / Nyquist bins are ignored.
    """
    R, T, F = angle.shape
    s0, s1 = dout.strides
    fr = np.lib.stride_tricks.as_strided(dout, shape=(R, T, W), strides=(s0, hop * s1, s1), writeable=False)
    g = fr * inverse_window(W, hop, np.float64)            # d/d(frame samples)
    G = np.fft.rfft(g, n=W, axis=-1)                       # sum_n g[n] e^{-j 2 pi f n / W}
    # d x[n] / d mag_f = (c_f / W) * Re( e^{j angle_f} e^{j 2 pi f n / W} ), c_f = 1 for f in {0, W/2} else 2
    c = np.full(F, 2.0)
    c[0] = 1.0
    if W % 2 == 0:
        c[-1] = 1.0
    ph = np.exp(1j * angle)
    if True:
        # DC/Nyquist: only the real part of Z contributes -> Re(ph) * Re-part of conj(G)
        dmag = (c / W) * np.real(ph * np.conj(G))
        dmag[..., 0] = (1.0 / W) * np.real(ph[..., 0]) * np.real(G[..., 0])
        if W % 2 == 0:
            dmag[..., -1] = (1.0 / W) * np.real(ph[..., -1]) * np.real(G[..., -1])
    return dmag.astype(dout.dtype)
