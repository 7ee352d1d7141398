"""Node builders for the STFT front / iSTFT back of a standalone Separator
(reference models/network.py:480-502 preprocessing, 584-607 postprocessing)."""
import numpy as np
import torch

from . import functional as F
from . import ops as K
from .graph import Node

_DFT_CACHE = {}


def dft_matrices(W, hop, device):
    """fp32 matrices built in float64: analysis D [W, 2F (+ zero columns up to a multiple of 4)] (periodic Hann folded in, [cos | -sin]) and synthesis
    Dinv [2F, W] (irfft weights c_f/W and tf.contrib.signal.inverse_stft_window_fn(hop) folded in)."""
    key = (W, hop, str(device))
    if key not in _DFT_CACHE:
        F_ = W // 2 + 1
        n = np.arange(W, dtype=np.float64)
        f = np.arange(F_, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / W)
        ang = 2.0 * np.pi * np.outer(n, f) / W                         # [W, F]
        D = np.concatenate([win[:, None] * np.cos(ang), -win[:, None] * np.sin(ang)], axis=1)
        denom = np.array([np.sum(win[(k % hop)::hop] ** 2) for k in range(W)])
        w_inv = win / denom
        c = np.full(F_, 2.0)
        c[0] = 1.0
        if W % 2 == 0:
            c[-1] = 1.0
        Dre = (c[:, None] / W) * np.cos(ang.T)                           # [F, W]
        Dim = -(c[:, None] / W) * np.sin(ang.T)
        Dim[0] = 0.0                                                     # irfft ignores Im of DC / Nyquist
        if W % 2 == 0:
            Dim[-1] = 0.0
        Dinv = np.concatenate([Dre, Dim], axis=0) * w_inv[None, :]
        # 2F = W + 2 columns: padded with zero columns to a multiple of 4 so that the analysis product's B rows and output rows are
        # 16-byte addressable (it otherwise runs on the scalar-load native-f32 kernel, 93 instead of ~35 us per STFT at the cfg4 shape);
        # consumers read the first 2F columns of each output row (ams_cplx_mag_phase: ld_ri)
        D = np.pad(D, ((0, 0), (0, (-D.shape[1]) % 4)))
        _DFT_CACHE[key] = (torch.from_numpy(D.astype(np.float32)).to(device).contiguous(),
                           torch.from_numpy(Dinv.astype(np.float32)).to(device).contiguous())
    return _DFT_CACHE[key]


def build_preprocessing(sep):
    """Separator.preprocessing: X = |stft(x_mix)|, X_non_mix = |stft(x_non_mix)|, y = one_hot(argmax_s)."""
    W, hop, S = sep.window_size, sep.hop_size, sep.S
    x_mix, x_non_mix = sep.x_mix, sep.x_non_mix

    def _stft_mix(run):
        xm = x_mix.value(run)
        return F.stft_mag_phase(xm, W, hop)                              # (mag [B,T,F], phasor [B*T, 2F])
    mixn = Node('stft_mix', _stft_mix)
    sep.stfts = mixn
    sep.X = Node('X', lambda run: mixn.value(run)[0])
    xs = sep.X
    sep.X_input = Node('X_input', lambda run: xs.value(run))
    sep.phasor = Node('phasor', lambda run: mixn.value(run)[1])

    def _nm(run):
        xn = x_non_mix.value(run)
        B, S_, L = xn.shape
        return F.stft_mag_phase(xn.reshape(B * S_, L), W, hop, want_phase=False)[0]       # rows (b,s): [B*S, T, F]
    sep.X_non_mix_rows = Node('X_non_mix_rows', _nm)
    rows = sep.X_non_mix_rows
    sep.X_non_mix = Node('X_non_mix', lambda run: rows.value(run).reshape(-1, S, rows.value(run).shape[1], sep.F).permute(0, 2, 3, 1))

    def _y(run):
        r = rows.value(run)
        B = r.shape[0] // S
        y = K.make_masks(r, B, S, sep.a, sep.b, False)
        return sep._weight_masks(y, run).reshape(B, r.shape[1], sep.F, S)
    sep.y = Node('y', _y)
    return sep.X


def build_postprocessing(sep):
    """Separator.postprocessing: re-attach the mixture phase, inverse STFT -> [B, S, L]."""
    W, hop, S = sep.window_size, sep.hop_size, sep.S
    holder = sep

    def _post(run):
        separated = holder.separated.value(run)                          # [B*S, T, F, 1]
        BS, T, Fq = separated.shape[:3]
        out = F.istft(separated.reshape(BS, T, Fq), holder.phasor.value(run), W, hop, S)
        return out.reshape(BS // S, S, -1)
    out = Node('output', _post)
    sep.output = out
    return out
