"""Oracle: whole training steps assembled from the per-op restatements.

Test infrastructure only -- see oracle/__init__.py.  Parameter dictionaries use the reference's
TF variable names: 'front/window/w', 'front/bases/bases', 'back/window/value',
'back/bases/value', 'prediction/forward_BLSTM_<i>/rnn/basic_lstm_cell/{kernel,bias}',
'prediction/backward_BLSTM_<i>/...', 'prediction/W' [Din, E*F], 'prediction/b', 'speaker_centroids'.
"""
import numpy as np
from . import front, stft, blstm, dense, dpcl, l41, separate


def lstm_names(scope, i):
    f = '%s/forward_BLSTM_%d/rnn/basic_lstm_cell/' % (scope, i)
    b = '%s/backward_BLSTM_%d/rnn/basic_lstm_cell/' % (scope, i)
    return f + 'kernel', f + 'bias', b + 'kernel', b + 'bias'


def stack_params(P, scope, nb_layers):
    return [tuple(P[n] for n in lstm_names(scope, i)) for i in range(nb_layers)]


def prediction_fwd(X, P, nb_layers, E, normalize=True):
    """DPCL.prediction / L41Model.prediction (dpcl.py:19-39, L41.py:21-45).  X [B,T,F] -> V [B,T,F,E]."""
    h, caches = blstm.blstm_stack_fwd(X, stack_params(P, 'prediction', nb_layers))
    u = dense.dense_fwd(h, P['prediction/W'], P['prediction/b'])
    if normalize:
        V, inv = dense.l2norm_fwd(u, E)
    else:
        V, inv = u.reshape(u.shape[:-1] + (u.shape[-1] // E, E)), None
    return V, (caches, h, V, inv)


def prediction_bwd(dV, cache, P, nb_layers):
    caches, h, V, inv = cache
    du = dense.l2norm_bwd(V, inv, dV) if inv is not None else dV
    du = du.reshape(du.shape[:2] + (-1,))
    dh, dW, db = dense.dense_bwd(h, P['prediction/W'], du)
    _, lg = blstm.blstm_stack_bwd(dh, caches, need_dx=False)
    grads = {'prediction/W': dW, 'prediction/b': db}
    for i, g in enumerate(lg):
        for n, v in zip(lstm_names('prediction', i), g):
            grads[n] = v
    return grads


def front_rep(x_mix, x_non_mix, P, hop):
    """Adapt.__init__ concat + Adapt.front path A (adapt.py:43-47, 95-126)."""
    B, S, L = x_non_mix.shape
    x = np.concatenate([x_mix, x_non_mix.reshape(B * S, L)], axis=0)
    f = front.front_filter(P['front/window/w'], P['front/bases/bases'])
    return front.conv_strided(x, f, hop)


def front_dpcl_loss(x_mix, x_non_mix, P, hop, nb_layers, E, want_grads=True):
    """cfg3(i) front_DPCL step: frozen front -> plugged DPCL -> affinity loss."""
    B, S, L = x_non_mix.shape
    y = front_rep(x_mix, x_non_mix, P, hop)
    X, X_nm = separate.split_front(y, B, S)
    Y, _ = separate.make_masks(np.abs(X_nm), 1.0, 0.0)
    V, cache = prediction_fwd(X, P, nb_layers, E)
    Bq, T, F, _ = V.shape
    Vf = V.reshape(B, T * F, E)
    Yf = Y.reshape(B, T * F, S)
    cost, terms = dpcl.dpcl_cost(Vf, Yf)
    if not want_grads:
        return cost, V, Y
    dV = dpcl.dpcl_cost_bwd(Vf, Yf).reshape(V.shape)
    return cost, prediction_bwd(dV, cache, P, nb_layers), V, Y


def stft_dpcl_loss(x_mix, x_non_mix, P, W, hop, nb_layers, E, want_grads=True, mask_spectra=None):
    """cfg1 STFT_DPCL step.  `mask_spectra` [B,T,F,S]: take the arg-max of THESE per-speaker magnitudes for the ideal
    masks instead of the float64 ones (tests/test_gpu_fullstep.py: at 1.3 M points a few bins hold two speakers within 1e-9 of the
    largest magnitude, and one label that falls the other way in float32 moves an untrained net's gradient by 1e-3)."""
    B, S, L = x_non_mix.shape
    X, X_nm, _ = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    Y, _ = separate.make_masks(X_nm if mask_spectra is None else mask_spectra, 1.0, 0.0)
    V, cache = prediction_fwd(X, P, nb_layers, E)
    T, F = V.shape[1:3]
    Vf, Yf = V.reshape(B, T * F, E), Y.reshape(B, T * F, S)
    cost, _ = dpcl.dpcl_cost(Vf, Yf)
    if not want_grads:
        return cost, V, Y
    dV = dpcl.dpcl_cost_bwd(Vf, Yf).reshape(V.shape)
    return cost, prediction_bwd(dV, cache, P, nb_layers), V, Y


def front_l41_loss(x_mix, x_non_mix, I, P, hop, nb_layers, E, normalize=True, want_grads=True, sampling=None, ns_rate=0.1):
    """front_L41 step (cfg5 shape): plugged L41Model on the frozen front."""
    B, S, L = x_non_mix.shape
    y = front_rep(x_mix, x_non_mix, P, hop)
    X, X_nm = separate.split_front(y, B, S)
    Y, _ = separate.make_masks(np.abs(X_nm), 1.0, -1.0)
    V, cache = prediction_fwd(X, P, nb_layers, E, normalize)
    # --sampling K --ns_method k-nearest (L41.py:91-116): the neighbour sets follow from the speaker table itself
    neg = l41.knearest_indices(P['speaker_centroids'], I, sampling, normalize) if sampling is not None else None
    cost = l41.l41_cost(V, Y, P['speaker_centroids'], I, normalize, neg, ns_rate)
    if not want_grads:
        return cost, V, Y
    dV, dspk = l41.l41_cost_bwd(V, Y, P['speaker_centroids'], I, normalize, neg, ns_rate)
    grads = prediction_bwd(dV, cache, P, nb_layers)
    grads['speaker_centroids'] = dspk
    return cost, grads, V, Y


def stft_l41_loss(x_mix, x_non_mix, I, P, W, hop, nb_layers, E, normalize=True, want_grads=True, sampling=None, ns_rate=0.1,
                  mask_spectra=None):
    """cfg4 STFT_L41 step (trainer.py:468-486 with L41Model): |STFT| magnitudes (network.py:480-502), masks
    y = one_hot(argmax_s |STFT(x_s)|) with (on, off) = (1, -1) (L41.py:9-10), 3xBLSTM -> Conv1D -> [l2norm], cost L41.py:47-186.
    `mask_spectra`: as in stft_dpcl_loss."""
    B, S, L = x_non_mix.shape
    X, X_nm, _ = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    Y, _ = separate.make_masks(X_nm if mask_spectra is None else mask_spectra, 1.0, -1.0)
    V, cache = prediction_fwd(X, P, nb_layers, E, normalize)
    # --sampling K --ns_method k-nearest (L41.py:91-116): the neighbour sets follow from the speaker table itself
    neg = l41.knearest_indices(P['speaker_centroids'], I, sampling, normalize) if sampling is not None else None
    cost = l41.l41_cost(V, Y, P['speaker_centroids'], I, normalize, neg, ns_rate)
    if not want_grads:
        return cost, V, Y
    dV, dspk = l41.l41_cost_bwd(V, Y, P['speaker_centroids'], I, normalize, neg, ns_rate)
    grads = prediction_bwd(dV, cache, P, nb_layers)
    grads['speaker_centroids'] = dspk
    return cost, grads, V, Y


def init_params(rng, dtype, front_W=None, N=None, D_in=None, layer_size=600, nb_layers=3, E=40, F=None,
                conv1d_scale=None, tot_speakers=None):
    """Random parameters with the reference's shapes/initialisers (ops.py:489-494)."""
    P = {}
    if front_W is not None:
        lw, lb = np.sqrt(3.0 / front_W), np.sqrt(6.0 / (front_W + N))
        P['front/window/w'] = rng.uniform(-lw, lw, front_W).astype(dtype)
        P['front/bases/bases'] = rng.uniform(-lb, lb, (front_W, N)).astype(dtype)
        P['back/window/value'] = rng.uniform(-lw, lw, front_W).astype(dtype)
        P['back/bases/value'] = rng.uniform(-lb, lb, (front_W, N)).astype(dtype)
    if D_in is not None:
        H = layer_size // 2
        d = D_in
        for i in range(nb_layers):
            lim = np.sqrt(6.0 / (d + H + 4 * H))
            kf, bf, kb, bb = lstm_names('prediction', i)
            P[kf] = rng.uniform(-lim, lim, (d + H, 4 * H)).astype(dtype)
            P[bf] = np.zeros(4 * H, dtype)
            P[kb] = rng.uniform(-lim, lim, (d + H, 4 * H)).astype(dtype)
            P[bb] = np.zeros(4 * H, dtype)
            d = layer_size
        if conv1d_scale is None:
            fan = np.sqrt(2.0 / float(layer_size + E * F))
            conv1d_scale = np.sqrt(2.0 / fan)            # the reference's nested-sqrt range (quirk C-8)
        P['prediction/W'] = rng.uniform(-conv1d_scale, conv1d_scale, (layer_size, E * F)).astype(dtype)
        P['prediction/b'] = np.zeros(E * F, dtype)
    if tot_speakers is not None:
        sd = np.sqrt(2.0 / E)
        v = rng.standard_normal((tot_speakers, E)) * sd
        P['speaker_centroids'] = np.clip(v, -2 * sd, 2 * sd).astype(dtype)
    return P


def init_enhance_params(rng, dtype, F, layer_size_enh, nb_layers_enh, scale=0.5):
    """Random 'enhance/*' variables with the shapes network.py:629-636 builds: BLSTM stack on [sep | X] (2F inputs) and a
    width-1 Conv1D back to F."""
    P = {}
    H = layer_size_enh // 2
    d = 2 * F
    for i in range(nb_layers_enh):
        lim = np.sqrt(6.0 / (d + H + 4 * H))
        kf, bf, kb, bb = lstm_names('enhance', i)
        P[kf] = rng.uniform(-lim, lim, (d + H, 4 * H)).astype(dtype)
        P[bf] = np.zeros(4 * H, dtype)
        P[kb] = rng.uniform(-lim, lim, (d + H, 4 * H)).astype(dtype)
        P[bb] = np.zeros(4 * H, dtype)
        d = layer_size_enh
    P['enhance/W'] = rng.uniform(-scale, scale, (layer_size_enh, F)).astype(dtype)
    P['enhance/b'] = np.zeros(F, dtype)
    return P
