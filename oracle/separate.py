"""Oracle: Separator glue -- masks, normalisation, silence weights, mask application
(reference models/network.py:357-400, 409-454, 504-521, 554-582, 610-660).

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np


def log10(x):
    """utils/ops.py:56-59: log(x)/log(10)."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.log(x) / np.log(np.asarray(10.0, dtype=x.dtype))


def split_front(y_front, B, S):
    """network.py:368-372: X [B,T,F] signed mixture rep; X_non_mix [B,T,F,S]."""
    T, F = y_front.shape[1:]
    X = y_front[:B]
    X_nm = y_front[B:].reshape(B, S, T, F).transpose(0, 2, 3, 1)
    return X, np.ascontiguousarray(X_nm)


def make_masks(X_nm_abs, a, b):
    """y = one_hot(argmax_s X_non_mix_abs, S, on=a, off=b); first index on ties
    (network.py:377-378 plugged [caller passes abs], :501-502 STFT)."""
    S = X_nm_abs.shape[-1]
    am = np.argmax(X_nm_abs, axis=-1)
    oh = (am[..., None] == np.arange(S)).astype(X_nm_abs.dtype)
    return oh * a + (1.0 - oh) * b, am


def function_mask(y, X, kind):
    """network.py:381-389: y *= g(|X|/max|X|) with g linear / sqrt / square."""
    r = np.abs(X) / np.abs(X).max(axis=(1, 2), keepdims=True)
    if kind == 'sqrt':
        r = np.sqrt(r)
    elif kind == 'square':
        r = r * r
    return y * r[..., None]


def silence_loss_mask(y, X, thr):
    """network.py:391-396: y *= (log10(max|X| / |X|) < thr)."""
    ax = np.abs(X)
    with np.errstate(divide='ignore'):
        m = (log10(ax.max(axis=(1, 2), keepdims=True) / ax) < thr).astype(y.dtype)
    return y * m[..., None]


def normalization01(X):
    """network.py:504-508."""
    mn = X.min(axis=(1, 2), keepdims=True)
    mx = X.max(axis=(1, 2), keepdims=True)
    return (X - mn) / (mx - mn)


def normalization_mean_std(X):
    """network.py:518-521: tf.nn.moments = population variance, no eps."""
    m = X.mean(axis=(1, 2), keepdims=True)
    v = ((X - m) ** 2).mean(axis=(1, 2), keepdims=True)
    return (X - m) / np.sqrt(v)


def stft_input_pipeline(X, pre_func, normalize, silent_db):
    """Separator.init_separator STFT branch (network.py:427-443)."""
    if pre_func == 'sqrt':
        X = np.sqrt(X)
    elif pre_func == 'log':
        X = log10(X + np.asarray(1e-12, X.dtype))
    if normalize == '01':
        X = normalization01(X)
    elif normalize == 'meanstd':
        X = normalization_mean_std(X)
    if silent_db > 0:
        mx = X.max(axis=(1, 2), keepdims=True)
        X = (mx - X < silent_db / 20.).astype(X.dtype) * X
    return X


def kmeans_silence_weights(X_input_abs, thr):
    """Kmeans_2.py:76-80: notsilent = log10(max(latent)/latent) < threshold, per utterance.  [B, TF]."""
    B = X_input_abs.shape[0]
    lat = X_input_abs.reshape(B, -1)
    with np.errstate(divide='ignore'):
        return (log10(lat.max(axis=1, keepdims=True) / lat) < thr).astype(X_input_abs.dtype)


def apply_masks(X_input, masks):
    """Separator.separate tail (network.py:577-581): X_input [B,T,F], masks [B,TF,S] -> [B*S, T, F]."""
    B, T, F = X_input.shape
    S = masks.shape[2]
    sep = X_input.reshape(B, T * F, 1) * masks
    return np.ascontiguousarray(sep.reshape(B, T, F, S).transpose(0, 3, 1, 2)).reshape(B * S, T, F)


def apply_masks_bwd(X_input, dsep, S):
    """d/d masks of apply_masks."""
    B, T, F = X_input.shape
    d = dsep.reshape(B, S, T * F).transpose(0, 2, 1)
    return d * X_input.reshape(B, T * F, 1)


def enhance_input(separated, X_input, S, normalize):
    """Separator.enhance head (network.py:612-627): concat([separated, tiled X], 3) -> [B*S, T, 2F]."""
    B, T, F = X_input.shape
    sep = separated.reshape(B, S, T, F)
    Xin = np.broadcast_to(X_input[:, None], (B, S, T, F))
    z = np.concatenate([sep, Xin], axis=3).reshape(B * S, T, 2 * F)
    if normalize:
        z = normalization_mean_std(z)
    return z


def enhance_output(y_net, X_input, S, nonlinearity):
    """Separator.enhance tail (network.py:640-660).  y_net [B*S, T, F] -> cost_in [B,TF,S], out [B*S,T,F]."""
    B, T, F = X_input.shape
    y = y_net.reshape(B, S, T * F).transpose(0, 2, 1)
    if nonlinearity == 'softmax':
        e = np.exp(y - y.max(axis=2, keepdims=True))
        y = e / e.sum(axis=2, keepdims=True)
    elif nonlinearity == 'tanh':
        y = np.tanh(y)
    cost_in = y * X_input.reshape(B, T * F, 1)
    out = np.ascontiguousarray(cost_in.transpose(0, 2, 1)).reshape(B * S, T, F)
    return cost_in, out
