"""Oracle: remaining whole-recipe restatements (pretraining step, inference pipelines).

Test infrastructure only -- see oracle/__init__.py.
"""
import numpy as np
from . import front, stft, separate, losses, kmeans, step


def pretrain_loss(x_mix, x_non_mix, P, hop, loss_kind, separation, overlap_coef=0.0, want_grads=True, beta=0.0, sparsity=0.01,
                  regularization=0.0, non_negativity=0.0):
    """experiments.training.pretraining step, path A (adapt.py:16-56, 95-252, 307-402), with the default-on terms of the CLI
    (utils/trainer.py:151-161: beta 1e-2 x sum kl_div(sparsity, p_hat), regularization 1e-4 applied twice to the two filter
    l2_losses, non_negativity applied twice to mean_b sum min(front, 0)^2 -- adapt.py:130-132, 312-316, 377-384)."""
    B, S, L = x_non_mix.shape
    x = np.concatenate([x_mix, x_non_mix.reshape(B * S, L)], axis=0)
    w1, b1, w2, b2 = P['front/window/w'], P['front/bases/bases'], P['back/window/value'], P['back/bases/value']
    f, f2 = front.front_filter(w1, b1), front.front_filter(w2, b2)
    y = front.conv_strided(x, f, hop)
    z = front.pretrain_separator(y, B, S, separation)
    back = front.synth_strided(z, f2, hop, L).reshape(B, S, L)
    loss, l2, sdr = losses.pretrain_cost(x_mix, x_non_mix, back, loss_kind)
    ov = front.overlap_metric(y, B, S)
    cost = loss + (overlap_coef * ov if overlap_coef != 0.0 else 0.0)
    Bt = y.shape[0]
    p_hat = None
    if beta != 0.0:
        p_hat, kl = front.sparsity_terms(y, sparsity)
        cost = cost + beta * kl
    reg, nn = losses.adapt_regularizers(f, f2, y, regularization, non_negativity)
    if regularization != 0.0:
        cost = cost + regularization * reg
    if non_negativity is not None and non_negativity != 0.0:
        cost = cost + non_negativity * nn
    if not want_grads:
        return cost, back
    dback = losses.pretrain_cost_bwd(x_non_mix, back, loss_kind).reshape(B * S, L)
    dz, df2 = front.synth_strided_bwd(z, f2, hop, dback)
    dy = front.pretrain_separator_bwd(y, B, S, separation, dz)
    if overlap_coef != 0.0:
        dy = dy + overlap_coef * front.overlap_metric_bwd(y, B, S)
    if beta != 0.0:
        dy = dy + beta * front.sparsity_terms_bwd(y, sparsity, p_hat)
    if non_negativity is not None and non_negativity != 0.0:
        dy = dy + non_negativity * non_negativity * 2.0 * np.minimum(y, 0.0) / Bt
    df = front.conv_strided_bwd_filter(x, dy, f.shape[0], hop)
    if regularization != 0.0:                            # d/df [lam * lam * 0.5 * sum f^2] = lam^2 f
        df = df + regularization * regularization * f
        df2 = df2 + regularization * regularization * f2
    dw1, db1 = front.front_filter_bwd(w1, b1, df)
    dw2, db2 = front.front_filter_bwd(w2, b2, df2)
    grads = {'front/window/w': dw1, 'front/bases/bases': db1, 'back/window/value': dw2, 'back/bases/value': db2}
    return cost, grads, back


def front_separate_infer(x_mix, x_non_mix, P, hop, nb_layers, E, init_idx, nb_tries, nb_steps, beta=None, with_silence=False,
                         threshold=2.0, end_assign=True, labels=None):
    """Front_Separator_Inference (trainer.py:420-434): front -> DPCL embeddings -> k-means masks -> back.  `labels` [B, TF]: use these
    instead of running k-means (tests/test_gpu_fullstep.py compares the synthesis on equal labels after counting the unequal ones)."""
    B, S, L = x_non_mix.shape
    y = step.front_rep(x_mix, x_non_mix, P, hop)
    X, _ = separate.split_front(y, B, S)
    T, Fq = X.shape[1:]
    V = None
    if labels is None:
        V, _ = step.prediction_fwd(X, P, nb_layers, E)
        emb = V.reshape(B, T * Fq, E)
        w = separate.kmeans_silence_weights(np.abs(X), threshold) if with_silence else None
        cent, labels, best = kmeans.kmeans(emb, init_idx, S, nb_tries, nb_steps, beta=beta, notsilent=w, assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, beta).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    f2 = front.front_filter(P['back/window/value'], P['back/bases/value'])
    out = front.synth_strided(sep, f2, hop, L).reshape(B, S, L)
    return out, labels, V


def stft_separate_infer(x_mix, x_non_mix, P, W, hop, nb_layers, E, init_idx, nb_tries, nb_steps, end_assign=True):
    """STFT_Separator_Inference (trainer.py:406-417): STFT -> DPCL -> hard k-means masks -> iSTFT with the mixture phase."""
    B, S, L = x_non_mix.shape
    X, _, ang = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    V, _ = step.prediction_fwd(X, P, nb_layers, E)
    T, Fq = X.shape[1:]
    cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, None).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    out = stft.istft(sep, np.repeat(ang, S, axis=0), W, hop).reshape(B, S, -1)
    return out, labels, V


def front_finetune_cost(x_mix, x_non_mix, P, hop, nb_layers, E, init_idx, nb_tries, nb_steps, beta, with_silence, threshold, end_assign,
                        loss_kind):
    """Front_Separator_Finetuning_Trainer objective (trainer.py:600-619 -> adapt.py:339-372), forward only: frozen front ->
    DPCL embeddings -> SOFT k-means masks -> back -> PIT l2 + (cross-batch, quirk C-3) sdr."""
    out, labels, V = front_separate_infer(x_mix, x_non_mix, P, hop, nb_layers, E, init_idx, nb_tries, nb_steps, beta=beta,
                                          with_silence=with_silence, threshold=threshold, end_assign=end_assign)
    loss, l2, sdr = losses.pit_cost_adapt(x_mix, x_non_mix, out, loss_kind)
    return loss, out


def _enhance_loss_core(X, X_nm, P, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, nonlinearity, end_assign, want_grads,
                       labels=None, info=None):
    """Shared tail of the *_enhance trainers: DPCL embeddings -> hard k-means masks -> enhance BLSTM stack (network.py:610-660) ->
    PIT squared error against the non-mix representation (network.py:662-693).  Gradients for the 'enhance/*' variables only.
    `info` (a dict) receives the k-means labels; `labels` [B, TF] replaces them (tests/test_gpu_fullstep.py: at 1.3 M points a few
    float32 embeddings lie within rounding of a cluster boundary; the test counts them, then compares the rest of the step on
    equal labels)."""
    from . import blstm, dense
    B, T, Fq = X.shape
    S = X_nm.shape[-1]
    if labels is None:
        V, _ = step.prediction_fwd(X, P, nb_layers, E)
        cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, assign_at_end=end_assign)
    if info is not None:
        info['labels'] = labels
    masks = kmeans.masks_from_labels(labels, S, None).astype(X.dtype)
    sep = separate.apply_masks(X, masks)                                    # [B*S, T, F]
    z = separate.enhance_input(sep, X, S, False)                            # [B*S, T, 2F]
    h, caches = blstm.blstm_stack_fwd(z, step.stack_params(P, 'enhance', nb_layers_enh))
    u = dense.dense_fwd(h, P['enhance/W'], P['enhance/b'])                  # [B*S, T, F]
    cost_in, out = separate.enhance_output(u, X, S, nonlinearity)           # [B,TF,S]
    cost, best_perm = losses.enhance_cost(X_nm, cost_in)
    if not want_grads:
        return cost
    est = cost_in.transpose(0, 2, 1)                                        # [B,S,TF]
    tgt = X_nm.reshape(B, -1, S).transpose(0, 2, 1)
    d_est = losses.pit_l2_bwd(tgt, est, best_perm, 'sum', 'sum', 1.0)       # [B,S,TF]
    d_cost_in = d_est.transpose(0, 2, 1)                                    # [B,TF,S]
    dy = d_cost_in * X.reshape(B, T * Fq, 1)
    if nonlinearity == 'softmax':
        ylog = u.reshape(B, S, T * Fq).transpose(0, 2, 1)
        e = np.exp(ylog - ylog.max(axis=2, keepdims=True))
        sm = e / e.sum(axis=2, keepdims=True)
        dlog = sm * (dy - (dy * sm).sum(axis=2, keepdims=True))
    elif nonlinearity == 'tanh':
        ylog = u.reshape(B, S, T * Fq).transpose(0, 2, 1)
        dlog = dy * (1.0 - np.tanh(ylog) ** 2)
    else:
        dlog = dy
    du = dlog.transpose(0, 2, 1).reshape(B * S, T, Fq)
    dh, dW, db = dense.dense_bwd(h, P['enhance/W'], du)
    _, lg = blstm.blstm_stack_bwd(dh, caches, need_dx=False)
    grads = {'enhance/W': dW, 'enhance/b': db}
    for i, g in enumerate(lg):
        for n, v in zip(step.lstm_names('enhance', i), g):
            grads[n] = v
    return cost, grads


def front_enhance_loss(x_mix, x_non_mix, P, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, nonlinearity='softmax',
                       end_assign=True, want_grads=True):
    """Front_Separator_Enhance_Trainer objective (trainer.py:621-630, adapt.py:457-469): frozen front + DPCL + hard k-means masks ->
    enhance stack -> PIT squared error against the signed non-mix representation."""
    B, S, L = x_non_mix.shape
    y = step.front_rep(x_mix, x_non_mix, P, hop)
    X, X_nm = separate.split_front(y, B, S)
    return _enhance_loss_core(X, X_nm, P, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, nonlinearity, end_assign, want_grads)


def stft_enhance_loss(x_mix, x_non_mix, P, W, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, nonlinearity='softmax',
                      end_assign=True, want_grads=True, labels=None, info=None):
    """STFT_Separator_enhance_Trainer objective (trainer.py:497-509 -> network.py:505-693): |STFT| -> DPCL -> hard k-means
    masks -> enhance stack -> PIT squared error against the non-mix magnitudes."""
    X, X_nm, _ = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    return _enhance_loss_core(X, X_nm, P, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, nonlinearity, end_assign, want_grads,
                              labels=labels, info=info)


def _enhance_forward(X, sep, P, S, nb_layers_enh, nonlinearity):
    from . import blstm, dense
    z = separate.enhance_input(sep, X, S, False)
    h, _ = blstm.blstm_stack_fwd(z, step.stack_params(P, 'enhance', nb_layers_enh))
    u = dense.dense_fwd(h, P['enhance/W'], P['enhance/b'])
    return separate.enhance_output(u, X, S, nonlinearity)                   # cost_in [B,TF,S], out [B*S,T,F]


def stft_finetune_cost(x_mix, x_non_mix, P, W, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, beta,
                       nonlinearity='softmax', end_assign=True):
    """STFT_Separator_FineTune_Trainer objective (trainer.py:511-536 -> network.py:505-607,697-725), forward only: |STFT| -> DPCL
    -> SOFT k-means masks -> enhance stack -> iSTFT with the mixture phase -> PIT 0.5*L2^2 on waveforms."""
    B, S, L = x_non_mix.shape
    X, _, ang = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    V, _ = step.prediction_fwd(X, P, nb_layers, E)
    T, Fq = X.shape[1:]
    cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, beta=beta, assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, beta).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    _, out = _enhance_forward(X, sep, P, S, nb_layers_enh, nonlinearity)
    wav = stft.istft(out, np.repeat(ang, S, axis=0), W, hop).reshape(B, S, -1)
    return losses.cost_finetuning(x_non_mix, wav)[0], wav


def front_enhance_finetune_cost(x_mix, x_non_mix, P, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, beta, with_silence,
                                threshold, end_assign, nonlinearity='softmax'):
    """Front_Separator_Enhance_Finetuning_Trainer objective (trainer.py:632-658 -> adapt.py:206-246,404-431), forward only: frozen
    front -> DPCL -> SOFT k-means masks -> enhance stack -> back end -> PIT 0.5*L2^2 on waveforms (Adapt.cost_finetuning, NOT the
    sdr+l2 Adapt.cost the plain finetuning trainer uses)."""
    B, S, L = x_non_mix.shape
    y = step.front_rep(x_mix, x_non_mix, P, hop)
    X, _ = separate.split_front(y, B, S)
    V, _ = step.prediction_fwd(X, P, nb_layers, E)
    T, Fq = X.shape[1:]
    w = separate.kmeans_silence_weights(np.abs(X), threshold) if with_silence else None
    cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, beta=beta, notsilent=w,
                                       assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, beta).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    _, out = _enhance_forward(X, sep, P, S, nb_layers_enh, nonlinearity)
    f2 = front.front_filter(P['back/window/value'], P['back/bases/value'])
    back = front.synth_strided(out, f2, hop, L).reshape(B, S, L)
    return losses.cost_finetuning(x_non_mix, back)[0], back


def front_separate_enhanced_infer(x_mix, x_non_mix, P, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps, beta=None,
                                  with_silence=False, threshold=2.0, end_assign=True, nonlinearity='softmax'):
    """Front_Separator_Enhanced_Inference (trainer.py:436-449): front -> DPCL -> k-means masks -> enhance stack -> back."""
    B, S, L = x_non_mix.shape
    y = step.front_rep(x_mix, x_non_mix, P, hop)
    X, _ = separate.split_front(y, B, S)
    V, _ = step.prediction_fwd(X, P, nb_layers, E)
    T, Fq = X.shape[1:]
    w = separate.kmeans_silence_weights(np.abs(X), threshold) if with_silence else None
    cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, beta=beta, notsilent=w,
                                       assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, beta).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    _, out = _enhance_forward(X, sep, P, S, nb_layers_enh, nonlinearity)
    f2 = front.front_filter(P['back/window/value'], P['back/bases/value'])
    return front.synth_strided(out, f2, hop, L).reshape(B, S, L)


def stft_separate_enhanced_infer(x_mix, x_non_mix, P, W, hop, nb_layers, E, nb_layers_enh, init_idx, nb_tries, nb_steps,
                                 end_assign=True, nonlinearity='softmax'):
    """STFT_Separator_Enhanced_Inference (trainer.py:390-404): |STFT| -> DPCL -> hard k-means -> enhance stack -> iSTFT."""
    B, S, L = x_non_mix.shape
    X, _, ang = stft.stft_preprocessing(x_mix, x_non_mix, W, hop)
    V, _ = step.prediction_fwd(X, P, nb_layers, E)
    T, Fq = X.shape[1:]
    cent, labels, best = kmeans.kmeans(V.reshape(B, T * Fq, E), init_idx, S, nb_tries, nb_steps, assign_at_end=end_assign)
    masks = kmeans.masks_from_labels(labels, S, None).astype(X.dtype)
    sep = separate.apply_masks(X, masks)
    _, out = _enhance_forward(X, sep, P, S, nb_layers_enh, nonlinearity)
    return stft.istft(out, np.repeat(ang, S, axis=0), W, hop).reshape(B, S, -1)


def pretrained_infer(x_mix, x_non_mix, P, hop, separation='mask'):
    """Pretrained_Inference (trainer.py:451-462): the pre-trained filterbank alone -- front -> oracle 'mask'/'perfect'
    separator (adapt.py:162-196) -> back."""
    B, S, L = x_non_mix.shape
    y = step.front_rep(x_mix, x_non_mix, P, hop)
    sep = front.pretrain_separator(y, B, S, separation)
    f2 = front.front_filter(P['back/window/value'], P['back/bases/value'])
    return front.synth_strided(sep, f2, hop, L).reshape(B, S, L)


def pretrain_loss_maxpool(x_mix, x_non_mix, P, Pool, hop, loss_kind, separation, want_grads=True):
    """experiments.training.pretraining --with_max_pool (path B; adapt.py:114-117, 210-223) with beta = reg = overlap_coef = 0."""
    B, S, L = x_non_mix.shape
    x = np.concatenate([x_mix, x_non_mix.reshape(B * S, L)], axis=0)
    w1, b1, w2, b2 = P['front/window/w'], P['front/bases/bases'], P['back/window/value'], P['back/bases/value']
    f, f2 = front.front_filter(w1, b1), front.front_filter(w2, b2)
    W = f.shape[0]
    y, am = front.front_maxpool(x, f, Pool, hop)
    z = front.pretrain_separator(y, B, S, separation)
    am_t = np.repeat(am[:B], S, axis=0)                              # mixture argmax tiled S times (adapt.py:212-218)
    back = front.synth_unpool(z, am_t, f2, L).reshape(B, S, L)
    loss, l2, sdr = losses.pretrain_cost(x_mix, x_non_mix, back, loss_kind)
    if not want_grads:
        return loss, back
    dback = losses.pretrain_cost_bwd(x_non_mix, back, loss_kind).reshape(B * S, L)
    dz, df2 = front.synth_unpool_bwd(z, am_t, f2, dback)
    dy = front.pretrain_separator_bwd(y, B, S, separation, dz)
    df = front.front_maxpool_bwd_filter(x, dy, am, W)
    dw1, db1 = front.front_filter_bwd(w1, b1, df)
    dw2, db2 = front.front_filter_bwd(w2, b2, df2)
    return loss, {'front/window/w': dw1, 'front/bases/bases': db1, 'back/window/value': dw2, 'back/bases/value': db2}, back, am


def pretrain_forward_avgpool(x_mix, x_non_mix, P, Pool, separation):
    """Forward of the --with_average_pool pretraining path (path C; adapt.py:118-120, 225-228): returns back [B,S,L]."""
    B, S, L = x_non_mix.shape
    x = np.concatenate([x_mix, x_non_mix.reshape(B * S, L)], axis=0)
    f = front.front_filter(P['front/window/w'], P['front/bases/bases'])
    f2 = front.front_filter(P['back/window/value'], P['back/bases/value'])
    y = front.front_avgpool(x, f, Pool)
    z = front.pretrain_separator(y, B, S, separation)
    up = front.upsample_nearest(z, Pool)
    return front.synth_strided(up, f2, 1, L).reshape(B, S, L), y
