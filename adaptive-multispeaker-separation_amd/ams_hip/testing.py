"""Helpers shared by tests, smoke() and bench.py: fabricate the 'pretrained Adapt' folder the front_* recipes load."""
import json
import os

import numpy as np

ADAPT_DEFAULTS = dict(
    dataset='synthetic', dataset_normalize=False, chunk_size=20480, nb_speakers=2, no_random_picking=True,
    validation_step=1000, men=True, women=True, sex=['M', 'F'], epochs=1, batch_size=64, learning_rate=1e-3,
    optimizer='Adam', decay_epoch=50, gradient_norm_clip=0.0, window_size=1024, filters=256, max_pool=256,
    with_max_pool=False, with_average_pool=False, hop_size=256, regularization=0.0, beta=0.0, sparsity=0.01,
    overlap_coef=0.0, overlap_value=0.1, non_negativity=0.0, loss='sdr+l2', separation='mask', type='pretraining',
    pretraining=True, synthetic_batches=4, synthetic_pool=2, no_summaries=True)

SEPARATOR_DEFAULTS = dict(
    normalize_separator='None', abs_input=False, pre_func='None', silence_mask_db=0, nb_layers=3, layer_size=600,
    embedding_size=40, no_normalize=True, recurrent_dropout=0.0, nb_tries=10, nb_steps=10, beta_kmeans=None,
    threshold=2.0, with_silence=False, end_assign=False, silence_loss=False, threshold_silence_loss=2.0,
    function_mask='None', sampling=None, ns_rate=0.1, ns_method='random', add_dilated=False)

ENHANCE_DEFAULTS = dict(normalize_enhance=False, nb_layers_enhance=3, layer_size_enhance=600, nonlinearity='softmax',
                        recurrent_dropout_enhance=0.0)


def front_back_init(W, N, seed=1):
    """front/back window ~ U(+-sqrt(3/W)), bases ~ U(+-sqrt(6/(W+N)))  (SURVEY 8d, Appendix A-9)."""
    rng = np.random.RandomState(seed)
    lw, lb = np.sqrt(3.0 / W), np.sqrt(6.0 / (W + N))
    return {
        'front/window/w': rng.uniform(-lw, lw, W).astype(np.float32),
        'front/bases/bases': rng.uniform(-lb, lb, (W, N)).astype(np.float32),
        'back/window/value': rng.uniform(-lw, lw, W).astype(np.float32),
        'back/bases/value': rng.uniform(-lb, lb, (W, N)).astype(np.float32),
    }


def write_checkpoint(folder, arrays, params, step=0):
    """Lay a folder out the way Network.save/tensorboard_init do: params JSON + model-<step>.npz + checkpoint."""
    os.makedirs(folder, exist_ok=True)
    np.savez(os.path.join(folder, 'model-%d.npz' % step), **{k.replace('/', '.'): v for k, v in arrays.items()})
    with open(os.path.join(folder, 'checkpoint'), 'w') as f:
        json.dump({'model_checkpoint_path': 'model-%d.npz' % step}, f)
    with open(os.path.join(folder, 'params'), 'w') as f:
        json.dump(params, f)
    return folder


def make_pretrained_adapt(folder, **overrides):
    params = dict(ADAPT_DEFAULTS)
    params.update(overrides)
    arrays = front_back_init(params['window_size'], params['filters'])
    return write_checkpoint(folder, arrays, params), params
