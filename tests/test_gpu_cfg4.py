"""GPU: BASELINE.json configs[3] -- "STFT + L41 (models/L41.py) with enhancing layer, 2 speakers" -- through the recipe classes the
reference's entry points build (experiments/training/STFT_L41.py -> STFT_Separator_Trainer(L41Model, 'STFT_L41'),
STFT_L41_enhance.py -> STFT_Separator_enhance_Trainer(L41Model, 'STFT_L41_enhance'); reference utils/trainer.py:468-500,
models/L41.py:47-186, models/network.py:610-693) vs the oracle, at reduced size."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import step as ostep, recipes as orec, optim as ooptim
from tests.test_gpu_recipes import base_args, one_train_step, check_step, _full_checkpoint, _infer


@pytest.mark.parametrize('normalize', [True, False])
def test_stft_l41_step(normalize):
    """experiments.training.STFT_L41: |STFT| -> 2xBLSTM -> Conv1D -> [l2norm] -> L41 cost against the speaker vectors, AMSGrad."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Trainer
    B, S, L, W, hop, LS, NL, E = 4, 2, 2048, 64, 32, 12, 2, 8
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=1e-3, no_normalize=normalize)
    a.pop('type')
    tr = STFT_Separator_Trainer(L41Model, 'STFT_L41', **a)
    dist, tfds = tr.prepare()
    names = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert 'speaker_centroids' in names and any(n.startswith('prediction/') for n in names)
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    assert P['speaker_centroids'].shape == (251, E)                      # tot_speakers of the pipeline (trainer.py:200-202)
    c_ref, g_ref, V, Y = ostep.stft_l41_loss(xm, xn, I, P, W, hop, NL, E, normalize)
    assert set(np.unique(Y)) == {-1.0, 1.0}                              # L41 masks (L41.py:9-10)
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3))


def test_stft_l41_step_with_knearest_negative_sampling():
    """`--sampling 3 --ns_method k-nearest --ns_rate 0.25` (utils/trainer.py:101-105, models/L41.py:69-116,143-147,165-166): the
    whole training step against the oracle -- cost, every gradient (speaker_centroids receives the negatives' share too), AMSGrad."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Trainer
    B, S, L, W, hop, LS, NL, E, K, rate = 4, 2, 2048, 64, 32, 12, 2, 8, 3, 0.25
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=1e-3, no_normalize=True, sampling=K, ns_method='k-nearest',
                  ns_rate=rate)
    a.pop('type')
    tr = STFT_Separator_Trainer(L41Model, 'STFT_L41', **a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, V, Y = ostep.stft_l41_loss(xm, xn, I, P, W, hop, NL, E, True, sampling=K, ns_rate=rate)
    c_plain = ostep.stft_l41_loss(xm, xn, I, P, W, hop, NL, E, True, want_grads=False)[0]
    assert c_ref > c_plain + 1e-3                                        # the negatives are part of the cost
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3))


@pytest.mark.parametrize('graph', [False, True])
def test_stft_l41_random_negative_sampling_trains(graph):
    """`--sampling 4` with the default ns_method 'random' (L41.py:117-139): a fresh set of non-mixture speakers per utterance and
    step, drawn on the device -- also from inside a replayed hipGraph.  The draw is not TensorFlow's, so: finite costs above the
    plain L41 cost of the same weights, and different costs on repeated evaluations of ONE batch (new negatives each time)."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Trainer
    B, S, L, W, hop, LS, NL, E = 3, 2, 2048, 64, 32, 12, 2, 8
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=0.0, no_normalize=True, sampling=4, ns_rate=0.5,
                  hip_graph=graph, no_summaries=True, synthetic_batches=1, synthetic_pool=1)
    a.pop('type')
    tr = STFT_Separator_Trainer(L41Model, 'STFT_L41', **a)
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    costs = []
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        tfds.initialize(tfds.TRAIN)
        for i in range(6):
            costs.append(float(model.train(feed, i)))
    torch.cuda.synchronize()
    assert np.all(np.isfinite(costs)) and min(costs) > 0.0
    assert len(set(np.round(costs[2:], 7))) > 1, costs                   # lr = 0, one batch: only the negatives change


@pytest.mark.parametrize('nonlinearity', ['softmax', 'tanh'])
def test_stft_l41_enhance_step(nonlinearity):
    """experiments.training.STFT_L41_enhance: restored STFT + L41 separator (checkpoint carries speaker_centroids) -> hard k-means
    masks -> enhance BLSTM stack -> PIT squared error; only enhance/* trains."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_enhance_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_l41enh_')
    rng = np.random.RandomState(43)
    B, S, L, W, hop, LS, NL, E, tries, steps, LSE, NLE, NSPK = 2, 2, 1024, 64, 16, 12, 2, 8, 2, 3, 8, 2, 251
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False, tot_speakers=NSPK)
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity=nonlinearity, learning_rate=1e-3, pretraining=False, tot_speakers=NSPK)
    a.pop('type')
    tr = STFT_Separator_enhance_Trainer(L41Model, 'STFT_L41_enhance', **a)
    dist, tfds = tr.prepare()
    names = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert names and all(n.startswith('enhance/') for n in names)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    assert np.array_equal(Pg['speaker_centroids'], P['speaker_centroids'].astype(np.float64))      # restored, frozen
    c_ref, g_ref = orec.stft_enhance_loss(xm, xn, Pg, W, hop, NL, E, NLE, idx, tries, steps, nonlinearity=nonlinearity)
    check_step(cost, c_ref, grads, g_ref, Pg, P_new, ooptim.AMSGrad(1e-3), tol=5e-4)


def test_stft_l41_enhanced_inference():
    """The chain cfg4 ends in: STFT_Separator_Enhanced_Inference(L41Model): |STFT| -> L41 embeddings -> hard k-means -> enhance -> iSTFT."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Enhanced_Inference
    tmp = tempfile.mkdtemp(prefix='ams_l41einf_')
    rng = np.random.RandomState(57)
    B, S, L, W, hop, LS, NL, E, tries, steps, LSE, NLE, NSPK = 2, 2, 2048, 64, 32, 12, 2, 8, 2, 3, 8, 1, 251
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False, enhance=(LSE, NLE),
                                         tot_speakers=NSPK)
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', out=False, tot_speakers=NSPK)
    a.pop('type')
    tr = STFT_Separator_Enhanced_Inference(L41Model, 'STFT_L41_enhance_inference', **a)
    xm, xn, out = _infer(tr, L)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref = orec.stft_separate_enhanced_infer(xm, xn, P64, W, hop, NL, E, NLE, idx, tries, steps)
    assert out.shape == out_ref.shape == (B, S, (T - 1) * hop + W)
    assert np.linalg.norm(out - out_ref) / np.linalg.norm(out_ref) < 1e-3
