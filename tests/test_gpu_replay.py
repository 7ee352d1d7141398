"""GPU: --hip_graph replay == eager launches for EVERY recipe tools/bench_configs.py replays (front_DPCL and the fine-tuning
recipe are covered in test_gpu_step.py): pre-training (path A), STFT_L41, STFT_L41_enhance (host-drawn k-means seeds refreshed
before each replay), front_L41 with S = 3.  Same seeds, same batches, 6 steps: per-step costs and the final weights must agree --
the graph holds the same kernels in the same order, so only atomics-free round-off identity is expected."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from tests.test_gpu_recipes import base_args

os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))
STEPS = 6


def _run(trainer, tfds, L):
    g, model = trainer.graph, trainer.model
    np.random.seed(7)                                          # host-drawn k-means seeds (Kmeans_2.py:61-66) follow the same stream
    costs = []
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        tfds.initialize(tfds.TRAIN)
        for i in range(STEPS):
            costs.append(float(model.train(feed, i)))
    torch.cuda.synchronize()
    return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}


def _compare(make):
    import utils.ops
    outs = []
    for graph in (False, True):
        utils.ops.rng.seed(42)
        torch.manual_seed(0)
        tr, tfds, L = make(graph)
        outs.append(_run(tr, tfds, L))
    (c_e, p_e), (c_g, p_g) = outs
    assert np.all(np.isfinite(c_e)) and np.allclose(c_e, c_g, rtol=1e-5, atol=0), (c_e, c_g)
    assert len(set(np.round(c_g, 9))) > 2                      # replays are not frozen on one batch
    for n in p_e:
        d = np.abs(p_e[n] - p_g[n]).max()
        assert d <= 1e-6 * max(1.0, np.abs(p_e[n]).max()), (n, d)


@pytest.mark.parametrize('loss,separation', [('sdr+l2', 'mask'), ('l2', 'perfect')])
def test_pretraining_replay_matches_eager(loss, separation):
    """cfg2 (README.md:23 flags): front -> mask/perfect separator -> back -> sdr+l2 cost, all four filterbank variables train."""
    from utils.trainer import Adapt_Pretrainer

    def make(graph):
        B, S, L, W, N, hop = 4, 2, 2048, 64, 16, 16
        a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, loss=loss,
                      separation=separation, beta=0.0, regularization=0.0, overlap_coef=1.0, learning_rate=1e-3, pretraining=True,
                      hip_graph=graph, no_summaries=True)
        a.pop('type')
        tr = Adapt_Pretrainer(**a)
        dist, tfds = tr.prepare()
        return tr, tfds, L
    _compare(make)


def test_pretraining_with_sparsity_and_regularisers_replay_matches_eager():
    """Same recipe with every optional term on (--beta, --regularization, --non_negativity): their scalar glue is captured too."""
    from utils.trainer import Adapt_Pretrainer

    def make(graph):
        B, S, L, W, N, hop = 3, 2, 1024, 64, 16, 16
        a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, loss='sdr+l2',
                      separation='mask', beta=0.01, sparsity=0.05, regularization=1e-3, overlap_coef=0.5, learning_rate=1e-3,
                      pretraining=True, hip_graph=graph, no_summaries=True)
        a.pop('type')
        tr = Adapt_Pretrainer(**a)
        dist, tfds = tr.prepare()
        return tr, tfds, L
    _compare(make)


def _stft_l41(graph, **kw):
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Trainer
    B, S, L, W, hop = 3, 2, 2048, 64, 32
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=12, nb_layers=2, embedding_size=8,
                  model_folder=None, learning_rate=1e-3, pretraining=False, tot_speakers=251, hip_graph=graph, no_summaries=True)
    a.update(kw)
    for k in ('filters', 'max_pool', 'type'):
        a.pop(k, None)
    tr = STFT_Separator_Trainer(L41Model, 'STFT_L41', **dict(a))
    dist, tfds = tr.prepare()
    return tr, tfds, L, a


def test_stft_l41_replay_matches_eager():
    _compare(lambda graph: _stft_l41(graph)[:3])


def test_stft_l41_enhance_replay_matches_eager():
    """cfg4's second stage: frozen L41 separator + hard k-means (seeds drawn on the host before every replay) + enhance stack."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_enhance_Trainer

    def make(graph):
        tr0, tfds0, L, a = _stft_l41(False)
        with tr0.graph.as_default():
            tr0.model.create_saver()
            tr0.model.save(0)
            folder = tr0.model._dir()
        del tr0
        a = dict(a)
        a.update(model_folder=folder, nb_tries=2, nb_steps=3, end_assign=True, nonlinearity='softmax', layer_size_enhance=8,
                 nb_layers_enhance=2, hip_graph=graph)
        tr = STFT_Separator_enhance_Trainer(L41Model, 'STFT_L41_enhance', **a)
        dist, tfds = tr.prepare()
        return tr, tfds, L
    _compare(make)


def test_front_l41_three_speakers_replay_matches_eager():
    """cfg5 family: plugged L41 on the frozen front, S = 3."""
    from ams_hip import testing
    from models.L41 import L41Model
    from utils.trainer import Front_Separator_Trainer

    def make(graph):
        tmp = tempfile.mkdtemp(prefix='ams_rp_')
        B, S, L, W, N, hop = 3, 3, 1024, 64, 16, 16
        folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                       batch_size=B, nb_speakers=S)
        a = base_args(**params)
        a.update(layer_size=12, nb_layers=2, embedding_size=8, model_folder=folder, model_previous=None, pretraining=False,
                 learning_rate=1e-3, tot_speakers=251, hip_graph=graph, no_summaries=True)
        a.pop('type')
        tr = Front_Separator_Trainer(L41Model, 'front_L41', **a)
        dist, tfds = tr.prepare()
        return tr, tfds, L
    _compare(make)


def test_a_replayed_step_sees_weights_written_behind_the_optimizer():
    """a captured front_DPCL step takes the FROZEN front's filter |w| * bases and its bound as constants of the capture.  When
    something other than the optimizer kernel writes weights (restore_model -> Network._weights_written), the next call must not replay
    the old filter: the step is captured again behind two eager steps.  Two models, same seeds: one trains eagerly, one replays; after
    three steps the front window of both is rewritten (x 1.5) through the same hook, and the costs of the following steps still agree."""
    from tests.smoke_step import build_front_dpcl
    import utils.ops
    cfg = dict(B=4, L=2048, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, no_summaries=True)
    costs = []
    for graph in (False, True):
        utils.ops.rng.seed(42)
        torch.manual_seed(0)
        np.random.seed(3)
        tmp = tempfile.mkdtemp(prefix='ams_rewrite_')
        trainer, tfds = build_front_dpcl(tmp, hip_graph=graph, **cfg)
        g, model = trainer.graph, trainer.model
        c = []
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: cfg['L']}
            tfds.initialize(tfds.TRAIN)
            for i in range(4):
                c.append(float(model.train(feed, i)))
            g.variables['front/window/w'].data.mul_(1.5)
            model._weights_written()
            for i in range(4, 9):
                c.append(float(model.train(feed, i)))
        torch.cuda.synchronize()
        costs.append(c)
    assert np.allclose(costs[0], costs[1], rtol=2e-5, atol=0), costs
    assert abs(costs[0][4] - costs[0][3]) > 1e-6 * abs(costs[0][3])        # the rewrite changed the cost: the check above is not vacuous


def test_staging_by_plain_copies_refreshes_the_waveform_bound():
    """Network._stage leaves max |waveform| for the front product (fp16x3 scales by it) when the fused staging launch
    applies; a later batch that has to come by plain copies (its tensors are not back to back) must not leave the OLD bound in place --
    a louder batch under a stale bound overflows fp16."""
    from models.network import Network
    B, S, L = 3, 2, 512
    flat = torch.zeros(B * L + B * S * L, device='cuda')
    static = [flat[:B * L].view(B, L), flat[B * L:].view(B, S, L)]
    g = torch.Generator(device='cuda').manual_seed(1)
    quiet = torch.rand(B * L + B * S * L, device='cuda', generator=g) * 0.01
    net = Network.__new__(Network)
    net._stage(static, [quiet[:B * L].view(B, L), quiet[B * L:].view(B, S, L)])          # back to back: the fused launch
    am = static[0]._ams_x_amax
    assert abs(float(am) - float(quiet.abs().max())) < 1e-9
    loud_m = torch.rand(B, L, device='cuda', generator=g) * 7.0                            # two separate tensors: plain copies
    loud_n = torch.rand(B, S, L, device='cuda', generator=g) * 9.0
    net._stage(static, [loud_m, loud_n])
    assert static[0]._ams_x_amax is am                                                    # same address: a captured step reads it
    assert abs(float(am) - float(max(loud_m.abs().max(), loud_n.abs().max()))) < 1e-6
    assert torch.equal(static[0], loud_m) and torch.equal(static[1], loud_n)


def test_a_captured_step_measures_both_forms_of_its_forward_products_and_keeps_one(monkeypatch):
    """models/network.py::_train_graphed with ops.PS_AUTOTUNE: the step is captured twice -- forward products from pre-split operand
    images (csrc/gemm_ps.hip) and split inside the product (csrc/gemm.hip) --, the two graphs are replayed in alternating blocks (every
    replay a real training step) and the faster one stays.  Whatever the choice on this board, the cost trajectory is the eager one:
    the two forms differ in summation order only."""
    from tests.smoke_step import build_front_dpcl
    from models.network import Network
    from ams_hip import ops
    import utils.ops
    monkeypatch.setattr(Network, '_PS_TUNE_BLOCK', 3)
    monkeypatch.setattr(Network, '_PS_RETUNE_EVERY', 5)         # a long run asks again: here after five replays
    monkeypatch.setattr(ops, 'PS_AUTOTUNE', True)
    ops.PS_TUNED.clear()
    cfg = dict(B=4, L=2048, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, no_summaries=True)
    costs, models = [], []
    for graph in (False, True):
        utils.ops.rng.seed(42)
        torch.manual_seed(0)
        np.random.seed(3)
        trainer, tfds = build_front_dpcl(tempfile.mkdtemp(prefix='ams_tune_'), hip_graph=graph, **cfg)
        g, model = trainer.graph, trainer.model
        c = []
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: cfg['L']}
            tfds.initialize(tfds.TRAIN)
            for i in range(2 + 12 + 5 + 12 + 2):              # two eager steps, 4 x 3 tuning replays, five replays, a second tuning, two more
                c.append(float(model.train(feed, i)))
                if graph and i == 6:
                    assert model._cg_state['tune'] is not None and len(model._cg_state['tune']['variants']) == 2
        torch.cuda.synchronize()
        costs.append(c)
        models.append(model)
    assert np.allclose(costs[0], costs[1], rtol=2e-5, atol=0), costs
    m = models[1]
    assert m._cg_state['tune'] is None and isinstance(m._ps_choice, bool)
    assert ops.PS_TUNED['presplit'] == m._ps_choice and ops.PS_TUNED['ms_presplit'] > 0 and ops.PS_TUNED['ms_in_product'] > 0
    assert ops.PS_TUNED['decisions'] == 2                        # decided at the start and once more after _PS_RETUNE_EVERY replays
    assert ops.PRESPLIT                                          # the process-wide default is untouched by a model's choice


def test_front_dpcl_finetuning_replay_matches_eager():
    """cfg3(ii) at the model's real size and a SMALL batch (8 utterances x 20480 points: the soft k-means backward then launches fewer
    chunks per utterance than its workspace query assumes -- the shape at which the word holding max |dx| was once read from the wrong
    place and the replayed step diverged while every test at 2-3 k points or at B = 64 passed): replayed = eager, all costs finite."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import bench_configs as bc
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Finetuning_Trainer
    B, L, S, N = 8, 20480, 2, 256
    tmp = tempfile.mkdtemp(prefix='ams_ft_replay_')
    tr0, tfds0, a0 = bc._front_dpcl_checkpoint(tmp, DPCL, 'front_DPCL', B, S, L, N)
    with tr0.graph.as_default():
        tr0.model.create_saver()
        tr0.model.save(0)
        folder = tr0.model._dir()
    del tr0

    def make(graph):
        a = dict(a0)
        a.update(model_folder=folder, nb_tries=1, nb_steps=10, beta_kmeans=10.0, with_silence=True, threshold=2.0, end_assign=True,
                 loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4, hip_graph=graph)
        tr = Front_Separator_Finetuning_Trainer(DPCL, 'front_L41_finetuning', **a)
        dist, tfds = tr.prepare()
        return tr, tfds, L
    _compare(make)
