"""GPU: whole training steps through the host mirror (reference API) vs the oracle."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def test_front_dpcl_training_step_matches_oracle():
    from tests.smoke_step import run_smoke
    errs = run_smoke(torch, np, verbose=False)
    assert max(errs.values()) < 2e-4, errs


def test_hip_graph_replay_matches_eager():
    """--hip_graph: 2 eager warm-up steps, capture, then replays -- parameters after 5 steps equal the all-eager run
    (same kernels, same accumulation order; the weight-gradient side stream is part of the captured graph)."""
    import tempfile
    from tests.smoke_step import build_front_dpcl

    def run(graph):
        tmp = tempfile.mkdtemp(prefix='ams_cg_')
        trainer, tfds = build_front_dpcl(tmp, B=4, L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, hip_graph=graph,
                                         no_summaries=True)
        g, model = trainer.graph, trainer.model
        costs = []
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 1024}
            tfds.initialize(tfds.TRAIN)
            for i in range(5):
                costs.append(float(model.train(feed, i)))
        torch.cuda.synchronize()
        return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}

    c_e, p_e = run(False)
    c_g, p_g = run(True)
    assert np.allclose(c_e, c_g, rtol=1e-6, atol=0), (c_e, c_g)
    for n in p_e:
        assert np.array_equal(p_e[n], p_g[n]) or np.abs(p_e[n] - p_g[n]).max() <= 1e-7 * max(1.0, np.abs(p_e[n]).max()), n


def test_ring_sync_arena_matches_private_buffers():
    """The sync buffers of a pass's ring launches come from one arena that pass_begin() zeroes on the side stream (ops._RingArena,
    `safe` bit 2 of ams_blstm_ring_fwd/bwd: no memset node in front of a ring), and the flat gradient buffer is zeroed there too
    (zero_grad(defer=True)): 5 eager steps end bit-identical to the same steps with a private buffer + memset per launch."""
    import tempfile
    from ams_hip import ops
    from tests.smoke_step import build_front_dpcl

    def run(arena):
        old, ops.RING_ARENA = ops.RING_ARENA, arena
        try:
            tmp = tempfile.mkdtemp(prefix='ams_ar_')
            trainer, tfds = build_front_dpcl(tmp, B=4, L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, no_summaries=True)
            g, model = trainer.graph, trainer.model
            costs = []
            with g.as_default():
                feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 1024}
                tfds.initialize(tfds.TRAIN)
                for i in range(5):
                    costs.append(float(model.train(feed, i)))
            torch.cuda.synchronize()
            ops.raise_on_ring_errors()
            return costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}
        finally:
            ops.RING_ARENA = old

    c_p, p_p = run(False)
    ops._ARENAS.clear()
    c_a, p_a = run(True)
    if ops.LSTM_RING != '0':
        ar = ops._ARENAS[torch.cuda.current_device()]
        assert len(ar.slots) == 4 and ar.next == 4 and ar.cleared == ops.PASS[0], (len(ar.slots), ar.next, ar.cleared, ops.PASS[0])
    assert c_p == c_a, (c_p, c_a)
    for n in p_p:
        assert np.array_equal(p_p[n], p_a[n]), n


def test_training_with_recurrent_dropout():
    """--recurrent_dropout 0.3 through the trainer: training passes take the DropoutWrapper form (fresh masks every step, so the same
    batch gives different costs), evaluation passes are the plain network (tf.cond(training, 1 - drop, 1.0): identical costs, equal to
    a model without the flag), gradients reach every variable and the step stays finite; the same under --hip_graph."""
    import tempfile
    from tests.smoke_step import build_front_dpcl

    def build(drop, graph=False):
        tmp = tempfile.mkdtemp(prefix='ams_rd_')
        tr, tfds = build_front_dpcl(tmp, B=4, L=1024, W=64, N=16, hop=16, layer_size=16, nb_layers=2, E=8, recurrent_dropout=drop,
                                    hip_graph=graph, no_summaries=True)
        return tr, tfds

    tr0, tfds0 = build(0.0)
    with tr0.graph.as_default():
        feed0 = {tfds0.handle: tfds0.get_handle(tfds0.TRAIN), tfds0.chunk_size: 1024}
        tfds0.initialize(tfds0.TRAIN)
        v_plain = tr0.model.valid_batch(feed0, 0)
        c_plain = float(tr0.model.train(feed0, 0))
    for graph in (False, True):
        tr, tfds = build(0.3, graph)
        g, model = tr.graph, tr.model
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 1024}
            tfds.initialize(tfds.TRAIN)
            v1 = model.valid_batch(feed, 0)
            assert v1 == v_plain, (v1, v_plain)                       # same seeds, same init, same batch: evaluation ignores the flag
            before = {v.ams_name: v.detach().clone() for v in model.trainable_variables}
            costs = [float(model.train(feed, i)) for i in range(5)]
        torch.cuda.synchronize()
        assert np.all(np.isfinite(costs)) and costs[0] != c_plain and len(set(costs)) == 5, (costs, c_plain)
        for v in model.trainable_variables:
            assert torch.isfinite(v).all() and not torch.equal(v, before[v.ams_name]), v.ams_name


def test_hip_graph_finetuning_step_refreshes_kmeans_seeds():
    """--hip_graph on a recipe whose k-means seeds come from the host RNG (front_*_finetuning): the captured kernels read a
    persistent index buffer that a pre-replay hook re-fills, so replayed steps see fresh seeds and the run equals the eager one
    when the global numpy RNG is seeded identically."""
    import os
    import tempfile
    from ams_hip import testing
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Trainer, Front_Separator_Finetuning_Trainer
    os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))

    def run(graph):
        tmp = tempfile.mkdtemp(prefix='ams_cgft_')
        B, S, L, W, N, hop = 3, 2, 1024, 64, 16, 16
        import utils.ops
        utils.ops.rng.seed(42)
        folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                       batch_size=B, nb_speakers=S)
        a = dict(params)
        a.update(testing.SEPARATOR_DEFAULTS)
        a.update(layer_size=12, nb_layers=2, embedding_size=8, model_folder=folder, model_previous=None, pretraining=False,
                 learning_rate=1e-3)
        a.pop('type')
        tr0 = Front_Separator_Trainer(DPCL, 'front_DPCL', **dict(a))
        tr0.prepare()
        with tr0.graph.as_default():
            tr0.model.create_saver()
            tr0.model.save(0)
            folder2 = tr0.model._dir()
        a.update(model_folder=folder2, nb_tries=2, nb_steps=3, beta_kmeans=4.0, with_silence=True, threshold=2.0, end_assign=True,
                 loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4, hip_graph=graph, no_summaries=True)
        tr = Front_Separator_Finetuning_Trainer(DPCL, 'front_L41_finetuning', **a)
        dist, tfds = tr.prepare()
        np.random.seed(7)
        costs = []
        with tr.graph.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
            for i in range(6):
                costs.append(float(tr.model.train(feed, i)))
        torch.cuda.synchronize()
        return costs

    c_e, c_g = run(False), run(True)
    assert np.allclose(c_e, c_g, rtol=1e-5), (c_e, c_g)
    assert len(set(np.round(c_g, 6))) > 2                     # the replayed steps are not frozen on one set of seeds / one batch


def test_training_from_tfrecord_files(tmp_path):
    """--dataset <name> + AMS_DATA_DIR: the reference's {split}_{M,F}.tfrecords drive the same trainer."""
    import os
    import tempfile
    from ams_hip import testing
    from data import tfrecord
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Trainer
    os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))
    rng = np.random.RandomState(3)
    L, B, S = 1024, 3, 2
    for split in ('train', 'valid', 'test', 'test_other'):
        for g_, base in (('M', 0), ('F', 100)):
            tfrecord.write_audio_records(str(tmp_path / ('%s_%s.tfrecords' % (split, g_))),
                                         [((0.05 * rng.randn(rng.randint(L + 1, 4 * L))).astype(np.float32), base + i) for i in range(6)])
    os.environ['AMS_DATA_DIR'] = str(tmp_path)
    try:
        folder, params = testing.make_pretrained_adapt(os.path.join(str(tmp_path), 'pre'), window_size=64, filters=16, hop_size=16,
                                                       chunk_size=L, batch_size=B, nb_speakers=S)
        a = dict(params)
        a.update(testing.SEPARATOR_DEFAULTS)
        a.update(layer_size=12, nb_layers=2, embedding_size=8, model_folder=folder, model_previous=None, pretraining=False,
                 learning_rate=1e-3, dataset='h5py_files/train-clean-100-8-s.h5', no_summaries=True)
        a.pop('type')
        tr = Front_Separator_Trainer(DPCL, 'front_DPCL', **a)
        dist, tfds = tr.prepare()
        n_train = tfds.length(tfds.TRAIN)
        assert n_train >= 1
        tfds.initialize(tfds.TRAIN)
        with tr.graph.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
            costs = [float(tr.model.train(feed, i)) for i in range(n_train)]
            run = tr.model.last_run
            xm, xn, I = (tr.model.x_mix.value(run), tr.model.x_non_mix.value(run), tr.model.I.value(run))
        assert all(np.isfinite(costs))
        assert xm.shape[1] == L and xn.shape[1:] == (S, L) and torch.allclose(xm, xn.sum(1), atol=1e-6)
        assert int(I[:, 0].max()) < 100 <= int(I[:, 1].min())           # speaker 0 from the M file, speaker 1 from the F file
        with pytest.raises(StopIteration):                               # end of epoch, like tf.errors.OutOfRangeError
            with tr.graph.as_default():
                for i in range(1000):
                    tr.model.train(feed, i)
    finally:
        os.environ.pop('AMS_DATA_DIR', None)


@pytest.mark.gpu
def test_entry_build_then_smoke_in_one_process():
    """__graft_entry__.build() followed by smoke() in ONE fresh process: the HIP libraries must not be loaded ahead of torch's own
    HIP runtime (two runtimes in a process make every launch fail with hipErrorNoDevice)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = 'import __graft_entry__ as g; g.build(); g.smoke(); print("ENTRY_OK")'
    out = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ENTRY_OK' in out.stdout, out.stderr[-2000:]
