"""GPU: a recurrence ring that cannot get all its workgroups resident must be SAFE, not only loud (csrc/lstm_ring.hip needs its whole
grid co-resident; reference utils/ops.py:358-383 has no such failure mode -- dynamic_rnn is a loop of independent launches).

The mechanism: every ring launch gets the device's sticky error word (ops.ring_error_word); a launch that abandons a bounded wait
sets it; the fused optimizers take the same word as their guard and leave weights and slots untouched; the host, at the sync it
makes anyway, repeats that batch on the per-step kernels (models/network.py::retrain_last / _eval_guarded)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


def test_optimizers_skip_when_the_guard_word_is_set(ops):
    torch.manual_seed(0)
    n = 5000
    for kind in ('amsgrad', 'rmsprop', 'momentum'):
        p, g = torch.randn(n, device='cuda'), torch.randn(n, device='cuda')
        slots = [torch.rand(n, device='cuda') for _ in range(3)]
        p0, s0 = p.clone(), [s.clone() for s in slots]
        word = torch.ones(1, dtype=torch.int32, device='cuda')

        def step(guard):
            if kind == 'amsgrad':
                ops.opt_amsgrad(p, g, slots[0], slots[1], slots[2], 1e-2, 0.9, 0.99, 1e-3, guard=guard)
            elif kind == 'rmsprop':
                ops.opt_rmsprop(p, g, slots[0], 1e-2, guard=guard)
            else:
                ops.opt_momentum(p, g, slots[0], 1e-2, guard=guard)
        step(word)
        torch.cuda.synchronize()
        assert torch.equal(p, p0) and all(torch.equal(a, b) for a, b in zip(slots, s0)), kind
        word.zero_()
        step(word)
        torch.cuda.synchronize()
        assert not torch.equal(p, p0) and not torch.equal(slots[0], s0[0]), kind
        p1 = p.clone()
        step(False)                                             # no guard at all
        torch.cuda.synchronize()
        assert not torch.equal(p, p1), kind


@pytest.mark.parametrize('hip_graph', [False, True])
def test_a_flagged_training_step_is_repeated_on_the_step_kernels(ops, hip_graph):
    """Two identical trainers.  In one, the sticky error word is raised while a training step is in flight (what a ring launch that
    gives up does): the optimizer must skip that update, Trainer's check must repeat the batch with the ring off, and both trainers
    must end on the same weights (ring and per-step kernels agree to ~1e-6)."""
    from tests.smoke_step import build_front_dpcl
    from models.network import Network
    cfg = dict(B=4, L=2048, W=64, N=16, hop=16, layer_size=24, nb_layers=2, E=8, no_summaries=True, hip_graph=hip_graph)
    runs = []
    for flagged in (False, True):
        torch.manual_seed(1)
        tmp = tempfile.mkdtemp(prefix='ams_guard_')
        trainer, tfds = build_front_dpcl(tmp, **cfg)
        g, model = trainer.graph, trainer.model
        before = Network.ring_fallbacks
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: cfg['L']}
            tfds.initialize(tfds.TRAIN)
            costs = []
            for step in range(5):
                snap = {v.ams_name: v.detach().clone() for v in model.trainable_variables}
                if flagged and step == 3:
                    ops.ring_error_word().fill_(1)              # "a ring launch of this step gave up"
                c = float(model.train(feed, step))
                if ops.ring_error_pending():                    # utils/trainer.py::Trainer.train does exactly this
                    for v in model.trainable_variables:         # the guarded optimizer left everything as it was
                        assert torch.equal(v.detach(), snap[v.ams_name]), v.ams_name
                    c = float(model.retrain_last(step))
                costs.append(c)
            torch.cuda.synchronize()
        assert not ops.ring_error_pending()
        assert Network.ring_fallbacks - before == (1 if flagged else 0)
        runs.append((costs, {v.ams_name: v.detach().cpu().numpy().copy() for v in model.trainable_variables}))
    (c0, p0), (c1, p1) = runs
    assert np.allclose(c0, c1, rtol=2e-5), (c0, c1)
    for n in p0:
        d = np.abs(p0[n] - p1[n]).max()
        assert d <= 1e-5 * max(1.0, np.abs(p0[n]).max()), (n, d)


def test_flagged_evaluation_batches_are_recomputed(ops):
    """valid_batch / infer / get_embeddings with the word raised DURING the evaluation: the batch is recomputed on the per-step kernels
    and the word cleared.  Raised BEFORE it (an unhandled training step): the batch is still evaluated on the per-step
    kernels but the word stays up for its owner (Trainer.train checks it right after the step).  The value equals the unflagged one."""
    from tests.smoke_step import build_front_dpcl
    from models.network import Network
    tmp = tempfile.mkdtemp(prefix='ams_guard_eval_')
    trainer, tfds = build_front_dpcl(tmp, B=4, L=2048, W=64, N=16, hop=16, layer_size=24, nb_layers=2, E=8, no_summaries=True)
    g, model = trainer.graph, trainer.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.VALID), tfds.chunk_size: 2048}
        vals = []
        for flagged in (None, 'during', 'before'):
            tfds.initialize(tfds.VALID)
            before = Network.ring_fallbacks
            node, orig, calls = model.cost_model, model.cost_model.fn, []
            if flagged == 'before':
                ops.ring_error_word().fill_(1)
            if flagged == 'during':
                def fn(run):
                    out = orig(run)
                    if not calls:
                        ops.ring_error_word().fill_(1)          # "a ring launch of this evaluation gave up"
                    calls.append(1)
                    return out
                node.fn = fn
            try:
                vals.append(model.valid_batch(feed, 0))
            finally:
                node.fn = orig
            assert Network.ring_fallbacks - before == (1 if flagged else 0)
            assert ops.ring_error_pending() == (flagged == 'before')
            ops.ring_errors_clear()
    assert abs(vals[0] - vals[1]) <= 2e-5 * abs(vals[0]) and abs(vals[0] - vals[2]) <= 2e-5 * abs(vals[0]), vals


def test_results_survive_foreign_uncapped_products_on_a_third_stream(ops):
    """A foreign stream that keeps every CU's registers full (uncapped 8-wave bf16x6 products, one workgroup per CU: no ring
    workgroup fits beside them) while a BLSTM stack runs: whatever happens to the rings -- they may get in, or give up and be
    repeated on the per-step kernels -- the embeddings must be the idle run's (1e-5) and no error may be left pending."""
    from tests.smoke_step import build_front_dpcl
    from models.network import Network
    tmp = tempfile.mkdtemp(prefix='ams_guard_load_')
    trainer, tfds = build_front_dpcl(tmp, B=64, L=4096, W=64, N=64, hop=64, layer_size=600, nb_layers=2, E=8, no_summaries=True)
    g, model = trainer.graph, trainer.model
    a, b = torch.randn(4096, 2048, device='cuda'), torch.randn(2048, 4096, device='cuda')
    side = torch.cuda.Stream()
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.VALID), tfds.chunk_size: 4096}
        tfds.initialize(tfds.VALID)
        emb = lambda: model._eval_guarded(feed, lambda run: model.sepNet.prediction.value(run))      # noqa: E731
        ref = emb().clone()
        torch.cuda.synchronize()
        before = Network.ring_fallbacks
        rng = np.random.RandomState(4)
        for rep in range(6):
            tfds.initialize(tfds.VALID)
            with torch.cuda.stream(side):
                for _ in range(int(rng.randint(4, 24))):
                    ops.gemm(a, b)                              # uncapped: 8 waves x ~230 VGPRs per CU
            got = emb()
            torch.cuda.synchronize()
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err < 1e-5, (rep, err, Network.ring_fallbacks - before)
        print('foreign load: %d of 6 batches were repeated on the per-step kernels' % (Network.ring_fallbacks - before))
    assert not ops.ring_error_pending()


def test_a_device_too_small_for_the_ring_selects_the_step_kernels():
    """ams_blstm_ring_sync_bytes consults the device: with fewer CUs than the ring's grid needs (a CPX partition, a CU mask) it
    returns 0 and callers use ams_blstm_recurrent_*.  AMS_LSTM_RING_CUS (testing aid, read once) pretends such a device."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from ams_hip._lib import load\nimport torch\ntorch.zeros(1).cuda()\nl = load()\n"
            "print(int(l.ams_blstm_ring_sync_bytes(64, 300, 0) != 0), int(l.ams_blstm_ring_sync_bytes(128, 300, 1) != 0))\n"
            % (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')))
    full = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    small = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMS_LSTM_RING_CUS='64'))
    mid = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMS_LSTM_RING_CUS='128'))
    assert full.stdout.split() == ['1', '1'], full.stderr[-500:]
    assert small.stdout.split() == ['0', '0'], small.stderr[-500:]            # 200 / 400 workgroups do not fit 2 x 64
    assert mid.stdout.split() == ['1', '0'], mid.stderr[-500:]                # 200 fit 2 x 128, 400 do not
