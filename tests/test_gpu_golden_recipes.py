"""GPU: the HIP recipes against the COMMITTED golden vectors of round 4 (tests/golden/make_golden.py::more) -- the trajectory of the
headline recipe and single steps of front_L41 (S = 3), front_DPCL_finetuning, STFT_L41_enhance and path-B pre-training -- and the
default product arithmetic (fp16x3 where bounds are at hand) against the native f32 MFMA products over many steps.

The fixtures hold inputs, weights, k-means seeds and the float64 oracle's outputs; each test builds the recipe through the reference
API mirror (utils/trainer.py), INJECTS the fixture's weights and batch, and compares.  Reference: utils/trainer.py:264-390 (loop),
models/network.py:228-232 (train), models/L41.py:150-178, models/network.py:610-724, models/adapt.py:115-117,210-243."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from tests.test_golden import load, rel, _split                                    # noqa: E402
from tests.test_gpu_recipes import base_args                                          # noqa: E402

os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))


def dev(a, dt=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()


def inject(tr, P):
    g = tr.graph
    with g.as_default():
        for n, v in g.variables.items():
            if n in P:
                v.data.copy_(dev(P[n]).view(v.shape))


def step_on(tr, tfds, L, inp, update=True):
    """One training step of tr.model on the FIXTURE's batch (what Network.train does, with the input nodes pre-filled)."""
    from ams_hip import functional as F
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        run = model._feeds(feed, True)
        run.cache[id(model.x_mix)] = dev(inp['x_mix'])
        run.cache[id(model.x_non_mix)] = dev(inp['x_non_mix'])
        if 'I' in inp:
            run.cache[id(model.I)] = dev(inp['I'], np.int32)
        opt = model.optimize
        opt.zero_grad()
        cost = model.cost_model.value(run)
        model._backward(cost)
        F.OVERLAP.join()
        grads = {v.ams_name: v.grad.detach().cpu().numpy().copy() for v in model.trainable_variables}
        if update:
            opt.step()
        torch.cuda.synchronize()
    return float(cost.detach().reshape(-1)[0]), grads, run


def check_grads(grads, G, tol=1e-3):
    assert sorted(grads) == sorted(G), (sorted(grads), sorted(G))
    scale = max(float(np.abs(G[n]).max()) for n in G)
    for n in G:
        err = float(np.abs(grads[n] - G[n]).max() / max(np.abs(G[n]).max(), 1e-2 * scale))
        assert err < tol, (n, err)


def test_front_dpcl_trajectory_matches_the_golden():
    """5 AMSGrad steps on one fixed batch: every cost within 1e-3 of the float64 oracle's, the final weights too."""
    from tests.smoke_step import build_front_dpcl
    d, t = load('front_dpcl_step.npz'), load('front_dpcl_traj.npz')
    B, S, L, W, N, hop, LS, NL, E = [int(v) for v in d['cfg']]
    tr, tfds = build_front_dpcl(tempfile.mkdtemp(prefix='ams_traj_'), B=B, L=L, W=W, N=N, hop=hop, layer_size=LS, nb_layers=NL, E=E, S=S,
                                no_summaries=True)
    inject(tr, {k[2:]: v for k, v in d.items() if k.startswith('P/')})
    inp = {'x_mix': d['x_mix'], 'x_non_mix': d['x_non_mix']}
    costs = [step_on(tr, tfds, L, inp)[0] for _ in range(5)]
    for i, (c, c_ref) in enumerate(zip(costs, t['costs'])):
        assert abs(c - c_ref) < 1e-3 * abs(c_ref), (i, costs, t['costs'])
    for n, v in tr.graph.variables.items():
        if n.startswith('prediction/'):
            # after 5 steps of lr 1e-3 a weight has moved by at most ~5e-3: compare the MOVEMENT, not the weight
            moved = t['P5/' + n] - d['P/' + n]
            err = np.abs((v.detach().cpu().numpy() - d['P/' + n]) - moved).max() / max(np.abs(moved).max(), 1e-30)
            assert err < 2e-2, (n, err)


def test_front_l41_three_speakers_matches_the_golden():
    from ams_hip import testing
    from models.L41 import L41Model
    from utils.trainer import Front_Separator_Trainer
    d = load('front_l41_step.npz')
    cfg, P, inp, G = _split(d)
    tmp = tempfile.mkdtemp(prefix='ams_gl41_')
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=cfg['W'], filters=cfg['N'], hop_size=cfg['hop'],
                                                   chunk_size=cfg['L'], batch_size=cfg['B'], nb_speakers=cfg['S'])
    a = base_args(**params)
    a.update(layer_size=cfg['LS'], nb_layers=cfg['NL'], embedding_size=cfg['E'], model_folder=folder, model_previous=None, pretraining=False,
             no_normalize=True, learning_rate=1e-3, tot_speakers=cfg['NSPK'])
    a.pop('type')
    tr = Front_Separator_Trainer(L41Model, 'front_L41', **a)
    dist, tfds = tr.prepare()
    inject(tr, P)
    cost, grads, run = step_on(tr, tfds, cfg['L'], inp, update=False)
    assert abs(cost - float(d['cost'])) < 1e-4 * abs(float(d['cost'])), (cost, float(d['cost']))
    with tr.graph.as_default(), torch.no_grad():
        V = tr.model.sepNet.prediction.value(run).detach().cpu().numpy()
        Y = tr.model.sepNet.y.value(run).cpu().numpy()
    assert rel(V.reshape(d['V'].shape), d['V']) < 1e-3                                 # north_star: embeddings within 1e-3
    assert np.array_equal(Y.reshape(d['Y'].shape), d['Y'])
    check_grads(grads, G)


def test_front_dpcl_finetuning_matches_the_golden():
    """Cost and separated waveforms vs the oracle; gradients vs the stored central differences of the float64 oracle."""
    from ams_hip import testing
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Finetuning_Trainer
    d = load('front_dpcl_finetuning_step.npz')
    cfg, P, inp, _ = _split(d)
    tmp = tempfile.mkdtemp(prefix='ams_gft_')
    params = dict(testing.ADAPT_DEFAULTS)
    params.update(testing.SEPARATOR_DEFAULTS)
    params.update(window_size=cfg['W'], hop_size=cfg['hop'], chunk_size=cfg['L'], batch_size=cfg['B'], nb_speakers=cfg['S'], layer_size=cfg['LS'],
                  nb_layers=cfg['NL'], embedding_size=cfg['E'], type='front_DPCL', pretraining=False, filters=cfg['N'])
    folder = testing.write_checkpoint(os.path.join(tmp, 'ckpt'), {k: v.astype(np.float32) for k, v in P.items()}, params)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=cfg['tries'], nb_steps=cfg['steps'], beta_kmeans=cfg['beta'], with_silence=True, threshold=2.0,
             end_assign=True, kmeans_init_indices=inp['idx'], loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4, pretraining=False)
    a.pop('type')
    tr = Front_Separator_Finetuning_Trainer(DPCL, 'front_L41_finetuning', **a)
    dist, tfds = tr.prepare()
    inject(tr, P)
    cost, grads, run = step_on(tr, tfds, cfg['L'], inp, update=False)
    assert abs(cost - float(d['cost'])) < 1e-3 * abs(float(d['cost'])), (cost, float(d['cost']))
    with tr.graph.as_default(), torch.no_grad():
        back = tr.model.back.value(run).detach().cpu().numpy()
    assert np.linalg.norm(back.reshape(d['back'].shape) - d['back']) / np.linalg.norm(d['back']) < 1e-3
    gmax = {n: float(np.abs(g).max()) for n, g in grads.items()}
    for name, fi, fd in zip(d['probe_names'], d['probe_index'], d['probe_fd']):
        got = float(grads[str(name)].reshape(-1)[int(fi)])
        assert abs(got - fd) < 2e-2 * max(abs(fd), 1e-3 * gmax[str(name)]), (str(name), int(fi), got, float(fd))


def test_stft_l41_enhance_matches_the_golden():
    from ams_hip import testing
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_enhance_Trainer
    d = load('stft_l41_enhance_step.npz')
    cfg, P, inp, G = _split(d)
    tmp = tempfile.mkdtemp(prefix='ams_genh_')
    params = dict(testing.ADAPT_DEFAULTS)
    for k in ('filters', 'max_pool'):
        params.pop(k)
    params.update(testing.SEPARATOR_DEFAULTS)
    params.update(window_size=cfg['W'], hop_size=cfg['hop'], chunk_size=cfg['L'], batch_size=cfg['B'], nb_speakers=cfg['S'], layer_size=cfg['LS'],
                  nb_layers=cfg['NL'], embedding_size=cfg['E'], type='STFT_DPCL', pretraining=False)
    sep_only = {k: v.astype(np.float32) for k, v in P.items() if not k.startswith('enhance/')}
    folder = testing.write_checkpoint(os.path.join(tmp, 'ckpt'), sep_only, params)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=cfg['tries'], nb_steps=cfg['steps'], end_assign=True, kmeans_init_indices=inp['idx'],
             layer_size_enhance=cfg['LSE'], nb_layers_enhance=cfg['NLE'], nonlinearity='softmax', learning_rate=1e-3, pretraining=False,
             tot_speakers=cfg['NSPK'])
    a.pop('type')
    tr = STFT_Separator_enhance_Trainer(L41Model, 'STFT_L41_enhance', **a)
    dist, tfds = tr.prepare()
    inject(tr, P)
    cost, grads, run = step_on(tr, tfds, cfg['L'], inp, update=False)
    assert abs(cost - float(d['cost'])) < 5e-4 * abs(float(d['cost'])), (cost, float(d['cost']))
    check_grads(grads, G)


def test_pretraining_with_max_pool_matches_the_golden():
    from utils.trainer import Adapt_Pretrainer
    d = load('pretraining_maxpool_step.npz')
    cfg, P, inp, G = _split(d)
    a = base_args(batch_size=cfg['B'], nb_speakers=cfg['S'], chunk_size=cfg['L'], window_size=cfg['W'], filters=cfg['N'], hop_size=cfg['hop'],
                  max_pool=cfg['Pool'], with_max_pool=True, loss='l2', separation='perfect', overlap_coef=0.0, optimizer='Adam',
                  learning_rate=1e-3, pretraining=True)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    inject(tr, P)
    cost, grads, run = step_on(tr, tfds, cfg['L'], inp, update=False)
    assert abs(cost - float(d['cost'])) < 1e-4 * abs(float(d['cost'])), (cost, float(d['cost']))
    with tr.graph.as_default(), torch.no_grad():
        back = tr.model.back.value(run).detach().cpu().numpy()
    assert rel(back.reshape(d['back'].shape), d['back']) < 1e-3
    check_grads(grads, G)


# ---- the default arithmetic against the native f32 products, step after step ----------------------------------
def _run_costs(n_steps, arith, f16x3, monkeypatch, **shape):
    from tests.smoke_step import build_front_dpcl
    from ams_hip import ops
    from ams_hip._lib import load as libload
    lib = libload()
    before = lib.ams_gemm_get_arith()
    lib.ams_gemm_set_arith(arith)
    monkeypatch.setattr(ops, 'F16X3', f16x3)
    try:
        tr, tfds = build_front_dpcl(tempfile.mkdtemp(prefix='ams_arith_'), no_summaries=True, **shape)
        g, model = tr.graph, tr.model
        costs = []
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: shape['L']}
            for i in range(n_steps):
                costs.append(float(model.train(feed, i)))
        ops.raise_on_ring_errors()
        return np.array(costs)
    finally:
        lib.ams_gemm_set_arith(before)


@pytest.mark.parametrize('n_steps,shape', [
    (200, dict(B=4, L=2048, W=64, N=16, hop=16, layer_size=24, nb_layers=2, E=8)),
    (20, dict(B=64, L=20480, W=1024, N=256, hop=256, layer_size=600, nb_layers=3, E=40, hip_graph=True)),
])
def test_default_arithmetic_tracks_native_f32_over_many_steps(n_steps, shape, monkeypatch):
    """The same model, data and optimizer under (a) the default products (fp16x3 wherever bounds are at hand, bf16x6 elsewhere),
    (b) bf16x6 everywhere (an EXACT split of the f32 operands), (c) the native f32 MFMA products: the cost of every step of (a) and of
    (b) within 1e-3 of (c).  200 steps at reduced size (4 batches in rotation), 20 at the benchmark shape under hipGraph replay."""
    c_def = _run_costs(n_steps, 1, True, monkeypatch, **shape)
    c_x6 = _run_costs(n_steps, 1, False, monkeypatch, **shape)
    c_f32 = _run_costs(n_steps, 0, True, monkeypatch, **shape)
    d_def = np.abs(c_def - c_f32) / np.abs(c_f32)
    d_x6 = np.abs(c_x6 - c_f32) / np.abs(c_f32)
    print('default vs native f32: max rel cost difference %.2e (step %d); bf16x6 vs native: %.2e; first %.6f last %.6f'
          % (d_def.max(), int(d_def.argmax()), d_x6.max(), c_f32[0], c_f32[-1]))
    assert np.isfinite(c_def).all() and np.isfinite(c_f32).all()
    if n_steps >= 100:
        assert c_f32[-20:].mean() < c_f32[:20].mean()    # the trajectory trains
    assert d_def.max() < 1e-3 and d_x6.max() < 1e-3, (d_def.max(), d_x6.max())
