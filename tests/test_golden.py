"""Golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py from the float64 oracle):
CPU: the oracle still reproduces them; GPU: the HIP path matches them."""
import os

import numpy as np
import pytest

from oracle import step as ostep, recipes as orec, kmeans as okm, stft as ostft

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    d = np.load(os.path.join(G, name))
    return {k: d[k] for k in d.files}


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def test_oracle_reproduces_front_dpcl_and_pretraining_goldens():
    d = load('front_dpcl_step.npz')
    B, S, L, W, N, hop, LS, NL, E = [int(v) for v in d['cfg']]
    P = {k[2:]: v for k, v in d.items() if k.startswith('P/')}
    cost, grads, V, Y = ostep.front_dpcl_loss(d['x_mix'], d['x_non_mix'], P, hop, NL, E)
    assert abs(cost - d['cost']) < 1e-12 and rel(V, d['V']) < 1e-12 and np.array_equal(Y, d['Y'])
    for k, v in grads.items():
        assert rel(v, d['G/' + k]) < 1e-10, k
    p = load('pretraining_step.npz')
    c, g, back = orec.pretrain_loss(d['x_mix'], d['x_non_mix'], P, hop, 'sdr+l2', 'mask', 1.0)
    assert abs(c - p['cost']) < 1e-12 and rel(back, p['back']) < 1e-12


def test_oracle_reproduces_kmeans_and_stft_goldens():
    k = load('kmeans_hard.npz')
    C, tries, iters = [int(v) for v in k['cfg']]
    cent, lab, best = okm.kmeans(k['X'], k['idx'], C, tries, iters, beta=None, notsilent=k['w'], assign_at_end=True)
    assert np.array_equal(lab, k['labels']) and np.array_equal(cent, k['centroids']) and np.array_equal(best, k['best'])
    s = load('stft.npz')
    st = ostft.stft(s['x'], 256, 128)
    assert rel(np.abs(st), s['mag']) < 1e-12 and rel(ostft.istft(np.abs(st), np.angle(st), 256, 128), s['rec']) < 1e-12


@pytest.mark.gpu
def test_hip_matches_front_dpcl_golden():
    import torch
    from ams_hip import functional as F, ops
    d = load('front_dpcl_step.npz')
    B, S, L, W, N, hop, LS, NL, E = [int(v) for v in d['cfg']]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    P = {k[2:]: dev(v).requires_grad_(k[2:].startswith('prediction/')) for k, v in d.items() if k.startswith('P/')}
    x = dev(np.concatenate([d['x_mix'], d['x_non_mix'].reshape(B * S, L)], 0))
    y = F.front_conv(x, F.front_filter(P['front/window/w'], P['front/bases/bases']), hop)
    Y = ops.make_masks(y[B:].contiguous(), B, S, 1.0, 0.0, True)
    h = y[:B].contiguous()
    for i in range(NL):
        kf, bf, kb, bb = ostep.lstm_names('prediction', i)
        h = F.blstm(h, P[kf], P[bf], P[kb], P[bb])
    u = F.dense(h, P['prediction/W'], P['prediction/b'])
    V, inv = F.l2norm_keep(u, E)
    out = F.dpcl_loss_from_u(u, V, inv, Y)
    out[0].backward()
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(d['cost'])) < 1e-4 * abs(float(d['cost']))
    assert rel(V.detach().cpu().numpy(), d['V']) < 1e-3                 # north_star: embeddings within 1e-3 rel fp32
    assert np.array_equal(Y.cpu().numpy().reshape(d['Y'].shape), d['Y'])
    for k, v in P.items():
        if k.startswith('prediction/'):
            assert rel(v.grad.cpu().numpy(), d['G/' + k]) < 1e-3, k


@pytest.mark.gpu
def test_hip_matches_kmeans_and_stft_goldens():
    import torch
    from ams_hip import functional as F
    k = load('kmeans_hard.npz')
    C, tries, iters = [int(v) for v in k['cfg']]
    cent, lab, best = F.kmeans(torch.from_numpy(k['X']).cuda(), torch.from_numpy(k['idx']).cuda(), C, tries, iters, None,
                               torch.from_numpy(k['w']).cuda(), True)
    torch.cuda.synchronize()
    assert np.array_equal(lab.cpu().numpy(), k['labels'])               # north_star: cluster assignment bit-exact
    assert np.array_equal(cent.cpu().numpy(), k['centroids']) and np.array_equal(best.cpu().numpy(), k['best'])
    s = load('stft.npz')
    mag, ph = F.stft_mag_phase(torch.from_numpy(s['x'].astype(np.float32)).cuda(), 256, 128)
    assert rel(mag.cpu().numpy(), s['mag']) < 1e-4
    rec = F.istft(mag, ph, 256, 128, 1)
    assert rel(rec.cpu().numpy(), s['rec']) < 1e-4


@pytest.mark.gpu
def test_hip_matches_pretraining_golden():
    """HIP pre-training step (Adapt_Pretrainer, loss sdr+l2, separation mask, overlap_coef 1.0 -- the README.md:23 recipe) on the
    inputs and filterbank stored in front_dpcl_step.npz vs pretraining_step.npz: cost, reconstructed waveforms, all 4 gradients."""
    import os
    import tempfile
    import torch
    from ams_hip import testing, functional as F
    from utils.trainer import Adapt_Pretrainer
    os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))
    d, gp = load('front_dpcl_step.npz'), load('pretraining_step.npz')
    B, S, L, W, N, hop, LS, NL, E = [int(v) for v in d['cfg']]
    a = dict(testing.ADAPT_DEFAULTS)
    a.update(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, loss='sdr+l2', separation='mask',
             overlap_coef=1.0, learning_rate=1e-3, pretraining=True)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    names = ('front/window/w', 'front/bases/bases', 'back/window/value', 'back/bases/value')
    with g.as_default():
        for n in names:
            v = g.variables[n]
            v.data.copy_(torch.from_numpy(d['P/' + n].astype(np.float32)).to(v.device))
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        run = model._feeds(feed, True)
        dev = lambda x, dt=np.float32: torch.from_numpy(np.ascontiguousarray(x, dtype=dt)).cuda()       # noqa: E731
        for node, t in zip((model.x_mix, model.x_non_mix), (dev(d['x_mix']), dev(d['x_non_mix']))):
            run.cache[id(node)] = t
        opt = model.optimize
        opt.zero_grad()
        cost = model.cost_model.value(run)
        cost.reshape(-1)[0].backward()
        F.OVERLAP.join()
        torch.cuda.synchronize()
        back = model.back.value(run).detach().cpu().numpy()
        grads = {v.ams_name: v.grad.detach().cpu().numpy() for v in model.trainable_variables}
    c = float(cost.detach().reshape(-1)[0])
    assert abs(c - float(gp['cost'])) < 1e-4 * abs(float(gp['cost'])), (c, float(gp['cost']))
    assert rel(back.reshape(gp['back'].shape), gp['back']) < 1e-3
    assert sorted(grads) == sorted(names)
    for n in names:
        assert rel(grads[n], gp['G/' + n]) < 1e-3, (n, rel(grads[n], gp['G/' + n]))


# ---------------------------------------------------------------------------------------------------------------------------
# Round 4: trajectory of the headline recipe + single steps of the recipes the round-1 fixtures do not touch (make_golden.py::more)
# ---------------------------------------------------------------------------------------------------------------------------
def _split(d):
    cfg = {k[4:]: (float(d[k]) if d[k].dtype.kind == 'f' else int(d[k])) for k in d if k.startswith('cfg/')}
    P = {k[2:]: v for k, v in d.items() if k.startswith('P/')}
    inp = {k[3:]: v for k, v in d.items() if k.startswith('in/')}
    G = {k[2:]: v for k, v in d.items() if k.startswith('G/')}
    return cfg, P, inp, G


def test_oracle_reproduces_the_trajectory_golden():
    """5 AMSGrad steps of front_DPCL on one fixed batch (SURVEY 8c harness row; reference utils/trainer.py:264-390,
    models/network.py:228-232): cost of every step and the weights after the fifth."""
    from oracle import optim as ooptim
    d, t = load('front_dpcl_step.npz'), load('front_dpcl_traj.npz')
    B, S, L, W, N, hop, LS, NL, E = [int(v) for v in d['cfg']]
    P = {k[2:]: v.copy() for k, v in d.items() if k.startswith('P/')}
    names = sorted(k for k in P if k.startswith('prediction/'))
    opt = ooptim.AMSGrad(1e-3)
    for i in range(5):
        c, g, _, _ = ostep.front_dpcl_loss(d['x_mix'], d['x_non_mix'], P, hop, NL, E)
        assert abs(c - t['costs'][i]) < 1e-10 * abs(t['costs'][i]), i
        opt.apply([P[n] for n in names], [g[n] for n in names])
    for n in names:
        assert rel(P[n], t['P5/' + n]) < 1e-10, n
    assert t['costs'][4] < 0.5 * t['costs'][0]                     # the fixture records a trajectory that actually moves


def test_oracle_reproduces_the_recipe_goldens():
    cfg, P, inp, G = _split(load('front_l41_step.npz'))
    c, g, V, Y = ostep.front_l41_loss(inp['x_mix'], inp['x_non_mix'], inp['I'], P, cfg['hop'], cfg['NL'], cfg['E'], True)
    d = load('front_l41_step.npz')
    assert abs(c - d['cost']) < 1e-12 and rel(V, d['V']) < 1e-12 and np.array_equal(Y, d['Y'])
    assert sorted(g) == sorted(G) and all(rel(g[k], G[k]) < 1e-10 for k in g)

    d = load('front_dpcl_finetuning_step.npz')
    cfg, P, inp, _ = _split(d)
    c, back = orec.front_finetune_cost(inp['x_mix'], inp['x_non_mix'], P, cfg['hop'], cfg['NL'], cfg['E'], inp['idx'], cfg['tries'],
                                       cfg['steps'], cfg['beta'], True, 2.0, True, 'sdr+l2')
    assert abs(c - d['cost']) < 1e-12 * abs(d['cost']) and rel(back, d['back']) < 1e-12

    d = load('stft_l41_enhance_step.npz')
    cfg, P, inp, G = _split(d)
    c, g = orec.stft_enhance_loss(inp['x_mix'], inp['x_non_mix'], P, cfg['W'], cfg['hop'], cfg['NL'], cfg['E'], cfg['NLE'], inp['idx'],
                                  cfg['tries'], cfg['steps'], nonlinearity='softmax')
    assert abs(c - d['cost']) < 1e-12 * abs(d['cost']) and all(rel(g[k], G[k]) < 1e-10 for k in g)

    d = load('pretraining_maxpool_step.npz')
    cfg, P, inp, G = _split(d)
    c, g, back, am = orec.pretrain_loss_maxpool(inp['x_mix'], inp['x_non_mix'], P, cfg['Pool'], cfg['hop'], 'l2', 'perfect')
    assert abs(c - d['cost']) < 1e-12 * abs(d['cost']) and rel(back, d['back']) < 1e-12 and np.array_equal(am, d['argmax'])
    assert all(rel(g[k], G[k]) < 1e-10 for k in g)
