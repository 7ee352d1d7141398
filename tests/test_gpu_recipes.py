"""GPU: whole recipes driven through the reference-API mirror (utils/trainer.py classes) vs the oracle."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import step as ostep, recipes as orec, optim as ooptim

os.environ.setdefault('AMS_LOG_DIR', tempfile.mkdtemp(prefix='ams_log_'))

# Separated waveforms vs the float64 oracle, relative L2.  north_star: masks within 1e-3.  Hard k-means labels are bit-exact given
# the same float32 embeddings (test_gpu_kernels2.py::test_kmeans_hard_bit_exact); here the oracle clusters its OWN float64
# embeddings, the seeds are injected, and no label flips at these sizes -- so the fp32 pipeline's round-off is all that is left.
INFER_TOL = 1e-3


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def snapshot(g):
    return {n: v.detach().cpu().numpy().astype(np.float64) for n, v in g.variables.items()}


def one_train_step(trainer, tfds, L):
    g, model = trainer.graph, trainer.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        P = snapshot(g)
        cost = float(model.train(feed, 0))
        run = model.last_run
        xm = model.x_mix.value(run).cpu().numpy().astype(np.float64)
        xn = model.x_non_mix.value(run).cpu().numpy().astype(np.float64)
        I = model.I.value(run).cpu().numpy()
        grads = {v.ams_name: v.grad.detach().cpu().numpy() for v in model.trainable_variables}
        P_new = {v.ams_name: v.detach().cpu().numpy() for v in model.trainable_variables}
    return P, cost, xm, xn, I, grads, P_new


def check_step(cost, c_ref, grads, g_ref, P, P_new, opt, tol=2e-4):
    errs = {'cost': abs(cost - c_ref) / max(abs(c_ref), 1e-30)}
    names = sorted(g_ref)
    assert sorted(grads) == names
    gscale = max(float(np.abs(g_ref[n]).max()) for n in names)
    for n in names:
        # a gradient that is analytically zero (e.g. a bias under a softmax over speakers) is compared on the scale of the others: its
        # f32 rounding noise is ~5e-6 of the largest gradient and moves with the arithmetic of the products in front of it (3e-2: with
        # 1e-2 the check passed or failed at 6e-4 vs 5e-4 depending on which tests had run before -- `-k dpcl`, round 5)
        errs['grad ' + n] = float(np.abs(grads[n] - g_ref[n]).max() / max(np.abs(g_ref[n]).max(), 3e-2 * gscale))
    plist = [P[n].copy() for n in names]
    opt.apply(plist, [g_ref[n] for n in names])
    pscale = max(float(np.abs(p).max()) for p in plist)
    for n, p in zip(names, plist):
        # zero-initialised biases with an analytically-zero gradient (AMSGrad normalises rounding noise): same floor idea
        errs['update ' + n] = float(np.abs(P_new[n] - p).max() / max(np.abs(p).max(), 1e-2 * pscale))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    assert worst[0][1] < tol, worst


def base_args(**kw):
    from ams_hip import testing
    a = dict(testing.ADAPT_DEFAULTS)
    a.update(testing.SEPARATOR_DEFAULTS)
    a.update(testing.ENHANCE_DEFAULTS)
    a.update(kw)
    return a


@pytest.mark.parametrize('loss,separation,overlap,optimizer', [('sdr+l2', 'mask', 1.0, 'Adam'), ('l2', 'perfect', 0.0, 'RMSProp'),
                                                                ('sdr', 'perfect', 0.5, 'SGD')])
def test_pretraining_step(loss, separation, overlap, optimizer):
    """experiments.training.pretraining (cfg2, default strided front): README.md:23 flags at reduced size."""
    from utils.trainer import Adapt_Pretrainer
    B, S, L, W, N, hop = 3, 2, 1024, 64, 16, 16
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, loss=loss, separation=separation,
                  overlap_coef=overlap, optimizer=optimizer, learning_rate=1e-3, pretraining=True)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, back = orec.pretrain_loss(xm, xn, P, hop, loss, separation, overlap)
    opt = {'Adam': ooptim.AMSGrad(1e-3), 'RMSProp': ooptim.RMSProp(1e-3), 'SGD': ooptim.Momentum(1e-3)}[optimizer]
    check_step(cost, c_ref, grads, g_ref, P, P_new, opt)


@pytest.mark.parametrize('beta,lam,nn,gain', [(1e-2, 1e-4, 0.1, 1.0), (1e-2, 1e-4, 0.0, 25.0), (0.5, 0.3, 0.7, 6.0)])
def test_pretraining_step_at_cli_default_regularisers(beta, lam, nn, gain):
    """experiments.training.pretraining with the terms the CLI turns ON by default (utils/trainer.py:151-161: --beta 1e-2,
    --regularization 1e-4; --non_negativity exercised too): beta * sum kl_div(sparsity, p_hat), lam * (lam * (l2_loss(f2) +
    l2_loss(f))), nn * (nn * mean_b sum min(front, 0)^2)  (models/adapt.py:130-132, 312-316, 377-384).  Cost, all four gradients
    and the AMSGrad update against the oracle.  `gain` scales the front filter so that p_hat = sum_b |y| sits below 1 (gradient
    through the clip), straddles 1, or mostly above (clipped: no gradient) -- the three regimes of utils/ops.py:46-49; the third
    row uses large coefficients so that every term is visible next to the loss."""
    from utils.trainer import Adapt_Pretrainer
    B, S, L, W, N, hop = 3, 2, 1024, 64, 16, 16
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, loss='sdr+l2', separation='perfect',
                  overlap_coef=0.001, optimizer='Adam', learning_rate=1e-3, pretraining=True, beta=beta, regularization=lam,
                  non_negativity=nn, sparsity=0.01)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    g = tr.graph
    g.variables['front/bases/bases'].data.mul_(gain)
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, back = orec.pretrain_loss(xm, xn, P, hop, 'sdr+l2', 'perfect', 0.001, beta=beta, sparsity=0.01, regularization=lam,
                                            non_negativity=nn)
    c_plain, g_plain, _ = orec.pretrain_loss(xm, xn, P, hop, 'sdr+l2', 'perfect', 0.001)
    # the terms are really in the objective and in the gradient (otherwise this test would pass with them dropped)
    assert abs(c_ref - c_plain) > 1e-6 * abs(c_plain)
    if beta > 0.1:                                       # (at the CLI's 1e-2 / 1e-4 the terms move the gradient by ~1e-6 of its size)
        assert max(rel(g_ref[n], g_plain[n]) for n in g_ref) > 1e-3
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3))


def test_stft_dpcl_step():
    """experiments.training.STFT_DPCL (cfg1) at reduced size."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_Trainer
    B, S, L, W, hop, LS, NL, E = 4, 2, 2048, 64, 32, 12, 2, 8
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=1e-3)
    a.pop('type')
    tr = STFT_Separator_Trainer(DPCL, 'STFT_DPCL', **a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, V, Y = ostep.stft_dpcl_loss(xm, xn, P, W, hop, NL, E)
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3))


@pytest.mark.parametrize('normalize', [True, False])
def test_front_l41_step(normalize):
    """experiments.training.front_L41 (cfg5 family) at reduced size, S = 3."""
    from tests.smoke_step import build_front_dpcl  # noqa: F401  (same folder helper)
    from ams_hip import testing
    from models.L41 import L41Model
    from utils.trainer import Front_Separator_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_l41_')
    B, S, L, W, N, hop, LS, NL, E = 3, 3, 1024, 64, 16, 16, 12, 2, 8
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                   batch_size=B, nb_speakers=S)
    a = base_args(**params)
    a.update(layer_size=LS, nb_layers=NL, embedding_size=E, model_folder=folder, model_previous=None, pretraining=False,
             no_normalize=normalize, learning_rate=1e-3)
    a.pop('type')
    tr = Front_Separator_Trainer(L41Model, 'front_L41', **a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, V, Y = ostep.front_l41_loss(xm, xn, I, P, hop, NL, E, normalize)
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3))


def _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, Fq, D_in, front=True, enhance=None, tot_speakers=None):
    from ams_hip import testing
    P = ostep.init_params(rng, np.float32, front_W=W if front else None, N=N, D_in=D_in, layer_size=LS, nb_layers=NL, E=E, F=Fq,
                          conv1d_scale=0.5, tot_speakers=tot_speakers)
    if enhance is not None:                       # (layer_size_enhance, nb_layers_enhance): a checkpoint left by an *_enhance run
        P.update(ostep.init_enhance_params(rng, np.float32, Fq, enhance[0], enhance[1]))
    params = dict(testing.ADAPT_DEFAULTS)
    if not front:
        for k in ('filters', 'max_pool'):
            params.pop(k)
    params.update(testing.SEPARATOR_DEFAULTS)
    params.update(window_size=W, hop_size=hop, chunk_size=L, batch_size=B, nb_speakers=S, layer_size=LS, nb_layers=NL,
                  embedding_size=E, type='front_DPCL' if front else 'STFT_DPCL', pretraining=False)
    if front:
        params.update(filters=N)
    return testing.write_checkpoint(os.path.join(tmp, 'ckpt'), P, params), params, P


@pytest.mark.parametrize('beta,with_silence', [(None, False), (None, True), (5.0, True)])
def test_front_separator_inference(beta, with_silence):
    """Front_Separator_Inference: front -> DPCL -> k-means masks -> back (trainer.py:420-434)."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Inference
    tmp = tempfile.mkdtemp(prefix='ams_inf_')
    rng = np.random.RandomState(11)
    B, S, L, W, N, hop, LS, NL, E, tries, steps = 2, 2, 2048, 64, 16, 16, 12, 2, 8, 2, 3
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N)
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, with_silence=with_silence, end_assign=True,
             kmeans_init_indices=idx, out=False)
    a.pop('type')
    tr = Front_Separator_Inference(DPCL, 'front_DPCL_inference', **a)
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: L}
        xm, xn, out = model.infer(feed, 0)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref, lab_ref, V_ref = orec.front_separate_infer(xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64), P64, hop,
                                                        NL, E, idx, tries, steps, beta=beta, with_silence=with_silence, end_assign=True)
    assert out.shape == (B, S, L)
    # embeddings agree to round-off; a label may flip only at a numerical tie, so compare the waveforms in norm
    err = np.linalg.norm(out.cpu().numpy() - out_ref) / np.linalg.norm(out_ref)
    assert err < INFER_TOL, err


def test_inference_recipe_keeps_weight_derivatives_until_the_weights_are_written():
    """Network.freeze_weights (inference recipes: no optimizer exists): bounds / gathered kernels derived from the variables are kept across
    passes -- a second pass gives the same bits -- and dropped by _weights_written() (restore, or whoever writes `.data` and says so): the
    next pass must see the new weights, exactly as a recipe built without the caches (AMS_NO_FREEZE) does."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Inference
    from ams_hip import ops
    rng = np.random.RandomState(13)
    B, S, L, W, N, hop, LS, NL, E, tries, steps = 2, 2, 2048, 64, 16, 16, 12, 2, 8, 5, 2
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    outs = {}
    for frozen in (True, False):
        tmp = tempfile.mkdtemp(prefix='ams_inf_')
        folder, params, P = _full_checkpoint(tmp, np.random.RandomState(11), W, N, hop, L, B, S, LS, NL, E, N, N)
        a = base_args(**params)
        a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=None, with_silence=False, end_assign=True,
                 kmeans_init_indices=idx, out=False)
        a.pop('type')
        if not frozen:
            os.environ['AMS_NO_FREEZE'] = '1'
        try:
            tr = Front_Separator_Inference(DPCL, 'front_DPCL_inference', **a)
            dist, tfds = tr.prepare()
        finally:
            os.environ.pop('AMS_NO_FREEZE', None)
        g, model = tr.graph, tr.model
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: L}
            o1 = model.infer(feed, 0)[2].clone()
            o2 = model.infer(feed, 0)[2].clone()
            Kf = g.variables['prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel']
            assert ops._frozen(Kf) == frozen and hasattr(Kf, '_ams_wcat') == frozen
            Kf.data.mul_(1.5)
            g.variables['prediction/W'].data.mul_(0.7)
            model._weights_written()
            assert not hasattr(Kf, '_ams_wcat')
            o3 = model.infer(feed, 0)[2].clone()
        assert torch.equal(o1, o2)
        assert torch.isfinite(o3).all() and not torch.equal(o1, o3)
        outs[frozen] = (o1, o3)
    for i in (0, 1):      # frozen: the rings' recurrent product runs as fp16x3 (it has a bound), otherwise bf16x6: same to round-off
        err = float((outs[True][i] - outs[False][i]).norm() / outs[False][i].norm())
        assert err < INFER_TOL, (i, err)


def test_stft_separator_inference():
    """STFT_Separator_Inference: STFT -> DPCL -> hard k-means -> iSTFT (trainer.py:406-417)."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_Inference
    tmp = tempfile.mkdtemp(prefix='ams_sinf_')
    rng = np.random.RandomState(12)
    B, S, L, W, hop, LS, NL, E, tries, steps = 2, 2, 2048, 64, 32, 12, 2, 8, 2, 3
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False)
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, out=False)
    a.pop('type')
    tr = STFT_Separator_Inference(DPCL, 'STFT_DPCL_inference', **a)
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: L}
        xm, xn, out = model.infer(feed, 0)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref, lab_ref, V_ref = orec.stft_separate_infer(xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64), P64, W, hop,
                                                       NL, E, idx, tries, steps, end_assign=True)
    assert out.shape == (B, S, (T - 1) * hop + W)
    err = np.linalg.norm(out.cpu().numpy() - out_ref) / np.linalg.norm(out_ref)
    assert err < INFER_TOL, err


def test_front_dpcl_finetuning_step():
    """experiments.training.front_DPCL_finetuning (cfg3(ii)): gradient reaches prediction/* only through the soft k-means
    masks, the back end and the PIT cost.  Cost vs the oracle; gradients vs central differences of the float64 oracle."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Finetuning_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_ft_')
    rng = np.random.RandomState(21)
    B, S, L, W, N, hop, LS, NL, E, tries, steps, beta = 2, 2, 1024, 64, 16, 16, 12, 2, 8, 1, 3, 4.0
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N)
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, with_silence=True, threshold=2.0, end_assign=True,
             kmeans_init_indices=idx, loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4, pretraining=False)
    a.pop('type')
    tr = Front_Separator_Finetuning_Trainer(DPCL, 'front_L41_finetuning', **a)
    dist, tfds = tr.prepare()
    assert all(v.ams_name.startswith('prediction/') for v in tr.model.trainable_variables)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    args = (hop, NL, E, idx, tries, steps, beta, True, 2.0, True, 'sdr+l2')
    c_ref, _ = orec.front_finetune_cost(xm, xn, Pg, *args)
    assert abs(cost - c_ref) < 1e-3 * abs(c_ref), (cost, c_ref)
    for name in ('prediction/W', 'prediction/forward_BLSTM_1/rnn/basic_lstm_cell/kernel', 'prediction/b'):
        g = grads[name]
        k = np.unravel_index(np.argmax(np.abs(g)), g.shape)          # probe the largest entry
        h = 1e-5 * max(1.0, abs(Pg[name][k]))
        Pp = {n: v.copy() for n, v in Pg.items()}
        Pp[name][k] += h
        cp, _ = orec.front_finetune_cost(xm, xn, Pp, *args)
        Pp[name][k] -= 2 * h
        cm, _ = orec.front_finetune_cost(xm, xn, Pp, *args)
        fd = (cp - cm) / (2 * h)
        assert abs(g[k] - fd) < 2e-2 * max(abs(fd), 1e-6), (name, g[k], fd)


def test_front_dpcl_enhance_step():
    """experiments.training.front_DPCL_enhance: enhance BLSTM stack on top of a frozen front + DPCL + hard k-means."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Enhance_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_enh_')
    rng = np.random.RandomState(31)
    B, S, L, W, N, hop, LS, NL, E, tries, steps, LSE, NLE = 2, 2, 1024, 64, 16, 16, 12, 2, 8, 2, 3, 8, 2
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N)
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', learning_rate=1e-3, pretraining=False)
    a.pop('type')
    tr = Front_Separator_Enhance_Trainer(DPCL, 'front_DPCL_enhance', **a)
    dist, tfds = tr.prepare()
    names = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert names and all(n.startswith('enhance/') for n in names)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref = orec.front_enhance_loss(xm, xn, Pg, hop, NL, E, NLE, idx, tries, steps)
    check_step(cost, c_ref, grads, g_ref, Pg, P_new, ooptim.AMSGrad(1e-3), tol=5e-4)


def test_pretraining_step_with_max_pool():
    """experiments.training.pretraining --with_max_pool (path B, SURVEY 8d cfg2 variant): fused conv+max-pool front, sparse back."""
    from utils.trainer import Adapt_Pretrainer
    B, S, L, W, N, hop, Pool = 2, 2, 1024, 64, 16, 128, 128
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, hop_size=hop, max_pool=Pool, with_max_pool=True,
                  loss='l2', separation='perfect', overlap_coef=0.0, optimizer='Adam', learning_rate=1e-3, pretraining=True)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref, back, am = orec.pretrain_loss_maxpool(xm, xn, P, Pool, hop, 'l2', 'perfect')
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3), tol=5e-4)


def test_pretraining_forward_with_average_pool():
    """--with_average_pool (path C): box-filtered strided conv / synthesis vs the oracle's dense conv + pool + up-sample form;
    gradients vs torch autograd of the dense CPU formulation."""
    import torch.nn.functional as TF
    from ams_hip import functional as F
    from oracle import front as ofront
    rng = np.random.RandomState(5)
    Bt, L, W, N, Pool, R = 3, 512, 32, 6, 64, 4
    x, f, f2 = rng.randn(Bt, L), rng.randn(W, N) / 6, rng.randn(W, N) / 6
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    ft, f2t = dev(f).requires_grad_(), dev(f2).requires_grad_()
    y = F.front_avgpool(dev(x), ft, Pool)
    assert rel(y.detach().cpu().numpy(), ofront.front_avgpool(x, f, Pool)) < 2e-5
    z = rng.randn(R, L // Pool, N)
    zt = dev(z).requires_grad_()
    out = F.synth_avgpool(zt, f2t, Pool, L)
    ref = ofront.synth_strided(ofront.upsample_nearest(z, Pool), f2, 1, L)
    assert rel(out.detach().cpu().numpy(), ref) < 2e-5
    dy, dout = rng.randn(*y.shape), rng.randn(R, L)
    (y * dev(dy)).sum().backward()
    (out * dev(dout)).sum().backward()
    # dense CPU reference with autograd
    pl, pr = (W - 1) // 2, W - 1 - (W - 1) // 2
    fc, f2c, zc = [torch.from_numpy(v).requires_grad_() for v in (f, f2, z)]
    X = TF.conv1d(TF.pad(torch.from_numpy(x)[:, None], (pl, pr)), fc.t()[:, None])                 # [Bt, N, L]
    yc = TF.avg_pool1d(X, Pool).permute(0, 2, 1)
    (yc * torch.from_numpy(dy)).sum().backward()
    up = zc.repeat_interleave(Pool, dim=1)
    oc = TF.conv_transpose1d(up.permute(0, 2, 1), f2c.t()[:, None])[:, 0, pl:pl + L]
    (oc * torch.from_numpy(dout)).sum().backward()
    assert rel(ft.grad.cpu().numpy(), fc.grad.numpy()) < 1e-4
    assert rel(f2t.grad.cpu().numpy(), f2c.grad.numpy()) < 1e-4 and rel(zt.grad.cpu().numpy(), zc.grad.numpy()) < 1e-4


def _fd_check(cost_fn, Pg, grads, names, tol=2e-2):
    """Central differences of the float64 oracle cost on the largest gradient entry of each named variable."""
    for name in names:
        g = grads[name]
        k = np.unravel_index(np.argmax(np.abs(g)), g.shape)
        h = 1e-5 * max(1.0, abs(Pg[name][k]))
        Pp = {n: v.copy() for n, v in Pg.items()}
        Pp[name][k] += h
        cp = cost_fn(Pp)
        Pp[name][k] -= 2 * h
        cm = cost_fn(Pp)
        fd = (cp - cm) / (2 * h)
        assert abs(g[k] - fd) < tol * max(abs(fd), 1e-6), (name, g[k], fd)


def test_stft_dpcl_enhance_step():
    """experiments.training.STFT_DPCL_enhance: enhance stack on top of a restored STFT + DPCL + hard k-means separator."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_enhance_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_senh_')
    rng = np.random.RandomState(41)
    B, S, L, W, hop, LS, NL, E, tries, steps, LSE, NLE = 2, 2, 1024, 64, 16, 12, 2, 8, 2, 3, 8, 2
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False)
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', learning_rate=1e-3, pretraining=False)
    a.pop('type')
    tr = STFT_Separator_enhance_Trainer(DPCL, 'STFT_DPCL_enhance', **a)
    dist, tfds = tr.prepare()
    names = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert names and all(n.startswith('enhance/') for n in names)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    c_ref, g_ref = orec.stft_enhance_loss(xm, xn, Pg, W, hop, NL, E, NLE, idx, tries, steps)
    check_step(cost, c_ref, grads, g_ref, Pg, P_new, ooptim.AMSGrad(1e-3), tol=5e-4)


def test_stft_dpcl_finetuning_step():
    """experiments.training.STFT_DPCL_finetuning: |STFT| -> DPCL -> soft k-means -> enhance -> iSTFT -> PIT waveform L2; the
    gradient reaches prediction/* through the soft masks and enhance/* directly."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_FineTune_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_sft_')
    rng = np.random.RandomState(43)
    B, S, L, W, hop, LS, NL, E, tries, steps, LSE, NLE, beta = 2, 2, 1024, 64, 16, 12, 2, 8, 1, 3, 8, 1, 4.0
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False, enhance=(LSE, NLE))
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, end_assign=True, kmeans_init_indices=idx,
             layer_size_enhance=LSE, nb_layers_enhance=NLE, nonlinearity='softmax', learning_rate=1e-4, optimizer='RMSProp',
             train=['enhance', 'prediction'], pretraining=False)
    a.pop('type')
    tr = STFT_Separator_FineTune_Trainer(DPCL, 'STFT_DPCL_finetuning', **a)
    dist, tfds = tr.prepare()
    tnames = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert any(n.startswith('enhance/') for n in tnames) and any(n.startswith('prediction/') for n in tnames)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    cost_fn = lambda Pp: orec.stft_finetune_cost(xm, xn, Pp, W, hop, NL, E, NLE, idx, tries, steps, beta)[0]   # noqa: E731
    c_ref = cost_fn(Pg)
    assert abs(cost - c_ref) < 1e-3 * abs(c_ref), (cost, c_ref)
    _fd_check(cost_fn, Pg, grads, ('enhance/W', 'prediction/W', 'prediction/forward_BLSTM_1/rnn/basic_lstm_cell/kernel'))


def test_front_dpcl_enhance_finetuning_step():
    """experiments.training.front_DPCL_enhance_finetuning: frozen front -> DPCL -> soft k-means -> enhance -> back -> PIT cost."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Enhance_Finetuning_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_eft_')
    rng = np.random.RandomState(47)
    B, S, L, W, N, hop, LS, NL, E, tries, steps, LSE, NLE, beta = 2, 2, 1024, 64, 16, 16, 12, 2, 8, 1, 3, 8, 1, 4.0
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N, enhance=(LSE, NLE))
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, with_silence=True, threshold=2.0, end_assign=True,
             kmeans_init_indices=idx, layer_size_enhance=LSE, nb_layers_enhance=NLE, nonlinearity='softmax', loss='sdr+l2',
             optimizer='RMSProp', learning_rate=1e-4, train=['enhance', 'prediction'], pretraining=False)
    a.pop('type')
    tr = Front_Separator_Enhance_Finetuning_Trainer(DPCL, 'front_DPCL_enhance_finetuning', **a)
    dist, tfds = tr.prepare()
    tnames = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert any(n.startswith('enhance/') for n in tnames) and any(n.startswith('prediction/') for n in tnames)
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    args = (hop, NL, E, NLE, idx, tries, steps, beta, True, 2.0, True)
    cost_fn = lambda Pp: orec.front_enhance_finetune_cost(xm, xn, Pp, *args)[0]   # noqa: E731
    c_ref = cost_fn(Pg)
    assert abs(cost - c_ref) < 1e-3 * abs(c_ref), (cost, c_ref)
    _fd_check(cost_fn, Pg, grads, ('enhance/W', 'prediction/W', 'prediction/forward_BLSTM_1/rnn/basic_lstm_cell/kernel'))


def _infer(tr, L):
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: L}
        xm, xn, out = model.infer(feed, 0)
    return xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64), out.cpu().numpy()


def test_front_separator_enhanced_inference():
    """Front_Separator_Enhanced_Inference: front -> DPCL -> k-means -> enhance stack -> back (trainer.py:436-449)."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Enhanced_Inference
    tmp = tempfile.mkdtemp(prefix='ams_einf_')
    rng = np.random.RandomState(51)
    B, S, L, W, N, hop, LS, NL, E, tries, steps, LSE, NLE = 2, 2, 2048, 64, 16, 16, 12, 2, 8, 2, 3, 8, 2
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N, enhance=(LSE, NLE))
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', out=False)
    a.pop('type')
    tr = Front_Separator_Enhanced_Inference(DPCL, 'front_DPCL_enhance_inference', **a)
    xm, xn, out = _infer(tr, L)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref = orec.front_separate_enhanced_infer(xm, xn, P64, hop, NL, E, NLE, idx, tries, steps)
    assert out.shape == (B, S, L)
    err = np.linalg.norm(out - out_ref) / np.linalg.norm(out_ref)
    assert err < INFER_TOL, err


def test_stft_separator_enhanced_inference():
    """STFT_Separator_Enhanced_Inference: |STFT| -> DPCL -> hard k-means -> enhance stack -> iSTFT (trainer.py:390-404)."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_Enhanced_Inference
    tmp = tempfile.mkdtemp(prefix='ams_seinf_')
    rng = np.random.RandomState(53)
    B, S, L, W, hop, LS, NL, E, tries, steps, LSE, NLE = 2, 2, 2048, 64, 32, 12, 2, 8, 2, 3, 8, 1
    Fq = W // 2 + 1
    folder, params, P = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False, enhance=(LSE, NLE))
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', out=False)
    a.pop('type')
    tr = STFT_Separator_Enhanced_Inference(DPCL, 'STFT_DPCL_enhance_inference', **a)
    xm, xn, out = _infer(tr, L)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref = orec.stft_separate_enhanced_infer(xm, xn, P64, W, hop, NL, E, NLE, idx, tries, steps)
    assert out.shape == out_ref.shape == (B, S, (T - 1) * hop + W)
    err = np.linalg.norm(out - out_ref) / np.linalg.norm(out_ref)
    assert err < INFER_TOL, err


@pytest.mark.parametrize('separation', ['mask', 'perfect'])
def test_pretrained_inference(separation):
    """Pretrained_Inference: the pre-trained filterbank with the oracle separator (trainer.py:451-462)."""
    from ams_hip import testing
    from utils.trainer import Pretrained_Inference
    tmp = tempfile.mkdtemp(prefix='ams_pinf_')
    B, S, L, W, N, hop = 3, 2, 2048, 64, 16, 16
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                   batch_size=B, nb_speakers=S, separation=separation)
    a = base_args(**params)
    a.update(model_folder=folder, out=False, separation=separation)
    a.pop('type')
    tr = Pretrained_Inference(None, 'pretrained_inference', **a)
    xm, xn, out = _infer(tr, L)
    P = {n: v.detach().cpu().numpy().astype(np.float64) for n, v in tr.graph.variables.items()}
    out_ref = orec.pretrained_infer(xm, xn, P, hop, separation)
    assert out.shape == (B, S, L)
    assert rel(out, out_ref) < 2e-4


def test_front_l41_inference_and_finetuning():
    """The same inference / fine-tuning trainers with the L41 separator (experiments.training.front_L41_finetuning): the
    checkpoint carries 'speaker_centroids', prediction/* is what k-means clusters."""
    from models.L41 import L41Model
    from utils.trainer import Front_Separator_Inference, Front_Separator_Finetuning_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_l41ft_')
    rng = np.random.RandomState(61)
    B, S, L, W, N, hop, LS, NL, E, tries, steps, beta, NSPK = 2, 2, 1024, 64, 16, 16, 12, 2, 8, 1, 3, 4.0, 251
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N, tot_speakers=NSPK)
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, with_silence=True, threshold=2.0, end_assign=True,
             kmeans_init_indices=idx, tot_speakers=NSPK, out=False, pretraining=False)
    a.pop('type')
    tr = Front_Separator_Inference(L41Model, 'front_L41_inference', **dict(a))
    xm, xn, out = _infer(tr, L)
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    out_ref, _, _ = orec.front_separate_infer(xm, xn, P64, hop, NL, E, idx, tries, steps, beta=beta, with_silence=True, end_assign=True)
    assert np.linalg.norm(out - out_ref) / np.linalg.norm(out_ref) < INFER_TOL

    a.update(loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4)
    tr = Front_Separator_Finetuning_Trainer(L41Model, 'front_L41_finetuning', **dict(a))
    dist, tfds = tr.prepare()
    Pg, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    args = (hop, NL, E, idx, tries, steps, beta, True, 2.0, True, 'sdr+l2')
    cost_fn = lambda Pp: orec.front_finetune_cost(xm, xn, Pp, *args)[0]   # noqa: E731
    c_ref = cost_fn(Pg)
    assert abs(cost - c_ref) < 1e-3 * abs(c_ref), (cost, c_ref)
    _fd_check(cost_fn, Pg, grads, ('prediction/W',))


def test_restore_from_tensorflow_bundle_matches_npz():
    """A model folder in the REFERENCE's on-disk format (text `checkpoint` + model-N.index/.data-*, `params` JSON) restores to
    the same inference output as the npz form (bundle written by ams_hip/tf_checkpoint.py)."""
    import shutil
    from ams_hip import tf_checkpoint
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Inference
    tmp = tempfile.mkdtemp(prefix='ams_tfck_')
    rng = np.random.RandomState(71)
    B, S, L, W, N, hop, LS, NL, E, tries, steps = 2, 2, 1024, 64, 16, 16, 12, 2, 8, 2, 3
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N)
    tf_folder = os.path.join(tmp, 'tf')
    os.makedirs(tf_folder)
    shutil.copy(os.path.join(folder, 'params'), os.path.join(tf_folder, 'params'))
    extra = dict(P)
    extra['global_epoch'] = np.array(0, np.int32)                    # the reference also saves optimizer state / counters
    extra['prediction/W/AMSGrad'] = np.zeros_like(P['prediction/W'])
    # the reference's Conv1D kernel is 3-D [1, Din, Dout] on disk (utils/ops.py:486-492); this build keeps [Din, Dout]
    extra['prediction/W'] = P['prediction/W'][None]
    assert extra['prediction/W'].ndim == 3
    tf_checkpoint.write_bundle(os.path.join(tf_folder, 'model-7'), extra)
    with open(os.path.join(tf_folder, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model-7"\nall_model_checkpoint_paths: "model-7"\n')
    T = -(-L // hop)
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    outs = []
    for fol in (folder, tf_folder):
        a = base_args(**params)
        a.update(model_folder=fol, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, out=False)
        a.pop('type')
        tr = Front_Separator_Inference(DPCL, 'front_DPCL_inference', **a)
        outs.append(_infer(tr, L)[2])
    assert np.array_equal(outs[0], outs[1])


def test_tf_eval_cli_on_a_pretrained_checkpoint():
    """python -m experiments.evaluation.tf_eval --model pretraining (reference experiments/evaluation/tf_eval.py:1-38): the CLI's
    running mean equals the batch-size-weighted mean of the ORACLE's SDR improvement (models/network.py:196-221) of the oracle's
    own reconstruction, batch by batch over the test split."""
    from ams_hip import testing
    from experiments.evaluation import tf_eval
    from oracle import losses as olosses
    from utils.trainer import Pretrained_Inference
    tmp = tempfile.mkdtemp(prefix='ams_tfeval_')
    B, S, L, W, N, hop = 3, 2, 2048, 64, 16, 16
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                   batch_size=B, nb_speakers=S, separation='perfect')
    argv = ['--model_folder', folder, '--model', 'pretraining', '--dataset', 'synthetic', '--batch_size', str(B), '--chunk_size', str(L),
            '--nb_speakers', str(S), '--window_size', str(W), '--filters', str(N), '--hop_size', str(hop), '--separation', 'perfect',
            '--men', '--women']
    sdr, n = tf_eval.main(argv)
    assert n >= 1 and np.isfinite(sdr)
    # the same split again (same command line), through the oracle
    tr, _ = tf_eval.build(argv)
    assert isinstance(tr, Pretrained_Inference)
    tot, cnt = 0.0, 0
    for xm, xn, imp in tr.sdr_improvement():
        xm, xn = xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64)
        P = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in tr.graph.variables.items()}
        back = orec.pretrained_infer(xm, xn, P, hop, 'perfect')
        val, _ = olosses.sdr_improvement(xm, xn, back)
        val = float(np.mean(val))
        assert abs(float(imp) - val) < 1e-3 * max(1.0, abs(val)), (float(imp), val)
        tot += val * B
        cnt += 1
    assert cnt == n and abs(sdr - tot / (cnt * B)) < 1e-3 * max(1.0, abs(sdr)), (sdr, tot / (cnt * B))
