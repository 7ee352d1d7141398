"""GPU: WHOLE training steps of BASELINE.json's configs at their FULL sizes against the float64 oracle -- cost, every gradient, every
updated weight -- through the recipe classes the reference's entry points build.  tests/test_gpu_benchshape.py does this for the
headline (cfg3(i) front_DPCL, B = 64); here the other configs, which until round 6 met the oracle as whole steps at reduced size only
(tests/test_gpu_recipes.py, tests/test_gpu_cfg4.py) and at full size kernel by kernel (tests/test_gpu_fullgeom.py):

    cfg1     STFT_DPCL               B = 4,   S = 2, W = 512, hop = 256                  (reference dpcl_stft_train.sh)
    cfg2     pretraining, path A     B = 64,  S = 2, W = 1024, hop = 256, N = 256        (README.md:23 flags)
    cfg3(ii) front_DPCL_finetuning   B = 64,  soft k-means beta = 10, 1 try x 10 steps, silence weights, back end, PIT cost, RMSProp
    cfg3     front_DPCL inference    B = 64,  hard k-means 10 x 10 -> masks -> back (embeddings, labels, waveforms)
    cfg4     STFT_L41                B = 64,  F = 257, 3 x BLSTM(600), E = 40
    cfg4     STFT_L41_enhance        B = 64,  frozen L41 + hard k-means 10 x 10 + enhance stack
    cfg5     front_L41               B = 128, S = 3, N = 512

All with L = 20480.  Tolerances: those of tests/test_gpu_benchshape.py -- cost 1e-4, gradients 2e-4 of the tensor's largest entry,
optimizer update 1e-5 (north_star allows 1e-3).  Two places of the path take an arg-max / arg-min of float32 quantities that may tie
-- the ideal masks of the STFT recipes and the hard k-means in front of the enhance stack: there the test counts the labels that
differ from the float64 chain, bounds them, and compares the rest of the step on equal labels (see _device_mask_spectra).
The float64 oracle takes 2-80 s per step on the GPU box's host."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import step as ostep, recipes as orec, optim as ooptim
from tests.test_gpu_recipes import base_args, one_train_step, _full_checkpoint

L = 20480
FWD_TOL, BWD_TOL = 1e-4, 2e-4


def check_step(cost, c_ref, grads, g_ref, P, P_new, opt, fwd_tol=FWD_TOL, bwd_tol=BWD_TOL, what=''):
    """Cost and every gradient against the float64 oracle (relative to the tensor's largest entry; a tensor whose gradient is
    analytically zero -- a bias under a softmax over speakers -- on the scale of the largest gradient), then the fused optimizer
    kernel on the DEVICE's gradients: with 0.5-1 M weights per tensor some have |g| within rounding of 0, AMSGrad turns their sign
    into +-lr, so an update computed from the oracle's gradient would test the sign of rounding noise
    (tests/test_gpu_benchshape.py::test_front_dpcl_step_at_benchmark_shape does the same)."""
    errs = {'cost': abs(cost - c_ref) / max(abs(c_ref), 1e-30)}
    names = sorted(g_ref)
    assert sorted(grads) == names
    gscale = max(float(np.abs(g_ref[n]).max()) for n in names)
    for n in names:
        errs['grad ' + n] = float(np.abs(grads[n] - g_ref[n]).max() / max(np.abs(g_ref[n]).max(), 3e-2 * gscale))
    plist = [P[n].copy() for n in names]
    opt.apply(plist, [grads[n].astype(np.float64) for n in names])
    for n, p in zip(names, plist):
        errs['update ' + n] = float(np.abs(P_new[n] - p).max() / max(np.abs(p).max(), 1e-30))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print('%s: cost %.6f oracle %.6f; worst' % (what, cost, c_ref), worst)
    if os.environ.get('AMS_TEST_REPORT_ONLY'):
        return
    assert errs['cost'] < fwd_tol, worst
    assert max(v for k, v in errs.items() if k.startswith('grad ')) < bwd_tol, worst
    assert max(v for k, v in errs.items() if k.startswith('update ')) < 1e-5, worst


@pytest.fixture(autouse=True)
def _arith():
    a = os.environ.get('AMS_TEST_ARITH')
    if a is None:
        yield
        return
    from ams_hip._lib import load
    lib = load()
    before = lib.ams_gemm_get_arith()
    lib.ams_gemm_set_arith(int(a))
    yield
    lib.ams_gemm_set_arith(before)


def _graphed_step(tr, tfds, opt_ref):
    """Two eager steps on the capture stream, the capture, and the first replay (network.py::_train_graphed): returns what
    one_train_step returns for the replayed step, with the oracle's optimizer slots kept in step on the device's gradients."""
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        for it in range(3):
            P = {n: v.detach().cpu().numpy().astype(np.float64) for n, v in g.variables.items()}
            cost = float(model.train(feed, it))
            if it < 2:
                names = sorted(v.ams_name for v in model.trainable_variables)
                gw = {v.ams_name: v.grad.detach().cpu().numpy().astype(np.float64) for v in model.trainable_variables}
                opt_ref.apply([P[n].copy() for n in names], [gw[n] for n in names])
        run = model.last_run
        xm = model.x_mix.value(run).cpu().numpy().astype(np.float64)
        xn = model.x_non_mix.value(run).cpu().numpy().astype(np.float64)
        I = model.I.value(run).cpu().numpy()
        grads = {v.ams_name: v.grad.detach().cpu().numpy() for v in model.trainable_variables}
        P_new = {v.ams_name: v.detach().cpu().numpy() for v in model.trainable_variables}
    return P, cost, xm, xn, I, grads, P_new


def _device_mask_spectra(xn, W, hop, what):
    """The per-speaker |STFT| the ideal masks are cut from (network.py:480-502), as the device computes them (float32), held against
    the float64 oracle; returns them for the oracle's arg-max.  Found at these sizes (round 6): of 1 299 392 bins at B = 64, 107 hold
    the two speakers within 1e-7 of the largest magnitude and 1-2 labels fall the other way in float32 (gap 2.3e-9; the native-f32
    DFT product flips one, fp16x3 -- three times closer to float64 -- two); ONE such label moves the gradients of an untrained net by
    1e-3 of their largest entry (a sum of ~1e6 incoherent terms), with the native f32 products exactly as with fp16x3.  The arg-max
    of a tie is not a property either arithmetic can hold, so: magnitudes to 2e-6, every differing label at a tie narrower than
    that, at most 1e-5 of the labels -- and the step is then compared on the labels the device's magnitudes give."""
    from ams_hip import functional as F
    from oracle import stft as ostft
    B, S, L_ = xn.shape
    ref = np.abs(ostft.stft(xn.reshape(B * S, L_), W, hop))
    mag, _ = F.stft_mag_phase(torch.from_numpy(xn.reshape(B * S, L_).astype(np.float32)).cuda(), W, hop, False)
    mag = mag.cpu().numpy().astype(np.float64)
    top = ref.max()
    assert np.abs(mag - ref).max() < 2e-6 * top
    T, Fq = ref.shape[1:]
    dev_nm = mag.reshape(B, S, T, Fq).transpose(0, 2, 3, 1)
    ref_nm = ref.reshape(B, S, T, Fq).transpose(0, 2, 3, 1)
    flips = dev_nm.argmax(-1) != ref_nm.argmax(-1)
    srt = np.sort(ref_nm, -1)
    gap = srt[..., -1] - srt[..., -2]
    print('%s: %d of %d ideal-mask labels differ from float64; widest tie among them %.3g of the largest magnitude'
          % (what, int(flips.sum()), flips.size, (gap[flips].max() / top) if flips.any() else 0.0))
    assert flips.sum() <= max(3, 1e-5 * flips.size)
    assert not flips.any() or gap[flips].max() < 2e-6 * top
    return np.ascontiguousarray(dev_nm)


def test_stft_dpcl_step_at_cfg1_size():
    """BASELINE configs[0]: STFT + DPCL, 2 speakers, batch 4 -- the flags of the reference's dpcl_stft_train.sh (window 512, hop 256,
    layer 600, embedding 40, chunk 20480): T = 79 frames x F = 257 bins, dense 600 -> 10280."""
    from models.dpcl import DPCL
    from utils.trainer import STFT_Separator_Trainer
    B, S, W, hop, LS, NL, E = 4, 2, 512, 256, 600, 3, 40
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=1e-3)
    a.pop('type')
    tr = STFT_Separator_Trainer(DPCL, 'STFT_DPCL', **a)
    dist, tfds = tr.prepare()
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    assert P['prediction/W'].shape[-1] == (W // 2 + 1) * E
    c_ref, g_ref, V, Y = ostep.stft_dpcl_loss(xm, xn, P, W, hop, NL, E, mask_spectra=_device_mask_spectra(xn, W, hop, 'cfg1'))
    assert V.size == B * 79 * 257 * E
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3), what='cfg1 STFT_DPCL B=4')


@pytest.mark.parametrize('graph', [False, True])
def test_pretraining_step_at_cfg2_size(graph):
    """BASELINE configs[1], the default strided front (path A): --filters 256 --max_pool 256 --nb_speakers 2, W = 1024, B = 64,
    --loss sdr+l2 --separation mask --overlap_coef 1.0; eager and as the replayed hipGraph.  Path B at this geometry:
    tests/test_gpu_fullgeom.py::test_maxpool_front_at_cfg2_geometry."""
    from utils.trainer import Adapt_Pretrainer
    B, S, W, N, hop = 64, 2, 1024, 256, 256
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, filters=N, max_pool=hop, hop_size=hop, loss='sdr+l2',
                  separation='mask', overlap_coef=1.0, beta=0.0, regularization=0.0, optimizer='Adam', learning_rate=1e-3,
                  pretraining=True, hip_graph=graph)
    a.pop('type')
    tr = Adapt_Pretrainer(**a)
    dist, tfds = tr.prepare()
    opt = ooptim.AMSGrad(1e-3)
    P, cost, xm, xn, I, grads, P_new = _graphed_step(tr, tfds, opt) if graph else one_train_step(tr, tfds, L)
    assert xm.shape == (B, L) and P['front/bases/bases'].size == W * N
    c_ref, g_ref, back = orec.pretrain_loss(xm, xn, P, hop, 'sdr+l2', 'mask', 1.0)
    check_step(cost, c_ref, grads, g_ref, P, P_new, opt, what='cfg2 pretraining A B=64 graph=%s' % graph)


def _front_checkpoint(tmp, rng, B, S, N, tot_speakers=None):
    W, hop, LS, NL, E = 1024, 256, 600, 3, 40
    folder, params, P = _full_checkpoint(tmp, rng, W, N, hop, L, B, S, LS, NL, E, N, N, tot_speakers=tot_speakers)
    return folder, params, P, (W, hop, LS, NL, E)


def test_front_dpcl_finetuning_step_at_cfg3_size():
    """BASELINE configs[2], second half: the whole chain front -> 3 x BLSTM -> soft k-means (beta 10, 1 try x
    10 steps, silence weights at threshold 2.0, end_assign) -> masks -> back -> PIT cost; prediction/* trains (the gradient reaches it only through the soft
    k-means), RMSProp; B = 64."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Finetuning_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_full_ft_')
    rng = np.random.RandomState(61)
    B, S, N, tries, steps, beta, thr = 64, 2, 256, 1, 10, 10.0, 2.0
    folder, params, P0, (W, hop, LS, NL, E) = _front_checkpoint(tmp, rng, B, S, N)
    T = L // hop
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=beta, with_silence=True, threshold=thr, end_assign=True,
             kmeans_init_indices=idx, loss='sdr+l2', optimizer='RMSProp', learning_rate=1e-4)
    a.pop('type')
    tr = Front_Separator_Finetuning_Trainer(DPCL, 'front_L41_finetuning', **a)
    dist, tfds = tr.prepare()
    assert all(v.ams_name.startswith('prediction/') for v in tr.model.trainable_variables)
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    args = (hop, NL, E, idx, tries, steps, beta, True, thr, True, 'sdr+l2')
    c_ref, out_ref = orec.front_finetune_cost(xm, xn, P, *args)
    assert out_ref.shape == (B, S, L)
    assert abs(cost - c_ref) < 1e-3 * abs(c_ref), (cost, c_ref)
    # the oracle of this recipe is forward-only: the gradient's largest entry against a central difference of the float64 cost
    g = grads['prediction/W']
    k = np.unravel_index(np.argmax(np.abs(g)), g.shape)
    h = 1e-5 * max(1.0, abs(P['prediction/W'][k]))
    Pp = {n: v.copy() for n, v in P.items()}
    Pp['prediction/W'][k] += h
    cp = orec.front_finetune_cost(xm, xn, Pp, *args)[0]
    Pp['prediction/W'][k] -= 2 * h
    cm = orec.front_finetune_cost(xm, xn, Pp, *args)[0]
    fd = (cp - cm) / (2 * h)
    assert abs(g[k] - fd) < 2e-2 * max(abs(fd), 1e-6), (g[k], fd)
    # RMSProp on the device's gradients (the update kernel's parity; gradient parity is the probe above)
    names = sorted(grads)
    plist = [P[n].copy() for n in names]
    ooptim.RMSProp(1e-4).apply(plist, [grads[n].astype(np.float64) for n in names])
    for n, p in zip(names, plist):
        assert np.abs(P_new[n] - p).max() <= 1e-5 * max(np.abs(p).max(), 1e-30), n


def test_front_dpcl_inference_at_cfg3_size():
    """The path north_star's "masks within 1e-3, cluster assignment bit-exact" is about, at B = 64: front -> 3 x BLSTM -> Conv1D ->
    l2norm -> hard k-means (10 tries x 10 steps, end_assign) -> masks -> back (trainer.py:420-434).  Embeddings against the float64
    oracle; the labels of float32 embeddings against those of float64 embeddings (counted: equal up to points within rounding of a
    cluster boundary -- on EQUAL embeddings they are held bit-exact by test_kmeans_hard_at_benchmark_shape); separated waveforms."""
    from models.dpcl import DPCL
    from utils.trainer import Front_Separator_Inference
    tmp = tempfile.mkdtemp(prefix='ams_full_inf_')
    rng = np.random.RandomState(71)
    B, S, N, tries, steps = 64, 2, 256, 10, 10
    folder, params, P0, (W, hop, LS, NL, E) = _front_checkpoint(tmp, rng, B, S, N)
    T = L // hop
    idx = np.stack([rng.choice(T * N, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, beta_kmeans=None, with_silence=False, end_assign=True,
             kmeans_init_indices=idx, out=False)
    a.pop('type')
    tr = Front_Separator_Inference(DPCL, 'front_DPCL_inference', **a)
    dist, tfds = tr.prepare()
    g, model = tr.graph, tr.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TEST), tfds.chunk_size: L}
        xm, xn, out, masks, emb = model._eval_guarded(feed, lambda run: [model.x_mix.value(run), model.x_non_mix.value(run),
                                                                         model.output.value(run), model.sepNet.masks.value(run),
                                                                         model.sepNet.embeddings.value(run)])
    P64 = {k: v.astype(np.float64) for k, v in P0.items()}
    out_ref, lab_ref, V_ref = orec.front_separate_infer(xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64), P64, hop,
                                                        NL, E, idx, tries, steps, end_assign=True)
    e_emb = np.abs(emb.cpu().numpy().reshape(V_ref.shape) - V_ref).max()            # unit-norm rows: absolute = relative to the norm
    lab_dev = masks.argmax(-1).cpu().numpy()
    differ = lab_dev != lab_ref
    o = out.cpu().numpy()
    e_own = np.linalg.norm(o - out_ref) / np.linalg.norm(out_ref)
    # the synthesis (masks -> back) on EQUAL labels
    out_eq = orec.front_separate_infer(xm.cpu().numpy().astype(np.float64), xn.cpu().numpy().astype(np.float64), P64, hop, NL, E, idx,
                                       tries, steps, end_assign=True, labels=lab_dev)[0]
    e_out = np.linalg.norm(o - out_eq) / np.linalg.norm(out_eq)
    per_utt = differ.reshape(B, -1).sum(1)
    print('cfg3 inference B=64: max |V - V64| %.3g; %d of %d labels differ from the float64 chain (in %d utterances, at most %d in one); '
          'waveforms %.3g on equal labels, %.3g against the float64 chain'
          % (e_emb, int(differ.sum()), differ.size, int((per_utt > 0).sum()), int(per_utt.max()), e_out, e_own))
    assert out.shape == (B, S, L)
    assert e_emb < 1e-5
    assert differ.sum() <= 1e-4 * differ.size
    assert e_out < 1e-5
    assert e_own < 1e-2


@pytest.mark.parametrize('graph', [False, True])
def test_stft_l41_step_at_cfg4_size(graph):
    """BASELINE configs[3], first half: STFT_L41 at B = 64, W = 512 (F = 257: the dense product's 10280 columns and the ring's input
    width 257 are the unaligned shapes), 3 x BLSTM(600), E = 40, 251 speaker vectors; eager and replayed."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_Trainer
    B, S, W, hop, LS, NL, E = 64, 2, 512, 256, 600, 3, 40
    a = base_args(batch_size=B, nb_speakers=S, chunk_size=L, window_size=W, hop_size=hop, layer_size=LS, nb_layers=NL,
                  embedding_size=E, model_folder=None, learning_rate=1e-3, tot_speakers=251, hip_graph=graph)
    a.pop('type')
    tr = STFT_Separator_Trainer(L41Model, 'STFT_L41', **a)
    dist, tfds = tr.prepare()
    opt = ooptim.AMSGrad(1e-3)
    P, cost, xm, xn, I, grads, P_new = _graphed_step(tr, tfds, opt) if graph else one_train_step(tr, tfds, L)
    assert P['speaker_centroids'].shape == (251, E)
    c_ref, g_ref, V, Y = ostep.stft_l41_loss(xm, xn, I, P, W, hop, NL, E, True, mask_spectra=_device_mask_spectra(xn, W, hop, 'cfg4'))
    assert V.size == B * 79 * 257 * E
    check_step(cost, c_ref, grads, g_ref, P, P_new, opt, what='cfg4 STFT_L41 B=64 graph=%s' % graph)


def test_stft_l41_enhance_step_at_cfg4_size():
    """BASELINE configs[3], second half: STFT_L41_enhance at B = 64 -- restored, frozen L41 separator -> hard k-means (10 tries x 10
    steps, end_assign) over 20303 points per utterance -> masks -> enhance 3 x BLSTM(600) -> PIT squared error; only enhance/* trains."""
    from models.L41 import L41Model
    from utils.trainer import STFT_Separator_enhance_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_full_enh_')
    rng = np.random.RandomState(67)
    B, S, W, hop, LS, NL, E, tries, steps, LSE, NLE, NSPK = 64, 2, 512, 256, 600, 3, 40, 10, 10, 600, 3, 251
    Fq = W // 2 + 1
    folder, params, P0 = _full_checkpoint(tmp, rng, W, None, hop, L, B, S, LS, NL, E, Fq, Fq, front=False, tot_speakers=NSPK)
    T = 1 + (L - W) // hop
    idx = np.stack([rng.choice(T * Fq, S, replace=False) for _ in range(B * tries)]).astype(np.int32)
    a = base_args(**params)
    a.update(model_folder=folder, nb_tries=tries, nb_steps=steps, end_assign=True, kmeans_init_indices=idx, layer_size_enhance=LSE,
             nb_layers_enhance=NLE, nonlinearity='softmax', learning_rate=1e-3, pretraining=False, tot_speakers=NSPK)
    a.pop('type')
    tr = STFT_Separator_enhance_Trainer(L41Model, 'STFT_L41_enhance', **a)
    dist, tfds = tr.prepare()
    names = sorted(v.ams_name for v in tr.model.trainable_variables)
    assert names and all(n.startswith('enhance/') for n in names)
    P, cost, xm, xn, I, grads, P_new = one_train_step(tr, tfds, L)
    with tr.graph.as_default():
        lab_dev = tr.model.masks.value(tr.model.last_run).argmax(-1).cpu().numpy()           # [B, TF] hard labels
    # 1. the oracle end to end: its own float64 embeddings and k-means.  The labels of float32 embeddings within rounding of a cluster
    #    boundary fall either way (bit-exact labels on EQUAL embeddings: tests/test_gpu_benchshape.py::test_kmeans_hard_at_benchmark_shape):
    #    counted here, and the cost still agrees
    info = {}
    c_own = orec.stft_enhance_loss(xm, xn, P, W, hop, NL, E, NLE, idx, tries, steps, nonlinearity='softmax', want_grads=False, info=info)
    differ = lab_dev != info['labels']
    print('cfg4 STFT_L41_enhance: %d of %d k-means labels differ from the float64 chain; cost %.6f oracle %.6f'
          % (int(differ.sum()), differ.size, cost, c_own))
    assert differ.sum() <= 1e-4 * differ.size
    assert abs(cost - c_own) < 1e-4 * abs(c_own)
    # 2. the trained part of the step (enhance stack, PIT cost, gradients, AMSGrad) on equal labels
    c_ref, g_ref = orec.stft_enhance_loss(xm, xn, P, W, hop, NL, E, NLE, idx, tries, steps, nonlinearity='softmax', labels=lab_dev)
    check_step(cost, c_ref, grads, g_ref, P, P_new, ooptim.AMSGrad(1e-3), what='cfg4 STFT_L41_enhance B=64')


@pytest.mark.parametrize('graph', [True])          # the replayed step (what tools/bench_configs.py times); eager: the other configs, 80 s here
def test_front_l41_step_at_cfg5_size(graph):
    """BASELINE configs[4]: front_L41, --nb_speakers 3 --filters 512, batch 128 per GPU: 512 signals x 80 frames through the front,
    ring input width 512, dense 600 -> 20480, L41 cost over 128 x 40960 points against 3 of 251 speaker vectors; as the replayed hipGraph."""
    from ams_hip import testing
    from models.L41 import L41Model
    from utils.trainer import Front_Separator_Trainer
    tmp = tempfile.mkdtemp(prefix='ams_full_l41_')
    B, S, W, N, hop, LS, NL, E = 128, 3, 1024, 512, 256, 600, 3, 40
    folder, params = testing.make_pretrained_adapt(os.path.join(tmp, 'pre'), window_size=W, filters=N, hop_size=hop, chunk_size=L,
                                                   batch_size=B, nb_speakers=S)
    a = base_args(**params)
    a.update(layer_size=LS, nb_layers=NL, embedding_size=E, model_folder=folder, model_previous=None, pretraining=False,
             learning_rate=1e-3, tot_speakers=251, hip_graph=graph)
    a.pop('type')
    tr = Front_Separator_Trainer(L41Model, 'front_L41', **a)
    dist, tfds = tr.prepare()
    opt = ooptim.AMSGrad(1e-3)
    P, cost, xm, xn, I, grads, P_new = _graphed_step(tr, tfds, opt) if graph else one_train_step(tr, tfds, L)
    assert xn.shape[:2] == (B, S) and P['prediction/W'].shape[-1] == N * E
    c_ref, g_ref, V, Y = ostep.front_l41_loss(xm, xn, I, P, hop, NL, E, True)
    assert V.size == B * 80 * N * E
    check_step(cost, c_ref, grads, g_ref, P, P_new, opt, what='cfg5 front_L41 B=128 graph=%s' % graph)
