"""GPU, BASELINE.json full sizes (SURVEY 8d cfg3 / cfg1 geometry): size-independent properties of the hot path, checked where the
oracle would take minutes -- utterance-shard invariance of the training step (what the N>1 data-parallel path relies on),
unit-norm embeddings, adjointness of the analysis / synthesis filterbank, STFT->iSTFT round trip, k-means invariants."""
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

B, S, L, W, N, HOP, LS, NL, E = 64, 2, 20480, 1024, 256, 256, 600, 3, 40


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.fixture(scope='module')
def full_model():
    from tests.smoke_step import build_front_dpcl
    tmp = tempfile.mkdtemp(prefix='ams_full_')
    trainer, tfds = build_front_dpcl(tmp, B=B, L=L, W=W, N=N, hop=HOP, layer_size=LS, nb_layers=NL, E=E, no_summaries=True)
    g = trainer.graph
    gen = torch.Generator(device='cpu').manual_seed(9)
    Wd = g.variables['prediction/W']
    Wd.data.copy_((torch.rand(Wd.shape, generator=gen) * 0.1 - 0.05).to(Wd.device))      # SURVEY 8(d) bench init
    return trainer, tfds


def _fetch_batch(trainer, tfds):
    g, model = trainer.graph, trainer.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        probe = model._feeds(feed, True)
        return [n.value(probe) for n in (model.x_mix, model.x_non_mix, model.I)]


def _cost_and_grads(trainer, tfds, batch, rows=None):
    """Forward + backward of the training objective on the given utterance rows of ONE fetched batch (no optimizer step)."""
    from ams_hip import functional as F
    g, model = trainer.graph, trainer.model
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        ins = batch if rows is None else [t[rows].contiguous() for t in batch]
        run = model._feeds(feed, True)
        for node, t in zip((model.x_mix, model.x_non_mix, model.I), ins):
            run.cache[id(node)] = t
        opt = model.optimize
        opt.zero_grad()
        cost = model.cost_model.value(run)
        cost.reshape(-1)[0].backward()
        F.OVERLAP.join()
        torch.cuda.synchronize()
        V = model.sepNet.prediction.value(run).detach()
        return float(cost.detach().reshape(-1)[0]), opt.flat_grad.clone(), V


def test_full_size_step_is_invariant_to_utterance_sharding(full_model):
    """cfg3(i), B=64: cost and flat gradient of the whole minibatch == mean over two 32-utterance shards (every
    loss on the path is a batch mean of per-utterance terms) -- the identity the RCCL all-reduce(sum)/world relies on."""
    trainer, tfds = full_model
    tfds.initialize(tfds.TRAIN)
    batch = _fetch_batch(trainer, tfds)
    c_all, g_all, V = _cost_and_grads(trainer, tfds, batch)
    c0, g0, _ = _cost_and_grads(trainer, tfds, batch, slice(0, B // 2))
    c1, g1, _ = _cost_and_grads(trainer, tfds, batch, slice(B // 2, B))
    assert np.isfinite(c_all)
    assert abs(c_all - 0.5 * (c0 + c1)) < 1e-5 * abs(c_all)
    gm = 0.5 * (g0 + g1)
    err = float((g_all - gm).abs().max() / g_all.abs().max())
    assert err < 2e-4, err                                   # different split-K / accumulation orders, fp32
    # embeddings: [B, T', F, E] unit vectors
    assert V.shape == (B, L // HOP, N, E)
    nrm = V.reshape(-1, E).norm(dim=1)
    assert float((nrm - 1).abs().max()) < 1e-5


def test_full_size_filterbank_adjointness(ops):
    """<analysis(x), y> == <x, synthesis(y)> for the same filters at [192, 20480] x (W=1024, hop=256, N=256): conv2d SAME and
    conv2d_transpose SAME are exact adjoints."""
    from ams_hip import functional as F
    gen = torch.Generator(device='cpu').manual_seed(3)
    Bt = B * (S + 1)
    x = torch.randn(Bt, L, generator=gen).cuda()
    f = (torch.randn(W, N, generator=gen) / 32).cuda()
    y = torch.randn(Bt, L // HOP, N, generator=gen).cuda()
    Ax = ops.front_conv(x, f, HOP)
    Aty = F.synth_strided(y, f, HOP, L)
    lhs = float((Ax.double() * y.double()).sum())
    rhs = float((x.double() * Aty.double().reshape(Bt, L)).sum())
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    # linearity at full size
    x2 = torch.randn(Bt, L, generator=gen).cuda()
    lin = ops.front_conv(x + 2 * x2, f, HOP) - (Ax + 2 * ops.front_conv(x2, f, HOP))
    assert float(lin.abs().max()) < 1e-3 * float(Ax.abs().max())


def test_full_size_stft_round_trip():
    """cfg1/cfg4 geometry (W=512, hop=256, L=20480): iSTFT(|STFT|, phase) reproduces the interior of the waveform; the first and
    last hop are covered by a single frame and are not reconstructed -- that behaviour is kept."""
    from ams_hip import functional as F
    Wd, hop = 512, 256
    gen = torch.Generator(device='cpu').manual_seed(4)
    x = torch.randn(B, L, generator=gen).cuda()
    mag, ph = F.stft_mag_phase(x, Wd, hop)
    T = 1 + (L - Wd) // hop
    assert mag.shape == (B, T, Wd // 2 + 1)
    out = F.istft(mag, ph, Wd, hop, 1)
    assert out.shape == (B, (T - 1) * hop + Wd)
    inner = slice(hop, (T - 1) * hop)
    err = float((out[:, inner] - x[:, inner]).abs().max())
    assert err < 1e-4, err


def test_full_size_kmeans_invariants(ops):
    """TF = 20480 bins, E = 40, 10 tries, 10 iterations (cfg3 inference geometry): labels in range, re-assigning the points to
    the returned centroids reproduces the returned labels (idempotence of the final assignment), planted clusters recovered."""
    gen = torch.Generator(device='cpu').manual_seed(5)
    TF, C, tries, steps, Bk = L // HOP * N, 2, 10, 10, 16
    centers = torch.randn(Bk, C, E, generator=gen)
    lab = torch.randint(0, C, (Bk, TF), generator=gen)
    X = (centers[torch.arange(Bk)[:, None], lab] + 0.4 * torch.randn(Bk, TF, E, generator=gen)).cuda()
    idx = torch.stack([torch.randperm(TF, generator=gen)[:C] for _ in range(Bk * tries)]).to(torch.int32).cuda()
    xn = ops.kmeans_normalize(X)
    cent, labels, best, _ = ops.kmeans_run(xn, idx, C, tries, steps)
    assert cent.shape == (Bk, C, E) and labels.shape == (Bk, TF)
    assert int(labels.min()) >= 0 and int(labels.max()) < C and int(best.min()) >= 0 and int(best.max()) < tries
    d = ((xn[:, :, None, :].double() - cent[:, None, :, :].double()) ** 2).sum(-1)
    relab = d.argmin(dim=-1).to(labels.dtype)
    agree = float((relab == labels).float().mean())
    assert agree > 0.9999, agree                              # ties / last-ulp distance differences only
    lc = labels.cpu().long()
    acc = torch.maximum((lc == lab).float().mean(1), (lc == 1 - lab).float().mean(1))
    assert float(acc.min()) > 0.99
