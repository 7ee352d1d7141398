"""GPU, the FULL-SIZE launch geometries of the BASELINE configs other than cfg3(i), each compared with the ORACLE (not only timed):

  cfg5      B=128 => 16 chains of the ring recurrence = 400 workgroups at 2 per CU, second chain group; dense 600 -> 512*40 = 20480
            columns; L41 loss with S=3 at TF = 80*512 = 40960                                      (models/L41.py:150-178)
  cfg4      STFT geometry F=257, T=79: dense 600 -> 10280 (not a multiple of any tile width) + l2norm + L41 loss, T odd
  cfg2 (B)  fused stride-1 conv + max-pool at W=1024 / N=256 / P=256 over whole 20480-sample rows, its sparse back end and the two
            gather gradients                                                                        (models/adapt.py:115-117, 210-243)
  cfg3(ii)  soft k-means forward + backward at L = TF = 20480, beta=10, 10 unrolled iterations, silence weights, final
            re-assignment                                                                           (models/Kmeans_2.py:86-188)

Tolerances are test_gpu_benchshape.py's (relative to the largest reference entry): forward 1e-4, backward 2e-4; the oracle is
float64 (parity vs the in-repo restatement of the reference; TensorFlow itself is not available -- DESIGN.md 2)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import front as ofront, blstm as oblstm, dense as odense, l41 as ol41

FWD_TOL, BWD_TOL = 1e-4, 2e-4
T, H, LS, E = 80, 300, 600, 40


def dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=dtype)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.mark.parametrize('ring', ['1', 'safe'])
def test_blstm_layer_at_cfg5_batch(ops, monkeypatch, ring):
    """One BLSTM layer at B=128, T=80, H=300, D=600 (cfg5: 16 chains = 400 ring workgroups, two per CU, the second chain group)
    forward + full backward against the float64 oracle, plain and write-through hand-off (utils/ops.py:358-383)."""
    B, D = 128, 600
    monkeypatch.setattr(ops, 'LSTM_RING', ring)
    assert ops.load().ams_blstm_ring_sync_bytes(B, H, 0) != 0 and ops.load().ams_blstm_ring_sync_bytes(B, H, 1) != 0, \
        'B=128 must run on the ring recurrence (the geometry under test), not on the per-step fallback'
    rng = np.random.RandomState(128)
    lim = np.sqrt(6.0 / (D + 5 * H))
    x = rng.randn(B, T, D) * 0.5
    Kf, Kb = rng.uniform(-lim, lim, (D + H, 4 * H)) * 2, rng.uniform(-lim, lim, (D + H, 4 * H)) * 2
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    out_ref, cache = oblstm.blstm_fwd(x, Kf, bf, Kb, bb)
    xd, Kfd, Kbd = dev(x), dev(Kf), dev(Kb)
    out, G, cst = ops.blstm_fwd(xd, Kfd, dev(bf), Kbd, dev(bb))
    e_fwd = rel(host(out), out_ref)
    dout = rng.randn(B, T, 2 * H) * 0.1
    dx_ref, (dKf_r, dbf_r, dKb_r, dbb_r) = oblstm.blstm_bwd(dout, cache)
    dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(xd, Kfd, Kbd, out, G, cst, dev(dout))
    errs = {'out': e_fwd, 'dx': rel(host(dx), dx_ref), 'dKf': rel(host(dKf), dKf_r), 'dKb': rel(host(dKb), dKb_r),
            'dbf': rel(host(dbf), dbf_r), 'dbb': rel(host(dbb), dbb_r)}
    print('blstm B=128 ring=%s' % ring, errs)
    assert ops.persist_errors() == 0
    ops.raise_on_ring_errors()
    assert e_fwd < FWD_TOL, errs
    assert max(v for k, v in errs.items() if k != 'out') < BWD_TOL, errs


def _dense_l41(Fq, Tq, S, seed):
    """Conv1D 600 -> E*F (utils/ops.py:486-503), reshape (column f*E+e), l2-normalise, L41 cost on 2 utterances, down to
    dW, db, dh and the speaker-vector gradient."""
    from ams_hip import functional as F
    Bq, NS = 2, 23
    rng = np.random.RandomState(seed)
    h = rng.randn(Bq, Tq, LS) * 0.5
    Wd = rng.uniform(-0.05, 0.05, (LS, E * Fq))
    bd = rng.randn(E * Fq) * 0.01
    spk = rng.randn(NS, E)
    I = np.stack([rng.choice(NS, S, replace=False) for _ in range(Bq)]).astype(np.int32)
    lab = rng.randint(0, S, (Bq, Tq, Fq))
    y = np.where(np.eye(S)[lab] > 0, 1.0, -1.0)                      # L41 masks: +1 dominant, -1 others (L41.py:9-10)
    u_ref = odense.dense_fwd(h, Wd, bd)
    V_ref, inv_ref = odense.l2norm_fwd(u_ref.reshape(Bq, -1), E)
    emb_ref = V_ref.reshape(Bq, Tq, Fq, E)
    c_ref = ol41.l41_cost(emb_ref, y, spk, I, True)
    de_ref, ds_ref = ol41.l41_cost_bwd(emb_ref, y, spk, I, True)
    du_ref = odense.l2norm_bwd(V_ref, inv_ref, de_ref.reshape(V_ref.shape)).reshape(u_ref.shape)
    dh_ref, dW_ref, db_ref = odense.dense_bwd(h, Wd, du_ref)

    ht, Wt, bt, st = (dev(a).requires_grad_(True) for a in (h, Wd, bd, spk))
    u = F.dense(ht, Wt, bt)
    e_u = rel(host(u), u_ref)
    V = F.l2norm(u.reshape(Bq, -1), E)
    e_v = rel(host(V), V_ref)
    c = F.l41_loss(V.reshape(Bq, Tq, Fq, E), dev(y), st, dev(I, np.int32), True)
    e_c = abs(float(c.detach()) - c_ref) / abs(c_ref)
    c.backward()
    F.OVERLAP.join()
    errs = {'u': e_u, 'V': e_v, 'cost': e_c, 'dh': rel(host(ht.grad), dh_ref), 'dW': rel(host(Wt.grad), dW_ref),
            'db': rel(host(bt.grad), db_ref), 'dspk': rel(host(st.grad), ds_ref)}
    return errs


def test_dense_l41_at_cfg5_width():
    """cfg5: F=512 => dense 600 -> 20480, TF = 40960, S=3."""
    errs = _dense_l41(512, T, 3, 55)
    print('dense+L41 cfg5 width', errs)
    assert max(errs[k] for k in ('u', 'V', 'cost')) < FWD_TOL, errs
    assert max(errs[k] for k in ('dh', 'dW', 'db', 'dspk')) < BWD_TOL, errs


def test_dense_l41_at_cfg4_stft_geometry():
    """cfg4: STFT W=512 => F=257, T=79: dense 600 -> 10280 (ragged in every tile configuration), TF = 20303, S=2."""
    errs = _dense_l41(257, 79, 2, 44)
    print('dense+L41 cfg4 geometry', errs)
    assert max(errs[k] for k in ('u', 'V', 'cost')) < FWD_TOL, errs
    assert max(errs[k] for k in ('dh', 'dW', 'db', 'dspk')) < BWD_TOL, errs


def test_maxpool_front_at_cfg2_geometry():
    """cfg2 path B at its own geometry: 4 whole rows of L=20480, W=1024, N=256, P=hop=256 => T'=80: fused conv+max-pool values and
    arg-max, the gather filter gradient, the sparse synthesis and its two gradients."""
    from ams_hip import functional as F
    Bt, L, W, N, P, hop, S = 4, 20480, 1024, 256, 256, 256, 1
    rng = np.random.RandomState(2)
    x, f = rng.randn(Bt, L), rng.randn(W, N) / np.sqrt(W)
    x32, f32_ = x.astype(np.float32).astype(np.float64), f.astype(np.float32).astype(np.float64)
    ft = dev(f).requires_grad_()
    y, am = F.front_maxpool(dev(x), ft, P, hop)
    y_ref, am_ref = ofront.front_maxpool(x32, f32_, P, hop)
    Tq = (L - P) // hop + 1
    assert y.shape == (Bt, Tq, N)
    e_y = rel(host(y), y_ref)
    amh = am.cpu().numpy()
    same = float((amh == am_ref).mean())
    # where the index differs, two conv outputs tie to fp32 round-off: the value AT our index must still be the window's maximum
    X = None
    if same < 1.0:
        X = ofront.conv_dense(x32, f32_)
        b, t, n = np.nonzero(amh != am_ref)
        ours = X[b, amh[b, t, n] // N, n]
        assert np.all(np.abs(ours - y_ref[b, t, n]) <= 1e-5 * np.abs(y_ref).max()), 'arg-max differs by more than a round-off tie'
        assert np.all((amh[b, t, n] // N >= t * hop) & (amh[b, t, n] // N < t * hop + P))
    dy = rng.randn(Bt, Tq, N)
    y.backward(dev(dy))
    e_df = rel(host(ft.grad), ofront.front_maxpool_bwd_filter(x32, dy, amh, W))
    vals, f2, dout = rng.randn(Bt * S, Tq, N), rng.randn(W, N) / np.sqrt(W), rng.randn(Bt * S, L)
    am_t = np.repeat(amh, S, axis=0)
    vt, f2t = dev(vals).requires_grad_(), dev(f2).requires_grad_()
    out = F.synth_unpool(vt, am, f2t, L, S, P, hop)
    e_out = rel(host(out), ofront.synth_unpool(vals, am_t, f2, L))
    out.backward(dev(dout))
    dv_ref, df2_ref = ofront.synth_unpool_bwd(vals, am_t, f2, dout)
    errs = {'y': e_y, 'argmax_same': same, 'df': e_df, 'out': e_out, 'dvals': rel(host(vt.grad), dv_ref), 'df2': rel(host(f2t.grad), df2_ref)}
    print('max-pool front cfg2 geometry', errs)
    assert same > 0.999 and e_y < 2e-5 and e_out < FWD_TOL, errs
    assert max(errs[k] for k in ('df', 'dvals', 'df2')) < BWD_TOL, errs


def test_soft_kmeans_at_cfg3_finetuning_geometry():
    """cfg3(ii) front_DPCL_finetuning's k-means: L = TF = 20480 bins, E=40, C=2, one try, 10 unrolled soft iterations at beta=10,
    silence weights, final re-assignment -- labels, centroids and dX against float64 torch autograd of the restated reference
    (tests/test_gpu_kmeans_soft.py::torch_soft_kmeans)."""
    from ams_hip import functional as F
    from tests.test_gpu_kmeans_soft import torch_soft_kmeans
    b, L, C, tries, iters, beta = 2, 20480, 2, 1, 10, 10.0
    rng = np.random.RandomState(20480)
    centers = rng.randn(C, E) * 1.5
    X = centers[rng.randint(0, C, (b, L))] + rng.randn(b, L, E) * 0.8
    w = (rng.rand(b, L) > 0.2).astype(np.float64)
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)])
    R1, R2 = rng.randn(b, L, C), rng.randn(b, C, E)
    Xt = torch.from_numpy(X).requires_grad_()
    sel_r, out_r, best_r = torch_soft_kmeans(Xt, torch.from_numpy(idx), C, tries, iters, beta, torch.from_numpy(w), True)
    ((out_r * torch.from_numpy(R1)).sum() + (sel_r * torch.from_numpy(R2)).sum()).backward()
    Xd = dev(X).requires_grad_()
    sel, out, best = F.kmeans(Xd, dev(idx, np.int32), C, tries, iters, beta, dev(w), True)
    assert np.array_equal(best.cpu().numpy(), best_r.numpy())
    e_lab = float(np.abs(host(out) - out_r.detach().numpy()).max())
    e_cent = rel(host(sel), sel_r.detach().numpy())
    ((out * dev(R1)).sum() + (sel * dev(R2)).sum()).backward()
    e_g = rel(host(Xd.grad), Xt.grad.numpy())
    print('soft k-means L=20480 beta=10: labels %.2e centroids %.2e dX %.2e' % (e_lab, e_cent, e_g))
    assert e_lab < 1e-3 and e_cent < 1e-4 and e_g < 1e-3
