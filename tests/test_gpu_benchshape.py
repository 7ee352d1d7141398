"""GPU, BENCHMARK shapes (BASELINE.json configs[2], SURVEY 8d cfg3: B=64, S=2, L=20480, W=1024, hop=256, N=256 => T'=80, TF=20480;
3xBLSTM(600) => H=300; E=40) compared with the ORACLE -- not with properties.  These are the launch geometries bench.py times:
the chain-per-XCD LSTM grid (8 chains only at B=64), the 5120x10240x600 dense product with the fused l2norm+DPCL pass at
TF=20480, the split-K in-place-frames front conv at W=1024 / N=256 / 192 rows, hard k-means at L=20480 with 10 restarts, and the
whole front_DPCL training step at B=64 (twin-interleaved weights, side-stream capped weight-gradient products, fused AMSGrad).

Tolerances (relative to the largest reference entry): forward 1e-4, backward 2e-4 -- 80 dependent fp32 steps x 3 layers against a
float64 oracle; north_star allows 1e-3.  k-means: np.array_equal."""
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import front as ofront, blstm as oblstm, dense as odense, dpcl as odpcl, kmeans as okm, step as ostep, optim as ooptim

B, S, L, W, N, HOP, LS, NL, E = 64, 2, 20480, 1024, 256, 256, 600, 3, 40
T = L // HOP
H = LS // 2
FWD_TOL, BWD_TOL = 1e-4, 2e-4


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


@pytest.mark.parametrize('D,ring', [(600, '1'), (256, '1'), (600, 'safe'), (600, '0')])
def test_blstm_layer_at_benchmark_shape(ops, monkeypatch, D, ring):
    """One BLSTM layer, B=64, T=80, H=300, D=600 (layers 1-2) / 256 (layer 0): forward and full backward through the DEFAULT
    recurrence -- 8 chain-per-XCD rings of 25 workgroups (csrc/lstm_ring.hip) -- plus its write-through hand-off and the per-step
    kernels (AMS_LSTM_XCD=2 grid) it falls back to (reference utils/ops.py:358-383)."""
    import os
    assert os.environ.get('AMS_LSTM_XCD', '2') == '2'
    monkeypatch.setattr(ops, 'LSTM_RING', ring)
    assert ops.load().ams_blstm_ring_sync_bytes(B, H, 0) != 0 and ops.load().ams_blstm_ring_sync_bytes(B, H, 1) != 0
    rng = np.random.RandomState(D)
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
    print('blstm bench shape D=%d' % D, errs)
    assert e_fwd < FWD_TOL, errs
    assert max(v for k, v in errs.items() if k != 'out') < BWD_TOL, errs
    assert ops.persist_errors() == 0
    ops.raise_on_ring_errors()


@pytest.mark.parametrize('D', [600, 256])
def test_forward_ring_as_fp16x3_at_benchmark_shape(ops, D):
    """The same layer with the bounds a training step supplies (functional.BLSTMLayer: input bound, ONE bound over all kernels):
    input projection AND the ring's recurrent product run as fp16x3 (amax_a / amax_b of ams_gemm_f32, amax_u of ams_blstm_ring_fwd).  Same forward
    tolerance as the bf16x6 form; the two forms differ (the arithmetic did change) by less than that tolerance."""
    rng = np.random.RandomState(D)
    lim = np.sqrt(6.0 / (D + 5 * H))
    x = rng.randn(B, T, D) * 0.5
    Kf, Kb = rng.uniform(-lim, lim, (D + H, 4 * H)) * 2, rng.uniform(-lim, lim, (D + H, 4 * H)) * 2
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    out_ref, _ = oblstm.blstm_fwd(x, Kf, bf, Kb, bb)
    xd, Kfd, Kbd, bfd, bbd = dev(x), dev(Kf), dev(Kb), dev(bf), dev(bb)
    bound_w = torch.maximum(ops.absmax(Kfd), ops.absmax(Kbd))
    out16, _, _ = ops.blstm_fwd(xd, Kfd, bfd, Kbd, bbd, amax=(ops.absmax(xd), bound_w))
    out6, _, _ = ops.blstm_fwd(xd, Kfd, bfd, Kbd, bbd)
    e16, e6 = rel(host(out16), out_ref), rel(host(out6), out_ref)
    d = float((out16 - out6).abs().max())
    print('forward ring D=%d: fp16x3 %.2e, bf16x6 %.2e, difference %.2e' % (D, e16, e6, d))
    assert e16 < FWD_TOL and e6 < FWD_TOL, (e16, e6)
    assert 0.0 < d < FWD_TOL, d
    assert ops.persist_errors() == 0
    ops.raise_on_ring_errors()


def test_dense_l2norm_dpcl_at_benchmark_shape(ops):
    """Conv1D 600 -> E*F = 10240 (utils/ops.py:486-503) + fused l2-normalise + DPCL loss (models/dpcl.py:41-87) at TF=20480, E=40,
    for B=2 utterances (160 rows of the 5120-row product: same tiles, same N and K), forward and backward down to dW, db, dh."""
    from ams_hip import functional as F
    Bq = 2
    rng = np.random.RandomState(7)
    h = rng.randn(Bq, T, LS) * 0.5
    Wd = rng.uniform(-0.05, 0.05, (LS, E * N))
    bd = rng.randn(E * N) * 0.01
    lab = rng.randint(0, S, (Bq, T * N))
    lab[0, :5000] = 0                                           # unbalanced classes
    Y = np.eye(S)[lab]
    u_ref = odense.dense_fwd(h, Wd, bd)                          # [Bq, T, F*E]
    V_ref, inv_ref = odense.l2norm_fwd(u_ref.reshape(Bq, -1), E)
    Vf = V_ref.reshape(Bq, T * N, E)
    c_ref, terms = odpcl.dpcl_cost(Vf, Y)
    dV_ref = odpcl.dpcl_cost_bwd(Vf, Y)
    du_ref = odense.l2norm_bwd(V_ref, inv_ref, dV_ref.reshape(V_ref.shape)).reshape(u_ref.shape)
    dh_ref, dW_ref, db_ref = odense.dense_bwd(h, Wd, du_ref)

    ht, Wt, bt = (dev(a).requires_grad_(True) for a in (h, Wd, bd))
    u = F.dense(ht, Wt, bt)
    assert rel(host(u), u_ref) < 2e-5
    cost, all_terms = F.dpcl_loss_u(u, dev(Y), E)
    o = host(all_terms)
    assert abs(o[0] - c_ref) < 2e-5 * abs(c_ref), (o[0], c_ref)
    for k in range(3):
        assert abs(o[1 + k] - terms[k]) < 2e-5 * max(1.0, abs(terms[k]))
    cost.backward(torch.ones(1, device='cuda'))
    F.OVERLAP.join()
    errs = {'dh': rel(host(ht.grad), dh_ref), 'dW': rel(host(Wt.grad), dW_ref), 'db': rel(host(bt.grad), db_ref)}
    print('dense+dpcl bench shape', errs)
    assert max(errs.values()) < BWD_TOL, errs
    # embeddings as the inference side materialises them
    V, inv = ops.l2norm_fwd(u.detach().reshape(Bq, -1), E)
    assert rel(host(V).reshape(V_ref.shape), V_ref) < 2e-5


def test_front_conv_at_benchmark_shape(ops):
    """Adapt.front path A (models/adapt.py:95-134) at the benchmark launch: 192 rows x 20480 samples, W=1024, hop=256, N=256 (the
    split-K in-place-frames product) and its filter gradient."""
    Bt = B * (S + 1)
    rng = np.random.RandomState(5)
    x = rng.randn(Bt, L) * 0.05
    w, bases = rng.uniform(-0.05, 0.05, W), rng.uniform(-0.07, 0.07, (W, N))
    f_ref = ofront.front_filter(w, bases)
    y_ref = ofront.conv_strided(x, f_ref, HOP)
    f = ops.front_filter(dev(w), dev(bases))
    y = ops.front_conv(dev(x), f, HOP)
    assert y.shape == (Bt, T, N)
    e_y = rel(host(y), y_ref)
    dy = rng.randn(*y_ref.shape) * 0.1
    df = ops.front_conv_bwd_filter(dev(x), dev(dy), W, HOP)
    df_ref = ofront.conv_strided_bwd_filter(x, dy, W, HOP)
    e_df = rel(host(df), df_ref)
    print('front conv bench shape', e_y, e_df)
    assert e_y < 2e-5 and e_df < BWD_TOL


def test_kmeans_hard_at_benchmark_shape(ops):
    """models/Kmeans_2.py:145-188 at L = TF = 20480, E = 40, C = 2, 10 restarts x 10 iterations, silence weights on, final
    re-assignment on: labels / centroids / best restart IDENTICAL to the float32 oracle."""
    from ams_hip import functional as F
    b, C, tries, iters = 2, 2, 10, 10
    rng = np.random.RandomState(31)
    centers = rng.randn(C, E).astype(np.float32) * 1.5
    lab_true = rng.randint(0, C, (b, T * N))
    X = (centers[lab_true] + rng.randn(b, T * N, E).astype(np.float32) * 0.9).astype(np.float32)
    w = (rng.rand(b, T * N) > 0.2).astype(np.float32)
    idx = np.stack([rng.choice(T * N, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    cent_ref, lab_ref, best_ref = okm.kmeans(X, idx, C, tries, iters, beta=None, notsilent=w, assign_at_end=True)
    cent, lab, best = F.kmeans(dev(X), dev(idx, np.int32), C, tries, iters, None, dev(w), True)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert np.array_equal(cent.cpu().numpy(), cent_ref)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)


@pytest.fixture()
def gemm_arith():
    """Switch the arithmetic of the dense products for one test (csrc/gemm.hip: 1 = bf16x6, the default; 0 = native f32 MFMA)."""
    from ams_hip._lib import load
    lib = load()
    before = lib.ams_gemm_get_arith()
    yield lib.ams_gemm_set_arith
    lib.ams_gemm_set_arith(before)


@pytest.mark.parametrize('hip_graph,arith', [(False, 1), (True, 1), (False, 0)])
def test_front_dpcl_step_at_benchmark_shape(hip_graph, arith, gemm_arith, monkeypatch):
    """The step bench.py times -- front_DPCL, B=64, full geometry -- against the float64 oracle: cost, every gradient, every
    updated weight after AMSGrad; eager and as the replayed hipGraph (3rd call = first replay), with the default bf16x6 products
    and, eager, with the native f32 MFMA products: the SAME tolerances hold for both arithmetics."""
    from tests.smoke_step import build_front_dpcl
    gemm_arith(arith)
    tmp = tempfile.mkdtemp(prefix='ams_benchshape_')
    trainer, tfds = build_front_dpcl(tmp, B=B, L=L, W=W, N=N, hop=HOP, layer_size=LS, nb_layers=NL, E=E, no_summaries=True,
                                     hip_graph=hip_graph)
    g, model = trainer.graph, trainer.model
    gen = torch.Generator(device='cpu').manual_seed(9)
    Wd = g.variables['prediction/W']
    Wd.data.copy_((torch.rand(Wd.shape, generator=gen) * 0.1 - 0.05).to(Wd.device))      # SURVEY 8(d) bench init
    with g.as_default():
        feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
        warm = 2 if hip_graph else 0                   # two eager steps on the capture stream, then capture + first replay
        opt_ref = ooptim.AMSGrad(1e-3)
        for it in range(warm + 1):
            P = {n: v.detach().cpu().numpy().astype(np.float64) for n, v in g.variables.items()}
            cost = float(model.train(feed, it))
            if it < warm:                              # keep the oracle's AMSGrad slots in step (fed with the device's gradients:
                names_w = sorted(v.ams_name for v in model.trainable_variables)     # the warm-up steps are not what is checked)
                gw = {v.ams_name: v.grad.detach().cpu().numpy().astype(np.float64) for v in model.trainable_variables}
                opt_ref.apply([P[n].copy() for n in names_w], [gw[n] for n in names_w])
        run = model.last_run
        xm = model.x_mix.value(run).cpu().numpy().astype(np.float64)
        xn = model.x_non_mix.value(run).cpu().numpy().astype(np.float64)
        grads = {v.ams_name: v.grad.detach().cpu().numpy() for v in model.trainable_variables}
        P_new = {v.ams_name: v.detach().cpu().numpy() for v in model.trainable_variables}
    c_ref, g_ref, V_ref, Y_ref = ostep.front_dpcl_loss(xm, xn, P, HOP, NL, E)
    names = sorted(g_ref)
    assert sorted(grads) == names
    errs = {'cost': abs(cost - c_ref) / abs(c_ref)}
    for n in names:
        errs['grad ' + n] = rel(grads[n], g_ref[n])
    # The fused AMSGrad kernel is checked on the DEVICE's gradients: with eps = 1e-3 (network.py:181-182) the update of an element
    # whose |g| is far below the tensor's largest is ~linear in g, so a gradient difference of 1e-5 of the tensor's max is a
    # percent-level difference of that element's update -- gradient parity is asserted above, optimizer parity here.
    plist = [P[n].copy() for n in names]
    opt_ref.apply(plist, [grads[n].astype(np.float64) for n in names])
    for n, p in zip(names, plist):
        errs['update ' + n] = rel(P_new[n], p)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print('front_DPCL B=64 step (hip_graph=%s, arith=%d): cost %.6f oracle %.6f; worst' % (hip_graph, arith, cost, c_ref), worst)
    assert errs['cost'] < FWD_TOL, worst
    assert max(v for k, v in errs.items() if k.startswith('grad ')) < BWD_TOL, worst
    assert max(v for k, v in errs.items() if k.startswith('update ')) < 1e-5, worst


@pytest.mark.parametrize('arith', ['fp16x3', 'bf16x6_f32'])
def test_ring_recurrence_is_bit_stable_under_uneven_load(ops, arith):
    """The in-launch hand-off of csrc/lstm_ring.hip (plain stores through the chain's L2 + L1-bypassing loads; granule tags forward,
    phase bits in the partial tiles backward) must
    not depend on what else the chip is doing: a stale or torn read would change bits.  One BLSTM layer at the benchmark shape,
    forward + BPTT, 12 times while a second stream keeps the CUs busy with large products (uneven: the load starts and stops at
    random points of the recurrence) -- every repetition must reproduce the idle run bit for bit, and no bounded wait may time out.
    arith: 'fp16x3' = both rings with the kernels' bound, as a training step runs them (forward fp16x3, backward fp16x3 with one scale
    per batch row); 'bf16x6_f32' = no bounds (forward bf16x6, backward on the f32 MFMAs).
    The side products are launched the way the product launches EVERYTHING that runs beside a ring (ams_hip/functional.py: side
    stream, residency cap): a ring needs all its workgroups resident at once, and an uncapped bf16x6 product (8 waves x ~230
    VGPRs, one workgroup per CU) admits no ring workgroup on a CU it occupies -- with several of those queued the ring's bounded
    wait CAN run out (observed: error word set, garbage output; ops.raise_on_ring_errors() is what the trainer calls)."""
    D = 256
    rng = np.random.RandomState(11)
    lim = np.sqrt(6.0 / (D + 5 * H))
    x = dev(rng.randn(B, T, D) * 0.5)
    Kf, Kb = dev(rng.uniform(-lim, lim, (D + H, 4 * H)) * 2), dev(rng.uniform(-lim, lim, (D + H, 4 * H)) * 2)
    bf, bb = dev(rng.randn(4 * H) * 0.1), dev(rng.randn(4 * H) * 0.1)
    dout = dev(rng.randn(B, T, 2 * H) * 0.1)
    assert ops.LSTM_RING == '1'
    from ams_hip._lib import load
    lib = load()

    bound_w = torch.maximum(ops.absmax(Kf), ops.absmax(Kb)) if arith == 'fp16x3' else None
    bound_x = ops.absmax(x) if arith == 'fp16x3' else None

    def layer():
        out, G, cst = ops.blstm_fwd(x, Kf, bf, Kb, bb, amax=(bound_x, bound_w) if bound_w is not None else None)
        grads = ops.blstm_bwd(x, Kf, Kb, out, G, cst, dout, amax_u=bound_w)
        return [out] + list(grads)
    ref = [t.clone() for t in layer()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 2048, device='cuda')
    bmat = torch.randn(2048, 4096, device='cuda')
    gen = np.random.RandomState(3)
    for rep in range(12):
        n_before, n_during = int(gen.randint(0, 3)), int(gen.randint(1, 6))
        with ops.lds_pad(50000), torch.cuda.stream(side):       # the product's residency cap for launches beside a ring
            for _ in range(n_before + n_during):
                ops.gemm(a, bmat)
        got = layer()
        torch.cuda.synchronize()
        for i, (g, r) in enumerate(zip(got, ref)):
            assert torch.equal(g, r), ('repetition %d: tensor %d differs from the idle run (max |diff| %.3e, ring error word %d)'
                                       % (rep, i, float((g - r).abs().max()), ops.persist_errors()))
    assert ops.persist_errors() == 0
    ops.raise_on_ring_errors()


def test_forward_ring_on_the_f32_pipe_still_matches():
    """The forward ring's recurrent product runs as bf16x6 on v_mfma_f32_16x16x32_bf16 by default (csrc/lstm_ring.hip, X6; every test
    above ran that).  AMS_LSTM_RING_X6=0 selects the round-2 form on v_mfma_f32_16x16x4_f32 (read once per process): the same
    benchmark-shape layer test in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_benchshape.py'), '-q', '-x', '-k',
                        'test_blstm_layer_at_benchmark_shape and (600-1 or 256-1)'], env=dict(os.environ, AMS_LSTM_RING_X6='0'),
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and '2 passed' in r.stdout, r.stdout[-1500:]
