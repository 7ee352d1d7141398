"""GPU parity: every HIP kernel, called through the C ABI (ctypes), against the float64 numpy oracle on
the same seeded inputs.  Spec tolerance (BASELINE.json north_star): 1e-3 relative fp32; the asserts below
use the much tighter bound an exact-f32 MFMA path should meet (`TOL`), so a layout bug cannot hide."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import front as ofront, blstm as oblstm, dense as odense, dpcl as odpcl, separate as osep, optim as ooptim

TOL = 2e-5          # observed fp32 round-off class; spec is 1e-3


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def rel(a, b):
    b = np.asarray(b, np.float64)
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.mark.parametrize('M,N,K,tA,tB', [
    (130, 70, 45, 0, 0), (130, 70, 45, 0, 1), (130, 70, 45, 1, 0), (130, 70, 45, 1, 1),
    (257, 1200, 600, 0, 0), (300, 1200, 2049, 1, 0), (512, 256, 1024, 0, 1), (64, 10240, 600, 0, 0),
    (33, 17, 5, 0, 0), (128, 128, 16, 0, 0), (600, 520, 5120, 1, 0),
    # branch-free 16-byte fetch path (all leading dimensions multiples of 4) with ragged tiles and a k-tail that is not a
    # multiple of the 8-deep tile; and its neighbours that must fall back to the guarded path (odd K / M / N)
    (132, 260, 604, 0, 0), (132, 260, 604, 0, 1), (132, 260, 604, 1, 0), (132, 260, 604, 1, 1),
    (4, 4, 4, 0, 0), (4, 8, 12, 1, 1), (260, 132, 2052, 1, 0), (132, 260, 603, 0, 1), (131, 260, 604, 1, 0), (132, 262, 604, 0, 0),
])
def test_gemm(ops, M, N, K, tA, tB):
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(*((K, M) if tA else (M, K)))
    B = rng.randn(*((N, K) if tB else (K, N)))
    bias = rng.randn(N)
    ref = (A.T if tA else A) @ (B.T if tB else B) + bias
    out = ops.gemm(dev(A), dev(B), transA=bool(tA), transB=bool(tB), bias=dev(bias))
    assert rel(host(out), ref) < TOL
    # accumulate on top
    C0 = rng.randn(M, N)
    c = dev(C0)
    ops.gemm(dev(A), dev(B), transA=bool(tA), transB=bool(tB), out=c, accumulate=True)
    assert rel(host(c), ref - bias + C0) < TOL


def test_gemm_strided_and_masked(ops):
    rng = np.random.RandomState(1)
    T, Bq, H = 7, 5, 12
    M = Bq * T
    out = rng.randn(M, 2 * H)
    dZ = rng.randn(M, 8 * H)
    # forward direction dU: sum_{t>=1} out[b,t-1,:H]^T dZ[b,t,:4H]
    ref = np.zeros((H, 4 * H))
    for b in range(Bq):
        for t in range(1, T):
            ref += np.outer(out[b * T + t - 1, :H], dZ[b * T + t, :4 * H])
    o, z = dev(out), dev(dZ)
    res = torch.empty((H, 4 * H), device='cuda', dtype=torch.float32)
    ops.gemm(o.view(-1), z.view(-1)[8 * H:], transA=True, out=res, M=H, N=4 * H, K=M - 1, lda=2 * H, ldb=8 * H, ldc=4 * H,
             mask=(T, T - 1))
    assert rel(host(res), ref) < TOL


@pytest.mark.parametrize('Bt,L,W,N,hop', [(3, 1000, 128, 8, 48), (6, 4096, 1024, 256, 256), (2, 2048, 256, 40, 64)])
def test_front_conv_and_filter(ops, Bt, L, W, N, hop):
    rng = np.random.RandomState(L)
    x, w, bases = rng.randn(Bt, L), rng.randn(W), rng.randn(W, N)
    f = ops.front_filter(dev(w), dev(bases))
    f_ref = ofront.front_filter(w, bases)
    assert rel(host(f), f_ref) < 1e-6
    y = ops.front_conv(dev(x), f, hop)
    y_ref = ofront.conv_strided(x, f_ref, hop)
    assert y.shape == y_ref.shape and rel(host(y), y_ref) < TOL
    dy = rng.randn(*y_ref.shape)
    df = ops.front_conv_bwd_filter(dev(x), dev(dy), W, hop)
    df_ref = ofront.conv_strided_bwd_filter(x, dy, W, hop)
    assert rel(host(df), df_ref) < TOL
    dw, db = ops.front_filter_bwd(dev(w), dev(bases), df)
    dw_ref, db_ref = ofront.front_filter_bwd(w, bases, df_ref)
    assert rel(host(dw), dw_ref) < TOL and rel(host(db), db_ref) < TOL


@pytest.mark.parametrize('Bt,L,W,N,hop', [(3, 1001, 128, 8, 48), (2, 1000, 128, 8, 50), (2, 1000, 126, 8, 48), (2, 1000, 128, 6, 48),
                                           (2, 1000, 128, 8, 48)])
def test_model_front_conv_at_shapes_the_16_byte_fetch_does_not_take(ops, Bt, L, W, N, hop):
    """functional.front_conv (what models/adapt.py calls) asks the launch to leave max |y| (it asked unconditionally and the
    scalar-fetch f32 form, which L, hop, pad_left, W or N not a multiple of 4 fall back to, rejects that -- arbitrary-length inference
    utterances raised AmsError).  Since ABI 4 the query takes the launch's geometry: such shapes run, get the right values, and their
    output carries no stale bound (the consumer measures); the aligned shape still gets its bound from the launch."""
    from ams_hip import functional as F
    rng = np.random.RandomState(L + W + N + hop)
    x, f = rng.randn(Bt, L), rng.randn(W, N)
    xd, fd = dev(x), dev(f)
    y = F.front_conv(xd, fd, hop)
    y_ref = ofront.conv_strided(x, f, hop)
    assert y.shape == y_ref.shape and rel(host(y), y_ref) < TOL
    T = -(-L // hop)
    pl = max((T - 1) * hop + W - L, 0) // 2
    takes = all(v % 4 == 0 for v in (L, hop, pl, W, N))
    tagged = getattr(y, '_ams_amax', None) is not None
    assert tagged == (takes and ops.F16X3)
    assert abs(float(ops.amax_of(y)) - np.abs(host(y)).max()) <= 1e-6 * np.abs(y_ref).max()


def test_make_masks(ops):
    rng = np.random.RandomState(2)
    B, S, T, F = 3, 3, 5, 7
    rep = rng.randn(B * S, T, F)
    rep[0, 0, 0] = rep[1, 0, 0] = 5.0                       # tie -> lowest index
    X_nm = rep.reshape(B, S, T, F).transpose(0, 2, 3, 1)
    for a, b, ab in ((1.0, 0.0, True), (1.0, -1.0, True), (1.0, 0.0, False)):
        Y_ref, am_ref = osep.make_masks(np.abs(X_nm) if ab else X_nm, a, b)
        Y, am = ops.make_masks(dev(rep), B, S, a, b, ab, want_argmax=True)
        torch.cuda.synchronize()
        assert np.array_equal(am.cpu().numpy().reshape(B, T, F), am_ref)
        assert np.array_equal(host(Y).reshape(B, T, F, S), Y_ref)


@pytest.mark.parametrize('B,S,TF,E,a,b', [(3, 2, 1000, 40, 1.0, 0.0), (2, 3, 4097, 40, 1.0, 0.0), (5, 2, 77, 8, 1.0, 0.25),
                                          (2, 4, 2561, 20, 0.7, 0.1)])
def test_make_masks_counted_for_the_fused_loss(ops, B, S, TF, E, a, b):
    """ams_dpcl_u_make_masks = ams_make_masks + ams_dpcl_u_count_labels in one pass (what a plugged DPCL training step runs): same
    labels, and the fused loss that picks the prepared workspace up returns the bits of the loss that counted by itself -- also for
    label values whose sums are not integers (the summation order is shared)."""
    rng = np.random.RandomState(B * TF + S)
    rep = dev(rng.randn(B * S, TF))
    U = dev(rng.randn(B, TF, E))
    Y0 = ops.make_masks(rep, B, S, a, b, True)
    out0, inv0, _, ws0 = ops.dpcl_loss_fwd_u(U, Y0)
    du0 = ops.dpcl_loss_bwd_u(U, Y0, inv0, ws0)
    for _ in range(2):
        Y1 = ops.make_masks(rep, B, S, a, b, True, dpcl_E=E)
        assert ops._DPCL_AHEAD[0] is not None and ops._DPCL_AHEAD[0][0] == Y1.data_ptr()
        ws_prepared = ops._DPCL_AHEAD[0][3]
        out1, inv1, _, ws1 = ops.dpcl_loss_fwd_u(U, Y1.reshape(B, -1, S))
        assert ws1 is ws_prepared and ops._DPCL_AHEAD[0] is None
        assert np.array_equal(host(Y1), host(Y0))
        assert np.array_equal(host(out1), host(out0)) and np.array_equal(host(inv1), host(inv0))
        assert np.array_equal(host(ops.dpcl_loss_bwd_u(U, Y1, inv1, ws1)), host(du0))
    # the oracle's loss on the same labels
    c_ref, _ = odpcl.dpcl_cost(odense.l2norm_fwd(host(U).reshape(B, -1), E)[0].reshape(B, TF, E), host(Y0))
    assert abs(float(host(out0)[0]) - c_ref) < TOL * max(1.0, abs(c_ref))


@pytest.mark.parametrize('ring', ['1', 'safe', '0'])
@pytest.mark.parametrize('B,T,D,H', [(5, 7, 12, 8), (20, 9, 24, 20), (3, 4, 16, 300), (17, 6, 10, 6), (33, 12, 8, 37), (4, 5, 6, 336),
                                       (2, 3, 4, 340), (6, 10, 600, 24), (5, 8, 256, 40), (3, 6, 644, 16)])
def test_blstm_layer(ops, monkeypatch, B, T, D, H, ring):
    """ring = '1': chain-per-XCD ring recurrence (csrc/lstm_ring.hip; plain-store hand-off where the chain shares an L2),
    'safe': the same with the placement-independent write-through hand-off forced, '0': per-step kernels (csrc/lstm.hip).
    H = 340 exceeds the ring's register-resident weight budget and takes the per-step path in every mode."""
    monkeypatch.setattr(ops, 'LSTM_RING', ring)
    assert (ops.load().ams_blstm_ring_sync_bytes(B, H, 0) != 0) == (H <= 336)
    rng = np.random.RandomState(B * T + H)
    lim = np.sqrt(6.0 / (D + 5 * H))
    x = rng.randn(B, T, D)
    Kf, Kb = rng.uniform(-lim, lim, (D + H, 4 * H)) * 3, rng.uniform(-lim, lim, (D + H, 4 * H)) * 3
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    out_ref, cache = oblstm.blstm_fwd(x, Kf, bf, Kb, bb)
    xd, Kfd, Kbd = dev(x), dev(Kf), dev(Kb)
    out, G, cst = ops.blstm_fwd(xd, Kfd, dev(bf), Kbd, dev(bb))
    assert rel(host(out), out_ref) < TOL
    dout = rng.randn(B, T, 2 * H)
    dx_ref, (dKf_r, dbf_r, dKb_r, dbb_r) = oblstm.blstm_bwd(dout, cache)
    dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(xd, Kfd, Kbd, out, G, cst, dev(dout))
    assert rel(host(dx), dx_ref) < 5 * TOL
    assert rel(host(dKf), dKf_r) < 5 * TOL and rel(host(dKb), dKb_r) < 5 * TOL
    assert rel(host(dbf), dbf_r) < 5 * TOL and rel(host(dbb), dbb_r) < 5 * TOL
    assert ops.persist_errors() == 0                            # no bounded in-launch wait timed out


@pytest.mark.parametrize('ring', ['1', 'safe'])
@pytest.mark.parametrize('B,T,D,H', [(5, 7, 12, 8), (20, 9, 24, 20), (3, 4, 16, 300), (33, 12, 8, 37), (4, 5, 6, 336), (64, 6, 40, 300),
                                       (6, 10, 600, 24), (18, 5, 16, 130), (7, 6, 16, 200), (9, 4, 16, 250)])
def test_backward_ring_as_fp16x3_one_scale_per_row(ops, monkeypatch, B, T, D, H, ring):
    """ams_blstm_ring_bwd with the kernels' bound (amax_u): the recurrent product da . U^T runs as fp16x3, computed transposed so that
    every batch row of da is scaled by its own maximum.  Same tolerances as the f32 MFMA form -- and PER BATCH ROW, with upstream
    gradients whose rows span 2^-60 .. 2^20: a scale shared by the 16 rows of a tile would leave the small rows no bits at all."""
    monkeypatch.setattr(ops, 'LSTM_RING', ring)
    monkeypatch.setattr(ops, 'F16X3', True)
    rng = np.random.RandomState(B * T + H + 1)
    lim = np.sqrt(6.0 / (D + 5 * H))
    x = rng.randn(B, T, D)
    Kf, Kb = rng.uniform(-lim, lim, (D + H, 4 * H)) * 3, rng.uniform(-lim, lim, (D + H, 4 * H)) * 3
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    out_ref, cache = oblstm.blstm_fwd(x, Kf, bf, Kb, bb)
    xd, Kfd, Kbd = dev(x), dev(Kf), dev(Kb)
    bound_u = torch.maximum(ops.absmax(Kfd), ops.absmax(Kbd))
    row_scale = 2.0 ** (20 - 80.0 * rng.permutation(B) / max(B - 1, 1))            # one magnitude per batch row
    dout = rng.randn(B, T, 2 * H) * row_scale[:, None, None]
    dx_ref, (dKf_r, dbf_r, dKb_r, dbb_r) = oblstm.blstm_bwd(dout, cache)
    got = {}
    for name, au in (('fp16x3', bound_u), ('f32', None)):
        out, G, cst = ops.blstm_fwd(xd, Kfd, dev(bf), Kbd, dev(bb))
        dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(xd, Kfd, Kbd, out, G, cst, dev(dout), amax_u=au)
        got[name] = host(dx)
        # every batch row to the tolerance of the whole (dx rows are independent of one another)
        for b in range(B):
            assert rel(got[name][b], dx_ref[b]) < 5 * TOL, (name, b, rel(got[name][b], dx_ref[b]))
        assert rel(host(dKf), dKf_r) < 5 * TOL and rel(host(dKb), dKb_r) < 5 * TOL
        assert rel(host(dbf), dbf_r) < 5 * TOL and rel(host(dbb), dbb_r) < 5 * TOL
    assert not np.array_equal(got['fp16x3'], got['f32'])        # the arithmetic did change
    assert ops.persist_errors() == 0
    ops.raise_on_ring_errors()


@pytest.mark.parametrize('B,TF,E,S', [(2, 300, 8, 2), (3, 5000, 40, 2), (2, 2049, 40, 3), (2, 77, 3, 2), (1, 700, 20, 4),
                                        (2, 2561, 40, 8)])
def test_l2norm_dpcl(ops, B, TF, E, S):
    rng = np.random.RandomState(TF)
    u = rng.randn(B, TF * E)
    u[0, :E] = 0.0                                           # exercises the eps clamp
    V_ref, inv_ref = odense.l2norm_fwd(u, E)
    V, inv = ops.l2norm_fwd(dev(u), E)
    assert rel(host(V).reshape(V_ref.shape), V_ref) < 1e-6
    lab = rng.randint(0, S, (B, TF))
    Y = np.eye(S)[lab]
    c_ref, terms = odpcl.dpcl_cost(V_ref.reshape(B, TF, E), Y)
    Vd, Yd = V.view(B, TF, E), dev(Y)
    out, ws = ops.dpcl_loss_fwd(Vd, Yd)
    o = host(out)
    assert abs(o[0] - c_ref) < TOL * max(1.0, abs(c_ref))
    for k in range(3):
        assert abs(o[1 + k] - terms[k]) < TOL * max(1.0, abs(terms[k]))
    dV_ref = odpcl.dpcl_cost_bwd(V_ref.reshape(B, TF, E), Y)
    dV = ops.dpcl_loss_bwd(Vd, Yd, ws)
    assert rel(host(dV), dV_ref) < 5 * TOL
    du_ref = odense.l2norm_bwd(V_ref, inv_ref, dV_ref.reshape(V_ref.shape))
    du = ops.dpcl_loss_bwd(Vd, Yd, ws, inv=inv)
    assert rel(host(du).reshape(du_ref.shape), du_ref) < 5 * TOL
    du2 = ops.l2norm_bwd(V, inv, dV.view(B, -1), E)
    assert rel(host(du2).reshape(du_ref.shape), du_ref) < 5 * TOL
    # fused training form: one pass over u gives 1/|u|, the loss terms and (optionally) V; backward from u
    Ud = dev(u).view(B, TF, E)
    out_u, inv_u, V_u, ws_u = ops.dpcl_loss_fwd_u(Ud, Yd, want_V=True)
    ou = host(out_u)
    assert abs(ou[0] - c_ref) < TOL * max(1.0, abs(c_ref))
    for k in range(3):
        assert abs(ou[1 + k] - terms[k]) < TOL * max(1.0, abs(terms[k]))
    assert rel(host(inv_u).reshape(-1), inv_ref.reshape(-1)) < 1e-6
    assert rel(host(V_u).reshape(V_ref.shape), V_ref) < 1e-6
    out_n, inv_n, V_n, _ = ops.dpcl_loss_fwd_u(Ud, Yd)
    assert V_n is None and np.array_equal(host(out_n), ou) and np.array_equal(host(inv_n), host(inv_u))
    du_u = ops.dpcl_loss_bwd_u(Ud, Yd, inv_u, ws_u)
    assert rel(host(du_u).reshape(du_ref.shape), du_ref) < 5 * TOL
    # label counts issued ahead, on another stream (what a training step does beside the recurrence): same bits, and the forward
    # picked the prepared workspace up; repeated on the same workspace (arrival counters are re-armed by every count pass)
    side = torch.cuda.Stream()
    for _ in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ws_a = ops.dpcl_count_labels_ahead(Yd, E)
        torch.cuda.current_stream().wait_stream(side)
        out_a, inv_a, _, ws_b = ops.dpcl_loss_fwd_u(Ud, Yd)
        assert ws_b is ws_a and ops._DPCL_AHEAD[0] is None
        assert np.array_equal(host(out_a), ou) and np.array_equal(host(inv_a), host(inv_u))
        assert np.array_equal(host(ops.dpcl_loss_bwd_u(Ud, Yd, inv_a, ws_b)), host(du_u))
    ops.dpcl_count_labels_ahead(Yd[:1].contiguous(), E)            # counts of ANOTHER label tensor are not picked up
    out_c, _, _, ws_c = ops.dpcl_loss_fwd_u(Ud, Yd)
    assert ws_c is not ws_a and np.array_equal(host(out_c), ou)
    up = dev(np.array([0.5]))
    du_h = ops.dpcl_loss_bwd_u(Ud, Yd, inv_u, ws_u, upstream=up)
    assert rel(host(du_h).reshape(du_ref.shape), 0.5 * du_ref) < 5 * TOL


def test_dpcl_backward_staged_form_still_agrees():
    """E = 40 takes the direct (LDS-free) backward kernel by default; AMS_DPCL_LDS=1 (read once per process) selects the staged
    one of round 2 for the same shapes.  Both are held to the oracle."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', os.path.abspath(__file__), '-k', 'test_l2norm_dpcl'],
                       env=dict(os.environ, AMS_DPCL_LDS='1'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_optimizers(ops):
    rng = np.random.RandomState(5)
    n = 100003
    p0, g = rng.randn(n), rng.randn(3, n)
    p = dev(p0)
    m, v, vh = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    ref = ooptim.AMSGrad(0.01)
    pr = p0.copy()
    b1p, b2p = 0.9, 0.99
    for k in range(3):
        ref.apply([pr], [g[k]])
        lr_t = 0.01 * np.sqrt(1 - b2p) / (1 - b1p)
        ops.opt_amsgrad(p, dev(g[k]), m, v, vh, lr_t, 0.9, 0.99, 1e-3)
        b1p *= 0.9
        b2p *= 0.99
    assert rel(host(p), pr) < 1e-5
    p, ms = dev(p0), torch.ones(n, device='cuda', dtype=torch.float32)
    r, pr = ooptim.RMSProp(0.01), p0.copy()
    for k in range(3):
        r.apply([pr], [g[k]])
        ops.opt_rmsprop(p, dev(g[k]), ms, 0.01)
    assert rel(host(p), pr) < 1e-5
    p, acc = dev(p0), torch.zeros(n, device='cuda', dtype=torch.float32)
    r, pr = ooptim.Momentum(0.01), p0.copy()
    for k in range(3):
        r.apply([pr], [g[k]])
        ops.opt_momentum(p, dev(g[k]), acc, 0.01)
    assert rel(host(p), pr) < 1e-5
    assert abs(host(ops.sumsq(dev(g[0])))[0] - np.sum(g[0].astype(np.float32).astype(np.float64) ** 2)) < 1e-4 * n


def test_global_norm_clip_and_weight_bound_stay_on_the_device(ops):
    """tf.clip_by_global_norm (models/network.py:185-190) through FlatOptimizer without a host round trip: the factor
    clip / max(|g| * pre_scale, clip) is computed by ams_clip_scale and multiplied in by the optimizer kernel (grad_scale_dev); and the
    kernels leave max |p| of what they wrote (bound_out) -- the next step's fp16x3 weight bound -- exactly."""
    from ams_hip.optim import FlatOptimizer
    rng = np.random.RandomState(6)
    for kind, clip in (('SGD', 0.5), ('SGD', 1e9), ('RMSProp', 0.5), ('Adam', 0.5)):
        vs = [torch.nn.Parameter(dev(rng.randn(300, 40))), torch.nn.Parameter(dev(rng.randn(77)))]
        p0 = [host(v).copy() for v in vs]
        opt = FlatOptimizer(vs, kind, 0.1, 50, clip)
        g = [rng.randn(*v.shape) for v in vs]
        for v, gi in zip(vs, g):
            v.grad.copy_(dev(gi))
        gn = np.sqrt(sum((gi.astype(np.float32).astype(np.float64) ** 2).sum() for gi in g))
        k = clip / max(gn, clip)
        opt.step()
        src = opt.flat._ams_src if hasattr(opt.flat, '_ams_src') else None
        if kind == 'SGD':
            for v, a, gi in zip(vs, p0, g):
                assert rel(host(v), a - 0.1 * k * gi) < 1e-5, (kind, clip)
        else:                                                     # the clipped update equals the unclipped update of the scaled gradient
            vs2 = [torch.nn.Parameter(dev(a)) for a in p0]
            opt2 = FlatOptimizer(vs2, kind, 0.1, 50, 0.0)
            for v, gi in zip(vs2, g):
                v.grad.copy_(dev(gi * k))
            opt2.step()
            for v, w in zip(vs, vs2):
                assert rel(host(v), host(w)) < 1e-5, (kind, clip)
        if src is not None:
            assert float(src.bound) == float(opt.flat.abs().max()), kind


@pytest.mark.parametrize('M,N,K', [(600, 10240, 5120), (64, 128, 40), (600, 256, 5120), (16, 64, 3000), (132, 388, 777)])
def test_gemm_at_b_colsum(ops, M, N, K):
    """dW = x^T dY and db = colsum(dY) from one pass over dY (Conv1D gradients, utils/ops.py:501-503): the fused launch against the
    oracle's two separate formulas, with and without accumulation, across split-K counts (K = 5120 splits, K = 40 does not)."""
    rng = np.random.RandomState(M + N + K)
    A, Bm = rng.randn(K, M), rng.randn(K, N)
    C0, b0 = rng.randn(M, N), rng.randn(N)
    out, bsum = dev(C0), dev(b0)
    assert ops.gemm_at_b_colsum(dev(A), dev(Bm), out, bsum, accumulate=True)
    assert rel(host(out), C0 + A.T.dot(Bm)) < TOL
    assert rel(host(bsum), b0 + Bm.sum(0)) < TOL
    out2, bsum2 = torch.empty(M, N, device='cuda'), torch.empty(N, device='cuda')
    assert ops.gemm_at_b_colsum(dev(A), dev(Bm), out2, bsum2, accumulate=False)
    assert rel(host(out2), A.T.dot(Bm)) < TOL and rel(host(bsum2), Bm.sum(0)) < TOL
    # a shape the fused form does not take (N not a multiple of 4): the wrapper says so instead of guessing
    assert not ops.gemm_at_b_colsum(dev(A), dev(Bm[:, :N - 1].copy()), torch.empty(M, N - 1, device='cuda'), torch.empty(N - 1, device='cuda'))


@pytest.mark.parametrize('B,T,D,H,twin', [(5, 7, 12, 10, False), (20, 9, 16, 24, False), (33, 6, 8, 12, True)])
def test_blstm_under_dropout_wrappers(B, T, D, H, twin):
    """--recurrent_dropout != 0 (utils/ops.py:363,373,379): the per-step recurrence with state dropout on c and h
    (ams_blstm_recurrent_fwd/bwd_dropout) plus the input / output masks, against the oracle with the SAME masks: output, dx and all
    four parameter gradients.  twin: the two direction kernels row-interleaved as FlatOptimizer stores them."""
    from ams_hip import ops
    rng = np.random.RandomState(B + H)
    keep = 0.75
    x = rng.randn(B, T, D)
    Kf, Kb = rng.randn(D + H, 4 * H) * 0.4, rng.randn(D + H, 4 * H) * 0.4
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    g = torch.Generator(device='cuda').manual_seed(B)
    masks = ops.blstm_dropout_masks(B, T, D, H, keep, 'cuda', generator=g)
    mh = {k: host(v) for k, v in masks.items()}
    assert set(np.unique(mh['h']).round(6)) <= {0.0, round(1.0 / keep, 6)} and 0.6 < (mh['in'] > 0).mean() < 0.9
    om = tuple({'in': mh['in'][d], 'h': mh['h'][:, :, d], 'c': mh['c'][:, :, d], 'out': mh['out'][:, :, d * H:(d + 1) * H]} for d in (0, 1))
    out_ref, cache = oblstm.blstm_fwd(x, Kf, bf, Kb, bb, om)
    dout = rng.randn(B, T, 2 * H)
    dx_ref, (dKf_ref, dbf_ref, dKb_ref, dbb_ref) = oblstm.blstm_bwd(dout, cache)
    if twin:
        blk = torch.empty(D + H, 2, 4 * H, device='cuda')
        blk[:, 0].copy_(dev(Kf))
        blk[:, 1].copy_(dev(Kb))
        kf, kb = blk[:, 0], blk[:, 1]
    else:
        kf, kb = dev(Kf), dev(Kb)
    xd = dev(x)
    y, saved = ops.blstm_fwd_dropout(xd, kf, dev(bf), kb, dev(bb), masks)
    assert rel(host(y), out_ref) < TOL
    dx, dKf, dbf, dKb, dbb = ops.blstm_bwd_dropout(dev(dout), xd, kf, kb, saved, masks)
    for mine, ref in ((dx, dx_ref), (dKf, dKf_ref), (dbf, dbf_ref), (dKb, dKb_ref), (dbb, dbb_ref)):
        assert rel(host(mine), ref) < 5 * TOL
