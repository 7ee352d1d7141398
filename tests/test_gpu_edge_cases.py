"""GPU: degenerate shapes and error behaviour of the C-ABI wrappers (smallest sizes, ragged tails, reference NaN quirks)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import blstm as oblstm, dpcl as odpcl, dense as odense, front as ofront, kmeans as okm

TOL = 2e-5


def dev(x, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=dtype)).cuda()


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


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (1, 300, 7), (129, 1, 129), (2, 2, 4097)])
def test_gemm_smallest_shapes(ops, M, N, K):
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A, B = rng.randn(M, K), rng.randn(K, N)
    for tA in (False, True):
        for tB in (False, True):
            out = ops.gemm(dev(A.T if tA else A), dev(B.T if tB else B), transA=tA, transB=tB)
            assert rel(host(out), A @ B) < TOL


def test_blstm_single_step_single_row(ops):
    rng = np.random.RandomState(0)
    for (B, T, D, H) in [(1, 1, 3, 4), (1, 5, 7, 300), (33, 1, 16, 8)]:
        x = rng.randn(B, T, D)
        Kf, Kb = rng.randn(D + H, 4 * H) * 0.3, rng.randn(D + H, 4 * H) * 0.3
        bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
        ref, cache = oblstm.blstm_fwd(x, Kf, bf, Kb, bb)
        out, G, cst = ops.blstm_fwd(dev(x), dev(Kf), dev(bf), dev(Kb), dev(bb))
        assert rel(host(out), ref) < TOL
        dout = rng.randn(B, T, 2 * H)
        dx_ref, (dKf_r, dbf_r, dKb_r, dbb_r) = oblstm.blstm_bwd(dout, cache)
        dx, dKf, dbf, dKb, dbb = ops.blstm_bwd(dev(x), dev(Kf), dev(Kb), out, G, cst, dev(dout))
        assert rel(host(dx), dx_ref) < 5 * TOL and rel(host(dKf), dKf_r) < 5 * TOL and rel(host(dbb), dbb_r) < 5 * TOL


def test_dpcl_single_point_and_single_utterance(ops):
    rng = np.random.RandomState(1)
    for (B, TF, E, S) in [(1, 1, 8, 2), (1, 3, 40, 2), (2, 255, 40, 3)]:
        u = rng.randn(B, TF * E)
        lab = rng.randint(0, S, (B, TF))
        lab[:, :S] = np.arange(S)[:min(S, TF)] if TF >= S else lab[:, :S]
        Y = np.eye(S)[lab]
        V_ref, inv_ref = odense.l2norm_fwd(u, E)
        c_ref, _ = odpcl.dpcl_cost(V_ref.reshape(B, TF, E), Y)
        out, inv, _, ws = ops.dpcl_loss_fwd_u(dev(u).view(B, TF, E), dev(Y))
        assert abs(host(out)[0] - c_ref) < 1e-4 * max(1.0, abs(c_ref))
        du_ref = odense.l2norm_bwd(V_ref, inv_ref, odpcl.dpcl_cost_bwd(V_ref.reshape(B, TF, E), Y).reshape(V_ref.shape))
        du = ops.dpcl_loss_bwd_u(dev(u).view(B, TF, E), dev(Y), inv, ws)
        # TF = 1: the loss is constant in u (a single unit vector), the true gradient is 0 and fp32 leaves rounding noise
        assert np.abs(host(du).reshape(du_ref.shape) - du_ref).max() < 1e-4 * max(np.abs(du_ref).max(), 1e-3)


def test_front_conv_signal_shorter_than_window(ops):
    rng = np.random.RandomState(2)
    for (Bt, L, W, N, hop) in [(1, 10, 64, 3, 16), (2, 100, 128, 5, 128), (1, 1, 8, 1, 1)]:
        x, f = rng.randn(Bt, L), rng.randn(W, N)
        y = ops.front_conv(dev(x), dev(f), hop)
        assert rel(host(y), ofront.conv_strided(x, f, hop)) < TOL


def test_kmeans_tiny_and_empty_cluster_nan(ops):
    """L barely above C; seeds that leave one cluster empty reproduce the reference's NaN centroid (0/0, Kmeans_2.py:164).  Only ONE
    update is compared: once a centroid is NaN the next argmin compares against NaN, which TF leaves unspecified."""
    rng = np.random.RandomState(3)
    b, L, E, C = 2, 5, 8, 2
    X = rng.randn(b, L, E).astype(np.float32)
    X[1, :] = X[1, 0]                                     # utterance 1: all points identical -> ties -> cluster 1 stays empty
    idx = np.array([[0, 1], [0, 1]], np.int32)
    xn = ops.kmeans_normalize(dev(X))
    cent, labels, best, _ = ops.kmeans_run(xn, torch.from_numpy(idx).cuda(), C, 1, 1)
    c_ref, l_ref, b_ref = okm.kmeans(X, idx, C, 1, 1, assign_at_end=True)
    got = host(cent)
    assert np.array_equal(np.isnan(got), np.isnan(c_ref))
    assert np.isnan(got[1, 1]).all() and not np.isnan(got[1, 0]).any() and not np.isnan(got[0]).any()
    assert np.array_equal(got[0].astype(np.float32), c_ref[0].astype(np.float32))          # the healthy utterance stays bit-exact
    assert np.array_equal(host(labels)[0].astype(np.int64), l_ref[0].astype(np.int64))


def test_error_behaviour(ops):
    from ams_hip._lib import AmsError
    a = torch.zeros(4, 4, device='cuda')
    with pytest.raises(AmsError):
        ops.gemm(a, torch.zeros(5, 4, device='cuda'))                       # inner dimensions differ
    with pytest.raises(AmsError):
        ops.gemm(torch.zeros(4, 4), a)                                      # host tensor: no CPU fallback
    with pytest.raises(AmsError):
        ops.l2norm_fwd(a.double(), 4)                                       # wrong dtype
    with pytest.raises(AmsError):
        ops.l2norm_fwd(torch.zeros(4, 8, device='cuda')[:, ::2], 4)         # non-contiguous
    xn = torch.zeros(2, 10, 8, device='cuda')
    with pytest.raises(AmsError):
        ops.kmeans_run(xn, torch.zeros(3, 2, dtype=torch.int32, device='cuda'), 2, 1, 1)   # init_idx must be [b*tries, C]
    with pytest.raises(AmsError):
        ops.dpcl_loss_bwd_u(torch.zeros(1, 4, 7, device='cuda'), torch.zeros(1, 4, 60, device='cuda'),
                            torch.zeros(1, 4, device='cuda'), torch.zeros(1024, device='cuda'))   # E + S > 64
