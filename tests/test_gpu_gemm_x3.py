"""Products from PRE-SPLIT operands (csrc/gemm_x3.hip: x3 images, LDS-DMA main loop) are f32 products: held against FLOAT64 next to the
native f32 MFMA kernel, like the in-loop-split bf16x6 kernel in tests/test_gpu_gemm_x6.py.

Same call sites (tf.matmul / conv1d k=1 / dynamic_rnn input projections and their gradients, reference utils/ops.py:366-383, 501-503).
Tested: the image itself (hi + mid + lo == x bit for bit, zero padding), both operand roles on both sides (ds_read_b128 and the
transposing ds_read_b64_tr_b16 path), ragged tiles and k-tails, offsets inside wider images, batching, split-K, bias / accumulate, the
time-shifted image of the recurrent-kernel gradients, both tile configurations incl. the residency-capped one, exactness on small
integers (transpose-detecting: asymmetric operands), and the mean signed error (truncating bf16 MFMA adder) at the step's own
weight-gradient shapes."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def f32(x):
    return np.asarray(x, np.float32).astype(np.float64)


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.fixture()
def lib():
    from ams_hip._lib import load
    l = load()
    yield l
    l.ams_x3_set_capped(0)
    l.ams_gemm_set_arith(1)


def x3_product(ops, A, B, tA, tB, **kw):
    """op(A) op(B) through x3 images; A given as [K, M] when tA, B as [N, K] when tB."""
    Ai, Bi = ops.x3_split(dev(A)), ops.x3_split(dev(B))
    M, K = (A.shape[1], A.shape[0]) if tA else A.shape
    N = B.shape[0] if tB else B.shape[1]
    return ops.gemm_x3(Ai, 1 if tA else 0, Bi, 0 if tB else 1, M, N, K, **kw)


def decode_image(img, R, C):
    """x3 image bytes -> (hi, mid, lo) float64 arrays of the PADDED logical matrix."""
    raw = img.buf.cpu().numpy().view(np.uint16)
    Rp, Cp = -(-R // 256) * 256, -(-C // 256) * 256
    u = raw.reshape(Rp // 8, Cp // 16, 3, 2, 8, 8)                     # [rb, cb, plane, kg, r8, c8]
    planes = u.transpose(2, 0, 4, 1, 3, 5).reshape(3, Rp, Cp)          # [plane, rb, r8, cb, kg, c8]
    f = (planes.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return f[0], f[1], f[2]


def test_x3_image_is_an_exact_split_with_zero_padding(ops):
    rng = np.random.RandomState(1)
    R, C = 300, 604
    X = (rng.randn(R, C) * np.exp(rng.uniform(-20, 20, (R, C)))).astype(np.float32)
    img = ops.x3_split(dev(X))
    hi, mid, lo = decode_image(img, R, C)
    s = (lo + mid) + hi                                                # exact in float64
    assert np.array_equal(s[:R, :C].astype(np.float32), X)
    assert np.all(s[R:] == 0) and np.all(s[:, C:] == 0)
    assert np.all(np.abs(mid[:R, :C]) <= np.abs(hi[:R, :C]) * 2.0 ** -7 + 1e-300)
    # a strided source (column slice of a wider buffer)
    wide = dev(np.concatenate([X, X], 1))
    img2 = ops.x3_split(wide[:, C:], R=R, C=C, ld=2 * C) if False else ops.x3_split(wide[:, C:])
    assert torch.equal(img2.buf, img.buf)


@pytest.mark.parametrize('M,N,K', [(132, 260, 604), (516, 772, 292), (128, 128, 32), (8, 8, 8), (600, 520, 5120), (5120, 600, 2400)])
@pytest.mark.parametrize('tA,tB', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_x3_matches_float64_like_native_f32(ops, lib, M, N, K, tA, tB):
    rng = np.random.RandomState(M + 3 * N + 7 * K + tA + 2 * tB)
    A = rng.randn(*((K, M) if tA else (M, K))) * np.exp(rng.uniform(-6, 6, size=((K, M) if tA else (M, K))))
    B = rng.randn(*((N, K) if tB else (K, N))) * np.exp(rng.uniform(-6, 6, size=((N, K) if tB else (K, N))))
    bias = rng.randn(N)
    A64, B64 = f32(A.T if tA else A), f32(B.T if tB else B)
    ref = A64 @ B64 + f32(bias)
    scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
    lib.ams_gemm_set_arith(0)
    e0 = np.abs(host(ops.gemm(dev(A), dev(B), transA=bool(tA), transB=bool(tB), bias=dev(bias))) - ref).max() / scale
    for capped in (0, 1):
        lib.ams_x3_set_capped(capped)
        e1 = np.abs(host(x3_product(ops, A, B, tA, tB, bias=dev(bias))) - ref).max() / scale
        assert e1 < 2e-6 and e1 <= 2.0 * e0 + 6e-8, (capped, e0, e1)
    lib.ams_x3_set_capped(0)
    C0 = rng.randn(M, N)
    c = dev(C0)
    x3_product(ops, A, B, tA, tB, out=c, accumulate=True)
    e2 = np.abs(host(c) - (A64 @ B64 + f32(C0))).max() / scale
    assert e2 < 2e-6 and e2 <= 2.0 * e0 + 6e-8, (e0, e2)


def test_x3_small_integers_are_exact(ops, lib):
    """Any layout / k-order / transposition mistake in the images or the fragment reads is a whole-number error."""
    rng = np.random.RandomState(5)
    for capped in (0, 1):
        lib.ams_x3_set_capped(capped)
        for (M, N) in ((196, 324), (520, 776), (520, 196), (196, 520)):
            for tA in (0, 1):
                for tB in (0, 1):
                    K = 100
                    A = rng.randint(-7, 8, size=(K, M) if tA else (M, K)).astype(np.float64)
                    B = rng.randint(-7, 8, size=(N, K) if tB else (K, N)).astype(np.float64)
                    ref = (A.T if tA else A) @ (B.T if tB else B)
                    c = host(x3_product(ops, A, B, tA, tB))
                    assert np.array_equal(c, ref), (capped, M, N, tA, tB, np.abs(c - ref).max())


def test_x3_identity_returns_the_operand_bit_for_bit(ops, lib):
    rng = np.random.RandomState(6)
    n = 256
    X = (rng.randn(n, n) * np.exp(rng.uniform(-20, 20, size=(n, n)))).astype(np.float32)
    eye = np.eye(n, dtype=np.float32)
    for tA, tB in ((0, 0), (0, 1), (1, 0), (1, 1)):
        left = host(x3_product(ops, X.T if tA else X, eye, tA, tB))
        assert np.array_equal(left.astype(np.float32), X), (tA, tB)
        right = host(x3_product(ops, eye, X.T if tB else X, tA, tB))
        assert np.array_equal(right.astype(np.float32), X), (tA, tB)


def test_x3_offsets_batches_and_the_shifted_image(ops, lib):
    """The recurrent-kernel gradients of one BLSTM layer (reference utils/ops.py:358-383 under tf.gradients): dU_fw[j, g] = sum_{b,t}
    h[b, t-1, j] dZ[b, t, g], dU_bw[j, g] = sum h[b, t+1, H + j] dZ[b, t, 4H + g] -- two products in ONE batched launch, A from the
    time-shifted image (column blocks [0, H) and [Hp, Hp + H)), B from column blocks [0, 4H) / [4H, 8H) of the dZ image."""
    rng = np.random.RandomState(7)
    for (T, Bq, H) in ((20, 16, 24), (80, 8, 300)):
        M = Bq * T
        out, dZ = rng.randn(M, 2 * H), rng.randn(M, 8 * H)
        o3, z3 = f32(out).reshape(Bq, T, 2 * H), f32(dZ).reshape(Bq, T, 8 * H)
        ref = np.stack([np.einsum('btj,btg->jg', o3[:, :-1, :H], z3[:, 1:, :4 * H]),
                        np.einsum('btj,btg->jg', o3[:, 1:, H:], z3[:, :-1, 4 * H:])])
        Hp = (H + 7) // 8 * 8
        hs = ops.x3_split_shifted(dev(out), T, H)
        zi = ops.x3_split(dev(dZ))
        res = torch.zeros((2, H, 4 * H), device='cuda')
        for capped in (0, 1):
            lib.ams_x3_set_capped(capped)
            res.zero_()
            ops.gemm_x3(hs, 1, zi, 1, H, 4 * H, M, out=res, ldc=4 * H, nbatch=2, a_m_zs=Hp, b_n_zs=4 * H, c_zs=H * 4 * H)
            err = np.abs(host(res) - ref).max() / (np.abs(ref).max() * 10)
            assert err < 2e-6, (T, H, capped, err)


@pytest.mark.parametrize('M,N,K', [(600, 10240, 5120), (600, 2400, 5120), (256, 2400, 5120), (304, 1200, 5120)])
def test_x3_weight_gradient_products_are_unbiased(ops, lib, M, N, K):
    """dW = x^T dY at the training step's own shapes, in the residency-capped configuration every weight-gradient product of the step
    runs in (and uncapped): |mean signed error| < 5e-9 of the term scale -- the bound tests/test_gpu_gemm_x6.py holds the uncapped
    two-accumulator bf16x6 kernel to -- and an rms error at the level of the native f32 MFMA kernel or of that default bf16x6 kernel
    (same six products, same two accumulator sets: for long same-sign chains both sit 1.2-1.4x above the native kernel, for zero-mean
    data 0.6x below it)."""
    rng = np.random.RandomState(M + N)
    for kind in ('randn', 'pos'):
        if kind == 'pos':
            A, B = rng.uniform(0.5, 1.0, (K, M)), rng.uniform(0.5, 1.0, (K, N))
        else:
            A, B = rng.randn(K, M), rng.randn(K, N)
        ref = f32(A).T @ f32(B)
        scale = np.abs(ref).mean() if kind == 'pos' else np.sqrt(K)
        rms = lambda d: float(np.sqrt((d ** 2).mean()))
        lib.ams_gemm_set_arith(0)
        d0 = (host(ops.gemm(dev(A), dev(B), transA=True)) - ref) / scale
        lib.ams_gemm_set_arith(1)
        d6 = (host(ops.gemm(dev(A), dev(B), transA=True)) - ref) / scale          # uncapped: the two-accumulator bf16x6 kernel
        for capped in (1, 0):
            lib.ams_x3_set_capped(capped)
            d1 = (host(x3_product(ops, A, B, 1, 0)) - ref) / scale
            print('%s %dx%dx%d capped=%d: native mean %.2e rms %.2e | bf16x6 (2 acc) mean %.2e rms %.2e | x3 mean %.2e rms %.2e'
                  % (kind, M, N, K, capped, d0.mean(), rms(d0), d6.mean(), rms(d6), d1.mean(), rms(d1)))
            assert abs(d1.mean()) < 5e-9, (kind, capped, d0.mean(), d1.mean())
            assert rms(d1) <= 1.05 * max(rms(d0), rms(d6)), (kind, capped, rms(d1), rms(d0), rms(d6))


def test_x3_split_with_column_sums(ops):
    """The pass that splits dU also yields db = colsum(dU) (utils/ops.py:501-503 under tf.gradients)."""
    rng = np.random.RandomState(3)
    for (R, C) in ((5120, 10240), (300, 604), (33, 70)):
        X = rng.randn(R, C)
        b0 = rng.randn(C)
        bt = dev(b0)
        img = ops.x3_split_colsum(dev(X), bt, True)
        assert torch.equal(img.buf, ops.x3_split(dev(X)).buf)
        refb = f32(b0) + f32(X).sum(0)
        assert np.abs(host(bt) - refb).max() < 2e-5 * np.abs(refb).max()


def test_x3_dense_forward_and_dx_at_benchmark_shape(ops, lib):
    rng = np.random.RandomState(11)
    A, B, bias = rng.randn(5120, 600), rng.randn(600, 10240) * 0.05, rng.randn(10240)
    ref = f32(A) @ f32(B) + f32(bias)
    scale = (np.linalg.norm(f32(A), axis=1)[:, None] * np.linalg.norm(f32(B), axis=0)[None, :]).max()
    e = np.abs(host(x3_product(ops, A, B, 0, 0, bias=dev(bias))) - ref).max() / scale
    assert e < 3e-7, e
    dU = rng.randn(5120, 10240)
    ref = f32(dU) @ f32(B).T
    scale = (np.linalg.norm(f32(dU), axis=1)[:, None] * np.linalg.norm(f32(B), axis=1)[None, :]).max()
    e = np.abs(host(x3_product(ops, dU, B, 0, 1)) - ref).max() / scale
    assert e < 3e-7, e
