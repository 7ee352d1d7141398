"""bf16x6 products (csrc/gemm.hip: gemm_x6_kernel) are f32 products: held against FLOAT64 next to the native f32 MFMA kernel.

The default arithmetic of every 16-byte-addressable product (tf.matmul / conv1d k=1 / dynamic_rnn input projections and their
gradients, reference utils/ops.py:366-383, 501-503; the strided analysis conv, models/adapt.py:122) splits both f32 operands
exactly into three bf16 terms and accumulates six bf16 MFMA products in f32.  The claim tested here: its error against float64
is of the SAME class as the native f32 MFMA kernel's (and far below the 1e-3 of the north star), on every operand loader, on
ragged tiles and k-tails, with split-K, batching, row masks, accumulation, bias and the fused column sums, and at the benchmark's
own shapes.  `ams_gemm_set_arith` switches between the two kernels inside one process.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import front as ofront


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy().astype(np.float64)


def f32(x):
    return np.asarray(x, np.float32).astype(np.float64)


def err(c, ref, scale):
    """max |c - ref| relative to the size of the terms that were summed (|A| |B| row/column norms): the quantity an f32
    accumulation error is proportional to; insensitive to cancellation in individual outputs."""
    return float(np.abs(np.asarray(c, np.float64) - ref).max() / scale)


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.fixture()
def arith(ops):
    from ams_hip._lib import load
    lib = load()
    before = lib.ams_gemm_get_arith()

    def set_(mode):
        lib.ams_gemm_set_arith(mode)
        assert lib.ams_gemm_get_arith() == mode
    yield set_
    lib.ams_gemm_set_arith(before)


def both(arith, fn):
    """fn() under the native f32 MFMA kernel and under bf16x6."""
    arith(0)
    c0 = fn()
    arith(1)
    c1 = fn()
    return c0, c1


# the x6 kernel must be at least as good as this multiple of the native kernel's error (both are ~K^0.5 * 2^-24 random walks;
# observed ratio 0.5-1.1), and absolutely below ABS (f32 class; the north star allows 1e-3)
RATIO, ABS = 2.0, 2e-6


def check_pair(e0, e1):
    assert e1 < ABS, (e0, e1)
    assert e1 <= RATIO * e0 + 6e-8, (e0, e1)          # + one f32 ulp: a 12-term product is nearly exact on the f32 pipe


@pytest.mark.parametrize('M,N,K,tA,tB', [
    (132, 260, 604, 0, 0), (132, 260, 604, 0, 1), (132, 260, 604, 1, 0), (132, 260, 604, 1, 1),
    (4, 4, 4, 0, 0), (4, 8, 12, 1, 1), (128, 128, 32, 0, 0), (128, 128, 36, 1, 0), (256, 384, 64, 0, 1),
    (260, 132, 2052, 1, 0), (600, 520, 5120, 1, 0), (512, 256, 1024, 0, 1), (64, 10240, 600, 0, 0), (300, 1200, 2048, 1, 0),
    (5120, 2400, 600, 0, 0), (5120, 600, 2400, 0, 1), (600, 2400, 5120, 1, 0),
    # the 8-wave 256 x 256 tile configuration (csrc/gemm.hip X6Cfg<1>) on every loader, ragged edges included; shapes with one
    # narrow side stay on 128 x 128
    (516, 772, 100, 0, 0), (516, 772, 100, 0, 1), (516, 772, 100, 1, 0), (516, 772, 100, 1, 1),
    (516, 132, 292, 0, 0), (516, 132, 292, 0, 1), (516, 132, 292, 1, 0), (516, 132, 292, 1, 1),
    (132, 516, 292, 0, 0), (132, 516, 292, 0, 1), (132, 516, 292, 1, 0), (132, 516, 292, 1, 1),
])
def test_x6_matches_float64_like_native_f32(ops, arith, M, N, K, tA, tB):
    rng = np.random.RandomState(M + 3 * N + 7 * K + tA + 2 * tB)
    # wide dynamic range inside every row: exercises all three terms of the split
    A = rng.randn(*((K, M) if tA else (M, K))) * np.exp(rng.uniform(-6, 6, size=((K, M) if tA else (M, K))))
    B = rng.randn(*((N, K) if tB else (K, N))) * np.exp(rng.uniform(-6, 6, size=((N, K) if tB else (K, N))))
    bias = rng.randn(N)
    A64, B64 = f32(A.T if tA else A), f32(B.T if tB else B)
    ref = A64 @ B64 + f32(bias)
    scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
    a, b, bv = dev(A), dev(B), dev(bias)
    c0, c1 = both(arith, lambda: host(ops.gemm(a, b, transA=bool(tA), transB=bool(tB), bias=bv)))
    check_pair(err(c0, ref, scale), err(c1, ref, scale))
    # accumulate on top of an existing C
    C0 = rng.randn(M, N)

    def acc():
        c = dev(C0)
        ops.gemm(a, b, transA=bool(tA), transB=bool(tB), out=c, accumulate=True)
        return host(c)
    c0, c1 = both(arith, acc)
    ref2 = A64 @ B64 + f32(C0)
    check_pair(err(c0, ref2, scale), err(c1, ref2, scale))


def test_x6_small_integers_are_exact(ops, arith):
    """Products of small integers are exact in every term of the split and in f32: any layout / k-order / transposition mistake in
    the bf16 images shows as a whole-number error.  Asymmetric operands (guide rule: symmetric inputs hide transposes)."""
    rng = np.random.RandomState(5)
    for M, N, K, tA, tB in [(m, n, 100, ta, tb) for (m, n) in ((196, 324), (520, 776), (520, 196), (196, 520))
                            for ta in (0, 1) for tb in (0, 1)]:
        A = rng.randint(-7, 8, size=(K, M) if tA else (M, K)).astype(np.float64)
        B = rng.randint(-7, 8, size=(N, K) if tB else (K, N)).astype(np.float64)
        ref = (A.T if tA else A) @ (B.T if tB else B)
        arith(1)
        c = host(ops.gemm(dev(A), dev(B), transA=bool(tA), transB=bool(tB)))
        assert np.array_equal(c, ref), (tA, tB, np.abs(c - ref).max())


def test_x6_split_is_exact_to_the_last_bit(ops, arith):
    """hi + mid + lo == x exactly: a product with the identity returns the operand bit for bit (every term is x * 1 in one of the
    three images; the dropped partial products are zero because the identity's mid and lo are), on both operand sides."""
    rng = np.random.RandomState(6)
    n = 256
    X = (rng.randn(n, n) * np.exp(rng.uniform(-20, 20, size=(n, n)))).astype(np.float32)
    eye = np.eye(n, dtype=np.float32)
    arith(1)
    for tA, tB in ((0, 0), (0, 1), (1, 0), (1, 1)):
        left = host(ops.gemm(dev(X.T if tA else X), dev(eye), transA=bool(tA), transB=bool(tB)))
        assert np.array_equal(left.astype(np.float32), X), (tA, tB)
        right = host(ops.gemm(dev(eye), dev(X.T if tB else X), transA=bool(tA), transB=bool(tB)))
        assert np.array_equal(right.astype(np.float32), X), (tA, tB)


def test_x6_masked_strided_and_batched(ops, arith):
    """The recurrent-kernel gradient form: time-shifted operands inside wider buffers, rows t == T-1 masked, both directions in one
    batched launch (reference utils/ops.py:358-383 under tf.gradients)."""
    rng = np.random.RandomState(7)
    T, Bq, H = 20, 16, 24
    M = Bq * T
    out = rng.randn(M, 2 * H)
    dZ = rng.randn(M, 8 * H)
    ref = np.zeros((2, H, 4 * H))
    for b in range(Bq):
        for t in range(1, T):
            ref[0] += np.outer(f32(out[b * T + t - 1, :H]), f32(dZ[b * T + t, :4 * H]))
            ref[1] += np.outer(f32(out[b * T + t - 1, H:]), f32(dZ[b * T + t, 4 * H:]))
    scale = np.abs(ref).max() * 10
    o, z = dev(out), dev(dZ)

    def run():
        res = torch.zeros((2, H, 4 * H), device='cuda', dtype=torch.float32)
        ops.gemm_batched2(o.view(-1), o.view(-1)[H:], z.view(-1)[8 * H:], z.view(-1)[8 * H + 4 * H:], res[0], res[1], True, False,
                          H, 4 * H, M - 1, 2 * H, 8 * H, 4 * H, mask=(T, T - 1))
        return host(res)
    c0, c1 = both(arith, run)
    check_pair(err(c0, ref, scale), err(c1, ref, scale))


@pytest.mark.parametrize('M,N,K', [(600, 10240, 5120), (256, 40, 640), (600, 40, 5120)])
def test_x6_fused_column_sums(ops, arith, M, N, K):
    """dW = x^T dY and db = colsum(dY) in one pass (utils/ops.py:501-503), both arithmetics."""
    rng = np.random.RandomState(M + N + K)
    A, Bm = rng.randn(K, M), rng.randn(K, N)
    ref, refb = f32(A).T @ f32(Bm), f32(Bm).sum(0)
    scale = (np.linalg.norm(f32(A), axis=0)[:, None] * np.linalg.norm(f32(Bm), axis=0)[None, :]).max()
    a, b = dev(A), dev(Bm)

    def run():
        out, bsum = torch.empty(M, N, device='cuda'), torch.empty(N, device='cuda')
        assert ops.gemm_at_b_colsum(a, b, out, bsum, accumulate=False)
        return host(out), host(bsum)
    (c0, s0), (c1, s1) = both(arith, run)
    check_pair(err(c0, ref, scale), err(c1, ref, scale))
    assert np.abs(s1 - refb).max() / np.abs(refb).max() < 2e-5 and np.abs(s0 - refb).max() / np.abs(refb).max() < 2e-5


@pytest.mark.parametrize('Bt,L,W,N,hop', [(6, 4096, 1024, 256, 256), (2, 2048, 256, 40, 64), (192, 20480, 1024, 256, 256)])
def test_x6_front_conv_and_filter_gradient(ops, arith, Bt, L, W, N, hop):
    """Strided analysis conv (models/adapt.py:122) and its filter gradient through the frames loaders, incl. the benchmark shape."""
    rng = np.random.RandomState(L + Bt)
    x, f = rng.randn(Bt, L), rng.randn(W, N)
    y_ref = ofront.conv_strided(f32(x), f32(f), hop)
    xd, fd = dev(x), dev(f)
    scale = np.sqrt(W) * np.abs(f32(x)).max() * np.abs(f32(f)).max()
    c0, c1 = both(arith, lambda: host(ops.front_conv(xd, fd, hop)))
    check_pair(err(c0, y_ref, scale), err(c1, y_ref, scale))
    if Bt <= 8:
        dy = rng.randn(*y_ref.shape)
        df_ref = ofront.conv_strided_bwd_filter(f32(x), f32(dy), W, hop)
        dyd = dev(dy)
        scale = np.sqrt(dy.shape[0] * dy.shape[1]) * np.abs(x).max() * np.abs(dy).max()
        c0, c1 = both(arith, lambda: host(ops.front_conv_bwd_filter(xd, dyd, W, hop)))
        check_pair(err(c0, df_ref, scale), err(c1, df_ref, scale))


def test_x6_dense_forward_at_benchmark_shape(ops, arith):
    """[B*T = 5120, 600] x [600, 10240] + bias (dense layer of models/dpcl.py:41-52 at batch 64), against float64."""
    rng = np.random.RandomState(11)
    A, B, bias = rng.randn(5120, 600), rng.randn(600, 10240) * 0.05, rng.randn(10240)
    ref = f32(A) @ f32(B) + f32(bias)
    scale = (np.linalg.norm(f32(A), axis=1)[:, None] * np.linalg.norm(f32(B), axis=0)[None, :]).max()
    a, b, bv = dev(A), dev(B), dev(bias)
    c0, c1 = both(arith, lambda: host(ops.gemm(a, b, bias=bv)))
    check_pair(err(c0, ref, scale), err(c1, ref, scale))


@pytest.mark.parametrize('K,kind', [(600, 'pos'), (5120, 'pos'), (600, 'randn'), (5120, 'randn')])
def test_x6_accumulation_is_unbiased(ops, arith, K, kind):
    """The bf16 MFMA adds its products to the accumulator with the bits below its guard bits truncated toward -inf; with all six
    partial products in one accumulator that is a coherent error of -0.3 .. -1 ulp per output (-6e-8 sqrt(K) at K = 5120), which
    sums over many outputs amplify (csrc/gemm.hip, SEP).  The default kernels keep the five small partial products in their own
    accumulator: the MEAN signed error must be at the native f32 kernel's level (measured -1e-10 .. -9e-10 against +-5e-10), and
    the rms error at or below it (measured 0.4-0.9x)."""
    rng = np.random.RandomState(K)
    M = N = 384                                      # 128 x 128 / 128 x 256 tiles, not residency-capped: the default path
    if kind == 'pos':
        A, B = rng.uniform(0.5, 1.0, (M, K)), rng.uniform(0.5, 1.0, (K, N))
    else:
        A, B = rng.randn(M, K), rng.randn(K, N)
    ref = f32(A) @ f32(B)
    scale = np.abs(ref).mean() if kind == 'pos' else np.sqrt(K)
    a, b = dev(A), dev(B)
    c0, c1 = both(arith, lambda: host(ops.gemm(a, b)))
    d0, d1 = (c0 - ref) / scale, (c1 - ref) / scale
    assert abs(d1.mean()) < 5e-9, (d0.mean(), d1.mean())
    assert np.sqrt((d1 ** 2).mean()) <= 1.05 * np.sqrt((d0 ** 2).mean()), (np.sqrt((d0 ** 2).mean()), np.sqrt((d1 ** 2).mean()))


@pytest.fixture()
def capped():
    """Launch products the way every weight-gradient product of a training step is launched (ams_hip/functional.py::_Overlap:
    side stream, residency cap of 2 workgroups per CU beside a recurrence ring)."""
    from ams_hip import ops
    with ops.lds_pad(50000):
        yield


def _bias_stats(c, ref, scale):
    d = (np.asarray(c, np.float64) - ref) / scale
    return float(d.mean()), float(np.sqrt((d ** 2).mean()))


# The residency-capped launches of the in-loop-split kernel run its ONE-accumulator form (a second accumulator set does not fit
# beside a recurrence-ring wave: csrc/gemm.hip, SEP).  All six partial products in one accumulator meet the bf16 MFMA's truncating
# adder: a COHERENT error toward -inf, measured -0.9e-7 of the term scale at K = 5120 on zero-mean data -- 20x the 5e-9 the
# two-accumulator form is held to, and what AMSGrad's first moment would integrate over steps.  These tests pin that number (it
# must not grow).  (An unbiased form of the capped launches -- products from pre-split operand images with two accumulator sets --
# was built and measured in round 3, -3 % step throughput, and left the tree in round 4: DESIGN.md 4.0b, commit 7fd39ac.)
CAPPED_BIAS_BOUND = 3e-7


@pytest.mark.parametrize('M,N,K', [(600, 10240, 5120), (600, 2400, 5120), (256, 2400, 5120)])
def test_x6_capped_weight_gradient_bias_is_bounded(ops, arith, capped, M, N, K):
    """dW = x^T dY (A_COL x B_ROW) at the step's own shapes -- dense 600 x 10240, projections 600 x 2400 and 256 x 2400, all K = B*T =
    5120 -- under the residency cap every weight-gradient product of the step runs with."""
    rng = np.random.RandomState(M + N)
    for kind in ('randn', 'pos'):
        A, B = (rng.randn(K, M), rng.randn(K, N)) if kind == 'randn' else (rng.uniform(0.5, 1.0, (K, M)), rng.uniform(0.5, 1.0, (K, N)))
        ref = f32(A).T @ f32(B)
        scale = np.sqrt(K) if kind == 'randn' else np.abs(ref).mean()
        a, b = dev(A), dev(B)
        c0, c1 = both(arith, lambda: host(ops.gemm(a, b, transA=True)))
        m0, r0 = _bias_stats(c0, ref, scale)
        m1, r1 = _bias_stats(c1, ref, scale)
        print('capped %s %dx%dx%d: native mean %.2e rms %.2e | bf16x6 (one accumulator) mean %.2e rms %.2e' % (kind, M, N, K, m0, r0, m1, r1))
        assert abs(m1) < CAPPED_BIAS_BOUND, (m0, m1)
        assert r1 <= 1.5 * r0, (r0, r1)


def test_x6_capped_recurrent_kernel_gradient_bias_is_bounded(ops, arith, capped):
    """The batched, time-shifted, masked form of the recurrent-kernel gradients at the step's shape (2 x [300 x 1200], K = 5120 rows
    of which every T-th is masked), capped like in the step."""
    rng = np.random.RandomState(77)
    T, Bq, H = 80, 64, 300
    M = Bq * T
    out, dZ = rng.randn(M, 2 * H), rng.randn(M, 8 * H)
    o3, z3 = f32(out).reshape(Bq, T, 2 * H), f32(dZ).reshape(Bq, T, 8 * H)
    ref = np.stack([np.einsum('btj,btg->jg', o3[:, :-1, :H], z3[:, 1:, :4 * H]),
                    np.einsum('btj,btg->jg', o3[:, :-1, H:], z3[:, 1:, 4 * H:])])
    o, z = dev(out), dev(dZ)

    def run():
        res = torch.zeros((2, H, 4 * H), device='cuda', dtype=torch.float32)
        ops.gemm_batched2(o.view(-1), o.view(-1)[H:], z.view(-1)[8 * H:], z.view(-1)[8 * H + 4 * H:], res[0], res[1], True, False,
                          H, 4 * H, M - 1, 2 * H, 8 * H, 4 * H, mask=(T, T - 1))
        return host(res)
    c0, c1 = both(arith, run)
    m0, r0 = _bias_stats(c0, ref, np.sqrt(M))
    m1, r1 = _bias_stats(c1, ref, np.sqrt(M))
    print('capped batched dU: native mean %.2e rms %.2e | bf16x6 (one accumulator) mean %.2e rms %.2e' % (m0, r0, m1, r1))
    assert abs(m1) < CAPPED_BIAS_BOUND and r1 <= 1.5 * r0, (m0, r0, m1, r1)


def test_mask_separator_nan_rows_agree_under_both_arithmetics(ops, arith):
    """Pre-training with --separation mask computes mix * (non_mix / mix) (models/adapt.py:179-184): NaN wherever the mixture
    representation is exactly zero (0 * inf), by the reference's design.  Those NaNs then enter the synthesis product z . f2^T
    (adapt.py:236-243).  The bf16x6 arithmetic turns Inf operands into NaN (inf - inf in its split) where the f32 MFMA propagates
    Inf -- but what reaches the product here is already NaN, so both arithmetics must mark exactly the same outputs as NaN and
    agree on every finite one."""
    rng = np.random.RandomState(8)
    B, S, T, N, W = 2, 2, 12, 16, 64
    y = rng.randn(B * (S + 1), T, N)
    y[0, 3, 5] = 0.0                       # mixture bin exactly zero, sources non-zero: nm / 0 = +-inf, 0 * inf = NaN
    y[1, 7, :] = 0.0                       # a whole silent mixture frame
    y[1 + B + 2, 7, 2] = 0.0               # with a zero source bin too: 0 / 0 = NaN
    z = ops.pretrain_separator_fwd(dev(y), B, S, 'mask')
    zh = host(z)
    assert np.isnan(zh).sum() >= 1 + N and not np.isinf(zh).any()
    f2 = rng.randn(W, N)
    zr = z.reshape(B * S * T, N)
    c0, c1 = both(arith, lambda: host(ops.gemm(zr, dev(f2), transB=True)))
    assert np.array_equal(np.isnan(c0), np.isnan(c1))
    assert not np.isinf(c0).any() and not np.isinf(c1).any()
    nan_rows = np.isnan(zh).reshape(B * S * T, N).any(axis=1)
    assert np.isnan(c1[nan_rows]).all() and np.isfinite(c1[~nan_rows]).all()
    fin = ~np.isnan(c0)
    assert np.abs(c0[fin] - c1[fin]).max() < 1e-5 * np.abs(c0[fin]).max()
