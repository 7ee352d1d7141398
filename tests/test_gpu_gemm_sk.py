"""GPU: stream-K scheduling of the 16-bit-pipe products (csrc/gemm.hip x6_body, include/ams.h: sk_scratch) against float64.

The reference's products are f32 tf.matmul / conv1d (utils/ops.py:366-383, :501-503).  Stream-K changes WHO computes which k range of an
output tile, never what is computed: whole-tile rounds, then every resident workgroup an equal share of the k-tiles of the tiles that are
left, partial tiles added by the tile's owner in a fixed order.  What must hold:
  * the result is the product (error vs float64 at the level of the launch without scratch, every operand layout, fp16x3 and bf16x6);
  * it is deterministic (bit-equal across repeats) and the scratch's flag words are zero again after every launch;
  * the scratch was actually used at the shapes the training step cares about (the cost model took stream-K there);
  * fused column sums and batched launches survive it."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


def _err(out, ref, A64, B64):
    scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
    return np.abs(out.double().cpu().numpy() - ref).max() / scale


def _scratch(ops, like):
    ops._sk(like)                                                   # make sure it exists
    return ops._SK[(like.device.index, torch.cuda.current_stream().cuda_stream)]


def _pair(rng, M, N, K, tA, tB):
    A = torch.from_numpy(rng.randn(*((K, M) if tA else (M, K))).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.randn(*((N, K) if tB else (K, N))).astype(np.float32)).cuda()
    return A, B, (A.T if tA else A).double().cpu().numpy(), (B.T if tB else B).double().cpu().numpy()


# the critical-path products of the B = 64 step: projections (K = 600 / 256), dense forward, dense dX, LSTM dX, first-layer dWx; plus ragged ones
SHAPES = [(5120, 2400, 600, 0, 0), (5120, 2400, 256, 0, 0), (5120, 10240, 600, 0, 0), (5120, 600, 10240, 0, 1), (5120, 600, 2400, 0, 1),
          (256, 2400, 5120, 1, 0), (1000, 1300, 388, 0, 0), (2052, 516, 1028, 1, 1)]


@pytest.mark.parametrize('M,N,K,tA,tB', SHAPES)
@pytest.mark.parametrize('f16', [True, False])
def test_stream_k_is_the_same_product(ops, M, N, K, tA, tB, f16):
    rng = np.random.RandomState(M + N + K + tA)
    A, B, A64, B64 = _pair(rng, M, N, K, tA, tB)
    ref = A64 @ B64
    bias = torch.from_numpy(rng.randn(N).astype(np.float32)).cuda() if not tA else None
    if bias is not None:
        ref = ref + bias.double().cpu().numpy()[None, :]
    bounds = (ops.absmax(A), ops.absmax(B)) if f16 else None
    sc = _scratch(ops, A)
    sc[2048:].fill_(float('nan'))                                   # slots: a hole in the hand-off would show as NaN
    torch.cuda.synchronize()
    out = ops.gemm(A, B, transA=bool(tA), transB=bool(tB), bias=bias, amax=bounds)
    out2 = ops.gemm(A, B, transA=bool(tA), transB=bool(tB), bias=bias, amax=bounds)
    torch.cuda.synchronize()
    assert int(sc[:2048].view(torch.int32).abs().sum()) == 0        # flags left zero
    used = bool(torch.isfinite(sc[2048:]).any())
    old, ops.SK = ops.SK, False
    try:
        plain = ops.gemm(A, B, transA=bool(tA), transB=bool(tB), bias=bias, amax=bounds)
    finally:
        ops.SK = old
    e_sk, e_plain = _err(out, ref, A64, B64), _err(plain, ref, A64, B64)
    assert torch.isfinite(out).all()
    assert torch.equal(out, out2)                                   # deterministic
    assert e_sk <= max(1.5 * e_plain, 3e-8), (e_sk, e_plain, used)
    if (M, N, K) in ((5120, 2400, 600), (5120, 10240, 600)):
        assert used, 'the cost model was expected to take stream-K at this shape'


def test_stream_k_with_fused_column_sums_and_accumulate(ops):
    """dW (+)= x^T dZ and db (+)= colsum(dZ) from one pass (ams_gemm_f32_at_b_colsum), uncapped as in the first layer's tail."""
    rng = np.random.RandomState(5)
    K, M, N = 5120, 256, 2400
    A = torch.from_numpy(rng.randn(K, M).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.randn(K, N).astype(np.float32)).cuda()
    C0 = torch.from_numpy(rng.randn(M, N).astype(np.float32)).cuda()
    b0 = torch.from_numpy(rng.randn(N).astype(np.float32)).cuda()
    ref = C0.double().cpu().numpy() + A.double().cpu().numpy().T @ B.double().cpu().numpy()
    refb = b0.double().cpu().numpy() + B.double().cpu().numpy().sum(0)
    res = {}
    for sk in (True, False):
        old, ops.SK = ops.SK, sk
        try:
            C, b = C0.clone(), b0.clone()
            assert ops.gemm_at_b_colsum(A, B, C, b, accumulate=True, amax=(ops.absmax(A), ops.absmax(B)))
            res[sk] = (C, b)
        finally:
            ops.SK = old
    for sk in (True, False):
        C, b = res[sk]
        assert np.abs(C.double().cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-6, sk
        assert np.abs(b.double().cpu().numpy() - refb).max() / np.abs(refb).max() < 2e-6, sk


def test_stream_k_batched(ops):
    """Two products of one shape in one launch (the two directions' recurrent-kernel gradients, uncapped)."""
    rng = np.random.RandomState(6)
    nb, K, M, N = 2, 5119, 300, 1200
    A = torch.from_numpy(rng.randn(nb, K, M).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.randn(nb, K, N).astype(np.float32)).cuda()
    C = torch.empty(nb, M, N, device='cuda')
    ops.gemm_batched(A, B, C, nb, K * M, K * N, M * N, True, False, M, N, K, M, N, N, amax=(ops.absmax(A), ops.absmax(B)))
    for z in range(nb):
        ref = A[z].double().cpu().numpy().T @ B[z].double().cpu().numpy()
        assert np.abs(C[z].double().cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-6


def test_front_conv_measures_its_output(ops):
    """ams_front_conv_fwd with bounds (fp16x3) and amax_y: y as before, and max |y| exactly what a pass over y finds."""
    rng = np.random.RandomState(7)
    Bt, L, W, N, hop = 48, 20480, 1024, 256, 256
    x = torch.from_numpy((rng.randn(Bt, L) * 0.05).astype(np.float32)).cuda()
    f = torch.from_numpy((rng.randn(W, N) * 0.03).astype(np.float32)).cuda()
    y0 = ops.front_conv(x, f, hop)
    y1 = ops.front_conv(x, f, hop, amax=(ops.absmax(x), ops.absmax(f)), measure=True)
    torch.cuda.synchronize()
    assert (y0 - y1).abs().max() / y0.abs().max() < 2e-6
    assert float(ops.amax_of(y1)) == float(y1.abs().max())


def test_stream_k_wherever_it_applies():
    """The default takes stream-K only for the tail of multi-round launches; AMS_GEMM_SK=2 (read once per process) takes it wherever it
    applies -- launches of fewer tiles than workgroups, 3-9 partial tiles per owner: the same tests, in a process of their own."""
    import os
    import subprocess
    import sys
    if os.environ.get('AMS_GEMM_SK') == '2':
        pytest.skip('this IS the forced run')
    env = dict(os.environ, AMS_GEMM_SK='2')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-m', 'gpu', '-k',
                        'same_product or column_sums or batched'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
