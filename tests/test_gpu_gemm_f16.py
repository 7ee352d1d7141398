"""GPU: fp16x3 product arithmetic (csrc/gemm.hip, include/ams.h: amax_a / amax_b of the product entry points) against float64, next to bf16x6 and the f32 MFMA.

The reference's products are f32 tf.matmul / conv2d.  fp16x3 issues each of them as three fp16 MFMA products of
operands scaled by a power of two taken from a per-tensor bound and split exactly into two fp16 terms.  What must hold:
  * error at the level of bf16x6 / the f32 MFMA on well-scaled data, every operand layout and tile configuration;
  * independence of the data's magnitude (1e-30 .. 1e+30: the scale is part of the arithmetic, fp16 has 5 exponent bits);
  * a loose bound (too high by 2^10) costs nothing, a wide dynamic range inside one operand costs what the header says;
  * the one-shot setting does not leak into the next launch; without bounds the launch is bit-identical to bf16x6."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


def _err(out, A64, B64):
    ref = A64 @ B64
    scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
    return np.abs(out.double().cpu().numpy() - ref).max() / scale


def _ops_pair(rng, M, N, K, tA, tB, sa=1.0, sb=1.0):
    A = torch.from_numpy((rng.randn(*((K, M) if tA else (M, K))) * sa).astype(np.float32)).cuda()
    B = torch.from_numpy((rng.randn(*((N, K) if tB else (K, N))) * sb).astype(np.float32)).cuda()
    A64 = (A.T if tA else A).double().cpu().numpy()
    B64 = (B.T if tB else B).double().cpu().numpy()
    return A, B, A64, B64


@pytest.mark.parametrize('M,N,K,tA,tB', [(512, 384, 600, 0, 0), (640, 600, 1024, 0, 1), (600, 1024, 640, 1, 0), (300, 1200, 512, 1, 1),
                                         (5120, 2400, 256, 0, 0), (132, 260, 68, 0, 0), (128, 128, 32, 1, 0), (4, 8, 4, 0, 0)])
def test_fp16x3_matches_float64_like_the_other_arithmetics(ops, M, N, K, tA, tB):
    from ams_hip._lib import load
    lib = load()
    rng = np.random.RandomState(M + 3 * N + 7 * K + tA + 2 * tB)
    A, B, A64, B64 = _ops_pair(rng, M, N, K, tA, tB)
    bounds = (ops.absmax(A), ops.absmax(B))
    assert abs(float(bounds[0]) - float(A.abs().max())) == 0.0
    e16 = _err(ops.gemm(A, B, transA=bool(tA), transB=bool(tB), amax=bounds), A64, B64)
    out6 = ops.gemm(A, B, transA=bool(tA), transB=bool(tB))
    e6 = _err(out6, A64, B64)
    lib.ams_gemm_set_arith(0)
    try:
        e32 = _err(ops.gemm(A, B, transA=bool(tA), transB=bool(tB)), A64, B64)
    finally:
        lib.ams_gemm_set_arith(1)
    assert e16 <= max(1.5 * max(e6, e32), 3e-8), (e16, e6, e32)
    # bounds are arguments, not state: the next launch without them is bf16x6 again, bit for bit
    assert torch.equal(ops.gemm(A, B, transA=bool(tA), transB=bool(tB)), out6)


@pytest.mark.parametrize('sa,sb', [(1e-30, 1.0), (1e+30, 1e-12), (3e-8, 2e-7), (7e4, 9e4), (1.0, 1e+25)])
def test_fp16x3_is_independent_of_the_operands_magnitude(ops, sa, sb):
    rng = np.random.RandomState(11)
    A, B, A64, B64 = _ops_pair(rng, 384, 512, 640, 0, 0, sa, sb)
    out = ops.gemm(A, B, amax=(ops.absmax(A), ops.absmax(B)))
    assert torch.isfinite(out).all()
    assert _err(out, A64, B64) < 1.5e-7


def test_a_loose_bound_costs_nothing_and_a_wide_range_costs_what_the_header_says(ops):
    rng = np.random.RandomState(12)
    A, B, A64, B64 = _ops_pair(rng, 256, 256, 512, 0, 0)
    tight = (ops.absmax(A), ops.absmax(B))
    loose = (tight[0] * 1024.0, tight[1] * 1024.0)
    e_t = _err(ops.gemm(A, B, amax=tight), A64, B64)
    e_l = _err(ops.gemm(A, B, amax=loose), A64, B64)
    assert e_l < 1.5 * e_t + 1e-9, (e_t, e_l)
    # one row of A at 2^-20 of the rest: that row's results are exact to bound * 2^-39 absolutely, i.e. ~2^-19 relatively
    A2 = A.clone()
    A2[7] *= 2.0 ** -20
    out = ops.gemm(A2, B, amax=(ops.absmax(A2), tight[1]))
    ref = A2.double().cpu().numpy() @ B64
    row_rel = np.abs(out[7].double().cpu().numpy() - ref[7]).max() / np.abs(ref[7]).max()
    rest = np.abs(np.delete(out.double().cpu().numpy(), 7, 0) - np.delete(ref, 7, 0)).max() / np.abs(ref).max()
    assert row_rel < 2.0 ** -15 and rest < 1e-6, (row_rel, rest)


def test_fp16x3_under_the_residency_cap_and_with_split_k(ops):
    """The capped (single-accumulator) variants and the split-K partial slabs take the same scaling."""
    from ams_hip._lib import load
    lib = load()
    rng = np.random.RandomState(13)
    for (M, N, K, tA, tB) in [(600, 2400, 5120, 1, 0), (600, 10240, 2560, 1, 0), (5120, 600, 2400, 0, 1)]:
        A, B, A64, B64 = _ops_pair(rng, M, N, K, tA, tB, 1e-4, 3.0)
        bounds = (ops.absmax(A), ops.absmax(B))
        with ops.lds_pad(50000):
            capped = ops.gemm(A, B, transA=bool(tA), transB=bool(tB), amax=bounds)
            capped6 = ops.gemm(A, B, transA=bool(tA), transB=bool(tB))
        free = ops.gemm(A, B, transA=bool(tA), transB=bool(tB), amax=bounds)
        # the capped variants run one accumulator (tests/test_gpu_gemm_x6.py pins what that costs bf16x6): same ceiling here
        e_c, e_c6, e_f = _err(capped, A64, B64), _err(capped6, A64, B64), _err(free, A64, B64)
        assert e_f < 1e-7 and e_c < max(2.0 * e_c6, 2e-7), (M, N, K, e_c, e_c6, e_f)


@pytest.mark.parametrize('M,N,K', [(600, 10240, 5120), (600, 2400, 5120), (256, 2400, 5120)])
def test_fp16x3_weight_gradient_bias_capped_and_free(ops, M, N, K):
    """The weight-gradient products of the training step run as fp16x3 UNDER THE RESIDENCY CAP (side stream, one accumulator set:
    gemm_x6_kernel<.., SEP = false, F16 = true>).  Signed mean error against float64 at the step's own shapes, zero-mean and
    all-positive data: the capped form is held to the bound tests/test_gpu_gemm_x6.py pins for capped bf16x6 (3e-7 of the term
    scale), the uncapped two-accumulator form to 2e-8 (5e-9 while these shapes were cut into 5-12 k-split slabs; under stream-K a
    workgroup's accumulation chain is up to 4x longer and the MFMA's truncating adder shows it on all-positive data: measured -1.1e-8,
    a thirtieth of the native kernel's rms); rms at the native f32 kernel's level."""
    from ams_hip._lib import load
    lib = load()
    rng = np.random.RandomState(M + N + 1)
    for kind in ('randn', 'pos'):
        if kind == 'randn':
            A, B = rng.randn(K, M), rng.randn(K, N)
        else:
            A, B = rng.uniform(0.5, 1.0, (K, M)), rng.uniform(0.5, 1.0, (K, N))
        A32, B32 = A.astype(np.float32), B.astype(np.float32)
        ref = A32.astype(np.float64).T @ B32.astype(np.float64)
        scale = np.sqrt(K) if kind == 'randn' else np.abs(ref).mean()
        a, b = torch.from_numpy(A32).cuda(), torch.from_numpy(B32).cuda()
        bounds = (ops.absmax(a), ops.absmax(b))
        lib.ams_gemm_set_arith(0)
        try:
            d0 = (ops.gemm(a, b, transA=True).double().cpu().numpy() - ref) / scale
        finally:
            lib.ams_gemm_set_arith(1)
        free = (ops.gemm(a, b, transA=True, amax=bounds).double().cpu().numpy() - ref) / scale
        with ops.lds_pad(50000):
            cap = (ops.gemm(a, b, transA=True, amax=bounds).double().cpu().numpy() - ref) / scale
        r0 = np.sqrt((d0 ** 2).mean())
        print('fp16x3 %s %dx%dx%d: native mean %.2e rms %.2e | free mean %.2e rms %.2e | capped mean %.2e rms %.2e'
              % (kind, M, N, K, d0.mean(), r0, free.mean(), np.sqrt((free ** 2).mean()), cap.mean(), np.sqrt((cap ** 2).mean())))
        assert abs(free.mean()) < 2e-8 and np.sqrt((free ** 2).mean()) <= 1.05 * r0, (kind, free.mean(), d0.mean())
        assert abs(cap.mean()) < 3e-7 and np.sqrt((cap ** 2).mean()) <= 1.5 * r0, (kind, cap.mean(), d0.mean())


def test_nan_and_inf_propagate(ops):
    rng = np.random.RandomState(14)
    A, B, _, _ = _ops_pair(rng, 128, 128, 64, 0, 0)
    A[3, 5] = float('nan')
    out = ops.gemm(A, B, amax=(ops.absmax(A), ops.absmax(B)))
    assert torch.isnan(ops.absmax(A)).all() and torch.isnan(out[3]).all() and torch.isfinite(out[4]).all()


def test_the_training_step_takes_fp16x3_wherever_bounds_are_at_hand(ops):
    """A front_DPCL step at a shape whose products all take the 16-byte-fetch path: every product, the front conv included (round 5),
    is launched with bounds -- a regression to bf16x6 would be silent otherwise -- and the only bounds MEASURED by a pass of their own
    in an EAGER step are the waveforms', the frozen filter's and the weights' (a replayed step measures none of the three: the staging
    launch, the frozen-filter cache and the optimizer kernel supply them); the front output's comes out of the conv launch, dZ and dU
    bring theirs from the kernels that wrote them."""
    import tempfile
    from tests.smoke_step import build_front_dpcl
    tmp = tempfile.mkdtemp(prefix='ams_f16_step_')
    ops.PASS[0] += 10                                             # models other tests left alive are no longer 'recently used' (ops.pass_begin)
    trainer, tfds = build_front_dpcl(tmp, B=16, L=4096, W=64, N=64, hop=64, layer_size=600, nb_layers=2, E=40, no_summaries=True)
    g, model = trainer.graph, trainer.model
    calls = []
    orig = ops.absmax

    def counting(t, out=None):
        calls.append(tuple(t.shape))
        return orig(t, out=out)
    ops.absmax = counting
    try:
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 4096}
            tfds.initialize(tfds.TRAIN)
            model.train(feed, 0)
            del calls[:]
            ops.PROFILE.reset(enabled=True)
            c = float(model.train(feed, 1))
            ops.PROFILE.enabled = False
        tags = [r[4] for r in ops.PROFILE.records if r[4].startswith('gemm')]
    finally:
        ops.absmax = orig
        ops.PROFILE.reset(False)
    assert np.isfinite(c)
    n16 = sum(t.startswith('gemm16') for t in tags)
    assert n16 >= 11 and len(tags) == n16, tags                  # front conv, 2 projections, dense fwd/dX/dW, 1 LSTM dX, 2 dWx, 2 dU
    assert len(calls) <= 3, calls                                 # the waveforms, the frozen filter, the optimizer's flat weight buffer


def test_audited_training_step_and_recapture_after_a_denial(ops):
    """Network.train_audited: one eager step with the range audit on, inside a hipGraph-replayed run.  Nothing is denied on sane data and
    the replay goes on; with the limit forced to zero every audited class is denied, the captured step is dropped, the next calls
    re-capture, and the step then runs those products as bf16x6 -- same cost trajectory to 1e-4."""
    import tempfile
    from tests.smoke_step import build_front_dpcl
    costs = {}
    for mode in ('plain', 'audited', 'denied'):
        tmp = tempfile.mkdtemp(prefix='ams_audit_')
        ops.PASS[0] += 10
        trainer, tfds = build_front_dpcl(tmp, B=8, L=4096, W=64, N=64, hop=64, layer_size=600, nb_layers=2, E=40, no_summaries=True, hip_graph=True)
        g, model = trainer.graph, trainer.model
        old_limit = ops.F16_AUDIT.limit
        cs = []
        try:
            with g.as_default():
                feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: 4096}
                tfds.initialize(tfds.TRAIN)
                for i in range(8):
                    if mode != 'plain' and i == 4:
                        if mode == 'denied':
                            ops.F16_AUDIT.limit = -1.0
                        c, new = model.train_audited(feed, i)
                        assert (len(new) > 0) == (mode == 'denied'), (mode, new)
                        assert ('_cg_state' in model.__dict__) == (mode != 'denied')
                    else:
                        c = model.train(feed, i)
                    cs.append(float(c))
        finally:
            ops.F16_AUDIT.limit = old_limit
            if mode == 'denied':
                assert len(ops.F16_AUDIT.denied) >= 5
            ops.F16_AUDIT.denied.clear()
        costs[mode] = np.array(cs)
        ops.raise_on_ring_errors()
    assert np.isfinite(costs['denied']).all()
    assert np.abs(costs['audited'] - costs['plain']).max() <= 1e-5 * np.abs(costs['plain']).max()
    assert np.abs(costs['denied'] - costs['plain']).max() <= 1e-4 * np.abs(costs['plain']).max()
