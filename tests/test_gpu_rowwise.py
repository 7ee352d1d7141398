"""GPU: ROW-WISE accuracy of the three product arithmetics on the REAL operands of one benchmark-shape training step.

VERDICT r03 / ADVICE r03: fp16x3 scales each operand by ONE power of two taken from a per-tensor bound, so an entry below bound * 2^-17
keeps fewer than 22 bits; every accuracy assertion so far was norm-wise over the whole product.  BPTT gradients (dZ, dU) are the
wide-dynamic-range operands where that could matter.  This file captures every product of a B = 64, T = 80, 3 x BLSTM(600), E = 40
front_DPCL step (operands, bounds and launch flags exactly as the step issues them -- second step after initialisation, so dZ / dU
carry a real BPTT), re-runs each one as fp16x3 (with the step's own bounds), bf16x6 and the native f32 MFMA kernel, and compares
every OUTPUT ROW and every OUTPUT COLUMN with a float64 product of the same f32 operands:

  * rows / columns whose operand slice lies within 2^17 of the bound the kernel scaled with ("in range": max |A[m, :]| >= bound_A 2^-17,
    max |B[:, n]| >= bound_B 2^-17): relative error (l2 over the row / column) <= 4 x max(native error of that row, median native error);
  * the rest: reported (count, their share of the output's energy, quantiles of their error relative to native) and held to an
    ABSOLUTE bound -- error <= 2^-20 of the largest in-range row norm -- i.e. what they lose is invisible next to the rows that matter.

Reference arithmetic: f32 tf.matmul / dynamic_rnn (reference utils/ops.py:366-383, 501-503)."""
import os
import tempfile

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

B, L, W, N, HOP, LS, NL, E = 64, 20480, 1024, 256, 256, 600, 3, 40


def _capture_step_products(pre_steps=1, silence=False, audit=None):
    """Run `pre_steps` eager training steps; record every product launch of the next one (operands cloned).  silence: the second half of
    every waveform of the batch is digital silence (exact zeros in x_mix and x_non_mix alike).  audit: a list that receives the range
    audit's report of that same step (ops.F16_AUDIT)."""
    from tests.smoke_step import build_front_dpcl
    from ams_hip import ops
    tmp = tempfile.mkdtemp(prefix='ams_rowwise_')
    trainer, tfds = build_front_dpcl(tmp, B=B, L=L, W=W, N=N, hop=HOP, layer_size=LS, nb_layers=NL, E=E, no_summaries=True)
    g, model = trainer.graph, trainer.model
    if silence:
        orig_next, done = tfds._next, set()

        def silent(run):
            mix, non_mix, ind = orig_next(run)                      # the pooled device tensors themselves (data/dataset.py::_next)
            if mix.data_ptr() not in done:
                mix[..., mix.shape[-1] // 2:] = 0.0
                non_mix[..., non_mix.shape[-1] // 2:] = 0.0
                done.add(mix.data_ptr())
            return mix, non_mix, ind
        tfds._next = silent
    # SURVEY 8(d) bench initialisation of the dense layer (the reference's +-12 would saturate nothing but is not what bench.py times)
    gen = torch.Generator(device='cpu').manual_seed(9)
    Wd = g.variables['prediction/W']
    Wd.data.copy_((torch.rand(Wd.shape, generator=gen) * 0.1 - 0.05).to(Wd.device))
    rec = []
    orig = {k: getattr(ops, k) for k in ('gemm', 'gemm_at_b_colsum', 'gemm_batched')}
    state = {'on': False}

    def view2d(t, rows, cols, ld):
        return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset()).clone()

    def gemm(A, B_, transA=False, transB=False, bias=None, out=None, accumulate=False, M=None, N=None, K=None, lda=None, ldb=None,
             ldc=None, mask=(0, 0), label='', amax=None):
        if state['on']:
            if M is None:
                Ac, Bc = A.clone(), B_.clone()
            else:
                Ac = view2d(A, K if transA else M, M if transA else K, lda)
                Bc = view2d(B_, N if transB else K, K if transB else N, ldb)
            rec.append({'name': label or 'gemm<%d,%d>' % (int(transA), int(transB)), 'A': Ac, 'B': Bc, 'tA': bool(transA), 'tB': bool(transB),
                        'mask': mask, 'amax': None if amax is None or amax[0] is None or amax[1] is None else (amax[0].clone(), amax[1].clone()),
                        'pad': ops.LDS_PAD[0]})
        return orig['gemm'](A, B_, transA, transB, bias, out, accumulate, M, N, K, lda, ldb, ldc, mask, label, amax)

    def gemm_at_b_colsum(A, B_, out, bsum, accumulate=True, amax=None, ldc=None):
        ok = orig['gemm_at_b_colsum'](A, B_, out, bsum, accumulate, amax, ldc)
        if state['on'] and ok:
            rec.append({'name': 'at_b_colsum', 'A': A.clone(), 'B': B_.clone(), 'tA': True, 'tB': False, 'mask': (0, 0),
                        'amax': None if amax is None else (amax[0].clone(), amax[1].clone()), 'pad': ops.LDS_PAD[0]})
        return ok

    def gemm_batched(A, B_, C, nbatch, a_zs, b_zs, c_zs, transA, transB, M, N, K, lda, ldb, ldc, accumulate=False, amax=None, mask=(0, 0)):
        if state['on']:
            for z in range(nbatch):
                Az = torch.as_strided(A, (K if transA else M, M if transA else K), (lda, 1), A.storage_offset() + z * a_zs).clone()
                Bz = torch.as_strided(B_, (N if transB else K, K if transB else N), (ldb, 1), B_.storage_offset() + z * b_zs).clone()
                rec.append({'name': 'batched[%d]' % z, 'A': Az, 'B': Bz, 'tA': bool(transA), 'tB': bool(transB), 'mask': mask,
                            'amax': None if amax is None else (amax[0].clone(), amax[1].clone()), 'pad': ops.LDS_PAD[0]})
        return orig['gemm_batched'](A, B_, C, nbatch, a_zs, b_zs, c_zs, transA, transB, M, N, K, lda, ldb, ldc, accumulate, amax, mask)

    ops.gemm, ops.gemm_at_b_colsum, ops.gemm_batched = gemm, gemm_at_b_colsum, gemm_batched
    try:
        with g.as_default():
            feed = {tfds.handle: tfds.get_handle(tfds.TRAIN), tfds.chunk_size: L}
            for i in range(pre_steps):
                model.train(feed, i)
            state['on'] = True
            if audit is not None:
                ops.F16_AUDIT.begin()
            model.train(feed, pre_steps)
            torch.cuda.synchronize()
            if audit is not None:
                denied = ops.F16_AUDIT.finish()
                audit.extend(ops.F16_AUDIT.report)
                assert not denied, denied
    finally:
        for k, v in orig.items():
            setattr(ops, k, v)
    ops.raise_on_ring_errors()
    return rec


def _errors(C, ref):
    """(row-wise, column-wise) l2 errors and norms, float64 numpy."""
    d = (C.double() - ref)
    return (d.norm(dim=1).cpu().numpy(), ref.norm(dim=1).cpu().numpy(), d.norm(dim=0).cpu().numpy(), ref.norm(dim=0).cpu().numpy())


@pytest.mark.parametrize('state', ['fresh', 'trained300', 'half_silent'])
def test_every_product_of_the_step_row_by_row(state):
    """fresh: step 2 of a freshly initialised model (round 4).  trained300: the same after 300 training steps (gates that have started
    to saturate, weights that have moved).  half_silent: a batch whose second half is digital silence -- exact
    zeros are outside the question (0 is exact in any arithmetic), what matters is what the silence does to the range of everything
    downstream.  The run-time guard (ops.F16_AUDIT) audits the SAME step: it must deny nothing, and its report is kept."""
    from ams_hip import ops
    from ams_hip._lib import load
    lib = load()
    audit = []
    recs = _capture_step_products(pre_steps={'fresh': 1, 'trained300': 300, 'half_silent': 1}[state], silence=state == 'half_silent',
                                  audit=audit)
    assert len(recs) >= 12, [r['name'] for r in recs]
    assert any(r['amax'] is not None for r in recs), 'the step passed no operand bounds: fp16x3 is not what it runs'
    lines, failures = [], []
    for i, r in enumerate(recs):
        A, Bm, tA, tB = r['A'], r['B'], r['tA'], r['tB']
        if r['mask'][0]:                                            # masked reduction rows (time-shifted h^T dZ): zero them in the reference
            period, skip = r['mask']
            k = torch.arange(A.shape[0] if tA else A.shape[1], device=A.device)
            keep = ((k % period) != skip).double()
        A64 = (A.t() if tA else A).double()
        B64 = (Bm.t() if tB else Bm).double()
        if r['mask'][0]:
            A64 = A64 * keep[None, :]
        ref = A64 @ B64
        kw = dict(transA=tA, transB=tB, mask=r['mask'])
        with ops.lds_pad(r['pad']):
            c6 = ops.gemm(A, Bm, **kw)
            c16 = ops.gemm(A, Bm, amax=r['amax'], **kw) if r['amax'] is not None else None
        # the forward products run from pre-split operand images outside an audited step (ops.forward_product, csrc/gemm_ps.hip): same
        # operands, same bounds, held to the same criterion
        cps = None
        if r['amax'] is not None and not tA and not tB and not r['mask'][0] and r['pad'] == 0 and Bm.shape[1] % 4 == 0 and ops.PRESPLIT:
            cps = ops.gemm_ps(ops.ps_pack_rows(A, r['amax'][0]), ops.ps_pack_cols(Bm, r['amax'][1]), A.shape[1], r['amax'])
        lib.ams_gemm_set_arith(0)
        try:
            c32 = ops.gemm(A, Bm, **kw)
        finally:
            lib.ams_gemm_set_arith(1)
        torch.cuda.synchronize()
        e32 = _errors(c32, ref)
        rowmax = A64.abs().amax(dim=1).cpu().numpy()
        colmax = B64.abs().amax(dim=0).cpu().numpy()
        bound_a = float(r['amax'][0]) if r['amax'] is not None else float(rowmax.max())
        bound_b = float(r['amax'][1]) if r['amax'] is not None else float(colmax.max())
        assert bound_a >= rowmax.max() and bound_b >= colmax.max(), 'operand bound below the operand maximum'
        for arith, c in (('fp16x3', c16), ('fp16x3 pre-split', cps), ('bf16x6', c6)):
            if c is None:
                continue
            e = _errors(c, ref)
            for axis, (err, nrm, err0, opmax, bound) in (('rows', (e[0], e[1], e32[0], rowmax, bound_a)),
                                                         ('cols', (e[2], e[3], e32[2], colmax, bound_b))):
                live = nrm > 0
                inr = live & (opmax >= bound * 2.0 ** -17)
                rel = np.where(live, err / np.maximum(nrm, 1e-300), 0.0)
                rel0 = np.where(live, err0 / np.maximum(nrm, 1e-300), 0.0)
                floor = np.median(rel0[inr]) if inr.any() else 0.0
                ratio = rel[inr] / np.maximum(rel0[inr], floor) if inr.any() else np.zeros(1)
                worst = float(ratio.max())
                out = live & ~inr
                big = nrm[inr].max() if inr.any() else 0.0
                o_abs = float((err[out] / big).max()) if out.any() and big > 0 else 0.0
                o_share = float((nrm[out] ** 2).sum() / max((nrm ** 2).sum(), 1e-300))
                o_q = np.quantile(rel[out] / np.maximum(rel0[out], 1e-300), [0.5, 0.99, 1.0]) if out.any() else np.zeros(3)
                lines.append('%2d %-18s %dx%dx%d pad=%-5d %s %s: in-range %d/%d  worst err/native %.2f (median %.2f; native median rel %.1e)'
                             '  | out of range %d (energy share %.1e): err/native q50 %.1f q99 %.1f max %.1f, abs err / largest row %.1e'
                             % (i, r['name'], ref.shape[0], ref.shape[1], A64.shape[1], r['pad'], arith, axis, int(inr.sum()), int(live.sum()),
                                worst, float(np.median(ratio)), floor, int(out.sum()), o_share, o_q[0], o_q[1], o_q[2], o_abs))
                if worst > 4.0:
                    failures.append((i, r['name'], arith, axis, 'in-range ratio', worst))
                if o_abs > 2.0 ** -20:
                    failures.append((i, r['name'], arith, axis, 'out-of-range abs', o_abs))
    lines.append('# run-time range audit of the same step (ops.F16_AUDIT): product class, operand, non-zero entries below bound * 2^-17, '
                 'their energy share, estimated relative error of the lost bits (limit %.1e)' % ops.F16_AUDIT.limit)
    for key, role, n_small, share, r, frac in audit:
        lines.append('#   %-60s %s  n_small %-10d energy share %.2e  est. rel. error %.2e  outer slices out of range %.2e'
                     % (str(key), role, int(n_small), share, r, frac))
    report = '\n'.join(lines)
    print(report)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'rowwise_products_%s.txt' % state), 'w') as f:
            f.write(report + '\n')
    assert audit and max(a[4] for a in audit) <= ops.F16_AUDIT.limit and max(a[5] for a in audit) <= ops.F16_AUDIT.row_limit
    assert not failures, failures


def test_range_audit_denies_a_product_whose_operand_leaves_the_fp16_range():
    """An operand with most of its rows 2^-30 below its bound: the audit's estimate exceeds the f32 level, the class is denied, and the
    next launch of that class runs as bf16x6 (bit-equal to a launch without bounds) although bounds are passed."""
    from ams_hip import ops
    rng = np.random.RandomState(3)
    A = torch.from_numpy(rng.randn(512, 256).astype(np.float32)).cuda()
    A[8:] *= 2.0 ** -30
    Bm = torch.from_numpy(rng.randn(256, 384).astype(np.float32)).cuda()
    am = (ops.absmax(A), ops.absmax(Bm))
    key = ('gemm', 'audit_test', 512, 384, 256, False, False)
    ops.F16_AUDIT.denied.discard(key)
    c16 = ops.gemm(A, Bm, amax=am, label='audit_test')
    c6 = ops.gemm(A, Bm, label='audit_test')
    assert not torch.equal(c16, c6)
    ops.F16_AUDIT.begin()
    ops.gemm(A, Bm, amax=am, label='audit_test')
    new = ops.F16_AUDIT.finish()
    try:
        assert new == [key], (new, ops.F16_AUDIT.report)
        rep = {r[1]: r for r in ops.F16_AUDIT.report}
        assert rep['A'][5] > 0.9 and rep['B'][5] == 0.0 and rep['B'][4] <= ops.F16_AUDIT.limit       # 504 of 512 rows of A out of range
        assert torch.equal(ops.gemm(A, Bm, amax=am, label='audit_test'), c6)          # denied: bf16x6 whatever the caller passes
        Aok = torch.from_numpy(rng.randn(512, 256).astype(np.float32)).cuda()
        ops.F16_AUDIT.begin()
        ops.gemm(Aok, Bm, amax=(ops.absmax(Aok), am[1]), label='audit_ok')
        assert ops.F16_AUDIT.finish() == []
    finally:
        ops.F16_AUDIT.denied.discard(key)
