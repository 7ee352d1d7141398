"""GPU: gradient of the unrolled soft k-means (needed by the front_*_finetuning recipes) against torch autograd of a
float64 CPU restatement of reference models/Kmeans_2.py (soft branch)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def torch_soft_kmeans(X, idx, C, tries, iters, beta, w, end, faithful=True):
    """float64 CPU restatement with autograd (Kmeans_2.py:40-188, beta branch)."""
    b, L, E = X.shape
    xn = X * torch.rsqrt(torch.clamp((X * X).sum(-1, keepdim=True), min=1e-12))
    cents, inert = [], []
    ones = torch.ones(L, dtype=X.dtype)
    for r in range(b * tries):
        x = xn[r // tries]
        wr = ones if w is None else (w[r % b] if faithful else w[r // tries])
        c = x[idx[r]]

        def lab_of(c, wv):
            d = (((x[:, None, :] - c[None]) ** 2) * wv[:, None, None]).sum(-1)
            e = torch.exp(-beta * d)
            return e / e.sum(1, keepdim=True)
        lab = lab_of(c, wr)
        for _ in range(iters):
            c = ((x * wr[:, None])[:, None, :] * lab[:, :, None]).sum(0) / lab.sum(0)[:, None]
            lab = lab_of(c, wr)
        d = ((x[:, None, :] - c[None]) ** 2).sum(-1)
        inert.append(((d * lab).sum(0) / lab.sum(0)).sum())
        cents.append((c, lab, x))
    best = torch.stack(inert).reshape(b, tries).argmin(1)
    sel, out = [], []
    for i in range(b):
        c, lab, x = cents[i * tries + int(best[i])]
        sel.append(c)
        if end:
            d = ((x[:, None, :] - c[None]) ** 2).sum(-1)
            e = torch.exp(-beta * d)
            lab = e / e.sum(1, keepdim=True)
        out.append(lab)
    return torch.stack(sel), torch.stack(out), best


@pytest.mark.parametrize('b,L,E,C,tries,iters,with_w,end', [(2, 3000, 40, 2, 2, 3, True, True), (2, 2500, 8, 3, 1, 4, True, False),
                                                              (1, 4200, 40, 2, 1, 2, False, True), (8, 20480, 40, 2, 1, 3, True, True)])
def test_soft_kmeans_backward(b, L, E, C, tries, iters, with_w, end, monkeypatch):
    from ams_hip import functional as F, ops
    tagged = []
    real_tag = ops.tag_amax
    monkeypatch.setattr(ops, 'tag_amax', lambda t, a: (tagged.append((t, a)), real_tag(t, a))[1])
    rng = np.random.RandomState(L + C)
    centers = rng.randn(C, E) * 1.5
    X = centers[rng.randint(0, C, (b, L))] + rng.randn(b, L, E) * 0.8
    w = (rng.rand(b, L) > 0.2).astype(np.float64) if with_w else None
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)])
    R1, R2 = rng.randn(b, L, C), rng.randn(b, C, E)
    beta = 3.0
    Xt = torch.from_numpy(X).requires_grad_()
    sel_r, out_r, best_r = torch_soft_kmeans(Xt, torch.from_numpy(idx), C, tries, iters, beta, None if w is None else torch.from_numpy(w), end)
    ((out_r * torch.from_numpy(R1)).sum() + (sel_r * torch.from_numpy(R2)).sum()).backward()

    Xd = torch.from_numpy(X.astype(np.float32)).cuda().requires_grad_()
    wd = None if w is None else torch.from_numpy(w.astype(np.float32)).cuda()
    sel, out, best = F.kmeans(Xd, torch.from_numpy(idx.astype(np.int32)).cuda(), C, tries, iters, beta, wd, end)
    assert np.array_equal(best.cpu().numpy(), best_r.numpy())
    assert np.abs(out.detach().cpu().numpy() - out_r.detach().numpy()).max() < 1e-3
    ((out * torch.from_numpy(R1.astype(np.float32)).cuda()).sum() + (sel * torch.from_numpy(R2.astype(np.float32)).cuda()).sum()).backward()
    g, g_ref = Xd.grad.cpu().numpy().astype(np.float64), Xt.grad.numpy()
    err = np.abs(g - g_ref).max() / np.abs(g_ref).max()
    assert err < 1e-3, err
    # ABI 5: the pass that writes dx (and the one that adds the seed rows' share) leaves max |dx| -- the bound the dense layer's gradient
    # products take instead of a pass over dx
    if ops.F16X3:
        dxs = [(t, a) for t, a in tagged if t.shape == Xd.shape]
        assert len(dxs) == 1
        assert float(dxs[0][1]) == float(dxs[0][0].abs().max())


@pytest.mark.parametrize('b,L,E,C,tries,iters,with_w,end', [(2, 3000, 40, 2, 1, 3, True, True), (2, 2500, 8, 3, 2, 4, True, False),
                                                              (1, 4200, 40, 2, 1, 2, False, True), (2, 1111, 20, 2, 1, 2, False, True),
                                                              (8, 20480, 40, 2, 1, 4, True, True)])
def test_soft_kmeans_from_the_unnormalised_embeddings(b, L, E, C, tries, iters, with_w, end, monkeypatch):
    """A fine-tuning step hands the k-means the embedding network's output BEFORE its Normalize layer (F.kmeans(pre_norm=...)): both
    normalisations in one pass (ams_l2norm2_fwd), both Jacobians in the pass that writes the gradient (ams_kmeans_soft_bwd inv / inv0).
    Forward: the bits of Normalize -> k-means(normalize_input).  Backward: against float64 autograd of the same composition, and close
    to the two-pass HIP form.  The last shape is a fine-tuning step's (8 utterances x 20480 points: the launch uses fewer chunks per
    utterance than the workspace query assumes -- the word that receives max |dx| must still be where the query says)."""
    from ams_hip import functional as F, ops
    tagged = []
    real_tag = ops.tag_amax
    monkeypatch.setattr(ops, 'tag_amax', lambda t, a: (tagged.append((t, a)), real_tag(t, a))[1])
    rng = np.random.RandomState(L + C + 1)
    centers = rng.randn(C, E) * 1.5
    U = (centers[rng.randint(0, C, (b, L))] + rng.randn(b, L, E) * 0.8) * np.exp(rng.randn(b, L, 1))      # rows of very different length
    w = (rng.rand(b, L) > 0.2).astype(np.float64) if with_w else None
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)])
    R1, R2 = rng.randn(b, L, C), rng.randn(b, C, E)
    beta = 3.0
    Ut = torch.from_numpy(U).requires_grad_()
    Vt = Ut * torch.rsqrt(torch.clamp((Ut * Ut).sum(-1, keepdim=True), min=1e-12))
    sel_r, out_r, best_r = torch_soft_kmeans(Vt, torch.from_numpy(idx), C, tries, iters, beta, None if w is None else torch.from_numpy(w), end)
    ((out_r * torch.from_numpy(R1)).sum() + (sel_r * torch.from_numpy(R2)).sum()).backward()

    idx_d = torch.from_numpy(idx.astype(np.int32)).cuda()
    wd = None if w is None else torch.from_numpy(w.astype(np.float32)).cuda()
    r1, r2 = torch.from_numpy(R1.astype(np.float32)).cuda(), torch.from_numpy(R2.astype(np.float32)).cuda()
    res = {}
    for form in ('fused', 'two_pass'):
        Ud = torch.from_numpy(U.astype(np.float32)).cuda().requires_grad_()
        if form == 'fused':
            never = lambda: (_ for _ in ()).throw(AssertionError('the normalised tensor was asked for'))      # noqa: E731
            sel, out, best = F.kmeans(never, idx_d, C, tries, iters, beta, wd, end, pre_norm=lambda: Ud)
        else:
            V = F.l2norm_keep(Ud.reshape(b, L * E), E)[0].reshape(b, L, E)
            sel, out, best = F.kmeans(V, idx_d, C, tries, iters, beta, wd, end)
        ((out * r1).sum() + (sel * r2).sum()).backward()
        res[form] = (sel.detach().cpu().numpy(), out.detach().cpu().numpy(), best.cpu().numpy(), Ud.grad.cpu().numpy().astype(np.float64))
    f, t = res['fused'], res['two_pass']
    assert np.array_equal(f[0], t[0]) and np.array_equal(f[1], t[1]) and np.array_equal(f[2], t[2])      # forward: the same bits
    assert np.array_equal(f[2], best_r.numpy())
    g_ref = Ut.grad.numpy()
    for name, g in (('fused', f[3]), ('two_pass', t[3])):
        err = np.abs(g - g_ref).max() / np.abs(g_ref).max()
        assert err < 1e-3, (name, err)
    assert np.abs(f[3] - t[3]).max() / np.abs(t[3]).max() < 1e-5
    if ops.F16X3:                                       # ABI 5: every soft backward left the bound of the gradient it wrote
        dxs = [(t_, a_) for t_, a_ in tagged if tuple(t_.shape) == (b, L, E)]
        assert len(dxs) == 2
        for t_, a_ in dxs:
            assert float(a_) == float(t_.abs().max())
