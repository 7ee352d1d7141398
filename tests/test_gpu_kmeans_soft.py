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
                                                              (1, 4200, 40, 2, 1, 2, False, True)])
def test_soft_kmeans_backward(b, L, E, C, tries, iters, with_w, end):
    from ams_hip import functional as F
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
