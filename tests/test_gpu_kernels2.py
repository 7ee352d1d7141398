"""GPU parity (second batch): synthesis, waveform statistics / costs, masks, STFT / iSTFT, L41 loss, k-means --
each through the C ABI against the float64 (or, for bit-exact k-means labels, float32) numpy oracle."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import front as ofront, stft as ostft, l41 as ol41, losses as olosses, separate as osep, kmeans as okm

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
def F():
    from ams_hip import functional as f
    return f


@pytest.fixture(scope='module')
def ops():
    from ams_hip import ops as o
    return o


@pytest.mark.parametrize('R,L,W,N,hop', [(3, 640, 64, 6, 16), (4, 4096, 1024, 256, 256), (2, 1000, 128, 8, 48)])
def test_synth_strided(F, R, L, W, N, hop):
    rng = np.random.RandomState(L)
    T = -(-L // hop)
    z, f2, dout = rng.randn(R, T, N), rng.randn(W, N), rng.randn(R, L)
    zt, ft = dev(z).requires_grad_(), dev(f2).requires_grad_()
    out = F.synth_strided(zt, ft, hop, L)
    assert rel(host(out), ofront.synth_strided(z, f2, hop, L)) < TOL
    out.backward(dev(dout))
    dz, df2 = ofront.synth_strided_bwd(z, f2, hop, dout)
    assert rel(host(zt.grad), dz) < TOL and rel(host(ft.grad), df2) < TOL


@pytest.mark.parametrize('S', [2, 3])
def test_pair_stats_and_costs(F, S):
    rng = np.random.RandomState(S)
    B, L = 5, 3000
    xn, bk = rng.randn(B, S, L) * 0.1, rng.randn(B, S, L) * 0.1
    xm = xn.sum(1)
    for kind in ('l2', 'sdr', 'l2+sdr'):
        bt = dev(bk).requires_grad_()
        p = F.pretrain_cost(dev(xm), dev(xn), bt)
        loss = p[0] if kind == 'l2' else p[1] if kind == 'sdr' else p[0] + p[1]
        lo, l2, sdr = olosses.pretrain_cost(xm, xn, bk, kind)
        assert abs(float(loss) - lo) < TOL * max(1.0, abs(lo))
        imp, _ = olosses.sdr_improvement(xm, xn, bk)
        assert abs(float(p[2]) - imp) < 1e-4 * max(1.0, abs(imp))
        loss.backward()
        assert rel(host(bt.grad), olosses.pretrain_cost_bwd(xn, bk, kind)) < 5 * TOL
    # PIT cost of the fine-tune recipes (0.5 sum_l, mean_s, min_perm, mean_b) + sub-gradient
    bt = dev(bk).requires_grad_()
    c = F.pit_l2(dev(xn), bt, 'sum', 'mean', 0.5)
    c_ref, best = olosses.cost_finetuning(xn, bk)
    assert abs(float(c) - c_ref) < TOL * max(1.0, abs(c_ref))
    c.backward()
    assert rel(host(bt.grad), olosses.pit_l2_bwd(xn, bk, best, 'sum', 'mean', 0.5)) < 5 * TOL
    # Adapt.cost non-pretraining branch, including the cross-batch SDR broadcast (quirk C-3)
    bt = dev(bk).requires_grad_()
    p = F.pit_cost_adapt(dev(xm), dev(xn), bt)
    lo, l2, sdr = olosses.pit_cost_adapt(xm, xn, bk, 'sdr+l2')
    assert abs(float(p[0]) - l2) < TOL * max(1.0, abs(l2)) and abs(float(p[1]) - sdr) < 1e-4 * max(1.0, abs(sdr))
    imp_ref, _ = olosses.sdr_improvement(xm, xn[:, None], bk, True)
    assert abs(float(p[2]) - imp_ref) < 1e-3 * max(1.0, abs(imp_ref))
    # gradient of the quirky sdr term vs torch autograd on the CPU restatement
    bt2 = torch.from_numpy(bk).requires_grad_()
    t_ = torch.from_numpy(xn)
    tn, an = (t_ ** 2).sum(-1), (bt2 ** 2).sum(-1)
    ts2 = ((t_[:, None] * bt2[None]).sum(-1)) ** 2
    sdr_t = (tn[:, None] * an[None]) / (ts2 + 1e-12)
    sdr_t.min(1)[0].sum(-1).mean().backward()
    p[1].backward()
    assert rel(host(bt.grad), bt2.grad.numpy()) < 1e-3


def test_overlap_metric(F):
    rng = np.random.RandomState(3)
    B, S, T, N = 3, 3, 5, 7
    y = rng.randn(B * (S + 1), T, N)
    yt = dev(y).requires_grad_()
    ov = F.overlap_metric(yt, B, S)
    assert abs(float(ov) - ofront.overlap_metric(y, B, S)) < 1e-6
    (ov * 3.0).backward()
    assert rel(host(yt.grad), 3.0 * ofront.overlap_metric_bwd(y, B, S)) < 1e-5


def test_apply_masks(F):
    rng = np.random.RandomState(4)
    B, T, Fq, S = 2, 5, 6, 3
    X, m = rng.randn(B, T, Fq), rng.rand(B, T * Fq, S)
    mt = dev(m).requires_grad_()
    sep = F.apply_masks(dev(X), mt)
    assert rel(host(sep), osep.apply_masks(X, m)) < 1e-6
    d = rng.randn(B * S, T, Fq)
    sep.backward(dev(d))
    assert rel(host(mt.grad), osep.apply_masks_bwd(X, d, S)) < 1e-6


@pytest.mark.parametrize('R,L,W,hop,S', [(4, 2048, 256, 128, 2), (6, 20480, 512, 256, 3), (2, 1500, 128, 32, 1)])
def test_stft_istft(F, R, L, W, hop, S):
    rng = np.random.RandomState(W)
    x = rng.randn(R, L)
    mag, ph = F.stft_mag_phase(dev(x), W, hop)
    s = ostft.stft(x, W, hop)
    assert rel(host(mag), np.abs(s)) < TOL
    T, Fq = mag.shape[1:]
    big = np.abs(s) > 1e-3 * np.abs(s).max()
    phn = host(ph).reshape(R, T, 2 * Fq)
    assert np.abs(phn[..., :Fq] - np.cos(np.angle(s)))[big].max() < 1e-3
    assert np.abs(phn[..., Fq:] - np.sin(np.angle(s)))[big].max() < 1e-3
    # inverse with the (mixture) phase tiled over S speakers: rows (b,s)
    B = R
    sep = rng.rand(B * S, T, Fq)
    ang = np.repeat(np.angle(s), S, axis=0)
    st = dev(sep).requires_grad_()
    out = F.istft(st, ph, W, hop, S)
    ref = ostft.istft(sep, ang, W, hop)
    assert out.shape[1] == (T - 1) * hop + W and rel(host(out), ref) < 5 * TOL
    dout = rng.randn(*ref.shape)
    out.backward(dev(dout))
    assert rel(host(st.grad), ostft.istft_bwd(ang, W, hop, dout)) < 5 * TOL


@pytest.mark.parametrize('normalize', [True, False])
def test_l41_loss(F, normalize):
    rng = np.random.RandomState(6)
    B, T, Fq, E, S, NS = 3, 4, 70, 40, 2, 11
    emb, spk = rng.randn(B, T, Fq, E) * 0.3, rng.randn(NS, E)
    I = np.array([[0, 3], [3, 5], [10, 1]], dtype=np.int32)
    y = np.where(rng.rand(B, T, Fq, S) > 0.5, 1.0, -1.0)
    et, st = dev(emb).requires_grad_(), dev(spk).requires_grad_()
    c = F.l41_loss(et, dev(y), st, dev(I, np.int32), normalize)
    c_ref = ol41.l41_cost(emb, y, spk, I, normalize)
    assert abs(float(c) - c_ref) < TOL * max(1.0, abs(c_ref))
    c.backward()
    de, ds = ol41.l41_cost_bwd(emb, y, spk, I, normalize)
    assert rel(host(et.grad), de) < 5 * TOL and rel(host(st.grad), ds) < 5 * TOL


@pytest.mark.parametrize('E,Fq', [(40, 70), (20, 33), (3, 50)])
@pytest.mark.parametrize('method,S,K', [(None, 2, 0), (None, 3, 0), ('k-nearest', 2, 5), ('random', 3, 16)])
def test_l41_loss_from_the_unnormalised_embeddings(F, E, Fq, method, S, K):
    """emb_is_u (K13 fused into K15): the loss takes the dense output BEFORE Normalize(3) (models/L41.py:43), normalises every point
    inside its own pass and returns the gradient w.r.t. that tensor.  Against the oracle's loss on l2-normalise(u) and its gradient
    pulled back through the normalise Jacobian -- rows of very different length, with and without negative sampling, 16-byte (E = 40,
    20) and 4-byte (E = 3) staging."""
    from oracle import dense as odense
    rng = np.random.RandomState(7 * E + S + K)
    B, T, NS, rate = 3, 4, 23, 0.3
    u = rng.randn(B, T, Fq * E) * np.exp(rng.randn(B, T, 1))
    u[0, 0, :E] = 0.0                                            # the epsilon clamp
    spk = rng.randn(NS, E)
    I = np.stack([rng.choice(NS, S, replace=False) for _ in range(B)]).astype(np.int32)
    lab = rng.randint(0, S, (B, T, Fq))
    y = np.where(np.eye(S)[lab] > 0, 1.0, -1.0)
    V, inv = odense.l2norm_fwd(u, E)                                            # [B,T,Fq,E]
    idx_ref = None
    if method == 'k-nearest':
        idx_ref = ol41.knearest_indices(spk, I, K, True)
    elif method == 'random':
        idx_ref = ol41.random_indices(I, NS, K, np.random.RandomState(3))
    c_ref = ol41.l41_cost(V, y, spk, I, True, idx_ref, rate) if idx_ref is not None else ol41.l41_cost(V, y, spk, I, True)
    dV, ds = ol41.l41_cost_bwd(V, y, spk, I, True, idx_ref, rate) if idx_ref is not None else ol41.l41_cost_bwd(V, y, spk, I, True)
    du_ref = odense.l2norm_bwd(V, inv, dV.reshape(V.shape)).reshape(u.shape)
    ut, st = dev(u).requires_grad_(), dev(spk).requires_grad_()
    idx = dev(idx_ref, np.int32) if idx_ref is not None else None
    c = F.l41_loss(ut, dev(y), st, dev(I, np.int32), True, neg_idx=idx, ns_rate=rate, from_u=True)
    assert abs(float(c) - c_ref) < TOL * max(1.0, abs(c_ref))
    c.backward()
    assert rel(host(ut.grad), du_ref) < 5 * TOL and rel(host(st.grad), ds) < 5 * TOL
    # and the two-pass form on the same input agrees (same loss, gradients to round-off)
    u2, s2 = dev(u).requires_grad_(), dev(spk).requires_grad_()
    c2 = F.l41_loss(F.l2norm(u2, E), dev(y), s2, dev(I, np.int32), True, neg_idx=idx, ns_rate=rate)
    c2.backward()
    assert abs(float(c2) - float(c)) < 1e-6 * max(1.0, abs(c_ref))
    assert rel(host(u2.grad), host(ut.grad)) < 1e-5


@pytest.mark.parametrize('normalize', [True, False])
@pytest.mark.parametrize('method,S,K', [('k-nearest', 2, 5), ('random', 2, 7), ('k-nearest', 3, 4), ('random', 3, 16)])
def test_l41_loss_negative_sampling(F, normalize, method, S, K):
    """--sampling K (models/L41.py:69-147,165-166): cost and both gradients against the oracle, with the neighbour sets the HIP
    side selects itself ('k-nearest': torch.topk == oracle argsort as SETS) or with injected sets ('random')."""
    rng = np.random.RandomState(60 + K)
    B, T, Fq, E, NS, rate = 3, 4, 70, 40, 23, 0.3
    emb, spk = rng.randn(B, T, Fq, E) * 0.3, rng.randn(NS, E)
    I = np.stack([rng.choice(NS, S, replace=False) for _ in range(B)]).astype(np.int32)
    lab = rng.randint(0, S, (B, T, Fq))
    y = np.where(np.eye(S)[lab] > 0, 1.0, -1.0)
    y[0, 0, :5] = -1.0                                          # bins with no dominant speaker: argmax -> 0
    et, st = dev(emb).requires_grad_(), dev(spk).requires_grad_()
    if method == 'k-nearest':
        idx_ref = ol41.knearest_indices(spk, I, K, normalize)
        idx = F.l41_knearest(st, dev(I, np.int32), K, normalize)
        assert idx.shape == (B, S, K)
        assert np.array_equal(np.sort(idx.cpu().numpy(), axis=2), np.sort(idx_ref, axis=2))
        assert all(I[b, s] in idx_ref[b, s] for b in range(B) for s in range(S))      # the speaker itself is its nearest neighbour
    else:
        idx_ref = ol41.random_indices(I, NS, K, np.random.RandomState(3))
        idx = dev(idx_ref, np.int32)
    c = F.l41_loss(et, dev(y), st, dev(I, np.int32), normalize, neg_idx=idx, ns_rate=rate)
    c_ref = ol41.l41_cost(emb, y, spk, I, normalize, idx_ref, rate)
    assert abs(float(c) - c_ref) < TOL * max(1.0, abs(c_ref))
    assert abs(c_ref - ol41.l41_cost(emb, y, spk, I, normalize)) > 1e-3               # the term is really there
    c.backward()
    de, ds = ol41.l41_cost_bwd(emb, y, spk, I, normalize, idx_ref, rate)
    assert rel(host(et.grad), de) < 5 * TOL and rel(host(st.grad), ds) < 5 * TOL


def test_l41_random_negatives_exclude_the_mixture(F):
    """L41.py:123-134: K distinct speakers per utterance, none of them in I[b]; a fresh draw per call."""
    torch.manual_seed(5)
    B, S, NS, K = 16, 3, 40, 9
    I = torch.stack([torch.randperm(NS)[:S] for _ in range(B)]).to(torch.int32).cuda()
    a, b2 = F.l41_random_negatives(I, NS, K), F.l41_random_negatives(I, NS, K)
    assert a.shape == (B, 1, K) and a.dtype == torch.int32
    an, In = a.cpu().numpy()[:, 0], I.cpu().numpy()
    for r in range(B):
        assert len(set(an[r])) == K and not (set(an[r]) & set(In[r])) and an[r].min() >= 0 and an[r].max() < NS
    assert not torch.equal(a, b2)


@pytest.mark.parametrize('b,L,E,C,tries,with_w,end', [(3, 5000, 40, 2, 2, False, True), (2, 4100, 40, 3, 3, True, False),
                                                       (2, 2500, 8, 2, 1, True, True), (1, 20480, 40, 2, 2, False, True),
                                                       (2, 4100, 40, 2, 3, 'real', False), (2, 3000, 40, 2, 2, 'mixed', True)])
def test_kmeans_hard_bit_exact(F, ops, b, L, E, C, tries, with_w, end):
    """Labels must be IDENTICAL to the float32 oracle (same summation order, no FMA, IEEE sqrt/div).  with_w True = the 0/1
    silence mask the reference builds (Kmeans_2.py:76-82; the kernel folds such a weight into one multiply per point); 'real' /
    'mixed' = arbitrary positive weights on every / every third point (the per-term form of Kmeans_2.py:175-181)."""
    rng = np.random.RandomState(L + C)
    centers = rng.randn(C, E).astype(np.float32) * 2.0
    lab_true = rng.randint(0, C, (b, L))
    X = (centers[lab_true] + rng.randn(b, L, E).astype(np.float32) * 0.7).astype(np.float32)
    w = (rng.rand(b, L) > 0.2).astype(np.float32) if with_w else None
    if with_w == 'real':
        w = rng.uniform(0.05, 1.7, (b, L)).astype(np.float32)
    elif with_w == 'mixed':
        w[:, ::3] = rng.uniform(0.05, 1.7, (b, L))[:, ::3].astype(np.float32)
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    cent_ref, lab_ref, best_ref = okm.kmeans(X, idx, C, tries, 4, beta=None, notsilent=w, assign_at_end=end)
    xn = ops.kmeans_normalize(dev(X))
    assert np.array_equal(host(xn).astype(np.float32), okm.l2_normalize_rows(X))
    cent, lab, best = F.kmeans(dev(X), dev(idx, np.int32), C, tries, 4, None, dev(w) if with_w else None, end)
    torch.cuda.synchronize()
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert np.array_equal(cent.cpu().numpy(), cent_ref)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)


def test_kmeans_soft_forward(F):
    rng = np.random.RandomState(9)
    b, L, E, C, tries = 2, 3000, 40, 2, 2
    centers = rng.randn(C, E) * 2.0
    X = centers[rng.randint(0, C, (b, L))] + rng.randn(b, L, E) * 0.7
    w = (rng.rand(b, L) > 0.2).astype(np.float64)
    idx = np.stack([rng.choice(L, C, replace=False) for _ in range(b * tries)]).astype(np.int32)
    cent_ref, lab_ref, best_ref = okm.kmeans(X, idx, C, tries, 5, beta=10.0, notsilent=w, assign_at_end=True)
    cent, lab, best = F.kmeans(dev(X), dev(idx, np.int32), C, tries, 5, 10.0, dev(w), True)
    assert np.array_equal(best.cpu().numpy(), best_ref)
    assert rel(host(cent), cent_ref) < 1e-4 and np.abs(host(lab) - lab_ref).max() < 1e-3


@pytest.mark.parametrize('Bt,L,W,N,P,hop,S', [(4, 512, 64, 16, 128, 128, 2), (6, 1024, 128, 40, 256, 128, 2), (2, 300, 32, 5, 40, 24, 1),
                                               (3, 2048, 1024, 256, 256, 256, 1), (2, 12000, 64, 8, 4200, 4200, 2)])   # last: pooling window wider than the LDS segment of the values' backward
def test_maxpool_front_and_sparse_synthesis(F, Bt, L, W, N, P, hop, S):
    """Path B: fused conv+max-pool (MFMA path when 128-aligned, generic kernel otherwise), gather filter gradient,
    sparse synthesis and its gradients -- vs the oracle (which itself is checked against dense unpool + conv_transpose)."""
    rng = np.random.RandomState(L + W)
    x, f = rng.randn(Bt, L), rng.randn(W, N) / np.sqrt(W)
    ft = dev(f).requires_grad_()
    y, am = F.front_maxpool(dev(x), ft, P, hop)
    y_ref, am_ref = ofront.front_maxpool(x.astype(np.float32).astype(np.float64), f.astype(np.float32).astype(np.float64), P, hop)
    T = (L - P) // hop + 1
    assert y.shape == (Bt, T, N) and rel(host(y), y_ref) < TOL
    amh = am.cpu().numpy()
    # arg-max may legitimately differ where two conv outputs tie to fp32 round-off: require the VALUE at our index to be the max
    same = (amh == am_ref)
    assert same.mean() > 0.999
    dy = rng.randn(Bt, T, N)
    y.backward(dev(dy))
    df_ref = ofront.front_maxpool_bwd_filter(x, dy, amh, W)
    assert rel(host(ft.grad), df_ref) < 5 * TOL
    # sparse synthesis with the "mixture" argmax tiled over S speakers
    B = Bt
    vals, f2, dout = rng.randn(B * S, T, N), rng.randn(W, N) / np.sqrt(W), rng.randn(B * S, L)
    am_t = np.repeat(amh, S, axis=0)
    vt, f2t = dev(vals).requires_grad_(), dev(f2).requires_grad_()
    out = F.synth_unpool(vt, am, f2t, L, S, P, hop)
    assert rel(host(out), ofront.synth_unpool(vals, am_t, f2, L)) < TOL
    out.backward(dev(dout))
    dv_ref, df2_ref = ofront.synth_unpool_bwd(vals, am_t, f2, dout)
    assert rel(host(vt.grad), dv_ref) < 5 * TOL and rel(host(f2t.grad), df2_ref) < 5 * TOL


@pytest.mark.parametrize('B,S,TF,nl', [(2, 2, 1000, 'softmax'), (3, 3, 777, 'tanh'), (1, 2, 4096, 'None'), (2, 4, 513, 'softmax')])
def test_enhance_output_stage(ops, B, S, TF, nl):
    """network.py:640-660 output stage and its backward vs torch autograd in float64."""
    rng = np.random.RandomState(B * 100 + S)
    u = rng.randn(B * S, TF)
    X = np.abs(rng.randn(B, TF)) + 0.1
    gc, gs = rng.randn(B, TF, S), rng.randn(B, S, TF)
    ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    y = ut.reshape(B, S, TF).transpose(1, 2)
    if nl == 'softmax':
        y = torch.softmax(y, dim=2)
    elif nl == 'tanh':
        y = torch.tanh(y)
    ci_ref = y * torch.tensor(X).reshape(B, TF, 1)
    sp_ref = ci_ref.transpose(1, 2)
    ((ci_ref * torch.tensor(gc)).sum() + (sp_ref * torch.tensor(gs)).sum()).backward()
    ud, Xd = dev(u).view(B * S, 1, TF), dev(X).view(B, 1, TF)
    ci, sp = ops.enhance_output_fwd(ud, Xd, S, nl)
    assert rel(host(ci), ci_ref.detach().numpy()) < 1e-5 and rel(host(sp), sp_ref.detach().numpy()) < 1e-5
    du = ops.enhance_output_bwd(ud, Xd, S, nl, dev(gc), dev(gs))
    assert rel(host(du).reshape(B * S, TF), ut.grad.numpy()) < 1e-4
    du1 = ops.enhance_output_bwd(ud, Xd, S, nl, dev(gc), None)
    du2 = ops.enhance_output_bwd(ud, Xd, S, nl, None, dev(gs))
    assert rel(host(du1) + host(du2), host(du)) < 1e-5


@pytest.mark.parametrize('normalize', [True, False])
def test_l41_speaker_vectors(ops, normalize):
    """L41.py:60-68 normalise + gather and its scatter backward (repeated speakers accumulate) vs torch autograd."""
    rng = np.random.RandomState(8)
    nspk, E, B, S = 17, 40, 6, 3
    table = rng.randn(nspk, E)
    I = rng.randint(0, nspk, (B, S))
    I[0, 0] = I[1, 1] = I[2, 2] = 5                                  # repeats across the batch
    g = rng.randn(B, S, E)
    tt = torch.tensor(table, dtype=torch.float64, requires_grad=True)
    sv = tt * torch.rsqrt(torch.clamp((tt * tt).sum(dim=1, keepdim=True), min=1e-12)) if normalize else tt
    vs_ref = sv[torch.tensor(I).long()]
    (vs_ref * torch.tensor(g)).sum().backward()
    Id = torch.tensor(I, dtype=torch.int32, device='cuda')
    vs = ops.l41_speaker_fwd(dev(table), Id, normalize)
    assert rel(host(vs), vs_ref.detach().numpy()) < 1e-6
    dt = ops.l41_speaker_bwd(dev(table), Id, dev(g), normalize)
    assert rel(host(dt), tt.grad.numpy()) < 1e-5


@pytest.mark.parametrize('pre,norm,silent_db', [('sqrt', 'meanstd', 0), ('log', '01', 0), (None, '01', 0), ('sqrt', None, 30.0),
                                                (None, 'meanstd', 12.0)])
def test_input_conditioning_matches_oracle(ops, pre, norm, silent_db):
    """Separator.init_separator STFT branch (network.py:427-443) chained exactly as the host mirror chains it."""
    from oracle import separate as osep
    rng = np.random.RandomState(17)
    X = np.abs(rng.randn(3, 37, 65)) + 1e-3
    ref = osep.stft_input_pipeline(X.copy(), pre, norm, silent_db)
    z = dev(X)
    if pre:
        z = ops.row_transform(z, pre=pre)
    if norm:
        z = ops.row_transform(z, norm=norm)
    if silent_db > 0:
        z = ops.row_transform(z, norm='silent', thr=silent_db / 20.)
    assert rel(host(z), ref) < 2e-5
    assert rel(host(ops.row_transform(dev(-X), pre='abs')), X) < 1e-7


@pytest.mark.parametrize('mode,sil', [('linear', None), ('sqrt', 2.0), ('square', None), (None, 1.0)])
def test_mask_weighting_and_silence_weights(ops, mode, sil):
    """network.py:381-396 mask weightings and Kmeans_2.py:76-80 'notsilent' weights."""
    from oracle import separate as osep
    rng = np.random.RandomState(23)
    B, T, Fq, S = 2, 31, 40, 3
    X = rng.randn(B, T, Fq)
    y = np.eye(S)[rng.randint(0, S, (B, T, Fq))]
    ref = y.copy()
    if mode:
        ref = osep.function_mask(ref, X, mode)
    if sil is not None:
        ref = osep.silence_loss_mask(ref, X, sil)
    got = ops.weight_masks(dev(X).view(B, T * Fq), dev(y).view(B, T * Fq, S), mode, sil)
    assert rel(host(got).reshape(ref.shape), ref) < 1e-5
    w_ref = osep.kmeans_silence_weights(np.abs(X), 1.5)
    w = ops.silence_weights(dev(np.abs(X)).view(B, T * Fq), 1.5)
    # a bin sitting exactly on the threshold may flip in fp32: allow a handful
    assert (host(w) != w_ref).sum() <= 2


@pytest.mark.parametrize('Bt,T,N,scale', [(9, 64, 16, 0.02), (192, 80, 256, 0.004), (5, 7, 3, 0.3)])
def test_sparsity_kl_and_regulariser_kernels(F, ops, Bt, T, N, scale):
    """p_hat = sum_b |y|, sum kl_div(p, p_hat) with both clip_by_value gates (models/adapt.py:130-132, utils/ops.py:46-54), the
    non-negativity energy (adapt.py:314-316) and sum-of-squares (tf.nn.l2_loss), forward and backward, against the oracle --
    including p_hat values exactly on and on either side of the clip bounds, and the benchmark's [192, 80 * 256] shape."""
    rng = np.random.RandomState(Bt * T)
    y = rng.randn(Bt, T, N) * scale
    y[:, 0, 0] = 0.0                                          # p_hat = 0 < 1e-10: clipped below, no gradient; sign(0) = 0
    y[:, 0, 1] = 0.0
    y[0, 0, 1] = 1.0                                          # p_hat == 1 exactly: inside both gates' closed ends
    y[:, 0, 2] = 2.0 / Bt * np.sign(rng.randn(Bt))            # p_hat = 2 > 1: clipped above
    p = 0.01
    y32 = y.astype(np.float32).astype(np.float64)
    p_hat_ref, kl_ref = ofront.sparsity_terms(y32, p)
    yt = dev(y).requires_grad_()
    ph = ops.abs_colsum(yt.detach().reshape(Bt, -1))
    assert rel(host(ph), p_hat_ref) < 1e-6
    kl = F.sparse_kl(yt.reshape(Bt, -1), p)
    assert abs(float(kl) - kl_ref) < 2e-5 * abs(kl_ref)
    (3.0 * kl).backward()
    g_ref = 3.0 * ofront.sparsity_terms_bwd(y32, p, np.asarray(host(ph), np.float64))
    assert rel(host(yt.grad), g_ref) < 2e-5
    # non-negativity and l2 terms
    yt2 = dev(y).requires_grad_()
    ne = F.negative_energy(yt2)
    ne_ref = (np.minimum(y32, 0.0) ** 2).reshape(Bt, -1).sum(1).mean()
    assert abs(float(ne) - ne_ref) < 2e-5 * ne_ref
    (0.5 * ne).backward()
    assert rel(host(yt2.grad), 0.5 * 2.0 * np.minimum(y32, 0.0) / Bt) < 1e-6
    yt3 = dev(y).requires_grad_()
    ss = F.sumsq(yt3)
    assert abs(float(ss) - (y32 ** 2).sum()) < 2e-5 * (y32 ** 2).sum()
    (0.25 * ss).sum().backward()
    assert rel(host(yt3.grad), 0.5 * y32) < 1e-6
