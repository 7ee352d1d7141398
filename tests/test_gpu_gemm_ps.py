"""GPU: products from pre-split fp16 operand images (csrc/gemm_ps.hip, include/ams.h "PS32") against float64 and against the in-product
fp16x3 form of ams_gemm_f32 -- same terms, same three products, f32 accumulation: the two differ only in summation order."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def unpack(img, K):
    """PS32 image -> (hi, lo) float64 [R, K] (the oracle of the layout: row pitch ceil(K / 32) * 128, k-tile = 32 hi | 32 lo)."""
    raw = img.cpu().numpy()
    R, pitch = raw.shape
    h = raw.view(np.float16).reshape(R, pitch // 128, 2, 32).astype(np.float64)
    return h[:, :, 0, :].reshape(R, -1)[:, :K], h[:, :, 1, :].reshape(R, -1)[:, :K], h


@pytest.mark.parametrize('R,K,ld', [(7, 600, 600), (130, 256, 300), (33, 45, 45), (64, 32, 32), (5, 1, 4)])
def test_pack_rows_is_the_exact_two_term_cut(R, K, ld):
    from ams_hip import ops
    rng = np.random.RandomState(R + K)
    x = (rng.randn(R, ld) * np.exp(rng.randn(R, ld) * 2)).astype(np.float32)
    xd = dev(x)[:, :K]
    bound = ops.absmax(xd.contiguous())
    img = ops.ps_pack_rows(xd, bound)
    hi, lo, full = unpack(img, K)
    b = float(bound)
    s = 2.0 ** (13 - np.floor(np.log2(b)))
    xs = x[:, :K].astype(np.float64) * s
    assert np.array_equal(hi, xs.astype(np.float16).astype(np.float64))               # hi = fp16(x s), round to nearest even
    assert np.array_equal(lo, (xs - hi).astype(np.float32).astype(np.float16).astype(np.float64))
    assert np.abs(hi + lo - xs).max() <= 2.0 ** -22 * np.abs(xs).max() + 2.0 ** -25   # 22 bits for entries within 2^17 of the bound
    Kp = (K + 31) // 32 * 32
    assert img.shape[1] == Kp * 4
    tail = full.reshape(R, -1, 2, 32)
    pad = np.concatenate([tail[:, :, 0, :].reshape(R, -1)[:, K:], tail[:, :, 1, :].reshape(R, -1)[:, K:]], axis=1)
    assert not pad.any()                                                                # k >= K: zeros in both planes


@pytest.mark.parametrize('K,N,ld', [(600, 2400, 2400), (256, 40, 48), (45, 130, 130), (33, 7, 8)])
def test_pack_cols_is_pack_rows_of_the_transpose(K, N, ld):
    from ams_hip import ops
    rng = np.random.RandomState(K + N)
    w = rng.randn(K, ld).astype(np.float32)
    wd = dev(w)[:, :N]
    bound = ops.absmax(wd.contiguous())
    a = ops.ps_pack_cols(wd, bound)
    b = ops.ps_pack_rows(wd.t().contiguous(), bound)
    assert torch.equal(a, b)


@pytest.mark.parametrize('M,N,K,bias', [(128, 256, 32, False), (128, 256, 64, True), (128, 256, 600, True), (256, 512, 96, False),
                                        (100, 40, 45, True), (5120, 2400, 600, True), (1000, 10240, 600, True), (5120, 2400, 256, False),
                                        (130, 260, 33, True), (640, 1200, 300, False)])
def test_product_matches_float64_and_the_in_product_cut(M, N, K, bias):
    from ams_hip import ops
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K).astype(np.float32)
    W = (rng.randn(K, N) * 0.3).astype(np.float32)
    b = rng.randn(N).astype(np.float32) if bias else None
    Ad, Wd, bd = dev(A), dev(W), (dev(b) if bias else None)
    am = (ops.absmax(Ad), ops.absmax(Wd))
    ai, bi = ops.ps_pack_rows(Ad, am[0]), ops.ps_pack_cols(Wd, am[1])
    out = torch.full((M, N), float('nan'), device='cuda')
    ops.gemm_ps(ai, bi, K, am, bias=bd, out=out)
    ref_gemm = ops.gemm(Ad, Wd, bias=bd, amax=am)
    torch.cuda.synchronize()
    rows = np.unique(np.concatenate([np.arange(min(M, 200)), np.linspace(0, M - 1, min(M, 300)).astype(int)]))
    ref = A[rows].astype(np.float64) @ W.astype(np.float64) + (b.astype(np.float64) if bias else 0.0)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    scale = np.linalg.norm(A[rows].astype(np.float64), axis=1)[:, None] * np.linalg.norm(W.astype(np.float64), axis=0)[None, :]
    err = np.abs(got[rows] - ref) / (scale + 1e-30)
    err_g = np.abs(ref_gemm.cpu().numpy()[rows] - ref) / (scale + 1e-30)
    assert err.max() < 2e-7 and err.max() <= 2.0 * err_g.max() + 1e-8, (err.max(), err_g.max())
    rel = np.linalg.norm(got[rows] - ref) / np.linalg.norm(ref)
    assert rel < 2e-6, rel


def test_rows_and_columns_past_the_edge_are_zeros_not_neighbours():
    """M and N that end inside a tile: the out-of-range rows of the LDS-DMA come back as zeros (buffer range check) and nothing is
    stored past the edge: a guard band around C stays untouched."""
    from ams_hip import ops
    M, N, K = 129, 260, 70
    rng = np.random.RandomState(9)
    A, W = rng.randn(M, K).astype(np.float32), rng.randn(K, N).astype(np.float32)
    Ad, Wd = dev(A), dev(W)
    am = (ops.absmax(Ad), ops.absmax(Wd))
    big = torch.full((M + 2, N + 8), 7.0, device='cuda')
    out = big[1:M + 1, 4:N + 4]
    ops.gemm_ps(ops.ps_pack_rows(Ad, am[0]), ops.ps_pack_cols(Wd, am[1]), K, am, out=out, ldc=big.stride(0))
    torch.cuda.synchronize()
    g = big.cpu().numpy()
    ref = A.astype(np.float64) @ W.astype(np.float64)
    assert np.abs(g[1:M + 1, 4:N + 4] - ref).max() < 1e-4 * np.abs(ref).max()
    g[1:M + 1, 4:N + 4] = 7.0
    assert (g == 7.0).all()


def test_launches_back_to_back_are_deterministic():
    from ams_hip import ops
    M, N, K = 5120, 2400, 600
    rng = np.random.RandomState(4)
    Ad, Wd = dev(rng.randn(M, K)), dev(rng.randn(K, N) * 0.1)
    am = (ops.absmax(Ad), ops.absmax(Wd))
    ai, bi = ops.ps_pack_rows(Ad, am[0]), ops.ps_pack_cols(Wd, am[1])
    first = ops.gemm_ps(ai, bi, K, am).clone()
    for _ in range(20):
        assert torch.equal(ops.gemm_ps(ai, bi, K, am), first)


def test_the_dma_pipeline_is_race_free_under_memory_pressure():
    """The main loop's correctness rests on hand-counted waits (`vmcnt(6)`: this wave's pieces of tile t have landed, tile t + 1 stays in
    flight) and one barrier per k-tile; a miscount shows up as RARE wrong tiles whenever a DMA happens to land late.  So: many launches of
    several shapes (whole rounds, a tail round, one k-tile, K with a padded tail) while another stream thrashes HBM and the L2 -- every
    output must equal the first one bit for bit, and the first one the float64 product."""
    from ams_hip import ops
    rng = np.random.RandomState(12)
    shapes = [(5120, 2400, 600), (1024, 1024, 2048), (640, 512, 32), (384, 768, 100), (128, 256, 3000)]
    cases = []
    for M, N, K in shapes:
        A, W = rng.randn(M, K).astype(np.float32), (rng.randn(K, N) * 0.2).astype(np.float32)
        Ad, Wd = dev(A), dev(W)
        am = (ops.absmax(Ad), ops.absmax(Wd))
        ai, bi = ops.ps_pack_rows(Ad, am[0]), ops.ps_pack_cols(Wd, am[1])
        first = ops.gemm_ps(ai, bi, K, am).clone()
        torch.cuda.synchronize()
        rows = np.linspace(0, M - 1, 64).astype(int)
        ref = A[rows].astype(np.float64) @ W.astype(np.float64)
        assert np.abs(first.cpu().numpy()[rows] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
        cases.append((ai, bi, K, am, first, torch.empty_like(first)))
    noise_stream = torch.cuda.Stream()
    big = torch.empty(64 * 1024 * 1024, device='cuda')                 # 256 MB: does not fit the Infinity Cache either
    bad = 0
    for rep in range(40):
        with torch.cuda.stream(noise_stream):
            big.add_(1.0)
            big.mul_(0.5)
        for ai, bi, K, am, first, out in cases:
            ops.gemm_ps(ai, bi, K, am, out=out)
            bad += int(not torch.equal(out, first))
    torch.cuda.synchronize()
    assert bad == 0, '%d of %d launches differed from the first launch of their shape' % (bad, 40 * len(cases))


_MAXPOOL_SCRIPT = r'''
import os, sys, numpy as np, torch
sys.path[:0] = [os.environ['AMS_ROOT'], os.path.join(os.environ['AMS_ROOT'], 'adaptive-multispeaker-separation_amd')]
from ams_hip import pooling
out = {}
for Bt, L, W, N, P, hop in ((3, 1024, 64, 16, 128, 128), (2, 2048, 1024, 256, 256, 256), (5, 1280, 100, 40, 128, 128), (1, 4096, 512, 300, 256, 128),
                            (1, 128, 16, 4, 128, 128), (2, 256, 33, 8, 128, 128), (7, 384, 96, 260, 128, 128)):
    rng = np.random.RandomState(Bt * 1000 + W)
    x = torch.from_numpy(rng.randn(Bt, L).astype(np.float32)).cuda()
    x[0, : L // 3] = 0.0                                        # digital silence at the head of a signal (the zero padding continues it)
    f = torch.from_numpy((rng.randn(W, N) / np.sqrt(W)).astype(np.float32)).cuda()
    y, am = pooling.front_maxpool_fwd(x, f, P, hop)
    torch.cuda.synchronize()
    out['y_%d_%d' % (L, W)] = y.cpu().numpy()
    out['a_%d_%d' % (L, W)] = am.cpu().numpy()
    out['w_%d_%d' % (L, W)] = np.array([pooling.load().ams_front_maxpool_workspace_bytes_w(Bt, L, N, W)])
np.savez(sys.argv[1], **out)
'''


def test_path_b_from_shifted_copy_images_equals_the_in_product_cut(tmp_path):
    """The stride-1 conv + max-pool partial of path B (models/adapt.py:115-117) on pre-split images (csrc/gemm_ps.hip: the signals as
    eight shifted copies, LDS-DMA main loop) against the form that cuts its operands inside the product (AMS_MAXPOOL_PS=0; the switch is
    read once per process, hence two processes): the same three fp16 products of the same terms, every output element accumulated in the
    same order (measured: identical bits; asserted: pooled values to 2e-6 of the largest, arg-max positions equal but for ties at that
    level); W = 16 ... 1024 (one k-tile, two, W not a multiple of 32), N = 4 ... 300 (below / at / above one 256-column tile), one tile per
    signal, silence at the head of a signal."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'maxpool_forms.py'
    script.write_text(_MAXPOOL_SCRIPT)
    res = {}
    for form in ('1', '0'):
        env = dict(os.environ, AMS_MAXPOOL_PS=form, AMS_ROOT=root)
        out = tmp_path / ('form%s.npz' % form)
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=env, timeout=600)
        res[form] = np.load(out)
    assert sorted(res['1'].files) == sorted(res['0'].files) and len(res['1'].files) == 21
    for k in [k for k in res['1'].files if k.startswith('y_')]:
        y1, y0, a1, a0 = res['1'][k], res['0'][k], res['1']['a' + k[1:]], res['0']['a' + k[1:]]
        # (the pre-split form asks for room for its images: the process that was to use it did)
        assert int(res['1']['w' + k[1:]][0]) > int(res['0']['w' + k[1:]][0])
        top = np.abs(y0).max()
        assert np.abs(y1 - y0).max() <= 2e-6 * top, (k, np.abs(y1 - y0).max() / top)
        same = float((a1 == a0).mean())
        print('path B forms', k, 'max |dy| / max |y| %.2g' % (np.abs(y1 - y0).max() / top), 'arg-max equal %.6f' % same)
        assert same > 0.999, (k, same)


def test_path_b_pipeline_is_race_free_under_memory_pressure():
    """The same screen for the conv variant of the kernel (shifted-copy images, fragments carried in registers from one k-tile to the
    next, piece loads that fill LDS rows 8 tile rows apart): launches of three shapes while another stream thrashes HBM and the L2 --
    pooled values and arg-max of every launch equal the first launch's bit for bit, and the first launch the float64 conv."""
    from ams_hip import pooling
    from oracle import front as ofront
    rng = np.random.RandomState(21)
    cases = []
    for Bt, L, W, N, P, hop in ((8, 4096, 1024, 256, 256, 256), (3, 2048, 96, 40, 128, 128), (16, 1024, 512, 260, 128, 128)):
        x = rng.randn(Bt, L).astype(np.float32)
        f = (rng.randn(W, N) / np.sqrt(W)).astype(np.float32)
        xd, fd = dev(x), dev(f)
        y0, a0 = pooling.front_maxpool_fwd(xd, fd, P, hop)
        torch.cuda.synchronize()
        y_ref, a_ref = ofront.front_maxpool(x[:1].astype(np.float64), f.astype(np.float64), P, hop)
        assert np.abs(y0[:1].cpu().numpy() - y_ref).max() <= 2e-5 * np.abs(y_ref).max()
        assert float((a0[:1].cpu().numpy() == a_ref).mean()) > 0.999
        cases.append((xd, fd, P, hop, y0.clone(), a0.clone()))
    noise_stream = torch.cuda.Stream()
    big = torch.empty(64 * 1024 * 1024, device='cuda')
    bad = 0
    for rep in range(25):
        with torch.cuda.stream(noise_stream):
            big.add_(1.0)
            big.mul_(0.5)
        for xd, fd, P, hop, y0, a0 in cases:
            y, a = pooling.front_maxpool_fwd(xd, fd, P, hop)
            bad += int(not (torch.equal(y, y0) and torch.equal(a, a0)))
    torch.cuda.synchronize()
    assert bad == 0, '%d of %d launches differed from the first launch of their shape' % (bad, 25 * len(cases))
