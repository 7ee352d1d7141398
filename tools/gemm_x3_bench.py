"""Pre-split (x3) products: correctness against float64 and timing at the training step's shapes, next to the in-loop-split bf16x6
kernel and the native f32 MFMA kernel (csrc/gemm.hip).    python tools/gemm_x3_bench.py [--reps 50] [--check 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from ams_hip import ops  # noqa: E402
from ams_hip._lib import load  # noqa: E402

# name, M, N, K, tA, tB   (tA: A given as [K, M]; tB: B given as [N, K])
SHAPES = [('dense fwd', 5120, 10240, 600, 0, 0), ('dense dX', 5120, 600, 10240, 0, 1), ('dense dW', 600, 10240, 5120, 1, 0),
          ('proj L1', 5120, 2400, 600, 0, 0), ('proj L0', 5120, 2400, 256, 0, 0), ('lstm dX', 5120, 600, 2400, 0, 1),
          ('lstm dWx', 600, 2400, 5120, 1, 0), ('square', 4096, 4096, 4096, 0, 0), ('ragged', 516, 772, 292, 0, 0),
          ('ragged T', 516, 772, 292, 1, 0), ('ragged NT', 516, 772, 292, 0, 1)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--check', type=int, default=1)
    ap.add_argument('--only', default='')
    ap.add_argument('--capped', type=int, default=0)
    a = ap.parse_args()
    lib = load()
    rng = np.random.RandomState(0)
    for name, M, N, K, tA, tB in SHAPES:
        if a.only and a.only not in name:
            continue
        A = (rng.randn(*((K, M) if tA else (M, K))) * np.exp(rng.uniform(-2, 2, size=((K, M) if tA else (M, K))))).astype(np.float32)
        B = (rng.randn(*((N, K) if tB else (K, N))) * np.exp(rng.uniform(-2, 2, size=((N, K) if tB else (K, N))))).astype(np.float32)
        Ad, Bd = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        Ai, Bi = ops.x3_split(Ad), ops.x3_split(Bd)
        roleA, roleB = (1 if tA else 0), (0 if tB else 1)
        lib.ams_x3_set_capped(a.capped)
        out = torch.empty((M, N), device='cuda')

        def x3():
            ops.gemm_x3(Ai, roleA, Bi, roleB, M, N, K, out=out)
        x3()
        torch.cuda.synchronize()
        line = '%-10s %5d x %5d x %5d  ' % (name, M, N, K)
        if a.check:
            A64, B64 = (A.T if tA else A).astype(np.float64), (B.T if tB else B).astype(np.float64)
            ref = A64 @ B64
            scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
            c = out.cpu().numpy().astype(np.float64)
            lib.ams_gemm_set_arith(1)
            c6 = ops.gemm(Ad, Bd, transA=bool(tA), transB=bool(tB)).cpu().numpy().astype(np.float64)
            lib.ams_gemm_set_arith(0)
            c0 = ops.gemm(Ad, Bd, transA=bool(tA), transB=bool(tB)).cpu().numpy().astype(np.float64)
            d = (c - ref) / scale
            line += 'err x3 %.2e (mean %+.1e) x6 %.2e f32 %.2e  ' % (np.abs(d).max(), d.mean(), np.abs(c6 - ref).max() / scale, np.abs(c0 - ref).max() / scale)
        t3 = timeit(x3, a.reps)
        ts = timeit(lambda: (ops.x3_split(Ad, out=Ai), ops.x3_split(Bd, out=Bi)), a.reps)
        lib.ams_gemm_set_arith(1)
        t6 = timeit(lambda: ops.gemm(Ad, Bd, transA=bool(tA), transB=bool(tB), out=out), a.reps)
        lib.ams_gemm_set_arith(0)
        t0 = timeit(lambda: ops.gemm(Ad, Bd, transA=bool(tA), transB=bool(tB), out=out), a.reps)
        lib.ams_gemm_set_arith(1)
        fl = 2.0 * M * N * K
        line += 'x3 %7.1f us %6.1f TF | split both %6.1f us | x6 %7.1f us %6.1f TF | f32 %7.1f us %6.1f TF' % (
            t3, fl / t3 / 1e6, ts, t6, fl / t6 / 1e6, t0, fl / t0 / 1e6)
        print(line, flush=True)
        lib.ams_x3_set_capped(0)


if __name__ == '__main__':
    main()
