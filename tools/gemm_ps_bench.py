"""Micro-benchmark: forward products from pre-split fp16 operand images (csrc/gemm_ps.hip) beside the in-product fp16x3 form
(csrc/gemm.hip), at the benchmark's forward shapes; pack launches timed separately.

    python tools/gemm_ps_bench.py [--reps 200]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops  # noqa: E402

SHAPES = [('dense fwd  x.W', 5120, 10240, 600), ('proj l2,3  x.Wx', 5120, 2400, 600), ('proj l1    x.Wx', 5120, 2400, 256),
          ('cfg5 dense', 10240, 20480, 600), ('square 4096', 4096, 4096, 4096)]


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--warm', type=int, default=100)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    rng = np.random.RandomState(0)
    for label, M, N, K in SHAPES:
        if a.only and a.only not in label:
            continue
        A = torch.from_numpy(rng.rand(M, K).astype(np.float32) * 2 - 1).cuda()
        W = torch.from_numpy((rng.randn(K, N) * 0.05).astype(np.float32)).cuda()
        bias = torch.zeros(N, device='cuda')
        out = torch.empty(M, N, device='cuda')
        am = (ops.absmax(A), ops.absmax(W))
        ai, bi = ops.ps_pack_rows(A, am[0]), ops.ps_pack_cols(W, am[1])
        t_ps = timed(lambda: ops.gemm_ps(ai, bi, K, am, bias=bias, out=out), a.reps, a.warm)
        t_g = timed(lambda: ops.gemm(A, W, bias=bias, out=out, amax=am), a.reps, a.warm)
        t_pr = timed(lambda: ops.ps_pack_rows(A, am[0], img=ai), a.reps, a.warm)
        t_pc = timed(lambda: ops.ps_pack_cols(W, am[1], img=bi), a.reps, a.warm)
        fl = 2.0 * M * N * K
        print('%-18s M=%5d N=%5d K=%4d  pre-split %7.1f us (%6.1f TF f32-eq, %.3f of the 16-bit pipe)   in-product %7.1f us (%6.1f TF)   '
              'pack rows %5.1f us  pack cols %5.1f us' % (label, M, N, K, t_ps, fl / t_ps * 1e-6, 3 * fl / t_ps * 1e-6 / 2500.0, t_g,
                                                          fl / t_g * 1e-6, t_pr, t_pc), flush=True)


if __name__ == '__main__':
    main()
