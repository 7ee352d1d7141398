"""Micro-benchmark: the step's products alone, native f32 MFMA kernel vs bf16x6 (csrc/gemm.hip), at the benchmark's shapes.

    python tools/gemm_x6_bench.py [--reps 20]

Prints one line per product and arithmetic: microseconds per launch, f32-equivalent TFLOP/s (2*M*N*K / t), and for bf16x6 the
fraction of the bf16 MFMA issue peak that the six products occupy (6 * 2*M*N*K / t / 2.5 PFLOP/s), plus the error of both against
float64 on a 256-row sample of the output.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops  # noqa: E402
from ams_hip._lib import load  # noqa: E402

# (label, M, N, K, transA, transB): front_DPCL step at B = 64, T = 80
SHAPES = [
    ('dense fwd      x.W', 5120, 10240, 600, 0, 0),
    ('dense dX    dU.W^T', 5120, 600, 10240, 0, 1),
    ('dense dW    x^T.dU', 600, 10240, 5120, 1, 0),
    ('proj l2,3      x.Wx', 5120, 2400, 600, 0, 0),
    ('proj l1        x.Wx', 5120, 2400, 256, 0, 0),
    ('lstm dX   dZ.Wx^T', 5120, 600, 2400, 0, 1),
    ('lstm dWx  x^T.dZ', 600, 2400, 5120, 1, 0),
    ('lstm dU   h^T.dZ', 300, 1200, 5120, 1, 0),
    ('square 4096', 4096, 4096, 4096, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--warm', type=int, default=100, help='untimed launches first (the clock ramps up from idle over milliseconds)')
    ap.add_argument('--only', default='', help='comma-separated substrings of the product labels to run')
    ap.add_argument('--modes', default='0,1,2', help='arithmetics to time: 0 native f32, 1 bf16x6, 2 fp16x3 (bounds measured once, outside the timing)')
    ap.add_argument('--pad', type=int, default=0, help='dynamic-LDS pad = residency cap of the launches (the step uses 50000 beside a ring)')
    a = ap.parse_args()
    lib = load()
    ops.LDS_PAD[0] = a.pad
    rng = np.random.RandomState(0)
    only = [w for w in a.only.split(',') if w]
    for label, M, N, K, tA, tB in SHAPES:
        if only and not any(w in label for w in only):
            continue
        A = torch.from_numpy(rng.randn(*((K, M) if tA else (M, K))).astype(np.float32)).cuda()
        B = torch.from_numpy(rng.randn(*((N, K) if tB else (K, N))).astype(np.float32)).cuda()
        out = torch.empty(M, N, device='cuda')
        rows = np.linspace(0, M - 1, min(M, 256)).astype(int)
        A64 = (A.T if tA else A).cpu().numpy().astype(np.float64)[rows]
        B64 = (B.T if tB else B).cpu().numpy().astype(np.float64)
        ref = A64 @ B64
        scale = (np.linalg.norm(A64, axis=1)[:, None] * np.linalg.norm(B64, axis=0)[None, :]).max()
        bounds = (ops.absmax(A), ops.absmax(B))
        for mode in [int(m) for m in a.modes.split(',')]:
            lib.ams_gemm_set_arith(1 if mode else 0)
            am = bounds if mode == 2 else None
            for _ in range(a.warm):
                ops.gemm(A, B, transA=bool(tA), transB=bool(tB), out=out, amax=am)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops.gemm(A, B, transA=bool(tA), transB=bool(tB), out=out, amax=am)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            tf = 2.0 * M * N * K / us * 1e-6
            err = np.abs(out.cpu().numpy().astype(np.float64)[rows] - ref).max() / scale
            extra = '  16-bit MFMA issue frac %.3f' % ((6 if mode == 1 else 3) * tf / 2500.0) if mode else ''
            print('%-20s M=%5d N=%5d K=%5d  %s  %8.1f us  %6.1f TFLOP/s(f32-eq)  err %.2e%s'
                  % (label, M, N, K, ('f32   ', 'bf16x6', 'fp16x3')[mode], us, tf, err, extra), flush=True)
    lib.ams_gemm_set_arith(1)


if __name__ == '__main__':
    main()
