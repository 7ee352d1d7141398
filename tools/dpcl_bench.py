"""The fused l2-normalise + DPCL loss kernels alone at the B = 64 step's shape (64 x 20480 points x 40): us per forward / backward
call and algorithmic GB/s.   python tools/dpcl_bench.py   (AMS_HIP_LIB selects a variant library)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops  # noqa: E402


def main():
    B, TF, E, S = 64, 20480, 40, 2
    g = torch.Generator(device='cuda').manual_seed(0)
    U = torch.randn(B, TF, E, device='cuda', generator=g)
    Y = torch.zeros(B, TF, S, device='cuda')
    Y[..., 0] = (torch.rand(B, TF, device='cuda', generator=g) > 0.5).float()
    Y[..., 1] = 1 - Y[..., 0]
    up = torch.ones(1, device='cuda')
    out, inv, _, ws = ops.dpcl_loss_fwd_u(U, Y)

    def t(fn, reps=100):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    f = t(lambda: ops.dpcl_loss_fwd_u(U, Y))
    b = t(lambda: ops.dpcl_loss_bwd_u(U, Y, inv, ws, upstream=up))
    print('%s: fwd chain %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)  cost %.6f'
          % (os.environ.get('AMS_HIP_LIB', 'default').split('/')[-1], f, 4.0 * B * TF * (E + S + 1) / f * 1e-3, b, 4.0 * B * TF * (2 * E + S + 1) / b * 1e-3, float(out[0])))


if __name__ == '__main__':
    main()
