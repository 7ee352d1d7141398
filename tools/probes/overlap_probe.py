"""Does the HBM-bound DPCL loss chain of one half of the batch hide behind the MFMA-bound dense products of the other half?
Times, at the B = 64 step's shapes: dense fwd (rows of 26 utterances), dense dX (rows of 38), the fused l2norm + DPCL forward and
backward (38 / 26 utterances), each alone and the pairs side by side on two streams.   python tools/probes/overlap_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops  # noqa: E402


def timeit(fn, reps=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    T, F, E, S, D = 80, 256, 40, 2, 600
    BA, BB = 38, 26
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand(64 * T, D, device='cuda', generator=g) * 2 - 1
    W = (torch.rand(D, F * E, device='cuda', generator=g) - 0.5) * 0.1
    b = torch.zeros(F * E, device='cuda')
    U = torch.empty(64 * T, F * E, device='cuda')
    Y = torch.zeros(64, T * F, S, device='cuda')
    Y[..., 0] = (torch.rand(64, T * F, device='cuda', generator=g) > 0.5).float()
    Y[..., 1] = 1 - Y[..., 0]
    ax, aw = ops.absmax(x), ops.absmax(W)
    side = torch.cuda.Stream()
    up = torch.ones(1, device='cuda')

    def fwd(lo, hi):
        ops.gemm(x[lo * T:hi * T], W, bias=b, out=U[lo * T:hi * T], amax=(ax, aw))

    fwd(0, 64)
    state = {}

    def loss(lo, hi):
        u = U[lo * T:hi * T].view(hi - lo, T * F, E)
        out, inv, _, ws = ops.dpcl_loss_fwd_u(u, Y[lo:hi].contiguous())
        state['dU%d' % lo] = ops.dpcl_loss_bwd_u(u, Y[lo:hi].contiguous(), inv, ws, upstream=up)

    loss(0, BA)
    loss(BA, 64)
    dU = torch.cat([state['dU0'].view(BA * T, F * E), state['dU%d' % BA].view(BB * T, F * E)])
    adu = ops.absmax(dU)

    def dx(lo, hi):
        ops.gemm(dU[lo * T:hi * T], W, transB=True, amax=(adu, aw))

    def pair(fa, fb):
        def run():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fb()
            fa()
            torch.cuda.current_stream().wait_stream(side)
        return run

    rows = [
        ('dense fwd 64 utt', lambda: fwd(0, 64)), ('dense fwd A=38', lambda: fwd(0, BA)), ('dense fwd B=26', lambda: fwd(BA, 64)),
        ('loss fwd+bwd 64', lambda: loss(0, 64)), ('loss A=38', lambda: loss(0, BA)), ('loss B=26', lambda: loss(BA, 64)),
        ('dense dX 64', lambda: dx(0, 64)), ('dense dX A=38', lambda: dx(0, BA)), ('dense dX B=26', lambda: dx(BA, 64)),
        ('fwd B || loss A', pair(lambda: fwd(BA, 64), lambda: loss(0, BA))),
        ('dX A || loss B', pair(lambda: dx(0, BA), lambda: loss(BA, 64))),
    ]
    for name, fn in rows:
        print('%-22s %8.1f us' % (name, timeit(fn)), flush=True)


if __name__ == '__main__':
    main()
