"""Cycle anatomy of one ring-recurrence workgroup (csrc/lstm_ring.hip, trace bit of the `safe` argument): shader cycles per phase
and step, thread 0 of (chain 0, member 0), benchmark shape B=64 T=80 H=300.   python tools/ring_anatomy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')):
    sys.path.insert(0, _p)
import numpy as np
import torch
from ams_hip import ops

B, T, D, H = 64, 80, 600, 300
rng = np.random.RandomState(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()      # noqa: E731
lim = np.sqrt(6.0 / (D + 5 * H))
x = dev(rng.randn(B, T, D) * 0.5)
Kf, Kb = dev(rng.uniform(-lim, lim, (D + H, 4 * H)) * 2), dev(rng.uniform(-lim, lim, (D + H, 4 * H)) * 2)
bf, bb, dout = dev(rng.randn(4 * H) * 0.1), dev(rng.randn(4 * H) * 0.1), dev(rng.randn(B, T, 2 * H) * 0.1)
lib = ops.load()
out, G, cst = ops.blstm_fwd(x, Kf, bf, Kb, bb)
G0 = G.clone()
Gz = torch.empty_like(G)
ops.gemm(x.view(B * T, D), ops.blstm_wcat(Kf, Kb, D), bias=torch.cat([bf, bb]), out=Gz, M=B * T, N=8 * H, K=D, lda=D, ldb=8 * H, ldc=8 * H)
ldu = Kf.stride(0)
_AU = ops.absmax(torch.cat([Kf[D:], Kb[D:]]))                                              # bound of the recurrent kernels
AU = _AU.data_ptr() if os.environ.get('AMS_ANATOMY_F16', '1') != '0' else None
p = lambda t: t.data_ptr()                                                                  # noqa: E731
st = torch.cuda.current_stream().cuda_stream
names = {'fwd': ['wait for h', 'MFMA + acc to LDS', 'barrier', 'gate epilogue + granule store', 'G/cst/out stores'],
         'bwd': ['wait for the tagged partial tiles', 'sum (LDS, barrier)', 'gate math + LDS write', 'barrier', 'dZ stores + MFMA + tile stores'] +
                (['(fine) dZ stores issued', '(fine) B operand read / scaled / split', '(fine) MFMA chains + rescale'] if os.environ.get('AMS_ANATOMY_FINE') else [])}   # (-DAMS_RING_DBG_ACK builds add slot 5: own stores acknowledged)
for mode, safe in (('plain stores (same L2)', 2), ('write-through', 3)):
    for kind in ('fwd', 'bwd'):
        n = lib.ams_blstm_ring_sync_bytes(B, H, int(kind == 'bwd'))
        sync = torch.zeros(n // 4 + 1, dtype=torch.float32, device='cuda')
        G.copy_(G0 if kind == 'bwd' else Gz)
        torch.cuda.synchronize()
        if kind == 'fwd':
            ops.check(lib.ams_blstm_ring_fwd(p(G), p(out), p(cst[0]), p(cst[1]), p(Kf[D:]), p(Kb[D:]), ldu, AU, p(sync), n, None, B, T, H, safe, st), 'f')
        else:
            ops.check(lib.ams_blstm_ring_bwd(p(G), p(cst[0]), p(cst[1]), p(dout), None, p(Kf[D:]), p(Kb[D:]), ldu, AU, p(sync), n, None, B, T, H, safe, st), 'b')
        torch.cuda.synchronize()
        w = sync[:64].view(torch.int64).cpu().numpy()
        ph = w[8:8 + len(names[kind])] / float(T)
        print('%s, %s: %.0f cycles per step' % (kind, mode, ph.sum()))
        for nm, v in zip(names[kind], ph):
            print('    %-34s %7.0f' % (nm, v))

# ---- the backward ring BESIDE a residency-capped weight-gradient product (the situation inside the training step): which phases stretch?
if '--beside' in sys.argv:
    side = torch.cuda.Stream()
    A = torch.randn(B * T, 2 * H, device='cuda')
    Bm = torch.randn(B * T, 10240, device='cuda')
    Cm = torch.empty(2 * H, 10240, device='cuda')
    for pad in (50000, 70000, 0):
        for kind in ('bwd', 'fwd'):
            n = lib.ams_blstm_ring_sync_bytes(B, H, int(kind == 'bwd'))
            sync = torch.zeros(n // 4 + 1, dtype=torch.float32, device='cuda')
            G.copy_(G0 if kind == 'bwd' else Gz)
            torch.cuda.synchronize()
            with torch.cuda.stream(side), ops.lds_pad(pad):
                for _ in range(3):
                    ops.gemm(A, Bm, transA=True, out=Cm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(200000)                         # let the product fill the chip first (~100 us)
            e0.record()
            if kind == 'bwd':
                ops.check(lib.ams_blstm_ring_bwd(p(G), p(cst[0]), p(cst[1]), p(dout), None, p(Kf[D:]), p(Kb[D:]), ldu, AU, p(sync), n, None, B, T, H, 2, st), 'b')
            else:
                ops.check(lib.ams_blstm_ring_fwd(p(G), p(out), p(cst[0]), p(cst[1]), p(Kf[D:]), p(Kb[D:]), ldu, AU, p(sync), n, None, B, T, H, 2, st), 'f')
            e1.record()
            torch.cuda.synchronize()
            w = sync[:64].view(torch.int64).cpu().numpy()
            ph = w[8:8 + len(names[kind])] / float(T)
            print('%s beside a dense dW product capped with a %d-byte LDS pad: %.0f cycles per step, launch %.0f us' % (kind, pad, ph.sum(), e0.elapsed_time(e1) * 1e3))
            for nm, v in zip(names[kind], ph):
                print('    %-34s %7.0f' % (nm, v))
