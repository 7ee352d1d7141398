"""Where a 16-bit-pipe product's time goes, per work item and phase (timing-anatomy build: make -C csrc stamp).

usage (GPU box): python tools/gemm_anatomy.py [--shape M,N,K,tA,tB ...] [--sk 0|1]
Thread 0 of every workgroup stamps the 100 MHz wall clock at: 0 item start, 1 first k-tile staged (before the first MFMA), 2 k loop done,
3 owner's partials added / column sums written, 4 next item's first fetch issued, 5 stores issued.  Printed per item index: mean over the
workgroups of each phase length (us), and the launch's span (first stamp 0 to last stamp 5)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')
os.environ.setdefault('AMS_HIP_LIB', os.path.join(PKG, 'ams_hip', 'libams_hip_stamp.so'))
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np      # noqa: E402
import torch            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', action='append', default=[])
    ap.add_argument('--sk', type=int, default=1)
    args = ap.parse_args()
    from ams_hip import ops
    from ams_hip._lib import load
    lib = load()
    dbg = ctypes.CDLL(os.environ['AMS_HIP_LIB']).ams_dbg_x6_stamps
    dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ops.SK = bool(args.sk)
    shapes = args.shape or ['5120,2400,600,0,0', '5120,2400,256,0,0', '5120,10240,600,0,0', '5120,600,10240,0,1', '5120,600,2400,0,1']
    rate = float(lib.ams_stamp_rate())
    for sh in shapes:
        M, N, K, tA, tB = [int(v) for v in sh.split(',')]
        A = torch.randn((K, M) if tA else (M, K), device='cuda')
        B = torch.randn((N, K) if tB else (K, N), device='cuda')
        bias = torch.randn(N, device='cuda') if not tA else None
        am = (ops.absmax(A), ops.absmax(B))
        for _ in range(5):
            ops.gemm(A, B, transA=bool(tA), transB=bool(tB), bias=bias, amax=am)
        torch.cuda.synchronize()
        dbg(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(A, B, transA=bool(tA), transB=bool(tB), bias=bias, amax=am)
        e1.record()
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 8 * 8, dtype=np.int64)
        dbg(buf.ctypes.data_as(ctypes.c_void_p), 0)
        st = buf.reshape(1024, 8, 8)
        live = st[:, :, 0] != 0
        t0 = st[:, :, 0][live].min()
        print('== %s sk=%d: event %.1f us, stamped span %.1f us, %d workgroups, items per workgroup up to %d' % (
            sh, args.sk, e0.elapsed_time(e1) * 1e3, (st[:, :, 5].max() - t0) / rate * 1e6, int(live[:, 0].sum()), int(live.sum(1).max())))
        print('   item  wgs  role(F/P/O)  nk    start   stage1    kloop   fixup  prefetch   stores   (us, mean over workgroups; start = since launch)')
        for i in range(8):
            m = live[:, i]
            if not m.any():
                break
            s = st[m, i, :]
            ph = [(s[:, k + 1] - s[:, k]).mean() / rate * 1e6 for k in range(5)]
            roles = [int((s[:, 6] == r).sum()) for r in range(3)]
            print('   %4d %4d  %3d/%3d/%3d %5.1f %8.1f %8.1f %8.1f %7.1f %8.1f %8.1f' % (
                i, int(m.sum()), roles[0], roles[1], roles[2], s[:, 7].mean(), (s[:, 0] - t0).mean() / rate * 1e6, *ph))


if __name__ == '__main__':
    main()
