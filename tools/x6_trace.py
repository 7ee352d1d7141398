"""Phase trace of the bf16x6 product kernel (a library built with -DAMS_X6_TRACE=1, passed as AMS_HIP_LIB): workgroup 0 stamps the
100 MHz wall clock at its phase boundaries; prints nanoseconds per phase and k-tile for the consumer (MFMA) and producer (split)
waves of the wave-specialised form.   python tools/x6_trace.py [M N K]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops  # noqa: E402
from ams_hip._lib import load  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
lib = load()
lib.ams_gemm_set_arith(1)
A = torch.randn(M, K, device='cuda')
B = torch.randn(K, N, device='cuda')
out = torch.empty(M, N, device='cuda')
for _ in range(3):
    ops.gemm(A, B, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 4))()
raw = ctypes.CDLL(os.environ['AMS_HIP_LIB'])
assert raw.ams_gemm_x6_trace_read(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64) * 10          # ns
mode = os.environ.get('AMS_GEMM_X6MODE', '2')
roles = ((0, 'fused stream', ('start->stream issued', 'stream issued->barrier passed')),) if mode == '2' else ((0, 'consumer', ('start->mfma issued', 'mfma issued->barrier passed')),
                         (1, 'producer', ('start->split+LDS writes done', '->fetch issued', '->barrier passed')))
for role, name, cols in roles:
    print(name)
    for kt in range(2, 14):
        r = t[role, kt]
        if role == 0:
            d = (r[1] - r[0], r[3] - r[1])
        else:
            d = (r[1] - r[0], r[2] - r[1], r[3] - r[2])
        nxt = t[role, kt + 1, 0] - r[0]
        print('  tile %2d  ' % kt + '  '.join('%s %5d ns' % (c, v) for c, v in zip(cols, d)) + '   | period %5d ns' % nxt)
