"""One product shape, a few launches (for counter sweeps): python tools/probes/one_gemm.py M,N,K,tA,tB [n]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')):
    sys.path.insert(0, _p)
import torch                        # noqa: E402
from ams_hip import ops             # noqa: E402
M, N, K, tA, tB = [int(v) for v in sys.argv[1].split(',')]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
A = torch.randn((K, M) if tA else (M, K), device='cuda')
B = torch.randn((N, K) if tB else (K, N), device='cuda')
am = (ops.absmax(A), ops.absmax(B))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n + 3):
    if i == 3:
        e0.record()
    ops.gemm(A, B, transA=bool(tA), transB=bool(tB), amax=am)
e1.record()
torch.cuda.synchronize()
print('%s group_m=%s sk=%s: %.1f us per launch' % (sys.argv[1], os.environ.get('AMS_GEMM_GROUP_M'), os.environ.get('AMS_GEMM_SK'), e0.elapsed_time(e1) * 1e3 / n))
