"""Hard k-means (10 tries x 10 iterations, b = 64, L = 20480, E = 40) with and without silence weights: ms per run.
   python tools/probes/km_w_bench.py      (AMS_KM_TRIES=0: one workgroup per try)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd')]
from ams_hip import ops
b, L, E, tries = 64, 20480, 40, 10
g = torch.Generator(device='cuda').manual_seed(0)
X = torch.randn(b, L, E, device='cuda', generator=g)
w = (torch.rand(b, L, device='cuda', generator=g) > 0.25).float()
idx = torch.stack([torch.randperm(L, device='cuda', generator=g)[:2] for _ in range(b * tries)]).to(torch.int32)
xn = ops.kmeans_normalize(X)
for name, ww in (('no weights', None), ('silence weights', w)):
    for _ in range(2):
        ops.kmeans_run(xn, idx, 2, tries, 10, w=ww)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.kmeans_run(xn, idx, 2, tries, 10, w=ww)
    e1.record(); torch.cuda.synchronize()
    print('AMS_KM_TRIES=%s %-16s %.3f ms per 10 x 10 run' % (os.environ.get('AMS_KM_TRIES', '1'), name, e0.elapsed_time(e1) / 5))
