"""Where and when the workgroups of kmeans_hard_tries_kernel ran (AMS_KT_DBG build): per CU the number of workgroups resident at once."""
import ctypes, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'adaptive-multispeaker-separation_amd'))
os.environ['AMS_HIP_LIB'] = os.path.abspath(sys.argv[1])
from ams_hip import ops
from ams_hip._lib import load
lib = load()
b, L, tries = 64, 20480, 10
nwg = b * (tries // 5) * 4 * 3
lib.ams_dbg_kt_buffer.restype = ctypes.c_void_p
ptr = lib.ams_dbg_kt_buffer(ctypes.c_size_t(nwg * 10 * 4))
print('occupancy (runtime):', lib.ams_dbg_kt_occupancy())
g = torch.Generator().manual_seed(1)
X = torch.randn(b, L, 40, generator=g).cuda()
idx = torch.stack([torch.randperm(L, generator=g)[:2] for _ in range(b * tries)]).to(torch.int32).cuda()
xn = ops.kmeans_normalize(X)
ops.kmeans_run(xn, idx, 2, tries, 1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (nwg * 10 * 4))()
from ctypes import byref
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpy(buf, ctypes.c_void_p(ptr), ctypes.c_size_t(nwg * 10 * 4 * 8), 2)
d = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 10, 4)
hw = (d[:, :, 0] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (d[:, :, 0] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
t0 = d[:, :, 1].astype(np.int64); t1 = d[:, :, 2].astype(np.int64)
base = t0.min()
print('span (100 MHz ticks):', t1.max() - base, ' WG lifetime median:', int(np.median(t1.max(1) - t0.min(1))))
key = xcc[:, 0] * 10000 + se[:, 0] * 1000 + sh[:, 0] * 100 + cu[:, 0]
print('distinct CUs used:', len(np.unique(key)))
print('waves of one WG on SIMDs:', [collections.Counter(simd[i].tolist()) for i in (0, 1, 700)])
# concurrency per CU: sweep
conc = []
for k in np.unique(key):
    m = key == k
    ev = sorted([(int(t0[i].min()), 1) for i in np.where(m)[0]] + [(int(t1[i].max()), -1) for i in np.where(m)[0]])
    c = mx = 0; area = 0; last = ev[0][0]
    for t, dl in ev:
        area += c * (t - last); last = t; c += dl; mx = max(mx, c)
    conc.append((mx, area / max(1, ev[-1][0] - ev[0][0]), m.sum()))
conc = np.array(conc)
print('per CU: max resident WGs: min %d median %d max %d; time-average resident: %.2f; WGs per CU: min %d max %d' %
      (conc[:, 0].min(), np.median(conc[:, 0]), conc[:, 0].max(), conc[:, 1].mean(), conc[:, 2].min(), conc[:, 2].max()))
