import sys, os
sys.path.insert(0,'/root/repo/adaptive-multispeaker-separation_amd'); sys.path.insert(0,'/root/repo')
import torch
from ams_hip import ops
x=torch.randn(192,20480,device='cuda')*0.05; f=torch.randn(1024,256,device='cuda')*0.03
am=(ops.absmax(x),ops.absmax(f))
for measure in (True,False):
    for _ in range(5): ops.front_conv(x,f,256,amax=am,measure=measure)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.front_conv(x,f,256,amax=am,measure=measure)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get('AMS_GEMM_X6CFG'),os.environ.get('AMS_GEMM_SK'),'measure',measure,'%.1f us'%(e0.elapsed_time(e1)*1e3/20))
