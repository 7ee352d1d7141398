import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/adaptive-multispeaker-separation_amd')
import numpy as np, torch
from utils import bss_eval as hb
from oracle import bss_eval as ob
rng=np.random.RandomState(0); S,L=2,20480
s=rng.randn(S,L); est=s[::-1]+0.3*rng.randn(S,L)
st=torch.tensor(s,device='cuda'); et=torch.tensor(est,device='cuda')
for _ in range(3): hb.bss_eval_pairs(st,et)
torch.cuda.synchronize(); t=time.time(); n=20
for _ in range(n): hb.bss_eval_pairs(st,et)
torch.cuda.synchronize(); g=(time.time()-t)/n
t=time.time(); ob.bss_eval_sources(s,est); c=time.time()-t
print('gpu ms per call (4 pairs, S=2, L=20480):', g*1e3, ' oracle numpy s:', c)
