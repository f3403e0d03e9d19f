import sys, numpy as np, math
sys.path.insert(0,'/root/repo')
import torch
from ffsubsync_amd import _native
pcm=np.zeros(480*8,np.int16)
pcm[480:780]=400
pcm[960:1260]=400; pcm[960]=399
pcm[1440:1920]=-32768
pcm[1920:1921]=1000
pcm[2400:2408]=1000
pcm[2880+8:2880+16]=1000
dev=torch.from_numpy(pcm).cuda()
true=[(pcm[i*480:(i+1)*480].astype(np.int64)**2).sum() for i in range(8)]
est=[]
for f in range(8):
    lo,hi=-10.0,100.0
    for _ in range(60):
        mid=(lo+hi)/2
        lab=_native.vad_energy(dev,480,mid,0.0).cpu().numpy()
        if lab[f]>0: lo=mid
        else: hi=mid
    est.append(10**(lo/10)*480)
for t,e in zip(true,est): print(t, round(e,1))
