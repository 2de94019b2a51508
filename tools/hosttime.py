import os, sys, time
ROOT='/root/repo'
for p in (ROOT, os.path.join(ROOT,'dsp-speech-recognition_amd')):
    sys.path.insert(0,p)
import numpy as np, torch
from features import _native as nat
from features.batch import FeaturePlan
dev=torch.device('cuda',0)
B,N,T=1024,16000,99
plan=FeaturePlan(samplerate=16000,winlen=0.025,winstep=0.01,numcep=13,nfilt=40,nfft=512,preemph=0.97,ceplifter=22,appendEnergy=True,winfunc=np.hamming)
layout=plan.layout(np.empty((B,N),dtype=np.float32))
waves=[0.25*torch.randn((B,N),device=dev) for _ in range(8)]
streams=[torch.cuda.Stream(dev) for _ in range(3)]
outs=[torch.empty((B*T,39),device=dev) for _ in range(4)]
sps=[s.cuda_stream for s in streams]
wp=[w.data_ptr() for w in waves]; op=[o.data_ptr() for o in outs]
def step(i):
    plan.run_raw(wp[i%8], nat.WAVE_F32, layout, op[i%4], 2, sps[i%3])
for i in range(50): step(i)
torch.cuda.synchronize()
for K in (400, 400, 2000, 4000, 400):
    t0=time.perf_counter()
    for i in range(K): step(i)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(f'K={K}: enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step')
