#!/usr/bin/env python3
"""Where does the NFFT=512 fused kernel lose accuracy with 64 filters?  
Per utterance / frame / coefficient error of fused vs generic kernel against the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
from features.batch import FeaturePlan
from features import _native as nat
from oracle import dsp_oracle
from test_gpu_batch import _batch

full = dict(samplerate=16000, nfft=512, lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True,
            winlen=0.032, winstep=0.008, nfilt=64, numcep=13)
for key, val in [a.split('=') for a in sys.argv[1:]]:
    full[key] = type(full[key])(val) if full[key] is not None else float(val)
plan = FeaturePlan(winfunc=np.hamming, **full)
lib = nat.load()
for dtype in (np.float32, np.int16):
    dense = _batch(51, 24, 8000, dtype=dtype)
    got, fo = plan.mfcc_batch(dense, delta_n=0) if False else plan.mfcc_batch(dense, delta_n=2)
    nat.check(lib.dsp_debug_force_generic(1))
    gen, _ = plan.mfcc_batch(dense, delta_n=2)
    nat.check(lib.dsp_debug_force_generic(0))
    worst = (0, None)
    for b in range(24):
        ref = dsp_oracle.mfcc_delta(dense[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **full)
        g = got[fo[b]:fo[b + 1]]; q = gen[fo[b]:fo[b + 1]]
        sc = np.max(np.abs(ref))
        e = np.abs(g - ref) / sc; eg = np.abs(q - ref) / sc
        t, k = np.unravel_index(np.argmax(e[:, :13]), e[:, :13].shape)
        if e[:, :13].max() > worst[0]:
            worst = (e[:, :13].max(), (b, t, k, eg[:, :13].max(), sc))
        # static part only
    print(dtype.__name__, 'worst static-coefficient error fused %.3g at (utt, frame, coef)=%s generic max %.3g scale %.3g' % (worst[0], worst[1][:3], worst[1][3], worst[1][4]))
    b, t, k = worst[1][:3]
    ref = dsp_oracle.mfcc_delta(dense[b].astype(np.float64), delta_n=2, winfunc=np.hamming, **full)
    print('  ref row', np.round(ref[t, :13], 4)); print('  fused  ', np.round(got[fo[b] + t, :13], 4)); print('  generic', np.round(gen[fo[b] + t, :13], 4))
    # log-mel of that frame from the oracle, to see which filters are small
    fb, en = dsp_oracle.fbank(dense[b].astype(np.float64), samplerate=16000, winlen=full['winlen'], winstep=full['winstep'], nfilt=full['nfilt'], nfft=512, lowfreq=0, highfreq=None, preemph=0.97, winfunc=np.hamming)
    print('  log fbank of the frame (first 16):', np.round(np.log(fb[t, :16]), 2), ' max', np.round(np.log(fb[t]).max(), 2))
