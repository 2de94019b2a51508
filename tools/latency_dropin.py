#!/usr/bin/env python3
"""Latency of the single-utterance drop-in calls (BASELINE configs[0] call pattern: model.py calls
mfcc / delta / basic_endpoint_detection once per utterance from a Python loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    sys.path.insert(0, p)
import numpy as np
import features
from oracle import dsp_oracle


def bench(fn, n=300):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


rng = np.random.default_rng(0)
x = 0.25 * rng.standard_normal(16000)
xi = np.round(3000 * rng.standard_normal(25600)).astype(np.int16)
cfg = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, preemph=0.97,
           ceplifter=22, appendEnergy=True, winfunc=np.hamming)
m = features.mfcc(x, **cfg)
rows = [
    ('mfcc 1 s (configs[0])', lambda: features.mfcc(x, **cfg), lambda: dsp_oracle.mfcc(x, **cfg)),
    ('delta [99,13] N=2', lambda: features.delta(m, 2), lambda: dsp_oracle.delta(m, 2)),
    ('fbank 1 s', lambda: features.fbank(x, **{k: v for k, v in cfg.items() if k not in ("numcep", "ceplifter", "appendEnergy")}),
     lambda: dsp_oracle.fbank(x, **{k: v for k, v in cfg.items() if k not in ("numcep", "ceplifter", "appendEnergy")})),
    ('basic_endpoint_detection 1.6 s int16', lambda: features.basic_endpoint_detection(xi, 16000),
     lambda: dsp_oracle.basic_endpoint_detection(xi, 16000)),
]
for name, gpu, cpu in rows:
    print(f'{name:40s} drop-in {bench(gpu):8.1f} us   oracle (NumPy, 1 core) {bench(cpu, 30):8.1f} us')
