"""GPU check of the matrix-pipe kernel against the fp64 oracle (and against the vector-pipe kernel).
Run on the GPU box: python tools/mfma512_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dsp-speech-recognition_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import dsp_oracle  # noqa: E402
from features.batch import FeaturePlan  # noqa: E402
import golden_cases as gc  # noqa: E402

CFG = dict(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512, lowfreq=0, highfreq=None,
           preemph=0.97, ceplifter=22, appendEnergy=True)


def batch(B, N, seed=3, dtype=np.float32, kinds=('white', 'tone', 'harmonic', 'siltail', 'vadf', 'ramp', 'zeros')):
    rng = np.random.default_rng(seed)
    x = np.empty((B, N), np.float64)
    for b in range(B):
        if b < 32:
            x[b] = np.asarray(gc.make_signal((kinds[b % len(kinds)], 100 + b, N)), np.float64)[:N]
        else:
            x[b] = 0.25 * rng.standard_normal(N) * (10.0 ** rng.uniform(-3, 1))
    if dtype == np.int16:
        return np.clip(np.round(x * 3000), -32768, 32767).astype(np.int16)
    return x.astype(np.float32)


def check(name, cfg, B, N, delta_n, dtype=np.float32, nref=40):
    plan = FeaturePlan(winfunc=np.hamming, **cfg)
    w = batch(B, N, dtype=dtype)
    got, fo = plan.mfcc_batch(w, delta_n=delta_n)
    worst, bad = 0.0, None
    for b in list(range(min(B, nref))) + [B - 1]:
        x = w[b].astype(np.float64)
        ref = dsp_oracle.mfcc_delta(x, delta_n=delta_n, winfunc=np.hamming, **cfg) if delta_n else dsp_oracle.mfcc(x, winfunc=np.hamming, **cfg)
        g = got[fo[b]:fo[b + 1]]
        if not np.all(np.isfinite(g)):
            err = float('inf')
        else:
            err = float(np.max(np.abs(g - ref)) / max(np.max(np.abs(ref)), 1e-300))
        if err > worst:
            worst, bad = err, b
    print(f'{name}: B={B} N={N} delta_n={delta_n} {np.dtype(dtype).name}: worst {worst:.3e} (utt {bad})', flush=True)
    if worst > 1e-4 and bad is not None:
        x = w[bad].astype(np.float64)
        ref = dsp_oracle.mfcc_delta(x, delta_n=delta_n, winfunc=np.hamming, **cfg) if delta_n else dsp_oracle.mfcc(x, winfunc=np.hamming, **cfg)
        g = got[fo[bad]:fo[bad + 1]]
        d = np.abs(g - ref)
        r, c = np.unravel_index(np.argmax(np.where(np.isfinite(d), d, 1e30)), d.shape)
        print('   worst at frame', r, 'col', c, 'got', g[r, c], 'ref', ref[r, c])
        np.set_printoptions(precision=4, linewidth=200, suppress=True)
        print('   got[0,:13]', g[0, :13])
        print('   ref[0,:13]', ref[0, :13])
        print('   per-frame max err (first 40):', np.max(d, 1)[:40])
        print('   per-col max err:', np.max(d, 0))
    return worst


if __name__ == '__main__':
    ok = True
    t0 = time.time()
    ok &= check('mfcc', CFG, 1024, 16000, 0) <= 1e-4
    ok &= check('rows', CFG, 1024, 16000, 2) <= 1e-4
    ok &= check('rows int16', CFG, 1024, 16000, 2, dtype=np.int16) <= 1e-4
    ok &= check('rows n26', dict(CFG, nfilt=26), 1024, 16000, 2) <= 1e-4
    ok &= check('rows odd N', CFG, 777, 12345, 2) <= 1e-4
    ok &= check('rows N1', CFG, 600, 8000, 1) <= 1e-4
    ok &= check('rows L320', dict(CFG, winlen=0.02), 640, 16000, 2) <= 1e-4
    ok &= check('short', CFG, 700, 1000, 2) <= 1e-4
    print('ALL OK' if ok else 'FAILED', f'{time.time() - t0:.1f}s')
    sys.exit(0 if ok else 1)
