#!/usr/bin/env python3
"""End-to-end front-end of the reference's classifier input (model.py:113-135, augment=False) on
raw int16 clips: python tools/kbench_model.py [--batch 256] [--rate 44100]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--rate', type=int, default=44100)
    ap.add_argument('--device-input', action='store_true', help='waveforms already resident on the GPU')
    args = ap.parse_args()
    from features.model_glue import ModelFeatureBatch
    from oracle import dsp_oracle
    rng = np.random.default_rng(3)
    clips = []
    for _ in range(args.batch):
        n = int(rng.uniform(1.0, 2.0) * args.rate)
        x = rng.normal(0, 30, n)
        blen = int(rng.uniform(0.5, 0.9) * n)
        b0 = int(rng.integers(0, n - blen))
        t = np.arange(blen) / args.rate
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * rng.uniform(100, 300) * t) * np.hanning(blen)
        clips.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    so = np.concatenate([[0], np.cumsum([len(c) for c in clips])]).astype(np.int64)
    flat = np.concatenate(clips)
    mfb = ModelFeatureBatch(rate=args.rate)
    src = torch.from_numpy(flat).cuda() if args.device_input else flat
    lay = mfb.pipe.prepare(so, delta_n=0)      # everything that depends only on the batch shape, built once
    for _ in range(3):
        inp, len0, ends = mfb.run(src, layout=lay)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        inp, len0, ends = mfb.run(src, layout=lay)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t1 = time.perf_counter()
    for c in clips[:8]:
        dsp_oracle.model_pipeline(c, args.rate)
    cpu = (time.perf_counter() - t1) / 8
    where = 'resident on the device' if args.device_input else 'from host memory'
    print(f'{args.batch} clips ({flat.nbytes / 1e6:.1f} MB int16 {where}) -> inp {tuple(inp.shape)}: '
          f'{dt * 1e3:.2f} ms per batch = {args.batch / dt:.0f} utt/s; NumPy oracle {cpu * 1e3:.1f} ms per clip '
          f'= {1 / cpu:.0f} utt/s on one core')


if __name__ == '__main__':
    main()
