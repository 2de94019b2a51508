#!/usr/bin/env python3
"""configs[3] end to end (VadMfccPipeline.launch on 1024 device-resident class-C int16 utterances), for rocprofv3
--stats: python tools/kbench_pipe.py [--reps 20] [--copy]   (--copy: the trimmed-copy form of the feature stage)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from kbench_vad import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--copy', action='store_true')
    ap.add_argument('--no-unit-variance', action='store_true')
    args = ap.parse_args()
    from features import _native as nat
    from features.pipeline import VadMfccPipeline
    dev = torch.device('cuda', 0)
    sigs = make_batch(args.batch)
    so = np.concatenate([[0], np.cumsum([len(s) for s in sigs])]).astype(np.int64)
    d_wave = torch.from_numpy(np.concatenate(sigs)).to(dev)
    pipe = VadMfccPipeline(rate=16000, frame=0.03, step=0.01, unit_variance=not args.no_unit_variance, winfunc=np.hamming, winlen=0.025,
                           winstep=0.01, numcep=13, nfilt=40, nfft=512, preemph=0.97, ceplifter=22, appendEnergy=True)
    pipe.copy_trimmed = args.copy
    lay = pipe.prepare(so, 2)
    d_feat = torch.empty((lay.frames_bound, lay.D), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(5):
        pipe.launch(d_wave.data_ptr(), nat.WAVE_I16, lay, d_feat.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        pipe.launch(d_wave.data_ptr(), nat.WAVE_I16, lay, d_feat.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    print(f'configs[3] pipeline ({"trimmed copy" if args.copy else "in place"}): {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us per {args.batch} utterances')


if __name__ == '__main__':
    main()
