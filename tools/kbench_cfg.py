#!/usr/bin/env python3
"""Times MFCC (+delta) for an arbitrary configuration, e.g. the production transform size of
model.py:74 (48 kHz, 30 ms / 10 ms, nfft=1536, 26 mel):

    python tools/kbench_cfg.py --rate 48000 --winlen 0.03 --nfft 1536 --nfilt 26 --batch 512
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rate', type=int, default=48000)
    ap.add_argument('--winlen', type=float, default=0.03)
    ap.add_argument('--winstep', type=float, default=0.01)
    ap.add_argument('--nfft', type=int, default=1536)
    ap.add_argument('--nfilt', type=int, default=26)
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--seconds', type=float, default=1.0)
    ap.add_argument('--delta', type=int, default=3)
    ap.add_argument('--reps', type=int, default=50)
    args = ap.parse_args()
    from features import _native as nat
    from features.batch import FeaturePlan
    dev = torch.device('cuda', 0)
    plan = FeaturePlan(samplerate=args.rate, winlen=args.winlen, winstep=args.winstep, numcep=13,
                       nfilt=args.nfilt, nfft=args.nfft, preemph=0.97, ceplifter=22, appendEnergy=True,
                       winfunc=np.hamming)
    B, N = args.batch, int(args.rate * args.seconds)
    layout = plan.layout(np.empty((B, N), dtype=np.float32))
    g = torch.Generator(device=dev).manual_seed(1)
    waves = [0.25 * torch.randn((B, N), device=dev, generator=g) for _ in range(4)]
    D = plan.width(args.delta)
    out = torch.empty((layout.total_frames, D), device=dev)
    st = torch.cuda.current_stream(dev)
    fast = nat.load().dsp_plan_has_fast_path(plan.plan.handle)
    for i in range(5):
        plan.run_raw(waves[i % 4].data_ptr(), nat.WAVE_F32, layout, out.data_ptr(), args.delta, st.cuda_stream)
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.reps):
            plan.run_raw(waves[i % 4].data_ptr(), nat.WAVE_F32, layout, out.data_ptr(), args.delta, st.cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.reps)
    frames = layout.total_frames
    byt = 4 * B * N + 4 * frames * D
    print(f'L={plan.L} S={plan.S} nfft={args.nfft} M={args.nfilt} fast_path={fast}: {frames} frames '
          f'{best * 1e3:.1f} us/step = {frames / best / 1e6:.3f} Gframes/s, {byt / best / 1e6:.1f} GB/s algorithmic')


if __name__ == '__main__':
    main()
