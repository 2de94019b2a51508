#!/usr/bin/env python3
"""Pitch-score kernel (row f-4) on a ragged batch of 10 kHz clips: python tools/kbench_pitch.py [--batch 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    sys.path.insert(0, p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--no-clip', action='store_true', help='skip the centre clipping (median sort)')
    args = ap.parse_args()
    from features import _native as nat
    from features import pitch as gp
    from oracle import dsp_oracle
    lib = nat.load()
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    lens = rng.integers(8000, 16000, args.batch)          # 0.8 .. 1.6 s at 10 kHz
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    x = torch.from_numpy(np.round(3000 * rng.standard_normal(so[-1])).astype(np.float32)).to(dev)
    L, S = 300, 100
    fo = nat.frame_offsets(so, L, S)
    d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
    out = torch.empty((int(fo[-1]), 180), device=dev)
    taps = gp._device_taps(L, 10000)
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        nat.check(lib.dsp_pitch_scores_batch(x.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), args.batch, int(fo[-1]), 0,
                                             L, S, taps.ptr, 0 if args.no_clip else 1, 20, 200, out.data_ptr(), st))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    c = x[:int(lens[0])].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    frames = dsp_oracle.to_frames(c, 10000, 0.03, 0.01)
    for fr in frames[:20]:
        dsp_oracle.pitch_frame_scores(dsp_oracle.center_clip(fr, False), 10000)
    cpu = (time.perf_counter() - t0) / 20
    print(f'{args.batch} clips, {int(fo[-1])} frames of 300 samples: {us:.0f} us per launch = {int(fo[-1]) / us:.2f} Mframes/s; '
          f'NumPy oracle {cpu * 1e6:.0f} us per frame on one core')


if __name__ == '__main__':
    main()
