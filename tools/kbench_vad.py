#!/usr/bin/env python3
"""Stage timings of the configs[3] pipeline (VAD features -> endpoint rule -> trim -> ragged MFCC)
on device-resident class-C utterances: python tools/kbench_vad.py [--batch 1024]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'dsp-speech-recognition_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_batch(B, seed=7, rate=16000):
    rng = np.random.default_rng(seed)
    sigs = []
    for _ in range(B):
        n = int(rng.uniform(1.0, 2.0) * rate)
        x = rng.normal(0, 30, n)
        blen = int(rng.uniform(0.5, 0.9) * n)
        b0 = int(rng.integers(0, n - blen))
        t = np.arange(blen) / rate
        x[b0:b0 + blen] += 8000 * np.sin(2 * np.pi * rng.uniform(100, 300) * t) * np.hanning(blen)
        sigs.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    return sigs


def timeit(fn, reps):
    for _ in range(3):
        fn()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    from features import _native as nat
    from features.batch import EndpointPlan, FeaturePlan, _BatchLayout
    lib = nat.load()
    dev = torch.device('cuda', 0)
    sigs = make_batch(args.batch)
    so = np.concatenate([[0], np.cumsum([len(s) for s in sigs])]).astype(np.int64)
    flat = np.concatenate(sigs)
    d_wave = torch.from_numpy(flat).to(dev)
    ep = EndpointPlan(16000, 0.03, 0.01)
    lay = _BatchLayout(ep.L, ep.S, args.batch, sample_offsets=so)
    nf = lay.total_frames
    d_amp = torch.empty(nf, dtype=torch.float64, device=dev)
    d_zcr = torch.empty(nf, dtype=torch.int32, device=dev)
    d_ep = torch.empty((args.batch, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def vad():
        nat.check(lib.dsp_vad_features_batch(d_wave.data_ptr(), nat.WAVE_I16, lay.p_sample, lay.p_frame, args.batch,
                                             nf, 0, ep.L, ep.S, 0, d_amp.data_ptr(), d_zcr.data_ptr(), st))

    def rule():
        nat.check(lib.dsp_endpoint_rule_batch(d_amp.data_ptr(), d_zcr.data_ptr(), lay.p_frame, args.batch, ep.L,
                                              float(ep.frame), float(ep.step), d_ep.data_ptr(), st))

    t_vad = timeit(vad, args.reps)
    t_rule = timeit(rule, args.reps)
    frames = d_ep.cpu().numpy().astype(np.int64)
    lens = np.diff(so)
    ends = np.stack([np.minimum((frames[:, 0] * ep.step * ep.rate).astype(np.int64), lens),
                     np.minimum((frames[:, 1] * ep.step * ep.rate).astype(np.int64), lens)], axis=1)
    dst = np.concatenate([[0], np.cumsum(ends[:, 1] - ends[:, 0])]).astype(np.int64)
    d_so = torch.from_numpy(so).to(dev)
    d_seg = torch.from_numpy(np.ascontiguousarray(ends.reshape(-1))).to(dev)
    d_dst = torch.from_numpy(dst).to(dev)
    d_trim = torch.empty(int(dst[-1]), dtype=torch.float32, device=dev)

    def trim():
        nat.check(lib.dsp_trim_scale_batch(d_wave.data_ptr(), nat.WAVE_I16, d_so.data_ptr(), d_seg.data_ptr(),
                                           d_dst.data_ptr(), args.batch, 1, d_trim.data_ptr(), st))

    t_trim = timeit(trim, args.reps)
    fp = FeaturePlan(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=40, nfft=512,
                     preemph=0.97, ceplifter=22, appendEnergy=True, winfunc=np.hamming)
    flay = _BatchLayout(fp.L, fp.S, args.batch, sample_offsets=dst)
    out = torch.empty((flay.total_frames, 39), device=dev)

    def mfcc():
        fp.run_raw(d_trim.data_ptr(), nat.WAVE_F32, flay, out.data_ptr(), 2, st)

    t_mfcc = timeit(mfcc, args.reps)
    nsamp = int(so[-1])
    print(f'B={args.batch}: {nsamp} samples ({nsamp * 2 / 1e6:.1f} MB int16), {nf} VAD frames, '
          f'{int(dst[-1])} kept samples, {flay.total_frames} MFCC frames')
    print(f'  vad features  {t_vad:8.1f} us  ({nsamp * 2 / t_vad / 1e3:.1f} GB/s of int16 read once)')
    print(f'  endpoint rule {t_rule:8.1f} us')
    print(f'  trim + scale  {t_trim:8.1f} us  ({(nsamp * 2 + int(dst[-1]) * 4) / t_trim / 1e3:.1f} GB/s)')
    print(f'  ragged MFCC+d {t_mfcc:8.1f} us  ({flay.total_frames / t_mfcc / 1e3:.3f} Gframes/s)')


if __name__ == '__main__':
    main()
