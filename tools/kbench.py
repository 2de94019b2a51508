#!/usr/bin/env python3
"""Kernel-level A/B harness: times the fused MFCC kernel (and the whole MFCC+delta step) for one or
more builds of libdsp_frontend.so in ONE process per library (DSP_FRONTEND_LIB selects the build).

    python tools/kbench.py [--reps 200] [--rounds 5]
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
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--nfilt', type=int, default=40)
    ap.add_argument('--streams', type=int, default=1)
    ap.add_argument('--check', default='', help='npy file: written if absent, else compared with the MFCC+delta output of batch 0')
    args = ap.parse_args()
    from features import _native as nat
    from features.batch import FeaturePlan
    dev = torch.device('cuda', 0)
    B, N, T = args.batch, 16000, 99
    plan = FeaturePlan(samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=args.nfilt, nfft=512,
                       preemph=0.97, ceplifter=22, appendEnergy=True, winfunc=np.hamming)
    layout = plan.layout(np.empty((B, N), dtype=np.float32))
    g = torch.Generator(device=dev).manual_seed(1)
    waves = [0.25 * torch.randn((B, N), device=dev, generator=g) for _ in range(8)]
    out = torch.empty((B * T, 39), device=dev)
    lib = nat.load()
    st = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)] if args.streams > 1 else [st]
    outs = [torch.empty((B * T, 39), device=dev) for _ in streams]

    def mfcc_only(i):
        k = i % len(streams)
        nat.check(lib.dsp_features_batch(plan.plan.handle, waves[i % 8].data_ptr(), nat.WAVE_F32, None, None, B,
                                         B * T, N, nat.OUT_MFCC, outs[k].data_ptr(), 13, None,
                                         streams[k].cuda_stream))

    def full(i):
        k = i % len(streams)
        plan.run_raw(waves[i % 8].data_ptr(), nat.WAVE_F32, layout, outs[k].data_ptr(), 2, streams[k].cuda_stream)

    if args.check:
        full(0)
        torch.cuda.synchronize()
        got = outs[0].cpu().numpy()
        if os.path.exists(args.check):
            ref = np.load(args.check)
            err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
            print(f'check vs {os.path.basename(args.check)}: normwise {err:.2e}' + ('  <-- MISMATCH' if not err <= 2e-5 else ''))
        else:
            np.save(args.check, got)
    import ctypes
    raw = ctypes.CDLL(nat.LIB_PATH)
    if hasattr(raw, 'dsp_debug_read_stamps'):           # diagnostic build: per-phase cycle sums
        buf = (ctypes.c_ulonglong * 16)()
        for i in range(20):
            mfcc_only(i)
        raw.dsp_debug_read_stamps(buf, 16)
        for i in range(100):
            mfcc_only(i)
        raw.dsp_debug_read_stamps(buf, 16)
        if os.environ.get('KBENCH_ROUNDS'):          # library built with -DF512_STAMPS -DF512_STAMP_ROUNDS
            for r_ in range(4):
                if buf[4 + r_]:
                    print(f'  round {r_}: {buf[r_] / buf[4 + r_]:9.0f} cycles per group, {buf[4 + r_] / 100:8.0f} groups per launch')
            print(f'  wave lifetime {buf[11] / max(buf[14], 1):9.0f} cycles, shader clock {buf[11] / max(buf[12], 1) * 0.1:5.2f} GHz')
            return
        groups = 100 * B * ((T + 7) // 8)
        names = ['loop/locate', 'hbm load + stage', 'pass1 reads+window+FFT32', 'untangle+twiddle', 'exchange',
                 '2 x FFT16', 'power+packed unit+energy', 'ps write', 'mel+log', 'DCT+allreduce', 'store']
        tot = sum(buf[i] for i in range(11))
        for i, nm in enumerate(names):
            print(f'  phase {i:2d} {nm:28s} {buf[i] / groups:8.0f} cycles/group  {100.0 * buf[i] / tot:5.1f} %')
        print(f'  total {tot / groups:8.0f} cycles per group (wave lifetime)')
        if buf[14]:
            nw = buf[14]
            print(f'  per wave: lifetime {buf[11] / nw:9.0f} cycles, prologue {buf[13] / nw:7.0f} cycles, in-loop '
                  f'{tot / nw:9.0f}; shader clock {buf[11] / max(buf[12], 1) * 0.1:5.2f} GHz; waves {nw / 100:.0f} per launch')
    res = {}
    for name, fn in (('mfcc', mfcc_only), ('mfcc+delta', full)):
        for i in range(10):
            fn(i)
        times = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(st)
            for s_ in streams:
                s_.wait_event(e0)
            for i in range(args.reps):
                fn(i)
            for s_ in streams:
                ev = torch.cuda.Event()
                ev.record(s_)
                st.wait_event(ev)
            e1.record(st)
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / args.reps)
        res[name] = (float(np.median(times)), float(np.min(times)))
    fr = B * T
    print(f"{os.path.basename(nat.LIB_PATH):34s} mfcc {res['mfcc'][0]*1e3:7.1f} us (min {res['mfcc'][1]*1e3:6.1f}) "
          f"= {fr/res['mfcc'][0]/1e6:6.2f} Gframes/s | +delta {res['mfcc+delta'][0]*1e3:7.1f} us "
          f"= {fr/res['mfcc+delta'][0]/1e6:6.2f} Gframes/s | streams={len(streams)}")


if __name__ == '__main__':
    main()
