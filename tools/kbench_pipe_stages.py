#!/usr/bin/env python3
"""configs[3] pipeline, marginal cost of every stage on the stream's timeline: HIP-event time per call of the first k
stages queued back to back (k = 1 .. 5), on 1024 device-resident class-C int16 utterances.  The differences are what each
stage adds end to end -- unlike the per-kernel durations of rocprofv3, which carry a fixed start-up share per kernel.
    python tools/kbench_pipe_stages.py [--reps 200]"""
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
    ap.add_argument('--reps', type=int, default=200)
    args = ap.parse_args()
    from features import _native as nat
    from features.pipeline import VadMfccPipeline
    dev = torch.device('cuda', 0)
    sigs = make_batch(args.batch)
    so = np.concatenate([[0], np.cumsum([len(s) for s in sigs])]).astype(np.int64)
    d_wave = torch.from_numpy(np.concatenate(sigs)).to(dev)
    pipe = VadMfccPipeline(rate=16000, frame=0.03, step=0.01, unit_variance=True, winfunc=np.hamming, winlen=0.025,
                           winstep=0.01, numcep=13, nfilt=40, nfft=512, preemph=0.97, ceplifter=22, appendEnergy=True)
    lay = pipe.prepare(so, 2)
    d_feat = torch.empty((lay.frames_bound, lay.D), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    lib = nat.load()
    ep, fp = pipe.endpoint, pipe.features
    wp, dt = d_wave.data_ptr(), nat.WAVE_I16
    handle = lay.vad.vad_handle(ep.L, ep.S)

    def s_vad():
        nat.check(lib.dsp_vad_features_layout_batch(handle, wp, dt, lay.vad.p_sample, lay.vad.p_frame, 0, lay.d_amp.ptr, lay.d_zcr.ptr, st))

    def s_rule():
        nat.check(lib.dsp_endpoint_rule_batch(lay.d_amp.ptr, lay.d_zcr.ptr, lay.vad.d_frame.ptr, lay.n_utt, ep.L, float(ep.frame), float(ep.step),
                                              lay.d_ep.ptr, st))

    def s_layout():
        nat.check(lib.dsp_endpoint_layout_segments_batch(lay.d_ep.ptr, lay.vad.p_sample, lay.n_utt, float(ep.step), float(ep.rate), None,
                                                         lay.d_seg.ptr, lay.d_dst_off.ptr, lay.d_frame_off.ptr, fp.plan.handle,
                                                         max(lay.frames_bound, 1), lay.d_work.ptr, lay.d_work.nbytes, st))

    def s_feat(delta_n):
        def f():
            nat.check(lib.dsp_mfcc_delta_segments_batch(fp.plan.handle, wp, dt, lay.vad.p_sample, lay.d_seg.ptr, lay.d_frame_off.ptr, lay.n_utt,
                                                        max(lay.frames_bound, 1), delta_n, 1 | 2, lay.d_work.ptr, lay.d_work.nbytes,
                                                        d_feat.data_ptr(), st))
        return f

    pipe.launch(wp, dt, lay, d_feat.data_ptr(), st)       # every intermediate buffer holds valid data from here on
    torch.cuda.synchronize()
    chains = [('VAD features', [s_vad]), ('+ endpoint rule', [s_vad, s_rule]), ('+ layout', [s_vad, s_rule, s_layout]),
              ('+ MFCC (cepstra only)', [s_vad, s_rule, s_layout, s_feat(0)]), ('+ delta rows (the whole pipeline)', [s_vad, s_rule, s_layout, s_feat(2)])]
    prev = 0.0
    for name, fns in chains:
        best = 1e30
        for _ in range(3):
            for _ in range(10):
                for f in fns:
                    f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                for f in fns:
                    f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.reps * 1e3)
        print(f'{name:36s} {best:7.1f} us per call   (+{best - prev:5.1f})')
        prev = best


if __name__ == '__main__':
    main()
