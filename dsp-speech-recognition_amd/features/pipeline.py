"""BASELINE.json configs[3]: energy / ZCR endpointing feeding the feature extractor, batched, with
variable-length outputs -- the device-side form of model.py:113-121 (endpoint_detect -> scale ->
feature_extract_mfcc, augment=False).

    waveforms (ragged, int16 or fp32)
      -> per-frame amplitude + ZCR, two-threshold rule        (dsp_vad_features_batch, dsp_endpoint_rule_batch)
      -> [left, right) per utterance                           (8 bytes per utterance come to the host)
      -> trimmed (optionally unit-variance) copy               (dsp_trim_scale_batch)
      -> MFCC (+ delta, delta-delta) of the trimmed clips      (dsp_mfcc_delta_batch, ragged fast path)

The endpoint rule must have seen a whole utterance before its first trimmed frame can be cut, so
the two framings (cfg.frame for VAD, winlen for MFCC) cannot share one pass; what is shared is the
device-resident waveform -- nothing but the endpoints and the final features crosses PCIe.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from .batch import EndpointPlan, FeaturePlan, _BatchLayout, _is_device_tensor, _wave_dtype_of


class _Borrowed:
    """Device memory owned by the caller (a tensor); only the address is kept."""

    def __init__(self, ptr):
        self.ptr = int(ptr)


def _ones(x):
    return np.ones((x,))


class VadMfccPipeline:
    def __init__(self, rate=16000, frame=0.03, step=0.01, unit_variance=False, **mfcc_kwargs):
        """mfcc_kwargs: arguments of base.mfcc (winlen, winstep, numcep, nfilt, nfft, preemph, ...).
        unit_variance=True reproduces model.py:63 (divide the trimmed clip by its population std)."""
        mfcc_kwargs.setdefault('samplerate', rate)
        self.rate = rate
        self.unit_variance = bool(unit_variance)
        self.endpoint = EndpointPlan(rate, frame, step)
        self.features = FeaturePlan(**mfcc_kwargs)

    def run(self, waves, sample_offsets, delta_n=2, download=True):
        """waves: 1-D host array (int16 or float) or torch-ROCm tensor (int16 / float32) of concatenated
        utterances.
        Returns (features [sum T_b, D] fp32, frame_offsets [B+1], endpoints [B, 2] in samples);
        with ``download=False`` the features stay on the device and the first element is the
        (DeviceBuffer, _BatchLayout) pair of the result instead."""
        nat.require_device()
        lib = nat.load()
        so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        B = len(so) - 1
        if _is_device_tensor(waves):                # torch-ROCm tensor: nothing crosses PCIe
            if not waves.is_contiguous():
                waves = waves.contiguous()
            dtype = _wave_dtype_of(waves)
            d_wave = _Borrowed(waves.data_ptr())
        else:
            wave, dtype = nat.as_wave(np.asarray(waves).reshape(-1))
            d_wave = nat.device_array('batch_wave', wave)

        # 1. endpoints (frame indices -> sample indices exactly as endpoint.py:64)
        ep = self.endpoint
        lay = _BatchLayout(ep.L, ep.S, B, sample_offsets=so, scratch='pipe_vad')
        nf = lay.total_frames
        d_amp = nat.SCRATCH.get('ep_amp', nf * 8)
        d_zcr = nat.SCRATCH.get('ep_zcr', nf * 4)
        d_ep = nat.SCRATCH.get('ep_batch', B * 8)
        ep.run_raw(d_wave.ptr, dtype, lay, d_amp.ptr, d_zcr.ptr, d_ep.ptr, None)
        frames = d_ep.download((B, 2), np.int32)
        # int((idx * step) * rate), the fp64 product order of endpoint.py:64, vectorised; numpy
        # slicing sig[left:right] clips at the end of the clip
        ends = ((frames.astype(np.float64) * ep.step) * ep.rate).astype(np.int64)
        ends = np.minimum(ends, np.diff(so)[:, None])

        # 2. trimmed copy on the device
        lens = np.maximum(ends[:, 1] - ends[:, 0], 0)
        dst_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        total = int(dst_off[-1])
        d_src_off = nat.device_array('trim_src_off', so)
        d_seg = nat.device_array('trim_seg', np.ascontiguousarray(ends.reshape(-1)))
        d_dst_off = nat.device_array('trim_dst_off', dst_off)
        d_trim = nat.SCRATCH.get('trim_wave', max(total, 1) * 4)
        nat.check(lib.dsp_trim_scale_batch(d_wave.ptr, dtype, d_src_off.ptr, d_seg.ptr, d_dst_off.ptr, B,
                                           1 if self.unit_variance else 0, d_trim.ptr, None))

        # 3. features of the trimmed clips (ragged layout)
        fp = self.features
        flay = _BatchLayout(fp.L, fp.S, B, sample_offsets=dst_off, scratch='pipe_mfcc')
        D = fp.width(delta_n)
        d_out = nat.SCRATCH.get('batch_out', flay.total_frames * D * 4)
        fp.run_raw(d_trim.ptr, nat.WAVE_F32, flay, d_out.ptr, delta_n, None)
        if not download:
            return (d_out, flay), flay.frame_offsets, ends
        out = d_out.download((flay.total_frames, D), np.float32)
        return out, flay.frame_offsets, ends
