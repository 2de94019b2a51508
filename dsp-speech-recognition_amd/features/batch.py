"""Batched entry points -- the form the hardware wants.  The reference calls ``mfcc`` once per
utterance from a Python loop (model.py:115-121); these classes take a whole batch
(concatenated waveforms + offsets, or a dense ``[B, N]`` array) and launch once.

Inputs may be host NumPy arrays (copied to the device and back) or device tensors exposing
``data_ptr()`` / ``is_cuda`` (torch-ROCm), in which case nothing crosses PCIe and the result is
returned as a torch tensor on the same device.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from . import _plan


def _ones(x):
    return np.ones((x,))


def _is_device_tensor(x):
    return hasattr(x, 'data_ptr') and getattr(x, 'is_cuda', False)


def _check_device(x):
    """A device tensor must live on the device the library is bound to (dsp_set_device): plans, scratch
    and kernels follow the current HIP device, a tensor elsewhere would be written across devices."""
    idx = getattr(getattr(x, 'device', None), 'index', None)
    cur = nat.current_device()
    if idx is not None and idx != cur:
        raise nat.DspError(f'tensor lives on cuda:{idx}, the library is on device {cur} (dsp_set_device)')


def _stream_ptr(stream):
    if stream is None:
        return None
    return int(getattr(stream, 'cuda_stream', stream))


class _BatchLayout:
    """Sample/frame offsets of one batch on host and (for ragged batches) on the device."""

    def __init__(self, L, S, n_utt, uniform_samples=0, sample_offsets=None, scratch=None):
        """``scratch``: name of a library-owned scratch slot for the device copies of the offsets (for
        layouts that live for one call; avoids two hipMalloc / hipFree pairs per call)."""
        self.n_utt = int(n_utt)
        self.uniform_samples = int(uniform_samples)
        if self.uniform_samples > 0:
            self.T = nat.frame_count(self.uniform_samples, L, S)
            self.frame_offsets = np.arange(self.n_utt + 1, dtype=np.int64) * self.T
            self.sample_offsets = np.arange(self.n_utt + 1, dtype=np.int64) * self.uniform_samples
            self.d_sample = self.d_frame = None
        else:
            self.sample_offsets = np.ascontiguousarray(sample_offsets, dtype=np.int64)
            assert self.sample_offsets.shape == (self.n_utt + 1,)
            self.frame_offsets = nat.frame_offsets(self.sample_offsets, L, S)
            self.T = 0
            if scratch is not None:
                self.d_sample = nat.device_array(scratch + '_so', self.sample_offsets)
                self.d_frame = nat.device_array(scratch + '_fo', self.frame_offsets)
            else:
                self.d_sample = nat.DeviceBuffer(self.sample_offsets.nbytes).upload(self.sample_offsets)
                self.d_frame = nat.DeviceBuffer(self.frame_offsets.nbytes).upload(self.frame_offsets)
        self.total_frames = int(self.frame_offsets[-1])
        self.total_samples = int(self.sample_offsets[-1])

    @property
    def p_sample(self):
        return self.d_sample.ptr if self.d_sample is not None else None

    @property
    def p_frame(self):
        return self.d_frame.ptr if self.d_frame is not None else None

    def vad_handle(self, L, S):
        """dsp_layout of this (ragged, long-lived) layout for (L, S) framing: the VAD kernel's index tables, built once
        instead of on every call (include/dsp_frontend.h: dsp_layout_create).  None for dense or scratch layouts."""
        if self.uniform_samples > 0 or not isinstance(self.d_frame, nat.DeviceBuffer):
            return None
        handles = self.__dict__.setdefault('_vad_handles', {})    # one per framing: the tables depend on (L, S)
        h = handles.get((int(L), int(S)))
        if h is None:
            import ctypes as C
            out = C.c_void_p(0)
            nat.check(nat.load().dsp_layout_create(self.d_frame.ptr, self.n_utt, self.total_frames, int(L), int(S), None,
                                                   C.byref(out)))
            nat.check(nat.load().dsp_stream_synchronize(None))
            h = handles[(int(L), int(S))] = _LayoutHandle(out.value)
        return h.handle


class _LayoutHandle:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        if self.handle:
            try:
                nat.load().dsp_layout_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


def _layout_for(L, S, waves, sample_offsets):
    shape = tuple(waves.shape)
    if sample_offsets is None:
        if len(shape) == 1:
            return _BatchLayout(L, S, 1, uniform_samples=shape[0])
        if len(shape) != 2:
            raise ValueError('waves must be [B, N] or 1-D with sample_offsets')
        return _BatchLayout(L, S, shape[0], uniform_samples=shape[1])
    so = np.asarray(sample_offsets, dtype=np.int64)
    lens = np.diff(so)
    if len(lens) and np.all(lens == lens[0]) and lens[0] > 0 and so[0] == 0:
        return _BatchLayout(L, S, len(lens), uniform_samples=int(lens[0]))
    return _BatchLayout(L, S, len(so) - 1, sample_offsets=so)


def _wave_dtype_of(x):
    name = str(x.dtype).replace('torch.', '')
    if name == 'int16':
        return nat.WAVE_I16
    if name == 'float32':
        return nat.WAVE_F32
    raise TypeError(f'device waveforms must be float32 or int16, got {x.dtype}')


class FeaturePlan:
    """One MFCC configuration (arguments of base.mfcc, base.py:8-10) bound to device tables.

    ``mfcc_batch`` returns ``[sum T_b, numcep]`` (or ``[sum T_b, 3*numcep]`` = mfcc | delta |
    delta-delta when ``delta_n >= 1``, base.py:70-79 applied twice) plus the frame offsets.
    """

    def __init__(self, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512,
                 lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True,
                 winfunc=_ones):
        self.plan = _plan.mfcc_plan(samplerate, winlen, winstep, numcep, nfilt, nfft, lowfreq,
                                    highfreq, preemph, ceplifter, appendEnergy, winfunc)
        self.L, self.S, self.C = self.plan.L, self.plan.S, self.plan.C

    def layout(self, waves, sample_offsets=None):
        return _layout_for(self.L, self.S, waves, sample_offsets)

    def width(self, delta_n=0):
        return self.C * (3 if delta_n >= 1 else 1)

    def run_raw(self, d_wave, wave_dtype, layout, d_out, delta_n=0, stream=None):
        """Launch on raw device pointers (ints).  Asynchronous w.r.t. the host."""
        lib = nat.load()
        st = _stream_ptr(stream)
        if delta_n >= 1:
            nat.check(lib.dsp_mfcc_delta_batch(self.plan.handle, d_wave, wave_dtype, layout.p_sample,
                                               layout.p_frame, layout.n_utt, layout.total_frames,
                                               layout.uniform_samples, int(delta_n), d_out, st))
        else:
            nat.check(lib.dsp_features_batch(self.plan.handle, d_wave, wave_dtype, layout.p_sample,
                                             layout.p_frame, layout.n_utt, layout.total_frames,
                                             layout.uniform_samples, nat.OUT_MFCC, d_out, 0, None, st))

    def mfcc_batch(self, waves, sample_offsets=None, delta_n=0, out=None, stream=None, layout=None):
        """waves: [B, N] (uniform) or 1-D concatenation with ``sample_offsets[B+1]``.
        Returns (features, frame_offsets)."""
        nat.require_device()
        if layout is None:
            layout = self.layout(waves, sample_offsets)
        D = self.width(delta_n)
        if _is_device_tensor(waves):
            import torch
            _check_device(waves)
            if not waves.is_contiguous():
                waves = waves.contiguous()
            if out is None:
                out = torch.empty((layout.total_frames, D), dtype=torch.float32, device=waves.device)
            if stream is None:
                stream = torch.cuda.current_stream(waves.device)
            self.run_raw(waves.data_ptr(), _wave_dtype_of(waves), layout, out.data_ptr(), delta_n, stream)
            return out, layout.frame_offsets
        wave, dtype = nat.as_wave(np.asarray(waves).reshape(-1))
        d_wave = nat.device_array('batch_wave', wave)
        d_out = nat.SCRATCH.get('batch_out', layout.total_frames * D * 4)
        self.run_raw(d_wave.ptr, dtype, layout, d_out.ptr, delta_n, None)
        return d_out.download((layout.total_frames, D), np.float32), layout.frame_offsets


class EndpointPlan:
    """Batched endpoint.basic_endpoint_detection (endpoint.py:34-66) at a fixed (rate, cfg.frame,
    cfg.step): per-frame amplitude + ZCR and the threshold state machine, all on the device.
    robust=True: endpoint.robust_endpoint_detection (endpoint.py:68-92) -- the autocorrelation gate of every
    frame (dsp_acr_gate_batch) is one more launch and the state machine runs in its gated, single-pass form."""

    def __init__(self, rate=16000, frame=0.03, step=0.01, robust=False):
        self.rate, self.frame, self.step = rate, frame, step
        self.L, self.S = int(rate * frame), int(step * rate)  # sigproc.py:19 truncation
        self.robust = bool(robust)
        if self.robust and rate // 50 > self.L:
            raise ValueError('robust endpointing: the gate\'s lags (rate // 50) reach past the frame')

    def layout(self, waves, sample_offsets=None):
        return _layout_for(self.L, self.S, waves, sample_offsets)

    def run_raw(self, d_wave, wave_dtype, layout, d_amp_sum, d_zcr, d_endpoints, stream=None):
        lib = nat.load()
        st = _stream_ptr(stream)
        handle = layout.vad_handle(self.L, self.S) if hasattr(layout, 'vad_handle') else None
        if handle is not None:      # long-lived ragged layout: its index tables were built once
            nat.check(lib.dsp_vad_features_layout_batch(handle, d_wave, wave_dtype, layout.p_sample, layout.p_frame, 0,
                                                        d_amp_sum, d_zcr, st))
        else:
            nat.check(lib.dsp_vad_features_batch(d_wave, wave_dtype, layout.p_sample, layout.p_frame,
                                                 layout.n_utt, layout.total_frames, layout.uniform_samples,
                                                 self.L, self.S, 0, d_amp_sum, d_zcr, st))
        if layout.d_frame is None:  # the rule kernel always takes explicit frame offsets
            layout.d_frame = nat.DeviceBuffer(layout.frame_offsets.nbytes).upload(layout.frame_offsets)
        if self.robust:
            d_voiced = nat.SCRATCH.get('ep_voiced', max(1, layout.total_frames))
            nat.check(lib.dsp_acr_gate_batch(d_wave, wave_dtype, layout.p_sample, layout.p_frame, layout.n_utt,
                                             layout.total_frames, layout.uniform_samples, self.L, self.S,
                                             int(self.rate) // 500, int(self.rate) // 50, 0.55, d_voiced.ptr, st))
            nat.check(lib.dsp_endpoint_rule_acr_batch(d_amp_sum, d_zcr, d_voiced.ptr, layout.d_frame.ptr, layout.n_utt, self.L,
                                                      float(self.frame), float(self.step), d_endpoints, st))
            return
        nat.check(lib.dsp_endpoint_rule_batch(d_amp_sum, d_zcr, layout.d_frame.ptr, layout.n_utt, self.L,
                                              float(self.frame), float(self.step), d_endpoints, st))

    def detect_batch(self, waves, sample_offsets=None, return_feature=False, layout=None):
        """-> int64 [B, 2] sample indices (left, right) per utterance, computed exactly as
        int(frame_index * cfg.step * rate) (endpoint.py:64); optionally amp/zcr per frame."""
        nat.require_device()
        if layout is None:
            layout = self.layout(waves, sample_offsets)
        nf, B = layout.total_frames, layout.n_utt
        d_amp = nat.SCRATCH.get('ep_amp', nf * 8)
        d_zcr = nat.SCRATCH.get('ep_zcr', nf * 4)
        d_ep = nat.SCRATCH.get('ep_batch', B * 8)
        if _is_device_tensor(waves):
            import torch
            _check_device(waves)
            if not waves.is_contiguous():
                waves = waves.contiguous()
            st = torch.cuda.current_stream(waves.device)
            self.run_raw(waves.data_ptr(), _wave_dtype_of(waves), layout, d_amp.ptr, d_zcr.ptr, d_ep.ptr, st)
            nat.check(nat.load().dsp_stream_synchronize(_stream_ptr(st)))
        else:
            wave, dtype = nat.as_wave(np.asarray(waves).reshape(-1))
            d_wave = nat.device_array('batch_wave', wave)
            self.run_raw(d_wave.ptr, dtype, layout, d_amp.ptr, d_zcr.ptr, d_ep.ptr, None)
        frames = d_ep.download((B, 2), np.int32)
        # fp64 product order of endpoint.py:64: (idx * step) * rate, then int() -- vectorised
        samples = ((frames.astype(np.float64) * self.step) * self.rate).astype(np.int64)
        if return_feature:
            amp = d_amp.download((nf,), np.float64) / self.L
            zcr = d_zcr.download((nf,), np.int32)
            return samples, amp, zcr, layout.frame_offsets
        return samples
