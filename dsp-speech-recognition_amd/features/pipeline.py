"""BASELINE.json configs[3]: energy / ZCR endpointing feeding the feature extractor, batched, with
variable-length outputs -- the device-side form of model.py:113-121 (endpoint_detect -> scale ->
feature_extract_mfcc).

    waveforms (ragged, int16 or fp32)
      -> per-frame amplitude + ZCR, two-threshold rule        (dsp_vad_features_batch, dsp_endpoint_rule_batch)
      -> [left, right) in samples, trimmed-batch offsets       (dsp_endpoint_layout_batch, on the device)
      -> MFCC + delta + delta-delta of sig[left:right] read IN PLACE; unit variance = a shift of c0 from fp64 sums
         the feature kernel accumulates while staging            (dsp_mfcc_delta_segments_batch)
         [plans / buffers that path does not serve, and delta_n = 0: trimmed (optionally unit-variance) fp32 copy
          (dsp_trim_scale_batch) -> dsp_mfcc_delta_batch / dsp_features_batch]

Nothing comes back to the host in the middle: the frame-index -> sample-index conversion
(endpoint.py:64), the clipping and both prefix sums run in one small kernel, and the feature kernels
are sized by an upper bound (the frame count of the untrimmed clips) while the true offsets stay in
device tables.  With a device tensor as input and ``download=False`` the whole call is a sequence of
asynchronous launches on the caller's stream -- no allocation, copy or synchronisation once the layout
of the batch (its sample offsets) has been prepared.

The endpoint rule must have seen a whole utterance before its first trimmed frame can be cut, so
the two framings (cfg.frame for VAD, winlen for MFCC) cannot share one pass; what is shared is the
device-resident waveform.
"""
from __future__ import annotations

import threading

import numpy as np

from . import _native as nat
from .batch import EndpointPlan, FeaturePlan, _BatchLayout, _is_device_tensor, _stream_ptr, _wave_dtype_of


class _Borrowed:
    """Device memory owned by the caller (a tensor); only the address is kept."""

    def __init__(self, ptr):
        self.ptr = int(ptr)


class _TensorBuffer:
    """Result rows held in a torch tensor (device-tensor inputs: torch's caching allocator instead of a
    hipMalloc / hipFree pair per call); same surface as DeviceBuffer where the callers need it."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()
        self.nbytes = tensor.numel() * tensor.element_size()

    def download(self, shape, dtype, stream=None):
        n = int(np.prod(shape))
        return self.tensor.reshape(-1)[:n].cpu().numpy().astype(dtype, copy=False).reshape(shape)


class _DeviceLayout:
    """What FeaturePlan.run_raw needs, with the offsets living only on the device."""

    def __init__(self, n_utt, d_sample, d_frame, frames_bound):
        self.n_utt, self.d_sample, self.d_frame = n_utt, d_sample, d_frame
        self.total_frames = int(frames_bound)        # upper bound; kernels read the truth from d_frame
        self.uniform_samples = 0

    @property
    def p_sample(self):
        return self.d_sample.ptr

    @property
    def p_frame(self):
        return self.d_frame.ptr


class PipelineLayout:
    """Everything about a batch that depends only on its sample offsets: device copies of the offset
    tables for the VAD framing, the frame-count bound of the feature stage, and the device buffers the
    stages hand to each other.  Build once per batch shape with ``VadMfccPipeline.prepare`` and reuse:
    ``run`` then allocates nothing."""

    def __init__(self, pipe, sample_offsets, delta_n):
        so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        self.sample_offsets = so
        self.n_utt = B = len(so) - 1
        ep, fp = pipe.endpoint, pipe.features
        self.vad = _BatchLayout(ep.L, ep.S, B, sample_offsets=so)
        # trimming never lengthens a clip, and the frame count is monotone in the length
        self.frames_bound = int(nat.frame_offsets(so, fp.L, fp.S)[-1])
        self.total_samples = int(so[-1])
        self.delta_n = int(delta_n)
        self.D = fp.width(delta_n)
        nf = self.vad.total_frames
        self.d_amp = nat.DeviceBuffer(nf * 8)
        self.d_zcr = nat.DeviceBuffer(nf * 4)
        self.d_ep = nat.DeviceBuffer(B * 8)
        self.d_seg = nat.DeviceBuffer(B * 16)
        self.d_dst_off = nat.DeviceBuffer((B + 1) * 8)
        self.d_frame_off = nat.DeviceBuffer((B + 1) * 8)
        self._d_trim = None                      # fp32 copy of the trimmed clips: only the fallback path needs it
        self.features = _DeviceLayout(B, self.d_dst_off, self.d_frame_off, self.frames_bound)
        # scratch of the in-place path (dsp_mfcc_delta_segments_batch): tables, statistics, dense cepstra
        self.d_work = None
        self.c0_shift_pending = False          # set by launch(): cepstra-only output read in place, unit variance not yet applied
        if delta_n >= 0:
            import ctypes as C
            nbytes = C.c_size_t(0)
            nat.check(nat.load().dsp_segments_workspace_bytes(fp.plan.handle, B, max(self.frames_bound, 1), C.byref(nbytes)))
            self.d_work = nat.DeviceBuffer(int(nbytes.value))

    @property
    def d_trim(self):
        if self._d_trim is None:
            self._d_trim = nat.DeviceBuffer(max(self.total_samples, 1) * 4)
        return self._d_trim


class VadMfccPipeline:
    def __init__(self, rate=16000, frame=0.03, step=0.01, unit_variance=False, **mfcc_kwargs):
        """mfcc_kwargs: arguments of base.mfcc (winlen, winstep, numcep, nfilt, nfft, preemph, ...).
        unit_variance=True reproduces model.py:63 (divide the trimmed clip by its population std)."""
        mfcc_kwargs.setdefault('samplerate', rate)
        self.rate = rate
        self.unit_variance = bool(unit_variance)
        self.copy_trimmed = False       # True: always go through the trimmed fp32 copy (A/B and tests)
        self.endpoint = EndpointPlan(rate, frame, step)
        self.features = FeaturePlan(**mfcc_kwargs)
        self._tls = threading.local()   # per thread: last few batch shapes seen by run()

    def prepare(self, sample_offsets, delta_n=2):
        nat.require_device()
        return PipelineLayout(self, sample_offsets, delta_n)

    def _cached_layout(self, sample_offsets, delta_n):
        """run(download=True) without an explicit layout: batches of a shape this thread has seen recently reuse
        their device tables and stage buffers (a training loop cycles through few batch shapes; building a
        layout allocates and uploads, which costs more than the kernels).  Results that stay on the device
        (download=False) never share a cached layout: their offsets live in the layout they are returned with."""
        cache = getattr(self._tls, 'layouts', None)
        if cache is None:
            cache = self._tls.layouts = {}
        so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        key = (nat.current_device(), int(delta_n), so.tobytes())
        lay = cache.pop(key, None)
        if lay is None:
            lay = PipelineLayout(self, so, delta_n)
            while len(cache) >= 4:
                cache.pop(next(iter(cache)))
        cache[key] = lay               # most recently used last
        return lay

    def launch(self, d_wave, wave_dtype, lay, d_out, stream=None, d_jitter=None, defer_c0_shift=False):
        """Queue the whole pipeline on `stream` (raw pointers, no host synchronisation, no allocation):
        features land in `d_out` ([lay.frames_bound, D] fp32, rows packed by lay.d_frame_off).
        ``defer_c0_shift`` (cepstra only, unit variance): the caller applies the -ln(var) of c0 itself
        (dsp_model_finalize_segments_batch does, from the statistics in lay.d_work) -- then the clips are read in
        place for delta_n = 0 as well and ``lay.c0_shift_pending`` says so; otherwise cepstra-only unit-variance
        output takes the trimmed, scaled copy."""
        lib = nat.load()
        st = _stream_ptr(stream)
        ep, fp = self.endpoint, self.features
        ep.run_raw(d_wave, wave_dtype, lay.vad, lay.d_amp.ptr, lay.d_zcr.ptr, lay.d_ep.ptr, st)
        lay.c0_shift_pending = False
        in_place = lay.d_work is not None and not self.copy_trimmed and \
            (lay.delta_n >= 1 or not self.unit_variance or defer_c0_shift)
        if in_place:      # one launch: segments, offsets AND the tables / zeroed statistics of the in-place feature stage
            nat.check(lib.dsp_endpoint_layout_segments_batch(lay.d_ep.ptr, lay.vad.p_sample, lay.n_utt, float(ep.step),
                                                             float(ep.rate), d_jitter, lay.d_seg.ptr, lay.d_dst_off.ptr,
                                                             lay.d_frame_off.ptr, fp.plan.handle, max(lay.frames_bound, 1),
                                                             lay.d_work.ptr, lay.d_work.nbytes, st))
        else:
            nat.check(lib.dsp_endpoint_layout_batch(lay.d_ep.ptr, lay.vad.p_sample, lay.n_utt, float(ep.step),
                                                    float(ep.rate), fp.L, fp.S, d_jitter, lay.d_seg.ptr,
                                                    lay.d_dst_off.ptr, lay.d_frame_off.ptr, st))
        if in_place:
            # the feature kernel reads sig[left:right] where it lies; unit variance becomes a shift of c0
            rc = lib.dsp_mfcc_delta_segments_batch(fp.plan.handle, d_wave, wave_dtype, lay.vad.p_sample, lay.d_seg.ptr,
                                                   lay.d_frame_off.ptr, lay.n_utt, max(lay.frames_bound, 1), lay.delta_n,
                                                   (1 if self.unit_variance else 0) | 2, lay.d_work.ptr,
                                                   lay.d_work.nbytes, d_out, st)
            if rc == nat.OK:
                lay.c0_shift_pending = bool(lay.delta_n == 0 and self.unit_variance)
                return
            if rc != 1:                      # 1 = "not served in place": take the copy below
                nat.check(rc)
        nat.check(lib.dsp_trim_scale_batch(d_wave, wave_dtype, lay.vad.p_sample, lay.d_seg.ptr, lay.d_dst_off.ptr,
                                           lay.n_utt, 1 if self.unit_variance else 0, lay.d_trim.ptr, st))
        fp.run_raw(lay.d_trim.ptr, nat.WAVE_F32, lay.features, d_out, lay.delta_n, st)

    def run(self, waves, sample_offsets=None, delta_n=2, download=True, layout=None, jitter=None, defer_c0_shift=False):
        """waves: 1-D host array (int16 or float) or torch-ROCm tensor (int16 / float32) of concatenated
        utterances; ``layout`` = a PipelineLayout from ``prepare`` (then sample_offsets and delta_n are taken
        from it).
        ``jitter``: optional int [B, 2] sample offsets added to (left, right) before trimming
        (model.py:54-60 draws them as -randint(0, 0.1 rate), +randint(0, 0.1 rate)).
        Returns (features [sum T_b, D] fp32, frame_offsets [B+1], endpoints [B, 2] in samples);
        with ``download=False`` nothing is copied or synchronised and the result is
        ((DeviceBuffer of [frames_bound, D] rows, PipelineLayout), None, None): the true frame offsets
        and endpoints are in layout.d_frame_off / layout.d_seg on the device."""
        nat.require_device()
        if layout is not None:
            lay = layout
        elif download:
            lay = self._cached_layout(sample_offsets, delta_n)
        else:
            lay = self.prepare(sample_offsets, delta_n)
        stream = None
        if _is_device_tensor(waves):                # torch-ROCm tensor: nothing crosses PCIe
            import torch
            if not waves.is_contiguous():
                waves = waves.contiguous()
            if waves.device.index != nat.current_device():
                raise nat.DspError(f'waveforms live on cuda:{waves.device.index}, the library is on device '
                                   f'{nat.current_device()} (dsp_set_device)')
            dtype = _wave_dtype_of(waves)
            d_wave = _Borrowed(waves.data_ptr())
            stream = torch.cuda.current_stream(waves.device)
        else:
            wave, dtype = nat.as_wave(np.asarray(waves).reshape(-1))
            d_wave = nat.device_array('batch_wave', wave)
        d_jit = None
        if jitter is not None:
            j = np.ascontiguousarray(jitter, dtype=np.int64).reshape(lay.n_utt, 2)
            d_jit = nat.device_array('pipe_jitter', j, _stream_ptr(stream)).ptr
        if _is_device_tensor(waves):
            d_out = _TensorBuffer(torch.empty(max(lay.frames_bound, 1) * lay.D, dtype=torch.float32,
                                              device=waves.device))
        else:
            d_out = nat.DeviceBuffer(max(lay.frames_bound, 1) * lay.D * 4)  # owned by the result
        self._last_wave = (d_wave.ptr, dtype)      # (ModelFeatureBatch's optional streams trim the same buffer again)
        self.launch(d_wave.ptr, dtype, lay, d_out.ptr, stream, d_jit, defer_c0_shift=defer_c0_shift and not download)
        if not download:
            return (d_out, lay), None, None
        st = _stream_ptr(stream)
        fo = lay.d_frame_off.download((lay.n_utt + 1,), np.int64, st)
        seg = lay.d_seg.download((lay.n_utt, 2), np.int64, st)
        out = d_out.download((int(fo[-1]), lay.D), np.float32, st)
        return out, fo, seg
