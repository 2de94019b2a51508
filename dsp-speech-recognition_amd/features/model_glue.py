"""The per-utterance glue model.py wraps around the feature path ("next" row f-1 of SURVEY 8f):
endpoint trim -> unit-variance scaling -> MFCC (with the (1, N) no-pre-emphasis quirk) -> global
mean removal -> delta(3) / delta-delta(3) -> per-coefficient z-score -> pad to 200 frames.

Round-1 form: composed from the GPU entry points of this package plus O(N) host reductions
(std of the trimmed clip, mean/std of the [T, 13] matrix).  Citations are file:line of the
reference's model.py.
"""
from __future__ import annotations

import numpy as np

from .base import delta, mfcc
from . import endpoint as _endpoint


def endpoint_detect(sig, rate, augment=False, rng=None):
    """model.py:52-64: trim to the detected endpoints (with augment=True widened by two draws of
    randint(0, int(0.1 * rate)) from ``rng`` -- a ``random.Random``, default the global generator the
    reference uses) and divide by the population standard deviation (sklearn
    ``scale(with_mean=False)``; a zero std divides by 1)."""
    left, right = _endpoint.basic_endpoint_detection(sig, rate)
    if augment:
        import random
        gen = rng if rng is not None else random
        hi = int(0.1 * rate)
        s_l, s_r = gen.randint(0, hi), gen.randint(0, hi)
        left, right = max(left - s_l, 0), max(right + s_r, 0)
    clip = np.asarray(sig[left:right], dtype=np.float64).reshape(-1, 1)
    sd = clip.std(axis=0)
    sd[sd == 0.0] = 1.0
    return clip / sd


def feature_extract_mfcc(sound, rate, nfft=1536):
    """model.py:66-88 -> ((mfcc0, mfcc1, mfcc2), min(T, 200))."""
    cfg = _endpoint.cfg
    m0 = mfcc(np.asarray(sound).reshape(1, -1), rate, winlen=cfg.frame, winstep=cfg.step, nfft=nfft,
              winfunc=np.hamming)
    m0 = m0 - np.mean(m0)
    m1 = delta(m0, 3)
    m2 = delta(m1, 3)
    mu, sd = m0.mean(axis=0), m0.std(axis=0)
    sd[sd == 0.0] = 1.0
    m0 = (m0 - mu) / sd
    return (m0, m1, m2), min(len(m0), 200)


def deviation(arr, smooth=1):
    """model.py:29-33: first difference at distance ``smooth``."""
    arr = np.asarray(arr, dtype=np.float64)
    return arr[smooth:] - arr[:len(arr) - smooth]


def feature_extract_pitch(sound, rate):
    """model.py:90-95, the optional pitch stream (cfg.use_pitch): pitch track / 150 and its
    first difference, both [T, 1]."""
    from .pitch import pitch_detect_sr
    cfg = _endpoint.cfg
    pitch0, _ = pitch_detect_sr(np.asarray(sound).reshape(-1), rate, winlen=cfg.frame, step=cfg.step)
    pitch0 = np.array(pitch0).reshape(-1, 1) / 150
    return [pitch0.reshape(-1, 1), deviation(pitch0).reshape(-1, 1)]


def feature_extract_timespace(sound, rate):
    """model.py:97-101, the optional amplitude stream (cfg.use_timefeat): z-scored frame amplitude
    (population std, zero -> 1 as sklearn.scale) and its first difference, both [T, 1]."""
    cfg = _endpoint.cfg
    a = np.asarray(_endpoint.amplitude_feature(np.asarray(sound).reshape(-1), rate, winlen=cfg.frame, step=cfg.step),
                   dtype=np.float64)
    sd = a.std()
    amp0 = ((a - a.mean()) / (sd if sd != 0 else 1.0)).reshape(-1, 1)
    return [amp0, deviation(amp0).reshape(-1, 1)]


def pad200(b):
    """model.py:35-39: zero-pad or truncate a [T, D] stream to exactly 200 frames."""
    b = np.asarray(b)
    if len(b) < 200:
        return np.pad(b, ((0, 200 - len(b)), (0, 0)), 'constant', constant_values=0)
    return np.array(b[:200])


def model_pipeline(sig, rate, augment=False, rng=None):
    """Raw int16 clip -> ((mfcc0, mfcc1, mfcc2), n) exactly as RNNModel.get_batch_full feeds the
    classifier per utterance (model.py:113-124; augment=True is the training call of model.py:144)."""
    return feature_extract_mfcc(endpoint_detect(sig, rate, augment=augment, rng=rng), rate)


def model_pipeline_aug(sig, rate, seed):
    """The training path with the jitter drawn from ``random.Random(seed)`` (test hook: what the
    reference computes after ``random.seed(seed)``)."""
    import random
    return model_pipeline(sig, rate, augment=True, rng=random.Random(seed))


def batch_to_rnn_input(features, frame_offsets, max_len=200):
    """[sum T_b, D] device (torch) features + frame offsets -> ([max_len, B, D], len0) exactly as
    model.py:35-50,131-135 lays a batch out for the classifiers (`inp[T, B, 39]`, zero padded or
    truncated to 200 frames).  Stays on the device: no host round trip between the HIP front-end and
    the PyTorch-ROCm RNN."""
    import torch
    fo = torch.as_tensor(np.asarray(frame_offsets), device=features.device)
    B = fo.numel() - 1
    lens = (fo[1:] - fo[:-1]).clamp(max=max_len)
    t = torch.arange(max_len, device=features.device)[:, None]            # [max_len, 1]
    src = (fo[:-1][None, :] + t).clamp(max=features.shape[0] - 1)          # [max_len, B]
    inp = features[src]                                                     # [max_len, B, D]
    inp = inp * (t < lens[None, :]).unsqueeze(-1).to(features.dtype)
    return inp, lens.cpu().numpy()


def model_finalize(mfcc0, frame_offsets, delta_n=3, max_len=200):
    """Host-array form of dsp_model_finalize_batch: ``mfcc0`` [sum T_b, C] (the raw MFCCs of a
    batch) -> (inp [max_len, B, 3C] fp32, len0 [B]) as model.py:75-78 + 35-50 build them."""
    from . import _native as nat
    nat.require_device()
    lib = nat.load()
    m = np.ascontiguousarray(mfcc0, dtype=np.float32)
    fo = np.ascontiguousarray(frame_offsets, dtype=np.int64)
    B, C = len(fo) - 1, m.shape[1]
    d_in = nat.device_array('fin_in', m)
    d_fo = nat.device_array('fin_fo', fo)
    d_out = nat.SCRATCH.get('fin_out', max_len * B * 3 * C * 4)
    d_len = nat.SCRATCH.get('fin_len', B * 4)
    nat.check(lib.dsp_model_finalize_batch(d_in.ptr, C, d_fo.ptr, B, C, int(delta_n), int(max_len), d_out.ptr,
                                           d_len.ptr, None))
    return d_out.download((max_len, B, 3 * C), np.float32), d_len.download((B,), np.int32)


class _ModelFeatureGraph:
    """What ModelFeatureBatch.capture returns: replay() re-runs the captured launches on whatever ``waves`` holds now."""

    def __init__(self, graph, waves, m0, inp, len0, layout):
        self.graph, self.waves, self.m0, self.inp, self.len0, self.layout = graph, waves, m0, inp, len0, layout

    def replay(self):
        self.graph.replay()
        return self.inp, self.len0


class ModelFeatureBatch:
    """Batched, device-resident form of RNNModel.get_batch_full (model.py:113-135): endpointing ->
    (optional endpoint jitter, model.py:54-60) -> trim -> unit variance -> MFCC on the (1, N) view (no
    pre-emphasis, sigproc.py:185) -> minus the utterance's scalar mean -> delta(3), delta(delta, 3) ->
    z-score of the static coefficients -> [200, B, 39] zero padded.  Every step is a kernel behind the C
    ABI, queued on one stream with no host round trip in between (features/pipeline.py)."""

    def __init__(self, rate, frame=0.03, step=0.01, nfft=1536, delta_n=3, max_len=200):
        from .pipeline import VadMfccPipeline
        self.rate = rate
        self.pipe = VadMfccPipeline(rate=rate, frame=frame, step=step, unit_variance=True, winlen=frame,
                                    winstep=step, nfft=nfft, preemph=0.0, winfunc=np.hamming)
        self.delta_n, self.max_len = delta_n, max_len

    def draw_jitter(self, n_utt, rng):
        """The augmentation of model.py:54-60 for a whole batch: (-randint(0, 0.1 rate), +randint(0, 0.1
        rate)) per utterance from ``rng`` (a random.Random, consumed in the reference's order: s_l then
        s_r, utterance by utterance)."""
        hi = int(0.1 * self.rate)
        j = np.empty((n_utt, 2), dtype=np.int64)
        for b in range(n_utt):
            j[b, 0] = -rng.randint(0, hi)
            j[b, 1] = rng.randint(0, hi)
        return j

    def run(self, waves, sample_offsets=None, jitter=None, layout=None, use_pitch=False, use_timefeat=False):
        """-> (inp [max_len, B, 39 (+2) (+2)] torch tensor on the library's device, len0 [B], endpoints [B, 2]).
        ``jitter``: int [B, 2] endpoint offsets (see draw_jitter) for the training path (augment=True);
        None = test path.  ``use_timefeat`` / ``use_pitch`` append the optional streams of model.py:125-128
        in the reference's order (pitch, then amplitude): the amplitude stream is computed on the device
        (dsp_vad_features_batch on the trimmed clips + dsp_model_timefeat_batch); the pitch stream runs the
        per-utterance pitch tracker on the trimmed clips (its smoothing / octave repair is sequential host
        logic), which costs a download and a host loop.  torch only owns the result tensors."""
        import torch
        from . import _native as nat
        from .batch import _is_device_tensor, _stream_ptr
        lib = nat.load()
        dev = waves.device if _is_device_tensor(waves) else torch.device('cuda', nat.current_device())
        stream = torch.cuda.current_stream(dev)
        if layout is None:       # this call consumes the layout's tables before it returns: a cached one is safe
            layout = self.pipe._cached_layout(sample_offsets, 0)
        # the clips are read where they lie (no trimmed fp32 copy): the MFCC kernel accumulates the unit-variance
        # statistics, the finalize kernel applies them to c0 as it reads
        (d_m0, lay), _, _ = self.pipe.run(waves, delta_n=0, download=False, layout=layout, jitter=jitter, defer_c0_shift=True)
        B, C = lay.n_utt, self.pipe.features.C
        inp = torch.empty((self.max_len, B, 3 * C), dtype=torch.float32, device=dev)
        len0 = torch.empty(B, dtype=torch.int32, device=dev)
        # host input: the pipeline ran on the legacy default stream, which orders against `stream`
        st = _stream_ptr(stream) if _is_device_tensor(waves) else None
        if lay.c0_shift_pending:
            nat.check(lib.dsp_model_finalize_segments_batch(d_m0.ptr, C, lay.d_frame_off.ptr, lay.d_seg.ptr, lay.d_work.ptr, B, C,
                                                            self.delta_n, self.max_len, inp.data_ptr(), len0.data_ptr(), st))
        else:
            nat.check(lib.dsp_model_finalize_batch(d_m0.ptr, C, lay.d_frame_off.ptr, B, C, self.delta_n, self.max_len,
                                                   inp.data_ptr(), len0.data_ptr(), st))
        extra = []
        if (use_pitch or use_timefeat) and lay.c0_shift_pending:
            # the optional streams work on the trimmed, scaled clips themselves: make that copy now (model.py:62-63)
            wave_ptr, wave_dtype = self.pipe._last_wave
            nat.check(lib.dsp_trim_scale_batch(wave_ptr, wave_dtype, lay.vad.p_sample, lay.d_seg.ptr, lay.d_dst_off.ptr,
                                               B, 1, lay.d_trim.ptr, st))
        if use_pitch:
            extra.append(self._pitch_streams(lay, st, dev))
        if use_timefeat:
            extra.append(self._timefeat_streams(lay, st, dev))
        seg = lay.d_seg.download((B, 2), np.int64, st)       # first host synchronisation of the default call
        nat.check(lib.dsp_stream_synchronize(st))            # d_m0 is freed on return: its consumer has finished
        if extra:
            inp = torch.cat([inp] + extra, dim=2)
        return inp, len0.cpu().numpy(), seg

    def enqueue(self, d_wave, wave_dtype, lay, d_m0, d_inp, d_len0, stream):
        """The default call (no optional streams, no jitter) as launches only -- raw device pointers, nothing allocated,
        nothing synchronised: endpointing, the layout glue, the feature kernel on the clips in place and the finalize
        kernel.  ``d_m0``: [lay.frames_bound, C] fp32 scratch, ``d_inp``: [max_len, B, 3 C] fp32, ``d_len0``: [B] int32."""
        from . import _native as nat
        from .batch import _stream_ptr
        lib = nat.load()
        st = _stream_ptr(stream)
        C = self.pipe.features.C
        self.pipe.launch(d_wave, wave_dtype, lay, d_m0, stream, None, defer_c0_shift=True)
        if lay.c0_shift_pending:
            nat.check(lib.dsp_model_finalize_segments_batch(d_m0, C, lay.d_frame_off.ptr, lay.d_seg.ptr, lay.d_work.ptr, lay.n_utt, C,
                                                            self.delta_n, self.max_len, d_inp, d_len0, st))
        else:
            nat.check(lib.dsp_model_finalize_batch(d_m0, C, lay.d_frame_off.ptr, lay.n_utt, C, self.delta_n, self.max_len,
                                                   d_inp, d_len0, st))

    def capture(self, waves, layout):
        """A HIP graph of ``enqueue`` over device-resident ``waves`` (torch tensor, int16 / float32) and a prepared layout:
        ``g = mfb.capture(waves, lay)``; put new clips of the same lengths into ``waves`` and call ``g.replay()`` ->
        (inp [max_len, B, 39], len0 [B] int32), both on the device, valid after the current stream's work (no host
        synchronisation).  One eager call runs first (it builds the layout's long-lived index tables)."""
        import torch
        from .batch import _wave_dtype_of
        dev = waves.device
        C, B = self.pipe.features.C, layout.n_utt
        m0 = torch.empty((max(layout.frames_bound, 1), C), dtype=torch.float32, device=dev)
        inp = torch.empty((self.max_len, B, 3 * C), dtype=torch.float32, device=dev)
        len0 = torch.empty(B, dtype=torch.int32, device=dev)
        dtype = _wave_dtype_of(waves)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.enqueue(waves.data_ptr(), dtype, layout, m0.data_ptr(), inp.data_ptr(), len0.data_ptr(), side)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            self.enqueue(waves.data_ptr(), dtype, layout, m0.data_ptr(), inp.data_ptr(), len0.data_ptr(), torch.cuda.current_stream(dev))
        return _ModelFeatureGraph(graph, waves, m0, inp, len0, layout)

    def _timefeat_streams(self, lay, st, dev):
        """[max_len, B, 2]: z-scored frame amplitude of the trimmed, scaled clips and its first difference
        (model.py:97-101), on the device -- at the amplitude stream's OWN framing (int(rate * cfg.frame),
        int(cfg.step * rate): to_frames truncates, sigproc.py:19; the MFCC framing rounds half up, so the two differ at
        e.g. 22.05 kHz): its frame offsets come from one small launch over the trimmed offsets."""
        import torch
        from . import _native as nat
        lib = nat.load()
        ep, fp = self.pipe.endpoint, self.pipe.features
        L2, S2 = int(self.rate * ep.frame), int(ep.step * self.rate)       # to_frames truncation, sigproc.py:19
        B = lay.n_utt
        if (L2, S2) == (fp.L, fp.S):
            p_fo, frames_bound = lay.d_frame_off.ptr, lay.frames_bound
        else:
            d_fo2 = torch.empty(B + 1, dtype=torch.int64, device=dev)
            nat.check(lib.dsp_resample_layout_batch(lay.d_dst_off.ptr, B, int(self.rate), 0, L2, S2, None, d_fo2.data_ptr(), st))
            p_fo, frames_bound = d_fo2.data_ptr(), int(lay.total_samples) // S2 + B + 1
        d_amp = torch.empty(max(frames_bound, 1), dtype=torch.float64, device=dev)
        d_zcr = torch.empty(max(frames_bound, 1), dtype=torch.int32, device=dev)
        nat.check(lib.dsp_vad_features_batch(lay.d_trim.ptr, nat.WAVE_F32, lay.d_dst_off.ptr, p_fo, B,
                                             frames_bound, 0, L2, S2, 0, d_amp.data_ptr(), d_zcr.data_ptr(), st))
        out = torch.empty((self.max_len, B, 2), dtype=torch.float32, device=dev)
        nat.check(lib.dsp_model_timefeat_batch(d_amp.data_ptr(), p_fo, B, L2, self.max_len,
                                               out.data_ptr(), st))
        nat.check(lib.dsp_stream_synchronize(st))            # d_amp / d_zcr / the offsets are released on return
        return out

    def _pitch_streams(self, lay, st, dev):
        """[max_len, B, 2]: pitch track / 150 and its first difference (model.py:90-95) of the trimmed, scaled
        clips, on the device: decimation to 10 kHz (an index selection, preprocess.py:21-28), frame scores, smoothing,
        arg-max, octave repair (pitch.py:96-206) and the [max_len, B, 2] layout -- six launches, no clip and no track
        crosses PCIe."""
        import torch
        from . import _native as nat
        from .pitch import pitch_tracks_device
        lib = nat.load()
        B = lay.n_utt
        cfg = _endpoint.cfg
        L, S = int(10000 * cfg.frame), int(cfg.step * 10000)
        d_pitch, d_fo = pitch_tracks_device(lay.d_trim.ptr, lay.d_dst_off.ptr, B, int(lay.total_samples), self.rate, L, S, st)
        out = torch.empty((self.max_len, B, 2), dtype=torch.float32, device=dev)
        nat.check(lib.dsp_model_pitchfeat_batch(d_pitch.ptr, d_fo.ptr, B, self.max_len, out.data_ptr(), st))
        return out
