"""Drop-in for the pitch-score path of the reference's ``features/pitch.py``:
``pitch_detect_sr`` as ``model.py:92`` calls it when ``cfg.use_pitch`` is set.

Per frame of the 10 kHz signal the reference clips at the median (pitch.py:145-155), band-passes with
a complex FIR built from an ideal band (sigproc.py:22-46), takes the magnitude and evaluates 180
autocorrelation lags (pitch.py:112-132) -- ~80 k multiply-adds per frame in Python loops.  Here that
is one kernel launch for all frames (``dsp_pitch_scores_batch``); the sequential O(T * 180) tail
(in-place smoothing, arg-max, octave repair) stays host logic, as in the reference.  The SVM /
plotting side of the reference's module (``pitch_feature``, ``pitch_model.py``) is out of scope.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from .preprocess import downsampling
from .sigproc import to_frames

MIN_SHIFT, MAX_SHIFT = 20, 200        # pitch.py:126-127: 50 .. 500 Hz on the 10 kHz lag grid
_taps_cache = {}


def bandpass_taps(N, rate, low_freq=0, high_freq=500, wintype='square'):
    """The complex FIR of sigproc.window (sigproc.py:35-45), fp64."""
    Hd = np.zeros(N)
    Hd[int(N * low_freq / rate):int(N * high_freq / rate)] = 1
    w = np.hamming(N) if wintype == 'hamming' else np.ones(N)
    return 2 * np.pi * w * np.fft.ifft(Hd, N)


def window(sig, rate, low_freq=0, high_freq=500, wintype='square'):
    """sigproc.window (sigproc.py:22-46): API-surface helper on the host; the scoring kernel applies
    the same filter on the device."""
    sig = np.asarray(sig, dtype=np.float64)
    return np.convolve(sig, bandpass_taps(len(sig), rate, low_freq, high_freq, wintype))[:len(sig)]


def center_clip(frame, binary=True):
    """pitch.py:145-155 (API-surface helper; fused into the scoring kernel on the device)."""
    frame = np.asarray(frame, dtype=np.float64)
    pos = frame[frame >= 0]
    med = np.median(pos) if len(pos) else np.nan
    up, dn = frame > med, frame < -med
    if binary:
        return np.where(up, 1, np.where(dn, -1, 0))
    return np.where(up, frame - med, np.where(dn, frame + med, 0.0))


def _device_taps(L, rate):
    key = (nat.current_device(), int(L), int(rate))
    buf = _taps_cache.get(key)
    if buf is None:
        h = bandpass_taps(L, rate, 50, 900, 'hamming')
        arr = np.stack([h.real, h.imag], axis=1).astype(np.float32)
        buf = nat.DeviceBuffer(arr.nbytes).upload(arr)
        _taps_cache[key] = buf
    return buf


def frame_scores_batch(sig10k, sample_offsets, L, S, rate=10000, clip=True):
    """[sum T_b, 180] fp32 scores for concatenated 10 kHz signals; returns (scores, frame_offsets)."""
    nat.require_device()
    lib = nat.load()
    so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
    fo = nat.frame_offsets(so, L, S)
    x = np.ascontiguousarray(sig10k, dtype=np.float32).reshape(-1)
    d_x = nat.device_array('pitch_sig', x if x.size else np.zeros(1, dtype=np.float32))
    d_so = nat.device_array('pitch_so', so)
    d_fo = nat.device_array('pitch_fo', fo)
    n_lags = MAX_SHIFT - MIN_SHIFT
    d_out = nat.SCRATCH.get('pitch_scores', int(fo[-1]) * n_lags * 4)
    nat.check(lib.dsp_pitch_scores_batch(d_x.ptr, d_so.ptr, d_fo.ptr, len(so) - 1, int(fo[-1]), 0, int(L), int(S),
                                         _device_taps(L, rate).ptr, 1 if clip else 0, MIN_SHIFT, MAX_SHIFT,
                                         d_out.ptr, None))
    return d_out.download((int(fo[-1]), n_lags), np.float32).astype(np.float64), fo


def pitch_detect_frame_sr(frame, rate):
    """pitch.py:112-132 for one (already clipped) frame -> list of 180 scores."""
    frame = np.asarray(frame, dtype=np.float64).reshape(-1)
    scores, _ = frame_scores_batch(frame, [0, len(frame)], len(frame), len(frame), rate=rate, clip=False)
    return list(scores[0])


def _rows_on_device(g, flags, bias=MIN_SHIFT, degree=2):
    """Score rows [T, n_lags] (fp64) through dsp_pitch_rows_batch as a batch of one -> (rows after the call, pitch)."""
    nat.require_device()
    lib = nat.load()
    rows = np.ascontiguousarray(np.asarray(g, dtype=np.float64))
    if rows.ndim != 2:
        rows = rows.reshape(len(rows), -1)
    T, n = rows.shape
    if T == 0 or n == 0:
        return rows, np.zeros(0)
    d_rows = nat.device_array('pitch_rows', rows)
    d_fo = nat.device_array('pitch_rows_fo', np.array([0, T], dtype=np.int64))
    d_pitch = nat.SCRATCH.get('pitch_rows_out', T * 8)
    nat.check(lib.dsp_pitch_rows_batch(d_rows.ptr, d_fo.ptr, 1, n, int(bias), int(degree), int(flags), d_pitch.ptr, None))
    return (d_rows.download((T, n), np.float64) if flags & 1 else rows), (d_pitch.download((T,), np.float64) if flags & 2 else None)


def smooth(g, degree=2):
    """pitch.py:157-164 (in-place running mean over rows [i - degree, i + degree), the rows before i already
    smoothed), on the device."""
    rows, _ = _rows_on_device(g, 1, degree=degree)
    return rows.tolist()


def max_pitch(g, bias=20):
    """pitch.py:166-172: 1 / (1e-4 (bias + first arg-max)) per row, on the device."""
    return list(_rows_on_device(g, 2, bias=bias)[1])


def robust_max_pitch(g, bias=20):
    """pitch.py:191-206: max_pitch, then the two octave-repair sweeps, on the device."""
    return list(_rows_on_device(g, 2 | 4, bias=bias)[1])


def pitch_tracks_batch(sig10k, sample_offsets, L, S, rate=10000):
    """pitch.pitch_detect_sr's whole per-utterance chain for concatenated 10 kHz signals, on the device: frame scores
    (dsp_pitch_scores_batch), then smoothing in place, arg-max and the two octave-repair sweeps
    (dsp_pitch_track_batch; pitch.py:157-206).  Returns (pitch [sum T_b] in Hz, fp64, frame_offsets)."""
    nat.require_device()
    lib = nat.load()
    so = np.ascontiguousarray(sample_offsets, dtype=np.int64)
    fo = nat.frame_offsets(so, L, S)
    x = np.ascontiguousarray(sig10k, dtype=np.float32).reshape(-1)
    d_x = nat.device_array('pitch_sig', x if x.size else np.zeros(1, dtype=np.float32))
    d_so = nat.device_array('pitch_so', so)
    d_fo = nat.device_array('pitch_fo', fo)
    n_lags = MAX_SHIFT - MIN_SHIFT
    d_scores = nat.SCRATCH.get('pitch_scores', int(fo[-1]) * n_lags * 4)
    d_pitch = nat.SCRATCH.get('pitch_track', int(fo[-1]) * 8)
    nat.check(lib.dsp_pitch_scores_batch(d_x.ptr, d_so.ptr, d_fo.ptr, len(so) - 1, int(fo[-1]), 0, int(L), int(S),
                                         _device_taps(L, rate).ptr, 1, MIN_SHIFT, MAX_SHIFT, d_scores.ptr, None))
    nat.check(lib.dsp_pitch_track_batch(d_scores.ptr, d_fo.ptr, len(so) - 1, n_lags, MIN_SHIFT, 2, d_pitch.ptr, None))
    return d_pitch.download((int(fo[-1]),), np.float64), fo


def pitch_tracks_device(d_clips, d_src_off, n_utt, n_samples_bound, rate, L, S, stream=None):
    """pitch.pitch_detect_sr for clips that are already on the device (fp32, concatenated, `rate` Hz): decimation to
    10 kHz (dsp_resample_layout_batch + dsp_decimate_batch), scores, smoothing, arg-max and octave repair, nothing
    leaves the device.  Returns (d_pitch [fp64, one per frame], d_frame_off [B+1]) as library scratch buffers."""
    lib = nat.load()
    if rate > 10000:
        d_so10 = nat.SCRATCH.get('pitch_so10', (n_utt + 1) * 8)
        d_fo = nat.SCRATCH.get('pitch_fo10', (n_utt + 1) * 8)
        d_x10 = nat.SCRATCH.get('pitch_x10', max(4, int(n_samples_bound) * 4))
        nat.check(lib.dsp_resample_layout_batch(d_src_off, n_utt, int(rate), 10000, int(L), int(S), d_so10.ptr, d_fo.ptr, stream))
        nat.check(lib.dsp_decimate_batch(d_clips, d_src_off, d_so10.ptr, n_utt, int(n_samples_bound), int(rate), 10000, d_x10.ptr, stream))
        p_x, p_so = d_x10.ptr, d_so10.ptr
    else:                       # downsampling keeps every sample when the clip is at 10 kHz or below (preprocess.py:21-28)
        d_fo = nat.SCRATCH.get('pitch_fo10', (n_utt + 1) * 8)
        nat.check(lib.dsp_resample_layout_batch(d_src_off, n_utt, int(rate), 0, int(L), int(S), None, d_fo.ptr, stream))
        p_x, p_so = d_clips, d_src_off
    frames_bound = int(n_samples_bound) // int(S) + n_utt + 1
    n_lags = MAX_SHIFT - MIN_SHIFT
    d_scores = nat.SCRATCH.get('pitch_scores', frames_bound * n_lags * 4)
    d_pitch = nat.SCRATCH.get('pitch_track', frames_bound * 8)
    nat.check(lib.dsp_pitch_scores_batch(p_x, p_so, d_fo.ptr, n_utt, frames_bound, 0, int(L), int(S), _device_taps(L, 10000).ptr, 1,
                                         MIN_SHIFT, MAX_SHIFT, d_scores.ptr, stream))
    nat.check(lib.dsp_pitch_track_batch(d_scores.ptr, d_fo.ptr, n_utt, n_lags, MIN_SHIFT, 2, d_pitch.ptr, stream))
    return d_pitch, d_fo


def pitch_detect_sr(sig, rate, winlen=0.0512, step=0.01):
    """pitch.py:96-110 -> (pitch per frame in Hz, frames of the 10 kHz signal)."""
    s = downsampling(np.asarray(sig).reshape(-1), rate, 10000)
    L, S = int(10000 * winlen), int(step * 10000)          # to_frames truncates (sigproc.py:19)
    pitch, _ = pitch_tracks_batch(s, [0, len(s)], L, S)
    frames = to_frames(s, 10000, winlen, step)
    return list(pitch), frames
