"""Drop-in for the energy / zero-crossing part of the reference's ``features/endpoint.py``.

Per-frame short-time amplitude and ZCR come from the HIP library (fp64 accumulation, exact for
int16 PCM); ``basic_endpoint_detection`` also runs the two-threshold state machine on the device
(one utterance per thread, fp64) so single calls and batches share one code path.
``amplitude_rule`` / ``zcr_rule`` keep their list-in / list-out Python form for callers that use
them directly: they are scalar control logic over a few hundred numbers, not arithmetic.

Citations are file:line of the reference.
"""
from __future__ import annotations

import numpy as np

from . import _run
from .sigproc import acr, to_frames


class _Cfg:
    """The two fields of the reference's global config the endpoint path reads
    (config.py:31-32; endpoint.py:40,64,151,154,210-211)."""
    frame = 0.03
    step = 0.01


def _resolve_cfg():
    # When dropped in under the reference's model.py, its own ``config.cfg`` is the live object
    # (CLI overrides mutate it, model.py:198-210) -- use it if it is already imported.
    import sys
    mod = sys.modules.get('config')
    c = getattr(mod, 'cfg', None) if mod is not None else None
    if c is not None and hasattr(c, 'frame') and hasattr(c, 'step'):
        return c
    return _Cfg()


cfg = _resolve_cfg()


def _frames_matrix(frames):
    f = np.asarray(frames)
    if f.ndim != 2:
        raise ValueError('frames must be a 2-D array [numframes, frame_len]')
    return f


def _vad_rows(frames, use_sq=False):
    """amp_sum / zcr of the rows of an explicit frame matrix (hop == length)."""
    f = _frames_matrix(frames)
    T, L = f.shape
    flat = f.reshape(-1) if f.dtype == np.int16 else np.ascontiguousarray(f, dtype=np.float32).reshape(-1)
    d_amp, d_zcr, T2 = _run.vad_features(flat, L, L, use_sq)
    assert T2 == T
    return d_amp.download((T,), np.float64), d_zcr.download((T,), np.int32), L


def get_amplitude(frames, window='square', use_sq=False):
    """Per-frame mean |x| (or mean x^2) as a Python list (endpoint.py:109-126).  window='square'
    convolves with ones(1), i.e. the identity -- the only form any caller uses.  'hamming'
    (never used by the reference's callers) smooths each rectified frame with hamming(L) first."""
    if isinstance(window, str) and window == 'hamming':
        f = _frames_matrix(frames)
        L = f.shape[-1]
        v = np.abs(f) if not use_sq else np.square(f)
        w = np.hamming(L)
        return [np.mean(np.convolve(row, w, 'same')) for row in v]
    amp_sum, _, L = _vad_rows(frames, use_sq)
    return [a for a in amp_sum / L]


def amplitude_feature(sig, rate, winlen, step):
    """get_amplitude(to_frames(...)) fused: frames never materialise (endpoint.py:128-131)."""
    L, S = int(rate * winlen), int(step * rate)
    d_amp, _, T = _run.vad_features(sig, L, S)
    return [a for a in d_amp.download((T,), np.float64) / L]


def get_zcr(frames):
    """Per-frame count of strictly opposite-sign neighbours (endpoint.py:182-198), list of ints."""
    _, zcr, _ = _vad_rows(frames)
    return [np.int64(z) for z in zcr]


def amplitude_rule(amp, mh=0.25, th=0.100, l_sil=0.100, r_sil=0.100, sigma=3, use_acr=False,
                   frames=None, rate=None):
    """Two-threshold segmentation of an amplitude track (endpoint.py:133-179) -> [(j, k), ...].
    M_L = mean+sigma*std of the quietest lead/tail frames, M_H = max(peak*mh, M_L); a run above
    M_H lasting >= th/cfg.frame frames is grown outwards while above M_L."""
    amp = list(amp)
    n = len(amp)

    def voiced(fr):  # endpoint.py:142-144
        lags = range(rate // 500, rate // 50)
        return max(acr(fr, lag) for lag in lags) / acr(fr, 0) > 0.55

    quiet = sorted(amp[:int(l_sil / cfg.step)] + amp[-int(r_sil / cfg.step):])[:-2]
    mu, sd = np.mean(quiet), np.std(quiet)
    min_run = th / cfg.frame
    low = mu + sigma * sd
    high = max(np.max(amp) * mh, low)
    segments = []
    i = 0
    while i < n:
        if amp[i] >= high:
            j = k = i
            while k < n and amp[k] > high:
                k += 1
            if k - j >= min_run:
                while j > 0 and amp[j] > low and (not use_acr or voiced(frames[j])):
                    j -= 1
                while k < n and amp[k] > low and (not use_acr or voiced(frames[k])):
                    k += 1
                segments.append((j, k))
            i = k
        i += 1
    return segments if segments else [(0, n)]


def zcr_rule(zcr, left, right, max_shift=0.400, l_sil=0, r_sil=0.100):
    """Widen (left, right) while the ZCR stays above mean+3*std of the trailing silence, by at
    most max_shift/cfg.frame frames (endpoint.py:201-220)."""
    zcr = list(zcr)
    limit = max_shift / cfg.frame
    ref = zcr[:int(l_sil / cfg.step)] + zcr[-int(r_sil / cfg.step):]
    thres = np.mean(ref) + 3 * np.std(ref)
    j = left
    while j > 0 and left - j <= limit and zcr[j] > thres:
        j -= 1
    k = right
    while k < len(zcr) and k - right <= limit and zcr[k] > thres:
        k += 1
    return j, k


def get_noise(amp, sep_point):
    """Mean frame amplitude outside the detected segments (endpoint.py:94-107); when the rule fell back
    to the whole clip there is no outside and the reference answers 1e30."""
    amp = np.asarray(amp, dtype=np.float64)
    segs = [(int(a), int(b)) for a, b in sep_point]
    if segs[0] == (0, len(amp)):
        return 1e30
    total, count, left = 0.0, 0, 0
    for lo, hi in segs:
        total += float(np.sum(amp[left:lo]))
        count += lo - left
        left = hi
    total += float(np.sum(amp[left:]))
    count += len(amp) - left
    return total / count


def basic_endpoint_detection(sig, rate, return_feature=False):
    """Energy + ZCR endpointing (endpoint.py:34-66) -> (left_sample, right_sample[, amp, zcr]).
    Framing (cfg.frame / cfg.step, int()-truncated sizes), amplitude, ZCR and the rule all run on
    the GPU; only the four result words (and amp/zcr when asked for) come back."""
    L, S = int(rate * cfg.frame), int(cfg.step * rate)
    d_amp, d_zcr, T = _run.vad_features(sig, L, S)
    left, right = _run.endpoint_rule(d_amp, d_zcr, T, L, cfg.frame, cfg.step)
    lo, hi = int(left * cfg.step * rate), int(right * cfg.step * rate)
    if not return_feature:
        return lo, hi
    amp = [a for a in d_amp.download((T,), np.float64) / L]
    zcr = [np.int64(z) for z in d_zcr.download((T,), np.int32)]
    return lo, hi, amp, zcr


def robust_endpoint_detection(sig, rate):
    """Autocorrelation-gated variant (endpoint.py:68-92): one amplitude-rule pass with mh = 0.5 that
    only grows a segment over frames whose normalised autocorrelation peak (lags rate//500 .. rate//50 - 1)
    exceeds 0.55.  A batch of one on the device: amplitude + ZCR (dsp_vad_features_batch), the gate of every
    frame (dsp_acr_gate_batch, fp64 sums) and the state machine with the gate as one more bit per frame
    (dsp_endpoint_rule_acr_batch); two result words come back.  Batches: features.batch.EndpointPlan(robust=True)."""
    L, S = int(rate * cfg.frame), int(cfg.step * rate)
    if rate // 50 > L:     # the gate's lags reach past the frame: no caller's cfg does this; the list form still serves it
        frames = to_frames(sig, rate, cfg.frame, step=cfg.step)
        seg = amplitude_rule(get_amplitude(frames), 0.5, frames=frames, use_acr=True, rate=rate)
        left2, right2 = zcr_rule(get_zcr(frames), seg[0][0], seg[-1][1])
        if right2 - left2 < 50:
            left2, right2 = 0, len(frames)
        return int(left2 * cfg.step * rate), int(right2 * cfg.step * rate)
    d_amp, d_zcr, T = _run.vad_features(sig, L, S)
    d_voiced = _run.acr_gate(sig, L, S, rate)
    left, right = _run.endpoint_rule(d_amp, d_zcr, T, L, cfg.frame, cfg.step, d_voiced)
    return int(left * cfg.step * rate), int(right * cfg.step * rate)
