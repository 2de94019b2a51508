"""Thin launch helpers shared by the drop-in functions: host array in -> HIP kernels -> host array
out (fp64, as the reference returns).  Single utterances are batches of one."""
from __future__ import annotations

import numpy as np

from . import _native as nat

_WIDTH = {nat.OUT_FRAMES: 'L', nat.OUT_MAGSPEC: 'K', nat.OUT_POWSPEC: 'K', nat.OUT_FBANK: 'M',
          nat.OUT_MFCC: 'C'}


def _signal_1d(sig):
    a = np.asarray(sig)
    if a.ndim != 1:
        a = a.reshape(-1)
    if a.size == 0:
        # framesig of an empty signal is one all-zero frame (sigproc.py:79-87)
        a = np.zeros(1, dtype=np.float32)
    return a


def features(plan, sig, out_kind):
    """Run one utterance through `plan` up to `out_kind`; returns fp32 array(s)."""
    lib = nat.load()
    wave, dtype = nat.as_wave(_signal_1d(sig))
    n = wave.shape[0]
    T = nat.frame_count(n, plan.L, plan.S)
    width = getattr(plan, _WIDTH[out_kind])
    d_wave = nat.device_array('wave', wave)
    d_out = nat.SCRATCH.get('out', T * width * 4)
    d_out2 = nat.SCRATCH.get('out2', T * 4)
    nat.check(lib.dsp_features_batch(plan.handle, d_wave.ptr, dtype, None, None, 1, T, n, out_kind,
                                     d_out.ptr, 0, d_out2.ptr, None))
    out = d_out.download((T, width), np.float32)
    if out_kind == nat.OUT_FBANK:
        return out, d_out2.download((T,), np.float32)
    return out


def rows_features(plan, rows, out_kind):
    """Treat each row of a [T, L] matrix as one frame (plan.S == plan.L, rectangular window)."""
    rows = np.asarray(rows)
    flat = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1)
    return features(plan, flat, out_kind)


def delta(feat32, N, want_dd=False):
    lib = nat.load()
    feat32 = np.ascontiguousarray(feat32, dtype=np.float32)
    T, D = feat32.shape
    d_in = nat.device_array('delta_in', feat32)
    d_out = nat.SCRATCH.get('delta_out', T * D * 4)
    d_dd = nat.SCRATCH.get('delta_dd', T * D * 4) if want_dd else None
    nat.check(lib.dsp_delta_batch(d_in.ptr, D, None, 1, T, T, D, int(N), d_out.ptr, D,
                                  d_dd.ptr if want_dd else None, D, None))
    out = d_out.download((T, D), np.float32)
    if want_dd:
        return out, d_dd.download((T, D), np.float32)
    return out


def scale_columns(x32, scale):
    lib = nat.load()
    x32 = np.ascontiguousarray(x32, dtype=np.float32)
    rows, cols = x32.shape
    d_x = nat.device_array('scale_x', x32)
    d_s = nat.device_array('scale_s', np.ascontiguousarray(scale, dtype=np.float32))
    nat.check(lib.dsp_scale_columns(d_x.ptr, rows, cols, d_s.ptr, None))
    return d_x.download((rows, cols), np.float32)


def preemphasis(sig, coeff):
    lib = nat.load()
    wave, dtype = nat.as_wave(sig)
    n = wave.shape[0]
    d_wave = nat.device_array('wave', wave)
    d_off = nat.device_array('pre_off', np.array([0, n], dtype=np.int64))
    d_out = nat.SCRATCH.get('out', n * 4)
    nat.check(lib.dsp_preemphasis_batch(d_wave.ptr, dtype, d_off.ptr, 1, n, float(coeff), d_out.ptr, None))
    return d_out.download((n,), np.float32)


def vad_features(sig, L, S, use_sq=False):
    """(amp_sum fp64 [T], zcr int32 [T], T) of one utterance framed at (L, S), rectangular."""
    lib = nat.load()
    wave, dtype = nat.as_wave(_signal_1d(sig))
    n = wave.shape[0]
    T = nat.frame_count(n, L, S)
    d_wave = nat.device_array('wave', wave)
    d_amp = nat.SCRATCH.get('vad_amp', T * 8)
    d_zcr = nat.SCRATCH.get('vad_zcr', T * 4)
    nat.check(lib.dsp_vad_features_batch(d_wave.ptr, dtype, None, None, 1, T, n, int(L), int(S),
                                         1 if use_sq else 0, d_amp.ptr, d_zcr.ptr, None))
    return d_amp, d_zcr, T


def endpoint_rule(d_amp, d_zcr, T, L, cfg_frame, cfg_step, d_voiced=None):
    """Frame indices (left, right) of one utterance from its amp / zcr rows on the device; with `d_voiced` (one byte
    per frame from acr_gate) the robust form of endpoint.py:68-92."""
    lib = nat.load()
    d_off = nat.device_array('ep_off', np.array([0, T], dtype=np.int64))
    d_ep = nat.SCRATCH.get('ep_out', 8)
    if d_voiced is None:
        nat.check(lib.dsp_endpoint_rule_batch(d_amp.ptr, d_zcr.ptr, d_off.ptr, 1, int(L), float(cfg_frame),
                                              float(cfg_step), d_ep.ptr, None))
    else:
        nat.check(lib.dsp_endpoint_rule_acr_batch(d_amp.ptr, d_zcr.ptr, d_voiced.ptr, d_off.ptr, 1, int(L), float(cfg_frame),
                                                  float(cfg_step), d_ep.ptr, None))
    ep = d_ep.download((2,), np.int32)
    return int(ep[0]), int(ep[1])


def acr_gate(sig, L, S, rate, thresh=0.55):
    """One byte per frame of one utterance framed at (L, S): the autocorrelation gate of endpoint.py:142-144 (lags
    rate // 500 .. rate // 50 - 1).  The wave buffer is the one vad_features uploaded (same scratch slot)."""
    lib = nat.load()
    wave, dtype = nat.as_wave(_signal_1d(sig))
    n = wave.shape[0]
    T = nat.frame_count(n, L, S)
    d_wave = nat.device_array('wave', wave)
    d_v = nat.SCRATCH.get('acr_voiced', T)
    nat.check(lib.dsp_acr_gate_batch(d_wave.ptr, dtype, None, None, 1, T, n, int(L), int(S), int(rate) // 500, int(rate) // 50,
                                     float(thresh), d_v.ptr, None))
    return d_v
