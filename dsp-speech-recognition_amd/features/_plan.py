"""Host-side logic of the feature path: everything the reference evaluates in Python per call
(sizes, window, mel triangles, DCT*lifter) is evaluated here ONCE per configuration, in fp64, and
handed to the HIP library as fp32 tables (include/dsp_frontend.h: dsp_plan_desc).
"""
from __future__ import annotations

import ctypes as C
import decimal
import math
import threading

import numpy as np

from . import _native as nat


def ones_window(n):
    return np.ones((n,))


def round_half_up(number):
    """Decimal ROUND_HALF_UP of the exact binary value (reference sigproc.py:55-56):
    0.01*22050 = 220.5 -> 221, where Python's round() gives 220."""
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'), rounding=decimal.ROUND_HALF_UP))


def frame_sizes(frame_len, frame_step):
    """(L, S) as framesig derives them (sigproc.py:77-78)."""
    return int(round_half_up(frame_len)), int(round_half_up(frame_step))


def hz2mel(hz):
    """HTK mel scale (base.py:34-35)."""
    return 2595 * np.log10(1 + hz / 700.)


def mel2hz(mel):
    """Inverse HTK mel scale (base.py:37-38)."""
    return 700 * (10 ** (mel / 2595.0) - 1)


def mel_edges(nfilt, nfft, samplerate, lowfreq, highfreq):
    """nfilt+2 FFT-bin edges floor((nfft+1)*hz/rate) (base.py:41-50); `highfreq or rate/2`."""
    highfreq = highfreq or samplerate / 2
    assert highfreq <= samplerate / 2, "highfreq is greater than samplerate/2"
    mels = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    return np.floor((nfft + 1) * mel2hz(mels) / samplerate)


def filterbank_matrix(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """Dense [nfilt, nfft/2+1] fp64 triangles with the reference's left-closed bin ranges
    (base.py:52-58).  Pure host logic: this is the table the kernels consume, not a compute path."""
    edges = mel_edges(nfilt, nfft, samplerate, lowfreq, highfreq)
    K = nfft // 2 + 1
    fb = np.zeros([nfilt, K])
    bins = np.arange(K, dtype=np.float64)
    for j in range(nfilt):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        rise = (bins >= lo) & (bins < mid)
        fall = (bins >= mid) & (bins < hi)
        fb[j, rise] = (bins[rise] - lo) / (mid - lo)
        fb[j, fall] = (hi - bins[fall]) / (hi - mid)
    return fb


def mel_csr(fb):
    """(start[M], count[M], weights[nnz]) covering each row's non-zero span."""
    M = fb.shape[0]
    start = np.zeros(M, dtype=np.int32)
    count = np.zeros(M, dtype=np.int32)
    w = []
    for j in range(M):
        nz = np.nonzero(fb[j])[0]
        if nz.size:
            start[j] = nz[0]
            count[j] = nz[-1] - nz[0] + 1
            w.append(fb[j, nz[0]:nz[-1] + 1])
    weights = np.concatenate(w) if w else np.zeros(0)
    return start, count, np.ascontiguousarray(weights, dtype=np.float32)


def dct_lifter_matrix(nfilt, numcep, ceplifter):
    """[C, M] = diag(lifter) @ DCT-II(ortho)[:C]  (base.py:13-14, 60-68; scipy.fftpack.dct closed
    form: c[k] = s_k * 2 * sum_n x[n] cos(pi k (2n+1) / 2M), s_0 = sqrt(1/4M), s_k = sqrt(1/2M))."""
    C_ = min(numcep, nfilt)
    n = np.arange(nfilt, dtype=np.float64)[None, :]
    k = np.arange(C_, dtype=np.float64)[:, None]
    mat = 2.0 * np.cos(np.pi * k * (2.0 * n + 1.0) / (2.0 * nfilt))
    mat *= math.sqrt(1.0 / (2.0 * nfilt))
    mat[0] *= math.sqrt(0.5)
    return lifter_vector(C_, ceplifter)[:, None] * mat


def lifter_vector(ncoeff, L=22):
    """1 + (L/2) sin(pi n / L); identity for L <= 0 (base.py:60-68)."""
    if L > 0:
        n = np.arange(ncoeff)
        return 1 + (L / 2.) * np.sin(np.pi * n / L)
    return np.ones(ncoeff)


class Plan:
    """Owns one dsp_plan handle plus the host copies of its tables."""

    def __init__(self, L, S, nfft, window, preemph=0.0, fb=None, dct=None, append_energy=False, host_dry_run=False):
        # host_dry_run (tests of the sanitizer build only): every host-side table builder runs, the tables stay in host
        # memory, no device is touched; the plan can be destroyed and nothing else (include/dsp_frontend.h)
        if not host_dry_run:
            nat.require_device()
        self.L, self.S, self.nfft = int(L), int(S), int(nfft)
        self.K = self.nfft // 2 + 1
        self.window = np.ascontiguousarray(window, dtype=np.float32)
        if self.window.shape != (self.L,):
            raise ValueError(f'winfunc returned shape {self.window.shape}, expected ({self.L},)')
        self.M = 0 if fb is None else int(fb.shape[0])
        self.C = 0 if dct is None else int(dct.shape[0])
        desc = nat.PlanDesc()
        desc.frame_len, desc.frame_step, desc.nfft = self.L, self.S, self.nfft
        desc.nfilt, desc.numcep = self.M, self.C
        desc.append_energy = 1 if append_energy else 0
        desc.preemph = float(preemph)
        desc.h_window = self.window.ctypes.data
        self._keep = [self.window]
        if fb is not None:
            start, count, weights = mel_csr(fb)
            if weights.size == 0:
                weights = np.zeros(1, dtype=np.float32)
            self._keep += [start, count, weights]
            desc.h_mel_start, desc.h_mel_count = start.ctypes.data, count.ctypes.data
            desc.h_mel_weights = weights.ctypes.data
        if dct is not None:
            d32 = np.ascontiguousarray(dct, dtype=np.float32)
            self._keep.append(d32)
            desc.h_dct = d32.ctypes.data
        h = C.c_void_p(0)
        if host_dry_run:
            nat.check(nat.load().dsp_debug_host_dry_run(1))
        try:
            nat.check(nat.load().dsp_plan_create(C.byref(desc), C.byref(h)))
        finally:
            if host_dry_run:
                nat.check(nat.load().dsp_debug_host_dry_run(0))
        self.handle = h.value

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            try:
                nat.load().dsp_plan_destroy(h)
            except Exception:
                pass
            self.handle = None


_cache = {}
_cache_lock = threading.Lock()
_CACHE_MAX = 64


def _cached(key, build):
    key = (nat.current_device(),) + tuple(key)      # a plan's tables live on the device it was built on
    with _cache_lock:
        p = _cache.get(key)
        if p is None:
            if len(_cache) >= _CACHE_MAX:
                _cache.pop(next(iter(_cache)))
            p = build()
            _cache[key] = p
        return p


def frame_plan(L, S, window, nfft=None, preemph=0.0):
    """Plan that stops at frames / spectra (no mel tables)."""
    L, S = int(L), int(S)
    if nfft is None:
        nfft = 16
        while nfft < L:
            nfft *= 2
        nfft = min(nfft, 4096)
    w = np.ascontiguousarray(window, dtype=np.float32)
    key = ('frame', L, S, int(nfft), float(preemph), w.tobytes())
    return _cached(key, lambda: Plan(L, S, nfft, w, preemph=preemph))


def mfcc_plan(samplerate, winlen, winstep, numcep, nfilt, nfft, lowfreq, highfreq, preemph,
              ceplifter, appendEnergy, winfunc, with_dct=True):
    """Plan for base.fbank / base.mfcc argument tuples."""
    L, S = frame_sizes(winlen * samplerate, winstep * samplerate)
    w = np.ascontiguousarray(np.asarray(winfunc(L), dtype=np.float64), dtype=np.float32)
    highfreq = highfreq or samplerate / 2
    key = ('mfcc', L, S, int(nfft), int(nfilt), int(numcep) if with_dct else -1, float(samplerate),
           float(lowfreq), float(highfreq), float(preemph), float(ceplifter), bool(appendEnergy),
           w.tobytes())

    def build():
        fb = filterbank_matrix(nfilt, nfft, samplerate, lowfreq, highfreq)
        dct = dct_lifter_matrix(nfilt, numcep, ceplifter) if with_dct else None
        return Plan(L, S, nfft, w, preemph=preemph, fb=fb, dct=dct, append_energy=appendEnergy)

    return _cached(key, build)
