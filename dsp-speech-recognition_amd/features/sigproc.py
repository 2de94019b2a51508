"""Drop-in for the reference's ``features/sigproc.py`` (same names, arguments, return types),
computed by the HIP library.  Citations are file:line of the reference.

Every function returns fresh float64 NumPy arrays like the reference; the arithmetic itself is
fp32 on the GPU (parity: <= 1e-4 normwise, see tests/).  No CPU fallback exists.
"""
from __future__ import annotations

import logging

import numpy
import numpy as np

from . import _native as nat
from . import _plan, _run
from ._plan import round_half_up  # sigproc.py:55-56 (host-side size rounding)


def _ones(x):
    return numpy.ones((x,))


def to_frames(sig, rate, t=0.020, step=0.010):
    """Rectangular framing with int()-TRUNCATED sizes (sigproc.py:11-19)."""
    return framesig(sig, int(rate * t), int(step * rate))


def framesig(sig, frame_len, frame_step, winfunc=_ones, stride_trick=True):
    """Overlapping frames times window (sigproc.py:66-98) -> [numframes, frame_len] float64.
    ``stride_trick`` is accepted for signature parity; both reference code paths give the same
    values and one kernel serves both."""
    L, S = _plan.frame_sizes(frame_len, frame_step)
    win = numpy.asarray(winfunc(L), dtype=numpy.float64)
    plan = _plan.frame_plan(L, S, win)
    sig = numpy.asarray(sig)
    if sig.ndim != 1:
        raise ValueError('framesig expects a 1-D signal (the reference concatenates it with 1-D zeros)')
    return _run.features(plan, sig, nat.OUT_FRAMES).astype(numpy.float64)


def _spectrum(frames, NFFT, kind):
    frames = numpy.asarray(frames)
    if frames.ndim != 2:
        raise ValueError('frames must be a 2-D array [numframes, frame_len]')
    L = frames.shape[1]
    NFFT = int(NFFT)
    if L > NFFT:  # sigproc.py:143-146
        logging.warning('frame length (%d) is greater than FFT size (%d), frame will be truncated. '
                        'Increase NFFT to avoid.', L, NFFT)
    plan = _plan.frame_plan(L, L, numpy.ones(L), nfft=NFFT)
    return _run.rows_features(plan, frames, kind).astype(numpy.float64)


def magspec(frames, NFFT):
    """|rfft(frames, NFFT)| per row (sigproc.py:136-148) -> [numframes, NFFT/2+1]."""
    return _spectrum(frames, NFFT, nat.OUT_MAGSPEC)


def powspec(frames, NFFT):
    """(1/NFFT) |rfft|^2 per row (sigproc.py:151-158)."""
    return _spectrum(frames, NFFT, nat.OUT_POWSPEC)


def logpowspec(frames, NFFT, norm=1):
    """10 log10 of the floored power spectrum, optionally peak-normalised (sigproc.py:161-175).
    The spectrum comes from the GPU; the floor/log/peak shift are O(T*K) host epilogue."""
    ps = powspec(frames, NFFT)
    ps[ps <= 1e-30] = 1e-30
    lps = 10 * numpy.log10(ps)
    return lps - numpy.max(lps) if norm else lps


def preemphasis(signal, coeff=0.95):
    """y[0]=x[0], y[n]=x[n]-coeff*x[n-1] (sigproc.py:178-185).

    The reference slices along axis 0, so a 2-D ``(1, N)`` input -- what model.py:74 passes --
    returns row 0 UNFILTERED, and ``(R, N)`` returns ``[row0, (rows[1:]-coeff*rows[:-1]).ravel()]``.
    Those shapes involve no per-sample filtering along time; only the 1-D case runs the kernel."""
    a = numpy.asarray(signal)
    if a.ndim == 1:
        if a.shape[0] == 0:
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')
        return _run.preemphasis(a, coeff).astype(numpy.float64)
    return preemphasis_axis0(a, coeff)


def preemphasis_axis0(a, coeff):
    """Row-wise (axis-0) behaviour of the reference on >=2-D input; see preemphasis()."""
    head = numpy.asarray(a[0], dtype=numpy.float64).ravel()
    if a.shape[0] == 1:
        return head
    tail = (a[1:] - coeff * a[:-1]).ravel()
    return numpy.concatenate((head, tail))


def deframesig(frames, siglen, frame_len, frame_step, winfunc=_ones):
    """Overlap-add inverse of framesig (sigproc.py:101-133).  Off the hot path (no caller in the
    reference); evaluated on the host."""
    frames = numpy.asarray(frames, dtype=numpy.float64)
    L, S = round_half_up(frame_len), round_half_up(frame_step)
    T = frames.shape[0]
    assert frames.shape[1] == L, '"frames" matrix is wrong size, 2nd dim is not equal to frame_len'
    padlen = (T - 1) * S + L
    if siglen <= 0:
        siglen = padlen
    win = numpy.asarray(winfunc(L), dtype=numpy.float64)
    rec, corr = numpy.zeros(padlen), numpy.zeros(padlen)
    for t in range(T):
        corr[t * S:t * S + L] += win + 1e-15
        rec[t * S:t * S + L] += frames[t]
    return (rec / corr)[:siglen]


def rolling_window(a, window, step=1):
    """All length-``window`` runs along the last axis, every ``step``-th one (sigproc.py:59-63; the
    reference builds them as a strided view, here they are gathered -- same values, fresh array)."""
    a = numpy.asarray(a)
    starts = numpy.arange(0, a.shape[-1] - window + 1, step)
    return a[..., starts[:, None] + numpy.arange(window)[None, :]]


def acr(frame, n):
    """Autocorrelation at lag n divided by the overlap length (sigproc.py:48-53); host scalar
    helper of the pitch / robust-endpoint code, off the hot path."""
    frame = numpy.asarray(frame, dtype=numpy.float64)
    if n == 0:
        return numpy.sum(frame * frame) / len(frame)
    return numpy.sum(frame[:-n] * frame[n:]) / (len(frame) - n)
