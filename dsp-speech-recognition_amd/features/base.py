"""Drop-in for the reference's ``features/base.py``: same names, signatures and defaults
(base.py:8-10,18-20,40,60,70), computed by the HIP library.  Citations are file:line of the
reference.  Returns float64 arrays like the reference; arithmetic is fp32 on the GPU.
"""
from __future__ import annotations

import numpy

from . import _native as nat
from . import _plan, _run
from . import sigproc
from ._plan import hz2mel, mel2hz  # base.py:34-38
from .sigproc import to_frames  # base.py:5 re-exports it


def _ones(x):
    return numpy.ones((x,))


def _prepare_signal(signal, preemph):
    """fbank's first step (base.py:22): pre-emphasis.  Returns (1-D signal, coeff for the kernel).
    1-D input: the filter is fused into the framing kernel.  >=2-D input: the reference's axis-0
    quirk applies (sigproc.py:185) -- for model.py:74's (1, N) that means NO filtering."""
    a = numpy.asarray(signal)
    if a.ndim == 1:
        if a.shape[0] == 0:
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')
        return a, float(preemph)
    return sigproc.preemphasis_axis0(a, preemph), 0.0


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13,
         nfilt=26, nfft=512, lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True,
         winfunc=_ones):
    """MFCC of one utterance (base.py:8-16) -> [numframes, numcep] float64: one fused kernel does
    pre-emphasis, framing*window, rFFT, |X|^2/nfft, mel, log, DCT-II(ortho)*lifter and the
    log-energy swap of column 0."""
    sig, pre = _prepare_signal(signal, preemph)
    plan = _plan.mfcc_plan(samplerate, winlen, winstep, numcep, nfilt, nfft, lowfreq, highfreq, pre,
                           ceplifter, appendEnergy, winfunc)
    return _run.features(plan, sig, nat.OUT_MFCC).astype(numpy.float64)


def fbank(signal, samplerate=16000, winlen=0.025, winstep=0.01,
          nfilt=26, nfft=512, lowfreq=0, highfreq=None, preemph=0.97,
          winfunc=_ones):
    """Mel filterbank energies and per-frame total energy (base.py:18-32) -> (feat[T, nfilt],
    energy[T]); exact zeros are replaced by float64 eps as the reference does before any log."""
    sig, pre = _prepare_signal(signal, preemph)
    plan = _plan.mfcc_plan(samplerate, winlen, winstep, 0, nfilt, nfft, lowfreq, highfreq, pre,
                           0, False, winfunc, with_dct=False)
    feat, energy = _run.features(plan, sig, nat.OUT_FBANK)
    return feat.astype(numpy.float64), energy.astype(numpy.float64)


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """[nfilt, nfft/2+1] triangular mel filterbank (base.py:40-58).  A table, built on the host in
    fp64 exactly once per configuration; the kernels consume its CSR form."""
    return _plan.filterbank_matrix(nfilt, nfft, samplerate, lowfreq, highfreq)


def lifter(cepstra, L=22):
    """Sinusoidal cepstral lifter (base.py:60-68); L <= 0 returns the input unchanged."""
    if L > 0:
        cepstra = numpy.asarray(cepstra)
        nframes, ncoeff = numpy.shape(cepstra)
        return _run.scale_columns(cepstra, _plan.lifter_vector(ncoeff, L)).astype(numpy.float64)
    return cepstra


def delta(feat, N):
    """Delta features with edge-replicated context (base.py:70-79)."""
    if N < 1:
        raise ValueError('N must be an integer >= 1')
    feat = numpy.asarray(feat)
    if feat.ndim != 2:
        raise ValueError('feat must be a 2-D array [numframes, ndim]')
    return _run.delta(feat, N).astype(numpy.float64)
