"""CPU checks of the frame-per-product matrix-pipe kernel's host side: the operand tables of csrc/mfma512t_tables.h,
pushed through an emulation of the kernel's data path (tools/mfma512t_emul.py: the kernel's fp16 / bf16 splits and the
MFMA operand maps in NumPy), must reproduce the oracle's MFCCs (base.py:8-16) -- a wrong table, K order, bin map or
scale shows up here without a GPU."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, normwise
from oracle import dsp_oracle
import golden_cases as gc


@pytest.fixture(scope='module')
def emul():
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    for src, so in (('mfma512_tables_c.cpp', '/tmp/m512_tab.so'), ('mfma512t_tables_c.cpp', '/tmp/m512t_tab.so')):
        subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tools', src)], check=True)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mfma512t_emul
    return mfma512t_emul


@pytest.mark.parametrize('nfilt,L,win,pattern,nblk', [(40, 400, np.hamming, 1, 17), (26, 400, np.hamming, 0, 18), (40, 320, np.hamming, 1, 17),
                                                      (13, 256, np.hamming, 0, 18), (40, 512, dsp_oracle._ones, 1, 17)])
def test_tables_reproduce_the_oracle(emul, nfilt, L, win, pattern, nblk):
    blob, lay = emul.build(L=L, S=160, nfilt=nfilt, win=win)
    assert lay['n_mtiles'] == (2 if nfilt + 1 <= 32 else 3) and lay['pattern'] == pattern and lay['n_wblocks'] == nblk
    for kind in ('white', 'tone', 'harmonic', 'ramp', 'siltail'):
        sig = gc.make_signal((kind, 30, 4000))
        cfg = dict(gc.BASE_CFG, nfilt=nfilt, winlen=L / 16000.0, winfunc=win)
        ref = dsp_oracle.mfcc(sig, **cfg)
        got = emul.mfcc_emul(sig, blob, lay, L=L, S=160)[:, :13]
        assert normwise(got, ref) <= 1e-4, (kind, normwise(got, ref))


def test_a_band_limited_filterbank_takes_the_full_pattern(emul):
    """lowfreq = 3000 Hz puts row tile 0 in the upper bin ranges: outside the sparse pattern, so every block is stored
    (the 16-frames-per-product kernel refuses this plan)."""
    blob, lay = emul.build(nfilt=40, lowfreq=3000)
    assert lay['pattern'] == 0 and lay['n_wblocks'] == 27
    sig = gc.make_signal(('white', 3, 4000))
    cfg = dict(gc.BASE_CFG, nfilt=40, lowfreq=3000, winfunc=np.hamming)
    assert normwise(emul.mfcc_emul(sig, blob, lay)[:, :13], dsp_oracle.mfcc(sig, **cfg)) <= 1e-4


def test_plans_the_kernel_does_not_serve_are_refused(emul):
    assert emul.build(nfilt=64)[1] == -1           # more rows than three tiles of 16
    assert emul.build(S=200)[1] == -1              # hop not a multiple of 16 samples
    assert emul.build(win=lambda n: 5.0 * np.hamming(n))[1] == -1     # stage-1 sums would leave the fp16 range


def test_bin_maps_cover_the_spectrum_once(emul):
    """Every FFT bin 1..255 that is not a multiple of 16 is exactly one power value of one lane (g, k1 = 1..15); the
    multiples of 16 (0 and 256 together) are the column-0 values."""
    def k2(rho):
        g, r = rho >> 2, rho & 3
        return (2 * g, 2 * g + 1, 14 - 2 * g, 15 - 2 * g)[r]
    seen = []
    for g in range(4):
        for k1 in range(1, 16):
            for r in range(4):
                k = k1 + 32 * k2(4 * g + r)
                b = k if k <= 256 else 512 - k
                assert 64 * g < b < 64 * g + 64
                seen.append(b)
    assert sorted(seen) == [b for b in range(1, 256) if b % 16]
