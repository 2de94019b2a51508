"""CPU checks of the matrix-pipe kernel's host side: the operand tables of csrc/mfma512_tables.h, pushed through an
emulation of the kernel's data path (tools/mfma512_emul.py: fp16 / bf16 roundings and the MFMA operand maps in NumPy),
must reproduce the oracle's MFCCs (base.py:8-16) -- a wrong table, K order or scale shows up here without a GPU."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, normwise
from oracle import dsp_oracle
import golden_cases as gc

SO = '/tmp/m512_tab.so'


@pytest.fixture(scope='module')
def emul():
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    subprocess.run(['g++', '-O2', '-shared', '-fPIC', '-o', SO, os.path.join(ROOT, 'tools', 'mfma512_tables_c.cpp')], check=True)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import mfma512_emul
    return mfma512_emul


@pytest.mark.parametrize('nfilt,L,win', [(40, 400, np.hamming), (26, 400, np.hamming), (40, 320, np.hamming), (13, 256, np.hamming),
                                          (40, 512, dsp_oracle._ones)])
def test_tables_reproduce_the_oracle(emul, nfilt, L, win):
    blob, lay = emul.build(L=L, S=160, nfilt=nfilt, win=win)
    assert lay['n_mtiles'] == (2 if nfilt + 1 <= 32 else 3) and len(lay['wblocks']) == (16 if nfilt <= 31 else 14)
    for kind in ('white', 'tone', 'ramp', 'siltail'):
        sig = gc.make_signal((kind, 30, 4000))
        cfg = dict(gc.BASE_CFG, nfilt=nfilt, winlen=L / 16000.0, winfunc=win)
        ref = dsp_oracle.mfcc(sig, **cfg)
        got = emul.mfcc_emul(sig, blob, lay, L=L, S=160)[:, :13]
        assert normwise(got, ref) <= 1e-4, (kind, normwise(got, ref))


def test_plans_the_kernel_does_not_serve_are_refused(emul):
    import ctypes as C
    from features import _plan as P
    lib = C.CDLL(SO)

    def rc(L=400, S=160, nfilt=40, lowfreq=0, numcep=13):
        window = np.ascontiguousarray(np.hamming(L), np.float32)
        fb = P.filterbank_matrix(nfilt, 512, 16000, lowfreq, None)
        st, cnt, w = P.mel_csr(fb)
        dct = np.ascontiguousarray(P.dct_lifter_matrix(nfilt, numcep, 22), np.float32)
        out = np.zeros(1 << 20, np.uint8)
        lay = np.zeros(64, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        return lib.m512_tables(L, S, 512, nfilt, numcep, 1, p(window), p(st), p(cnt), p(w), p(dct), p(out), out.size, p(lay))

    assert rc() == 0
    assert rc(nfilt=64) == -1          # more rows than three tiles of 16
    assert rc(S=200) == -1             # hop not a multiple of 16 samples
    assert rc(lowfreq=3000) == -3      # filters of the lowest tile reach beyond the lowest quarter of the spectrum


def test_host_fp16_rounding_is_ieee(emul):
    """The table builder's own fp32 -> fp16 conversion (round to nearest even, subnormals kept) against NumPy's, over
    normals, subnormals, ties, the overflow edge and signed zeros."""
    import ctypes as C
    lib = C.CDLL(SO)
    lib.m512_half.restype = C.c_uint16
    lib.m512_half.argtypes = [C.c_float]
    rng = np.random.default_rng(1)
    xs = np.concatenate([
        rng.standard_normal(2000) * 10.0 ** rng.uniform(-9, 5, 2000),
        np.float32(2.0) ** np.arange(-28, 17),
        (1.0 + np.arange(0, 64) / 2048.0) * 2.0 ** -3,               # ties and near-ties between fp16 neighbours
        np.arange(0, 40) * 2.0 ** -25,                                # the subnormal grid and its half steps
        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, 6.1e-5, 5.96e-8, 2.98e-8, 2.9e-8],
    ]).astype(np.float32)
    with np.errstate(over='ignore'):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.m512_half(float(x)) for x in xs], np.uint16)
    assert np.array_equal(got, want), [(float(x), hex(g), hex(w)) for x, g, w in zip(xs, got, want) if g != w][:5]
