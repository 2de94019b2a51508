"""CPU emulation of kernels_mfma512t.h from the tables the library builds (csrc/mfma512t_tables.h).

Follows the kernel's data movement operand by operand -- the frame's windowed samples as the stage-1 A fragment, the
DFT-32 matrix as B, the per-lane twiddles, stage 2 on the stage-1 result as it stands, the column-0 product over 16
frames, power, the bf16 packets of the mel exchange, log2, the DCT product, the scale correction -- with the MFMA
operand maps of the MI355X guide, so a wrong table or K order shows up here, on the CPU, before a GPU run.  Test
infrastructure only (compares with the fp64 oracle).   g++ -O2 -shared -fPIC -o /tmp/m512t_tab.so tools/mfma512t_tables_c.cpp
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'dsp-speech-recognition_amd'))
from oracle import dsp_oracle as o  # noqa: E402
import golden_cases as gc  # noqa: E402
from features import _plan as P  # noqa: E402
from mfma512_emul import block_f16, bf16split, mfma3  # noqa: E402

XBITS, WSH = 10, 16
SO = '/tmp/m512t_tab.so'


def f16split(x):
    """The kernel's data split (m512t_split2): hi = the value with its low 13 mantissa bits cleared, rounded to fp16 only
    where that is not exact (fp16 subnormals), lo = fp16(x - the cleared value)."""
    x = np.asarray(x, np.float32)
    h32 = (x.view(np.uint32) & np.uint32(0xffffe000)).view(np.float32)
    with np.errstate(over='ignore'):
        hi = h32.astype(np.float16).astype(np.float32)
        lo = (x - h32).astype(np.float16).astype(np.float32)
    return hi, lo


def n1_of(g, j):
    return 8 * (j >> 1) + 2 * g + (j & 1)


def build(L=400, S=160, nfilt=40, numcep=13, lifter=22, append_energy=True, rate=16000, win=np.hamming, lowfreq=0, highfreq=None):
    lib = C.CDLL(SO)
    window = np.ascontiguousarray(win(L), np.float32)
    fb = P.filterbank_matrix(nfilt, 512, rate, lowfreq, highfreq)
    st, cnt, w = P.mel_csr(fb)
    dct = np.ascontiguousarray(P.dct_lifter_matrix(nfilt, numcep, lifter), np.float32)
    out = np.zeros(1 << 20, np.uint8)
    lay = np.zeros(128, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib.m512t_tables(L, S, 512, nfilt, numcep, int(append_energy), p(window), p(st), p(cnt), p(w), p(dct), p(out), out.size, p(lay))
    if rc != 0:
        return None, rc
    names = ('off_f1', 'off_f2', 'off_m0', 'off_dm', 'off_tw', 'off_win', 'off_rowsum', 'off_w', 'bytes', 'n_wblocks', 'n_mtiles', 'erow', 'pattern')
    d = {k: int(lay[i]) for i, k in enumerate(names)}
    d['wblocks'] = [(int(lay[13 + 2 * b]), int(lay[14 + 2 * b])) for b in range(d['n_wblocks'])]
    return out[:d['bytes']].copy(), d


def block_bf16(blob, off):
    raw = (blob[off:off + 1024].view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(64, 8)
    A = np.zeros((16, 32), np.float32)
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = raw[lane]
    return A


def tile_mfcc(blob, lay, pre, HS=10):
    """pre: pre-emphasised fp32 samples from the tile's first frame on (zero padded), >= 16 * (15 HS + 32) of them.
    Returns cep[16 frames, 16 coefs].  The scale is per HALF tile (8 frames), from the half's samples."""
    F1 = [(block_f16(blob, lay['off_f1'] + u * 2048), block_f16(blob, lay['off_f1'] + u * 2048 + 1024)) for u in range(2)]   # [col k1][K]
    F2 = [(block_f16(blob, lay['off_f2'] + u * 2048), block_f16(blob, lay['off_f2'] + u * 2048 + 1024)) for u in range(2)]
    M0 = [(block_f16(blob, lay['off_m0'] + u * 2048), block_f16(blob, lay['off_m0'] + u * 2048 + 1024)) for u in range(2)]
    tw = blob[lay['off_tw']: lay['off_tw'] + 2048].view(np.float32).reshape(64, 8)
    wo = blob[lay['off_win']: lay['off_win'] + 2048].view(np.float32).reshape(64, 8)
    escs = np.zeros(16, np.int32)
    Pmain = np.zeros((16, 16, 16), np.float32)       # [frame][rho][k1]
    B0h = np.zeros((32, 16), np.float32)             # [K][frame]
    B0l = np.zeros((32, 16), np.float32)
    lanes = np.arange(64)
    gs, cs = lanes >> 4, lanes & 15
    for h in range(2):
        seg = pre[16 * HS * 8 * h: 16 * HS * 8 * h + 16 * (7 * HS + 32)]
        m = float(np.max(np.abs(seg)))
        e = 0 if m == 0 else int(np.floor(np.log2(m))) + 1
        esc = XBITS - e - 1 if m else 0
        sc = np.float32(2.0 ** esc)
        img = (seg * sc).astype(np.float32)
        for fl in range(8):
            f = 8 * h + fl
            escs[f] = esc
            fr = img[16 * HS * fl: 16 * HS * fl + 512]
            A = np.zeros((16, 32), np.float32)       # A[row n2 = c][K = 8 g + j]
            for j in range(8):
                A[cs, 8 * gs + j] = fr[16 * n1_of(gs, j) + cs] * wo[lanes, j]
            Ah, Al = f16split(A)
            # B operand blocks are stored lane-ordered with "row" = column k1: block_f16 gives [k1][K] -> transpose
            Dre = mfma3(Ah, Al, F1[0][0].T, F1[0][1].T)           # [n2][k1]
            Dim = mfma3(Ah, Al, F1[1][0].T, F1[1][1].T)
            Cc = np.zeros((16, 16), np.float32)
            Ss = np.zeros((16, 16), np.float32)
            for r in range(4):
                Cc[4 * gs + r, cs] = tw[lanes, r]
                Ss[4 * gs + r, cs] = tw[lanes, 4 + r]
            Yre = ((Dre * Cc).astype(np.float32) - (Dim * Ss).astype(np.float32)).astype(np.float32)
            Yim = ((Dre * Ss).astype(np.float32) + (Dim * Cc).astype(np.float32)).astype(np.float32)
            Yreh, Yrel = f16split(Yre)
            Yimh, Yiml = f16split(Yim)
            Bh = np.zeros((32, 16), np.float32)
            Bl = np.zeros((32, 16), np.float32)
            for g in range(4):
                for j in range(8):
                    n2, part = 4 * g + (j & 3), j >> 2
                    Bh[8 * g + j] = (Yimh if part else Yreh)[n2]
                    Bl[8 * g + j] = (Yiml if part else Yrel)[n2]
            B0h[:, f] = Bh[:, 0]
            B0l[:, f] = Bl[:, 0]
            Zre = mfma3(F2[0][0], F2[0][1], Bh, Bl)               # [rho][k1]
            Zim = mfma3(F2[1][0], F2[1][1], Bh, Bl)
            Pmain[f] = Zre * Zre + Zim * Zim
    Z0re = mfma3(M0[0][0], M0[0][1], B0h, B0l)                    # [m][frame]
    Z0im = mfma3(M0[1][0], M0[1][1], B0h, B0l)
    P0 = Z0re * Z0re + Z0im * Z0im
    Ph, Pl = bf16split(Pmain)
    P0h, P0l = bf16split(P0)
    nt = lay['n_mtiles']
    E = np.zeros((3, 16, 16), np.float32)                         # [tile][row][frame]
    for b, (step, tile) in enumerate(lay['wblocks']):
        Wh, Wl = block_bf16(blob, lay['off_w'] + b * 2048), block_bf16(blob, lay['off_w'] + b * 2048 + 1024)
        if step < 8:
            # the exchange: K element (gq, j) = value r = j & 3 of packet p = 8 s + 2 gq + (j >> 2) = lane p of the powers
            Bhi = np.zeros((32, 16), np.float32)
            Blo = np.zeros((32, 16), np.float32)
            for gq in range(4):
                for j in range(8):
                    p_ = 8 * step + 2 * gq + (j >> 2)
                    g, k1, r = p_ >> 4, p_ & 15, j & 3
                    Bhi[8 * gq + j] = Ph[:, 4 * g + r, k1]
                    Blo[8 * gq + j] = Pl[:, 4 * g + r, k1]
            E[tile] += ((Wh @ Bhi).astype(np.float32) + (Wl @ Bhi).astype(np.float32) + (Wh @ Blo).astype(np.float32)).astype(np.float32)
        else:
            Bp = np.zeros((32, 16), np.float32)
            for gq in range(4):
                for j in range(8):
                    Bp[8 * gq + j] = (P0l if j >= 4 else P0h)[4 * gq + (j & 3), :]
            E[tile] += ((Wh @ Bp).astype(np.float32) + (Wl @ Bp).astype(np.float32)).astype(np.float32)
    corr = (2 * escs + WSH).astype(np.float32)                    # per frame
    zval = np.float32(np.log2(2.220446049250313e-16)) + corr
    with np.errstate(divide='ignore'):
        LE = np.where(E == 0, zval[None, None, :], np.log2(E).astype(np.float32)).astype(np.float32)
    LEh, LEl = f16split(LE)
    cep = np.zeros((16, 16), np.float32)
    for step in range(2):
        Bh = np.zeros((32, 16), np.float32)
        Bl = np.zeros((32, 16), np.float32)
        for gam in range(4):
            for j in range(8):
                if step == 0:
                    tile, row = j >> 2, 4 * gam + (j & 3)
                else:
                    if j >= 4:
                        continue
                    tile, row = 2, 4 * gam + j
                if tile < nt:
                    Bh[8 * gam + j] = LEh[tile, row]
                    Bl[8 * gam + j] = LEl[tile, row]
        off = lay['off_dm'] + step * 2048
        cep += mfma3(block_f16(blob, off), block_f16(blob, off + 1024), Bh, Bl)
    rowsum = blob[lay['off_rowsum']:lay['off_rowsum'] + 64].view(np.float32)
    cep = cep - corr[None, :] * rowsum[:, None]
    return cep.T                                                  # [frame][coef]


def mfcc_emul(sig, blob, lay, L=400, S=160, preemph=0.97):
    x = np.asarray(sig).astype(np.float32)
    pre = np.empty_like(x)
    pre[0] = x[0]
    pre[1:] = x[1:] - np.float32(preemph) * x[:-1]
    T = 1 if len(x) <= L else 1 + int(np.ceil((len(x) - L) / S))
    HS = S // 16
    span = 16 * (15 * HS + 32)
    buf = np.zeros((T + 16) * S + span, np.float32)
    buf[:len(pre)] = pre
    out = np.zeros((T, 16), np.float32)
    for t0 in range(0, T, 16):
        cep = tile_mfcc(blob, lay, buf[t0 * S: t0 * S + span], HS)
        nv = min(16, T - t0)
        out[t0:t0 + nv] = cep[:nv]
    return out


def main():
    worst = 0.0
    for nfilt, L, S, win, name in ((40, 400, 160, np.hamming, 'base'), (26, 400, 160, np.hamming, 'nfilt26'),
                                   (40, 320, 160, np.hamming, 'L320'), (40, 512, 160, o._ones, 'ones512')):
        blob, lay = build(L=L, S=S, nfilt=nfilt, win=win)
        print(name, {k: v for k, v in lay.items() if k != 'wblocks'}, 'blocks', lay['wblocks'])
        kinds = ('white', 'int16', 'tone', 'harmonic', 'zeros', 'siltail', 'vad', 'ramp') if name == 'base' else ('white', 'tone')
        for kind in kinds:
            sig = gc.make_signal((kind, 30, 16000))
            cfg = dict(gc.BASE_CFG, nfilt=nfilt, winlen=L / 16000.0, winstep=S / 16000.0)
            cfg['winfunc'] = win
            ref = o.mfcc(sig, **cfg)
            got = mfcc_emul(sig, blob, lay, L=L, S=S)[:, :13]
            err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300)
            worst = max(worst, err)
            print(f'  {kind:9s} err {err:.3e}')
    print(f'worst {worst:.3e}')


if __name__ == '__main__':
    main()
