"""CPU emulation of kernels_mfma512.h from the tables the library builds (csrc/mfma512_tables.h).

Follows the kernel's data movement operand by operand (stage-1 A / B fragments, (re, im) packing, the 4 x 4 lane-group
transpose, stage 2, power, bf16 mel blocks, log2, DCT product, scale correction) with the MFMA operand maps of the
MI355X guide, so a wrong table or a wrong K order shows up here, on the CPU, before a GPU run.  Test infrastructure
only (compares with the fp64 oracle).   g++ -O2 -shared -fPIC -o /tmp/m512_tab.so tools/mfma512_tables_c.cpp
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'dsp-speech-recognition_amd'))
from oracle import dsp_oracle as o  # noqa: E402
import golden_cases as gc  # noqa: E402
from features import _plan as P  # noqa: E402

KAP = np.array([[0, 2, 4, 6], [15, 13, 11, 9], [1, 3, 5, 7], [14, 12, 10, 8]])
XBITS, WSH = 10, 16


def mbin(s, kr):
    if s == 0:
        return 32 * kr if kr <= 7 else 16 + 32 * (15 - kr)
    return s + 32 * kr if kr <= 7 else 512 - s - 32 * kr


def build(L=400, S=160, nfilt=40, numcep=13, lifter=22, append_energy=True, rate=16000, win=np.hamming):
    lib = C.CDLL('/tmp/m512_tab.so')
    window = np.ascontiguousarray(win(L), np.float32)
    fb = P.filterbank_matrix(nfilt, 512, rate, 0, None)
    st, cnt, w = P.mel_csr(fb)
    dct = np.ascontiguousarray(P.dct_lifter_matrix(nfilt, numcep, lifter), np.float32)
    out = np.zeros(1 << 20, np.uint8)
    lay = np.zeros(64, np.int32)
    rc = lib.m512_tables(L, S, 512, nfilt, numcep, int(append_energy), window.ctypes.data_as(C.c_void_p),
                         st.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                         dct.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.size,
                         lay.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    d = dict(off_a1=lay[0], off_a2=lay[1], off_a2p=lay[2], off_dm=lay[3], off_w=lay[4], off_rowsum=lay[5],
             bytes=lay[6], n_wblocks=lay[7], n_mtiles=lay[8], sa1=lay[9], KR=lay[10],
             wblocks=[(lay[11 + 2 * b], lay[12 + 2 * b]) for b in range(lay[7])])
    return out[:lay[6]].copy(), d


def block_f16(blob, off):
    """1 KB lane-ordered block -> A[16 rows][32 k] as float32 (A[row l & 15][8 (l >> 4) + j])."""
    raw = blob[off:off + 1024].view(np.float16).reshape(64, 8).astype(np.float32)
    A = np.zeros((16, 32), np.float32)
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = raw[lane]
    return A


def block_bf16(blob, off):
    raw = (blob[off:off + 1024].view(np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(64, 8)
    A = np.zeros((16, 32), np.float32)
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = raw[lane]
    return A


def f16split(x):
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32)


def bf16split(x):
    x = np.asarray(x, np.float32)
    hi = bf16(x)
    lo = bf16(x - hi)
    return hi, lo


def mfma3(Ah, Al, Bh, Bl):
    return ((Ah @ Bh).astype(np.float32) + (Ah @ Bl).astype(np.float32) + (Al @ Bh).astype(np.float32)).astype(np.float32)


def tile_mfcc(blob, lay, pre, HS=10):
    """pre: pre-emphasised fp32 samples of the tile (zero padded), >= 16 * (15 HS + 32) of them.
    Returns cep[16 frames, 16 coefs]."""
    m = float(np.max(np.abs(pre)))
    # the kernel bounds |y| by 2 max|x| from the raw samples; here: the exponent of max|y| plus one
    e = 0 if m == 0 else int(np.floor(np.log2(m))) + 1
    esc = XBITS - e - 1 if m else 0
    sc = np.float32(2.0 ** esc)
    xh, xl = f16split(pre * sc)
    img_h = xh[:16 * (15 * HS + 32)].reshape(-1, 16).T          # [n2][row]
    img_l = xl[:16 * (15 * HS + 32)].reshape(-1, 16).T
    acc = np.zeros((16, 2, 16, 16), np.float32)                 # [n2][t][row][frame]
    for n2 in range(16):
        Bh = np.stack([img_h[n2, HS * n: HS * n + 32] for n in range(16)], 1)   # [k = n1][frame]
        Bl = np.stack([img_l[n2, HS * n: HS * n + 32] for n in range(16)], 1)
        for t in range(2):
            off = lay['off_a1'] + ((n2 * 2 + t) * 2) * 1024
            acc[n2, t] = mfma3(block_f16(blob, off), block_f16(blob, off + 1024), Bh, Bl)
    Yh, Yl = f16split(acc)
    A2 = [(block_f16(blob, lay['off_a2'] + u * 2048), block_f16(blob, lay['off_a2'] + u * 2048 + 1024)) for u in range(2)]
    A2p = [(block_f16(blob, lay['off_a2p'] + u * 2048), block_f16(blob, lay['off_a2p'] + u * 2048 + 1024)) for u in range(2)]
    Pw = np.zeros((16, 4, 4, 16), np.float32)                   # [slot][g][i'][frame]
    for s in range(16):
        # B2[k = 8 gam + j][frame] = Y_part(j & 1)[slot s][n2 = 4 gam + (j >> 1)]
        Bh = np.zeros((32, 16), np.float32)
        Bl = np.zeros((32, 16), np.float32)
        for gam in range(4):
            for j in range(8):
                Bh[8 * gam + j] = Yh[4 * gam + (j >> 1), j & 1, s]
                Bl[8 * gam + j] = Yl[4 * gam + (j >> 1), j & 1, s]
        A = A2p if s == 0 else A2
        re = mfma3(A[0][0], A[0][1], Bh, Bl)                    # [rho][frame]
        im = mfma3(A[1][0], A[1][1], Bh, Bl)
        pw = re * re + im * im
        for g in range(4):
            for ip in range(4):
                Pw[s, g, ip] = pw[4 * g + ip]
    Ph, Pl = bf16split(Pw)
    E = np.zeros((lay['n_mtiles'], 16, 16), np.float32)         # [tile][row][frame]
    for b, (step, tile) in enumerate(lay['wblocks']):
        ip, h = step >> 1, step & 1
        Bh = np.zeros((32, 16), np.float32)
        Bl = np.zeros((32, 16), np.float32)
        for gam in range(4):
            for j in range(8):
                Bh[8 * gam + j] = Ph[8 * h + j, gam, ip]
                Bl[8 * gam + j] = Pl[8 * h + j, gam, ip]
        off = lay['off_w'] + b * 2048
        E[tile] += mfma3(block_bf16(blob, off), block_bf16(blob, off + 1024), Bh, Bl)
    corr = np.float32(2 * esc + WSH)
    zval = np.float32(np.log2(2.220446049250313e-16)) + corr
    with np.errstate(divide='ignore'):
        LE = np.where(E == 0, zval, np.log2(E).astype(np.float32)).astype(np.float32)
    # DCT product: step 0 element (gam, j) = filter 16 (j >> 2) + 4 gam + (j & 3); step 1: 32 + 4 gam + j
    nt = lay['n_mtiles']
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
    cep = cep - corr * rowsum[:, None]
    return cep.T                                                # [frame][coef]


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
