"""CPU emulation of the numerics of the MFMA DFT kernel (csrc/kernels_mfma512.h) -- design study.

The 512-point real DFT of a 400-sample frame is computed as two dense stages on the matrix pipe:
  n = 16 n1 + n2, k = k1 + 32 k2
  stage 1 (per column n2):  Y[k1, n2] = sum_n1 A1[n2][k1, n1] x[16 n1 + n2],
           A1[n2][k1, n1] = w[16 n1 + n2] W32^(n1 k1) W512^(n2 k1)     (window and twiddle folded in)
  stage 2 (per row k1):     X[k1 + 32 k2] = sum_n2 W16^(n2 k2) Y[k1, n2]
with every operand split into an fp16 (hi, lo) pair and three products (hi*hi, hi*lo, lo*hi) accumulated
in fp32 -- what v_mfma_f32_16x16x32_f16 does.  This script reproduces those roundings in NumPy and
measures the MFCC error against the fp64 oracle on the golden signals, to decide whether the scheme
meets the 1e-4 normwise bar.  Test infrastructure only (imports the oracle).
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from oracle import dsp_oracle as o  # noqa: E402
import golden_cases as gc  # noqa: E402

L, S, NFFT = 400, 160, 512
XBITS = int(os.environ.get('XBITS', 10))       # tile scale: max |x| in [2^(XBITS-1), 2^XBITS)
SA1 = float(2.0 ** int(os.environ.get('SA1', 1)))
SA2 = float(2.0 ** int(os.environ.get('SA2', 0)))
FLUSH = int(os.environ.get('FLUSH', 0))        # 1: fp16 subnormals flushed to zero in the operands
TILE = int(os.environ.get('TILE', 16))
NPROD = int(os.environ.get('NPROD', 3))


def f16(x):
    y = np.asarray(x, np.float32).astype(np.float16)
    if FLUSH:
        y = np.where(np.abs(y.astype(np.float32)) < 2.0 ** -14, np.float16(0), y)
    return y


def split(x):
    x = np.asarray(x, np.float32)
    hi = f16(x)
    lo = f16(x - hi.astype(np.float32))
    return hi.astype(np.float32), lo.astype(np.float32)


def mm3(Ah, Al, Bh, Bl):
    """fp32-accumulated three-product matmul (products of fp16 pairs are exact in fp32)."""
    acc = (Ah @ Bh).astype(np.float32)
    if NPROD >= 2:
        acc = acc + (Ah @ Bl).astype(np.float32)
    if NPROD >= 3:
        acc = acc + (Al @ Bh).astype(np.float32)
    return acc.astype(np.float32)


def build_tables(win):
    n1 = np.arange(32)
    k1 = np.arange(17)
    A1 = []
    for n2 in range(16):
        n = 16 * n1 + n2
        w = np.where(n < L, win[np.minimum(n, L - 1)], 0.0)
        ph = -2j * np.pi * (np.outer(k1, n1) / 32.0 + n2 * k1[:, None] / 512.0)
        M = w[None, :] * np.exp(ph)                     # [17, 32] complex
        rows = [M[0].real]                              # k1 = 0: real (twiddle 1)
        for k in range(1, 16):
            rows += [M[k].real, M[k].imag]
        # k1 = 16: (-1)^n1 w x summed -> real; its twiddle W512^(16 n2) goes into stage 2
        rows.append((w * np.cos(np.pi * n1)))
        A1.append(np.stack(rows) * SA1)                 # [32, 32]
    A1 = np.stack(A1)                                   # [16, 32, 32]
    return A1


def stage2_tables():
    n2 = np.arange(16)
    k2 = np.arange(16)
    F = np.exp(-2j * np.pi * np.outer(k2, n2) / 16.0)  # [k2, n2]
    # complex rows k1 = 1..15: real 32 x 32 acting on (re n2 | im n2)
    A2 = np.block([[F.real, -F.imag], [F.imag, F.real]]) * SA2
    # rows 0 and 16 packed: inputs (R0[n2] | R16[n2]); row 0: k2 = 0..8; row 16: twiddle W512^(16 n2), k2 = 0..7
    F0 = F[:9]
    T16 = np.exp(-2j * np.pi * 16 * n2 / 512.0)
    F16 = F[:8] * T16[None, :]
    top = np.concatenate([F0.real, F0.imag[1:8]], 0)    # 9 re + 7 im (k2 = 1..7) = 16 rows
    bot = np.concatenate([F16.real, F16.imag], 0)       # 8 re + 8 im = 16 rows
    A2p = np.zeros((32, 32))
    A2p[:16, :16] = top
    A2p[16:, 16:] = bot
    return A2, A2p * SA2


def powspec_mfma(sig_pre, T):
    """sig_pre: pre-emphasised fp32 signal (zero padded); returns P[T, 257] in fp32 semantics."""
    win = np.hamming(L)
    A1 = build_tables(win)
    A2, A2p = stage2_tables()
    A1h, A1l = split(A1)
    A2h, A2l = split(A2)
    A2ph, A2pl = split(A2p)
    P = np.zeros((T, 257), np.float32)
    for t0 in range(0, T, TILE):
        fr = np.arange(t0, min(T, t0 + TILE))
        lo_s, hi_s = fr[0] * S, fr[-1] * S + 512
        seg = sig_pre[lo_s:hi_s].astype(np.float32)
        m = float(np.max(np.abs(seg))) if seg.size else 0.0
        e = 0 if m == 0 else XBITS - (int(np.floor(np.log2(m))) + 1)
        sc = np.float32(2.0 ** e)
        xh, xl = split(seg * sc)
        Y = np.zeros((len(fr), 32, 16), np.float32)     # [frame, row, n2]
        for n2 in range(16):
            idx = (fr[:, None] - fr[0]) * S + 16 * np.arange(32)[None, :] + n2
            Bh, Bl = xh[idx].T, xl[idx].T               # [32 n1, frames]
            Y[:, :, n2] = mm3(A1h[n2], A1l[n2], Bh, Bl).T
        Yh, Yl = split(Y)
        for k1 in range(1, 16):
            Bh = np.concatenate([Yh[:, 2 * k1 - 1, :], Yh[:, 2 * k1, :]], 1).T   # [32, frames]
            Bl = np.concatenate([Yl[:, 2 * k1 - 1, :], Yl[:, 2 * k1, :]], 1).T
            Z = mm3(A2h, A2l, Bh, Bl)                   # [32, frames]: re k2 | im k2
            pw = Z[:16] ** 2 + Z[16:] ** 2
            for k2 in range(16):
                k = k1 + 32 * k2
                if k > 256:
                    k = 512 - k
                P[fr, k] = pw[k2]
        Bh = np.concatenate([Yh[:, 0, :], Yh[:, 31, :]], 1).T
        Bl = np.concatenate([Yl[:, 0, :], Yl[:, 31, :]], 1).T
        Z = mm3(A2ph, A2pl, Bh, Bl)
        re0 = Z[0:9]
        im0 = np.concatenate([np.zeros((1, len(fr)), np.float32), Z[9:16], np.zeros((1, len(fr)), np.float32)])
        for k2 in range(9):
            P[fr, 32 * k2] = re0[k2] ** 2 + im0[k2] ** 2
        for k2 in range(8):
            P[fr, 16 + 32 * k2] = Z[16 + k2] ** 2 + Z[24 + k2] ** 2
        P[fr] *= np.float32((1.0 / (float(sc) * SA1 * SA2)) ** 2 / NFFT)
    return P


def mfcc_mfma(sig, nfilt=40, numcep=13, ceplifter=22, preemph=0.97):
    sig = np.asarray(sig)
    x = sig.astype(np.float32)
    pre = np.empty_like(x)
    pre[0] = x[0]
    pre[1:] = x[1:] - np.float32(preemph) * x[:-1]
    T, padlen = o.frame_geometry(len(x), L, S)[:2] if hasattr(o, 'frame_geometry') else (None, None)
    T = 1 if len(x) <= L else 1 + int(np.ceil((len(x) - L) / S))
    buf = np.zeros((T - 1) * S + 512 + 16, np.float32)
    buf[:len(pre)] = pre
    P = powspec_mfma(buf, T).astype(np.float64)
    fb = o.get_filterbanks(nfilt, NFFT, 16000, 0, 8000)
    energy = P.sum(1)
    energy = np.where(energy == 0, o.EPS, energy)
    feat = (P.astype(np.float32) @ fb.T.astype(np.float32)).astype(np.float64)
    feat = np.where(feat == 0, o.EPS, feat)
    feat = np.log(feat)
    feat = feat @ o.dct2_ortho_matrix(nfilt, numcep).T
    feat = o.lifter(feat, ceplifter)
    feat[:, 0] = np.log(energy)
    return feat, P


def main():
    worst = 0.0
    for kind in ('white', 'white32', 'uniform', 'int16', 'tone', 'harmonic', 'zeros', 'siltail', 'ramp', 'vad',
                 'vadf', 'bursts'):
        for seed in (30, 31):
            sig = gc.make_signal((kind, seed, 16000))
            ref = o.mfcc(sig, **{k: (np.hamming if k == 'winfunc' else v) for k, v in gc.BASE_CFG.items()})
            got, P = mfcc_mfma(sig)
            err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-300)
            # power spectrum error, relative to the frame's largest bin
            pre = o.preemphasis(sig.astype(np.float64), 0.97)
            fr = o.framesig(pre, L, S, np.hamming)
            Pref = o.powspec(fr, NFFT)
            perr = np.max(np.abs(P - Pref) / np.maximum(Pref.max(1, keepdims=True), 1e-300))
            # an fp32 FFT for comparison
            X32 = np.fft.rfft(fr.astype(np.float32), NFFT).astype(np.complex64)
            P32 = (np.abs(X32) ** 2 / NFFT).astype(np.float32)
            p32err = np.max(np.abs(P32 - Pref) / np.maximum(Pref.max(1, keepdims=True), 1e-300))
            worst = max(worst, err)
            print(f'{kind:9s} seed {seed}: mfcc err {err:.3e}   powspec err/framemax {perr:.3e} (fp32 pocketfft {p32err:.3e})')
    print(f'worst {worst:.3e}  (bar 1e-4)  XBITS={XBITS} SA1={SA1} FLUSH={FLUSH} TILE={TILE} NPROD={NPROD}')


if __name__ == '__main__':
    main()
