"""Drop-in for the reference's ``features/preprocess.py``."""
from __future__ import annotations

import numpy as np

from .sigproc import preemphasis  # preprocess.py:11-19 is the same filter as sigproc.py:178-185


def downsampling(sig, src_rate, dst_rate):
    """preprocess.py:21-28 keeps sample i whenever ``i * dst_rate / src_rate > kept - 1 + 1e-8``.
    For decimation (ratio < 1) at most one tick is crossed per step, so the kept indices are the
    first positions whose value exceeds -1 + 1e-8, 0 + 1e-8, 1 + 1e-8, ... -- one searchsorted."""
    sig = np.asarray(sig)
    n = len(sig)
    if n == 0:
        return np.array([])
    if dst_rate > src_rate:           # not a decimation: fall back to the literal loop
        ticks, keep = -1, []
        for i in range(n):
            if i * dst_rate / src_rate > ticks + 1e-8:
                ticks += 1
                keep.append(i)
        return sig[np.asarray(keep, dtype=np.int64)]
    vals = (np.arange(n, dtype=np.int64) * int(dst_rate)) / int(src_rate)
    ticks = np.arange(-1, int(np.floor(vals[-1])) + 1, dtype=np.float64) + 1e-8
    idx = np.searchsorted(vals, ticks, side='right')
    return sig[idx[idx < n]]
