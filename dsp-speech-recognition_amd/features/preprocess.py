"""Drop-in for the reference's ``features/preprocess.py``."""
from __future__ import annotations

import numpy as np

from .sigproc import preemphasis  # preprocess.py:11-19 is the same filter as sigproc.py:178-185


def downsampling(sig, src_rate, dst_rate):
    """Keep sample i whenever i*dst/src passes the next integer tick (preprocess.py:21-28).
    Sequential index selection; pitch-side helper, out of the hot path."""
    sig = np.asarray(sig)
    ticks = -1
    keep = []
    for i in range(len(sig)):
        if i * dst_rate / src_rate > ticks + 1e-8:
            ticks += 1
            keep.append(i)
    return sig[np.asarray(keep, dtype=np.int64)] if keep else np.array([])
