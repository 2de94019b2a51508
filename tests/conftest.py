"""pytest configuration: markers, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dsp-speech-recognition_amd')
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    path = os.path.join(ROOT, 'tests', 'golden', 'golden.npz')
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def normwise(a, b):
    """max|a-b| / max|b| -- the parity metric of SURVEY.md section 8d."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b)) if b.size else 0.0
    num = np.max(np.abs(a - b)) if b.size else 0.0
    return num / den if den > 0 else num
