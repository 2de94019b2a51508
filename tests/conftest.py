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


_MEASURED = {}


def record(name, value):
    """Keep the worst measured value of a parity metric; the session writes them to
    gpurun_out/parity_measured.json so every tolerance in the tests can be traced to a measurement."""
    value = float(value)
    _MEASURED[name] = max(_MEASURED.get(name, 0.0), value)
    return value


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_measured.json'), 'w') as fh:
            json.dump(dict(sorted(_MEASURED.items())), fh, indent=1)
    except OSError:
        pass
    tr = session.config.pluginmanager.get_plugin('terminalreporter')
    if tr is not None:
        tr.write_line('measured parity (worst per metric): ' + ', '.join(f'{k}={v:.3g}' for k, v in sorted(_MEASURED.items())))
