"""Pin the CPU oracle to the reference: every golden case (outputs of the real reference, see
tests/golden/make_golden.py) must be reproduced by oracle/dsp_oracle.py to fp64 round-off."""
import warnings

import numpy as np
import pytest

from golden_cases import CASES, run_case
from oracle import dsp_oracle

INT_KEYS = {'endpoints', 'zcr', 'len'}


@pytest.mark.parametrize('case', CASES, ids=[c['name'] for c in CASES])
def test_oracle_matches_reference(case, golden):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        res = run_case(case, dsp_oracle)
    for key, val in res.items():
        ref = golden[f"{case['name']}/{key}"]
        assert val.shape == ref.shape, (case['name'], key, val.shape, ref.shape)
        if key in INT_KEYS or case['fn'] in ('get_zcr', 'amplitude_rule'):
            assert np.array_equal(val, ref), (case['name'], key)
        else:
            scale = max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
            tol = 1e-9 if case['fn'].startswith('model_feature_extract_mfcc') else 1e-12
            assert np.max(np.abs(val - ref)) <= tol * scale, (case['name'], key,
                                                              np.max(np.abs(val - ref)))


def test_golden_covers_every_case(golden):
    names = {k.split('/')[0] for k in golden}
    assert names == {c['name'] for c in CASES}
