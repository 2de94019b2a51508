#!/usr/bin/env python3
"""Generate tests/golden/golden.npz by running the ACTUAL reference on tests/golden_cases.py.

Runs only in the build container, where the reference checkout exists (default /root/reference;
override with REFERENCE_ROOT).  Nothing of the reference is copied: it is imported from where it
lies, called on seeded synthetic inputs, and only the numeric outputs are stored.  The reference
never travels to the GPU box; the fixtures do.

    python tests/golden/make_golden.py            # rewrites tests/golden/golden.npz (+ manifest)

The reference's ``config`` module creates ./log/ at import, so we chdir to a scratch directory
first, and force a headless matplotlib backend.
"""
import io
import json
import os
import sys
import tempfile
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REF = os.environ.get('REFERENCE_ROOT', '/root/reference')


def load_reference():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    scratch = tempfile.mkdtemp(prefix='refscratch_')
    os.chdir(scratch)
    sys.path.insert(0, REF)
    import features  # noqa: the reference package
    import model  # noqa: reference driver (for the model.py glue rows)

    api = types.SimpleNamespace()
    for name in ('preemphasis', 'framesig', 'to_frames', 'magspec', 'powspec', 'logpowspec',
                 'deframesig', 'get_filterbanks', 'fbank', 'mfcc', 'lifter', 'delta',
                 'get_amplitude', 'get_zcr', 'amplitude_rule', 'zcr_rule', 'amplitude_feature',
                 'basic_endpoint_detection', 'robust_endpoint_detection', 'downsampling', 'center_clip',
                 'pitch_detect_frame_sr', 'pitch_detect_sr', 'get_noise', 'rolling_window'):
        setattr(api, name, getattr(features, name))
    # NOTE: ``features.preemphasis`` resolves to preprocess.preemphasis (star-import order); the
    # sigproc one is what fbank calls.  They are identical; record the sigproc one.
    api.preemphasis = features.sigproc.preemphasis

    def model_pipeline(sig, rate):
        base = model._ModelBase
        sound = base.endpoint_detect(None, sig, rate, augment=False)
        return base.feature_extract_mfcc(None, sound, rate)

    api.model_pipeline = model_pipeline

    def model_pipeline_aug(sig, rate, seed):
        import random
        random.seed(seed)                      # model.py:55 draws from the global generator
        base = model._ModelBase
        sound = base.endpoint_detect(None, sig, rate, augment=True)
        return base.feature_extract_mfcc(None, sound, rate)

    api.model_pipeline_aug = model_pipeline_aug
    bare = object.__new__(model._ModelBase)            # the methods only use self.deviation
    api.model_feature_extract_pitch = bare.feature_extract_pitch
    api.model_feature_extract_timespace = bare.feature_extract_timespace
    return api


def main():
    sys.path.insert(0, TESTS)
    import numpy as np
    from golden_cases import CASES, run_case

    api = load_reference()
    out = {}
    manifest = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for case in CASES:
            res = run_case(case, api)
            manifest[case['name']] = {k: [list(v.shape), str(v.dtype)] for k, v in res.items()}
            for k, v in res.items():
                out[f"{case['name']}/{k}"] = v
    buf = io.BytesIO()
    np.savez_compressed(buf, **out)
    path = os.path.join(HERE, 'golden.npz')
    with open(path, 'wb') as f:
        f.write(buf.getvalue())
    with open(os.path.join(HERE, 'golden_manifest.json'), 'w') as f:
        json.dump({'numpy': np.__version__, 'cases': manifest}, f, indent=1, sort_keys=True)
    print(f'wrote {path}: {len(CASES)} cases, {len(out)} arrays, {len(buf.getvalue()) / 1e6:.2f} MB')


if __name__ == '__main__':
    main()
