#!/usr/bin/env python3
"""Generate tests/golden/rnn_golden.npz by running the ACTUAL reference classifier (rnn_clf.RNN over
layers.DynamicEncoder) on the CPU of the build container (cfg.cuda = False).

Nothing of the reference is copied: it is imported from /root/reference, its parameters are overwritten with
seeded values (features/classifier.py::fill_parameters, so that the fixture needs no 7 MB state_dict), it is called on a
seeded [200, B, 39] input, and only the input, the lengths, the parameter names and the logits are stored.

    python tests/golden/make_rnn_golden.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('REFERENCE_ROOT', '/root/reference')
SEED = 20260410


def main():
    import importlib.util
    import numpy as np
    import torch
    # our helper, loaded by path: `import features` must resolve to the REFERENCE package below
    spec = importlib.util.spec_from_file_location('_clf', os.path.join(ROOT, 'dsp-speech-recognition_amd', 'features',
                                                                      'classifier.py'))
    ours = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ours)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    os.chdir(tempfile.mkdtemp(prefix='refscratch_'))      # the reference's config creates ./log/ at import
    sys.path.insert(0, REF)
    import config
    config.cfg.cuda = False
    import rnn_clf
    torch.manual_seed(0)
    ref = rnn_clf.RNN().eval()
    names = ours.fill_parameters(ref, SEED)
    rng = np.random.default_rng(SEED + 1)
    len0 = np.array([200, 57, 131, 200, 12, 99, 180, 64], dtype=np.int64)      # unsorted on purpose
    T, B = 200, len(len0)
    inp = (rng.standard_normal((T, B, 39)) * np.array([3.0] + [1.0] * 38)).astype(np.float32)
    inp[:, 4] = -np.abs(inp[:, 4])                                              # exercises the zero rows under max pooling
    for b, n in enumerate(len0):
        inp[n:, b] = 0.0
    with torch.no_grad():
        logits = ref(torch.from_numpy(inp), len0).numpy()
    out = os.path.join(HERE, 'rnn_golden.npz')
    np.savez_compressed(out, inp=inp, len0=len0, logits=logits.astype(np.float32), names=np.array(names),
                        seed=np.int64(SEED))
    print('wrote', out, logits.shape, float(np.abs(logits).max()))


if __name__ == '__main__':
    main()
