#!/usr/bin/env python3
"""Generate tests/golden/clf_golden.npz by running the ACTUAL reference classifiers rnn_clf.HRNN, HRNN_Att and Transformer
(over layers.DynamicEncoder / SelfAttn / LayerNormalization and transformer.TransformerEncoder) on the CPU of the build
container (cfg.cuda = False), on the input of tests/golden/rnn_golden.npz.

Nothing of the reference is copied: it is imported from /root/reference, its parameters are overwritten with seeded values
(features/classifier.py::fill_parameters), and only parameter names and outputs are stored.  The three classes call
F.dropout with its default training=True in every mode (rnn_clf.py:73,116,184,199), so their logits are random; stored are
  * what is deterministic in the reference as it stands: ``feat`` of HRNN / HRNN_Att (return_feature=True) and the
    attention block's output of Transformer (attn_enc(inp): five of its 200 rows and its sum over time), all in eval mode;
  * ``*_nodrop``: logits and features of the same call with torch.nn.functional.dropout replaced by the identity for its
    duration -- the reference's own layers and wiring behind the dropout calls, made deterministic.

    python tests/golden/make_clf_golden.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('REFERENCE_ROOT', '/root/reference')
SEED = 20260504


def main():
    import importlib.util
    import numpy as np
    import torch
    import torch.nn.functional as F
    spec = importlib.util.spec_from_file_location('_clf', os.path.join(ROOT, 'dsp-speech-recognition_amd', 'features', 'classifier.py'))
    ours = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ours)
    g = np.load(os.path.join(HERE, 'rnn_golden.npz'))
    inp, len0 = torch.from_numpy(g['inp']), g['len0']
    os.environ.setdefault('MPLBACKEND', 'Agg')
    os.chdir(tempfile.mkdtemp(prefix='refscratch_'))      # the reference's config creates ./log/ at import
    sys.path.insert(0, REF)
    import config
    config.cfg.cuda = False
    import rnn_clf
    out = {'seed': np.int64(SEED)}
    real_dropout = F.dropout
    for k, (name, cls) in enumerate((('hrnn', rnn_clf.HRNN), ('hrnn_att', rnn_clf.HRNN_Att), ('transformer', rnn_clf.Transformer))):
        torch.manual_seed(0)
        ref = cls().eval()
        out[name + '_names'] = np.array(ours.fill_parameters(ref, SEED + k))
        with torch.no_grad():
            if name != 'transformer':
                out[name + '_feat'] = ref(inp, len0, return_feature=True)[1].numpy()
            else:
                a = ref.attn_enc(inp)[0].numpy()                       # [200, 8, 39]: five rows and the sum over time are kept
                out[name + '_attn_rows'] = a[[0, 11, 56, 130, 199]]
                out[name + '_attn_sum'] = a.astype(np.float64).sum(0)
            F.dropout = lambda x, *a, **kw: x
            try:
                lo, feat = ref(inp, len0, return_feature=True)
            finally:
                F.dropout = real_dropout
            out[name + '_logits_nodrop'] = lo.numpy()
            out[name + '_feat_nodrop'] = feat.numpy()
    path = os.path.join(HERE, 'clf_golden.npz')
    np.savez_compressed(path, **out)
    for k_, v in out.items():
        print(k_, getattr(v, 'shape', v))


if __name__ == '__main__':
    main()
