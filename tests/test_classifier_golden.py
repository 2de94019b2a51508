"""features/classifier.py::RNNHead restates the forward pass of the reference's rnn_clf.RNN
(rnn_clf.py:12-34, layers.py:42-76).  tests/golden/rnn_golden.npz holds logits the REAL reference class produced
on the CPU of the build container (tests/golden/make_rnn_golden.py); the stand-in must reproduce them with the
same seeded weights -- which pins its packing, direction sum, unsort, length-normalised average and
zero-row-including max pooling to the reference."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def rnn_golden():
    return np.load(os.path.join(HERE, 'golden', 'rnn_golden.npz'))


def _head(seed, names):
    import torch
    from features.classifier import RNNHead, fill_parameters
    torch.manual_seed(0)
    head = RNNHead().eval()
    assert fill_parameters(head, int(seed)) == [str(n) for n in names]     # same parameters, same order as the reference
    return head


def test_standin_reproduces_reference_logits_on_cpu(rnn_golden):
    import torch
    g = rnn_golden
    head = _head(g['seed'], g['names'])
    with torch.no_grad():
        got = head(torch.from_numpy(g['inp']), g['len0']).numpy()
    assert got.shape == g['logits'].shape == (8, 20)
    assert np.max(np.abs(got - g['logits'])) <= 1e-5 * max(1.0, float(np.max(np.abs(g['logits']))))
    # the detail a masked rewrite gets wrong: utterance 4 has negative features only in column 0 ... its max pool is
    # taken over the zero rows behind its 12 frames too (rnn_clf.py:31)
    assert g['len0'][4] == 12


@pytest.mark.gpu
def test_standin_reproduces_reference_logits_on_rocm(rnn_golden):
    import torch
    g = rnn_golden
    dev = torch.device('cuda', 0)
    head = _head(g['seed'], g['names']).to(dev)
    with torch.no_grad():
        got = head(torch.from_numpy(g['inp']).to(dev), g['len0']).cpu().numpy()
    assert np.max(np.abs(got - g['logits'])) <= 1e-4 * max(1.0, float(np.max(np.abs(g['logits']))))


# ---- HRNN, HRNN_Att, Transformer (rnn_clf.py:36-120,166-203): tests/golden/clf_golden.npz from the real classes ------------------

@pytest.fixture(scope='module')
def clf_golden():
    return np.load(os.path.join(HERE, 'golden', 'clf_golden.npz'))


def _make(kind, g):
    import torch
    from features import classifier as C
    cls, k = {'hrnn': (C.HRNNHead, 0), 'hrnn_att': (C.HRNNAttHead, 1), 'transformer': (C.TransformerHead, 2)}[kind]
    torch.manual_seed(0)
    head = cls().eval()
    names = C.fill_parameters(head, int(g['seed']) + k)
    assert names == [str(n) for n in g[kind + '_names']], kind      # same parameters, same names, same order as the reference
    return head


def _check(kind, g, rnn_g, dev, tol):
    import torch
    head = _make(kind, g).to(dev)
    inp = torch.from_numpy(rnn_g['inp']).to(dev)
    scale = lambda a: max(1.0, float(np.max(np.abs(a))))
    with torch.no_grad():
        res = head(inp, rnn_g['len0'], dropout=False)
        lo, feat = res[0].cpu().numpy(), res[1].cpu().numpy()
        assert np.max(np.abs(feat - g[kind + '_feat_nodrop'])) <= tol * scale(g[kind + '_feat_nodrop'])
        assert np.max(np.abs(lo - g[kind + '_logits_nodrop'])) <= tol * scale(g[kind + '_logits_nodrop'])
        if kind == 'transformer':
            a = res[2].cpu().numpy()
            assert np.max(np.abs(a[[0, 11, 56, 130, 199]] - g['transformer_attn_rows'])) <= tol * scale(g['transformer_attn_rows'])
            assert np.max(np.abs(a.astype(np.float64).sum(0) - g['transformer_attn_sum'])) <= 200 * tol * scale(g['transformer_attn_rows'])
        else:       # the reference as it stands: features are in front of its always-on dropout
            assert np.max(np.abs(feat - g[kind + '_feat'])) <= tol * scale(g[kind + '_feat'])
        # the always-on dropout of the reference (rnn_clf.py:73,116,199): a fifth of the logits zeroed, the rest scaled by 1.25
        torch.manual_seed(5)
        lo_d = head(inp, rnn_g['len0'])[0].cpu().numpy()
        kept = lo_d != 0
        assert 0.5 < kept.mean() < 0.98
        if kind != 'transformer':       # (the Transformer's logits also see the dropout behind the attention block)
            assert np.allclose(lo_d[kept], 1.25 * lo[kept], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('kind', ['hrnn', 'hrnn_att', 'transformer'])
def test_hierarchical_and_transformer_standins_reproduce_the_reference_on_cpu(kind, clf_golden, rnn_golden):
    import torch
    _check(kind, clf_golden, rnn_golden, torch.device('cpu'), 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['hrnn', 'hrnn_att', 'transformer'])
def test_hierarchical_and_transformer_standins_reproduce_the_reference_on_rocm(kind, clf_golden, rnn_golden):
    import torch
    _check(kind, clf_golden, rnn_golden, torch.device('cuda', 0), 2e-4)
