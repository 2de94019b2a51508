"""PyTorch-ROCm stand-in of the reference's vanilla recurrent classifier, for SURVEY row f-3 (device-resident
features feed the RNN without a host round trip) and for the configs[4] throughput figure of bench.py.

The reference's own classes (rnn_clf.py, layers.py) run unchanged on PyTorch-ROCm; they are not part of this
package and do not travel to the GPU box.  This module restates the forward pass of ``rnn_clf.RNN``
(rnn_clf.py:12-34) over ``layers.DynamicEncoder`` (layers.py:42-76) so that the tests can pin it to logits the
REAL reference class produced (tests/golden/make_rnn_golden.py -> tests/golden/rnn_golden.npz):

    sort by length -> pack_padded_sequence -> 3-layer bidirectional GRU(39 -> 200) -> pad_packed_sequence
    -> forward + backward halves summed -> unsort -> sum over time / length  ||  max over time -> Linear(400 -> 20)

Two details of the reference that a "masked pooling" rewrite would get wrong, kept here on purpose:
the time axis of the GRU output is max(len0), not the padded 200, and the max pooling runs over the zero rows
``pad_packed_sequence`` leaves behind shorter utterances (an utterance whose activations are all negative pools to 0).
"""
import numpy as np
import torch
from torch import nn


class RNNHead(nn.Module):
    """Same parameter names, shapes and forward pass as the reference's ``RNN`` (``enc.gru.*``, ``out.*``)."""

    class _Enc(nn.Module):
        def __init__(self, input_size, hidden_size, n_layers):
            super().__init__()
            self.hidden_size = hidden_size
            self.gru = nn.GRU(input_size, hidden_size, n_layers, dropout=0.0, bidirectional=True)

    def __init__(self, feat_size=39, hidden=200, layers=3, classes=20):
        super().__init__()
        self.enc = RNNHead._Enc(feat_size, hidden, layers)
        self.out = nn.Linear(2 * hidden, classes)

    def forward(self, inp, len0):
        """inp: [T, B, 39] (zero beyond each utterance's length), len0: lengths (numpy / list / tensor) -> [B, 20]."""
        lens = torch.as_tensor(np.asarray(len0.cpu() if torch.is_tensor(len0) else len0), dtype=torch.int64)
        order = torch.argsort(lens, descending=True, stable=True)          # layers.py:64 np.argsort(-input_lens)
        unsort = torch.argsort(order).to(inp.device)
        packed = nn.utils.rnn.pack_padded_sequence(inp[:, order.to(inp.device)], lens[order])     # layers.py:70
        y, _ = self.enc.gru(packed)
        y, _ = nn.utils.rnn.pad_packed_sequence(y)                          # [max(len0), B, 2H], zeros behind each end
        h = self.enc.hidden_size
        y = (y[:, :, :h] + y[:, :, h:])[:, unsort]                          # layers.py:73-74
        avg = y.sum(0) / lens.to(inp.device, inp.dtype).unsqueeze(1)        # rnn_clf.py:29-30
        mx = y.max(0).values                                                # rnn_clf.py:31 (over the padded rows too)
        return self.out(torch.cat([avg, mx], dim=1))


def fill_parameters(module, seed):
    """Deterministic weights for parity fixtures: every parameter, in ``named_parameters()`` order, drawn from
    numpy's ``default_rng(seed)`` as uniform(-0.08, 0.08) float32 -- the same call fills the reference class in
    tests/golden/make_rnn_golden.py, so both sides hold identical weights without a multi-megabyte fixture."""
    rng = np.random.default_rng(seed)
    names = []
    with torch.no_grad():
        for name, p in module.named_parameters():
            v = rng.uniform(-0.08, 0.08, size=tuple(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(v))
            names.append(name)
    return names
