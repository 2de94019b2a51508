"""PyTorch-ROCm stand-ins of the reference's recurrent classifiers (RNN, HRNN, HRNN_Att, Transformer), for SURVEY row f-3 (device-resident
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


class _DynEnc(nn.Module):
    """layers.DynamicEncoder (layers.py:42-76): sort by length (np.argsort(-lens): stable), pack, bidirectional GRU, pad back
    to max(lens) rows (zeros behind each end), forward + backward halves SUMMED, unsort.  Parameter names ``gru.*``."""

    def __init__(self, input_size, hidden_size, n_layers, dropout=0.0):
        super().__init__()
        self.hidden_size = hidden_size
        self.gru = nn.GRU(input_size, hidden_size, n_layers, dropout=dropout, bidirectional=True)

    def forward(self, x, lens):
        lens = torch.as_tensor(np.asarray(lens.cpu() if torch.is_tensor(lens) else lens), dtype=torch.int64)
        order = torch.argsort(lens, descending=True, stable=True)
        unsort = torch.argsort(order).to(x.device)
        packed = nn.utils.rnn.pack_padded_sequence(x[:, order.to(x.device)], lens[order])
        y, _ = self.gru(packed)
        y, _ = nn.utils.rnn.pad_packed_sequence(y)
        h = self.hidden_size
        return (y[:, :, :h] + y[:, :, h:])[:, unsort].contiguous()


def _pool_head(y, lens, out, attn=None):
    """The pooled features of rnn_clf.py:68-72 / 107-115: sum over time / length (or the self-attention vector) || max over
    time -- both over the zero rows pad_packed_sequence leaves behind shorter utterances -- and the pre-dropout logits."""
    n = torch.as_tensor(np.asarray(lens), dtype=y.dtype, device=y.device)
    first = attn(y) if attn is not None else y.sum(0) / n.unsqueeze(1)
    feat = torch.cat([first, y.max(0).values], dim=1)
    return out(feat), feat


class HRNNHead(nn.Module):
    """rnn_clf.HRNN (rnn_clf.py:36-77): 2-layer bidirectional GRU at the frame rate, every ``hir``-th output row
    (t = 0, hir, 2 hir, .. of max(len0) rows) into a 1-layer bidirectional GRU with lengths ceil(len0 / hir), pooled.
    The reference applies F.dropout(out, 0.2) to the logits in EVERY mode (training=True is F.dropout's default,
    rnn_clf.py:73): ``dropout=True`` does the same, ``False`` returns the logits in front of it.  forward -> (logits, feat)."""
    hir = 10

    def __init__(self, feat_size=39):
        super().__init__()
        self.hidden_size = 200
        self.enc1 = _DynEnc(feat_size, 200, 2, dropout=0.2)
        self.enc2 = _DynEnc(200, 200, 1)
        self.out = nn.Linear(400, 20)

    def _levels(self, inp, len0):
        len0 = np.asarray(len0.cpu() if torch.is_tensor(len0) else len0)
        len1 = (len0 + self.hir - 1) // self.hir                      # rnn_clf.py:52
        y = self.enc1(inp, len0)[:, :, -self.hidden_size:]            # rnn_clf.py:58
        return self.enc2(y[0::self.hir], len1), len1                  # rnn_clf.py:61-65

    def forward(self, inp, len0, dropout=True):
        y2, len1 = self._levels(inp, len0)
        out, feat = _pool_head(y2, len1, self.out)
        return (torch.nn.functional.dropout(out, 0.2) if dropout else out), feat


class _SelfAttn(nn.Module):
    """layers.SelfAttn (layers.py:78-95): softmax over ALL rows of the padded output, zero rows included."""

    def __init__(self, hidden_size):
        super().__init__()
        self.attn = nn.Linear(hidden_size, hidden_size)
        self.v = nn.Linear(hidden_size, 1)

    def forward(self, y):
        y = y.transpose(0, 1)                                                   # [B, T, H]
        w = torch.softmax(self.v(torch.tanh(self.attn(y))).squeeze(2), 1)       # [B, T]
        return torch.bmm(w.unsqueeze(1), y).squeeze(1)


class HRNNAttHead(HRNNHead):
    """rnn_clf.HRNN_Att (rnn_clf.py:79-120): hir = 5, the attention vector in place of the average pool.  Parameter order
    as in the reference, whose __init__ re-assigns enc1 / enc2 / out created by HRNN.__init__ and appends attn."""
    hir = 5

    def __init__(self, feat_size=39):
        super().__init__(feat_size)
        self.attn = _SelfAttn(200)

    def forward(self, inp, len0, dropout=True):
        y2, len1 = self._levels(inp, len0)
        out, feat = _pool_head(y2, len1, self.out, self.attn)
        return (torch.nn.functional.dropout(out, 0.2) if dropout else out), feat


class _LayerNorm(nn.Module):
    """layers.LayerNormalization (layers.py:125-143): unbiased std, eps added to sigma, and NO normalisation at all when
    the SECOND axis has one element."""

    def __init__(self, d, eps=1e-3):
        super().__init__()
        self.eps = eps
        self.a_2 = nn.Parameter(torch.ones(d))
        self.b_2 = nn.Parameter(torch.zeros(d))

    def forward(self, z):
        if z.size(1) == 1:
            return z
        mu, sigma = z.mean(-1, keepdim=True), z.std(-1, keepdim=True)
        return (z - mu) / (sigma + self.eps) * self.a_2 + self.b_2


class _MHA(nn.Module):
    """transformer.MultiHeadAttention + ScaledDotProductAttention (transformer.py:32-121) with the reference's quirks:
    tanh on the queries, temperature sqrt(d_model), and ``nn.Softmax()`` WITHOUT a dim on a 3-D tensor, i.e. a softmax over
    axis 0 -- across the (head x utterance) axis, not over the keys."""

    def __init__(self, n_head, d_model, d_k, d_v):
        super().__init__()
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Parameter(torch.empty(n_head, d_model, d_k))
        self.w_ks = nn.Parameter(torch.empty(n_head, d_model, d_k))
        self.w_vs = nn.Parameter(torch.empty(n_head, d_model, d_v))
        self.layer_norm = _LayerNorm(d_model)
        self.proj = nn.Linear(n_head * d_v, d_model)
        for w in (self.w_qs, self.w_ks, self.w_vs):
            nn.init.xavier_normal_(w)

    def forward(self, x):                                       # x: [T, B, d_model]
        q = x.transpose(0, 1)                                   # [B, T, d]
        B, T, d = q.shape
        rep = q.repeat(self.n_head, 1, 1).view(self.n_head, -1, d)
        qs = torch.bmm(rep, self.w_qs).view(-1, T, self.d_k)
        ks = torch.bmm(rep, self.w_ks).view(-1, T, self.d_k)
        vs = torch.bmm(rep, self.w_vs).view(-1, T, self.d_v)
        att = torch.bmm(torch.tanh(qs), ks.transpose(1, 2)) / float(np.power(d, 0.5))
        att = torch.softmax(att, 0)                             # transformer.py:40,58: implicit dim of a 3-D input is 0
        o = torch.bmm(att, vs)
        o = torch.cat(torch.split(o, B, dim=0), dim=-1)
        return self.layer_norm(self.proj(o) + q).transpose(0, 1)


class _PosFFN(nn.Module):
    """transformer.PositionwiseFeedForward (transformer.py:123-139): two 1 x 1 convolutions = two per-position linear maps."""

    def __init__(self, d, d_inner):
        super().__init__()
        self.w_1 = nn.Conv1d(d, d_inner, 1)
        self.w_2 = nn.Conv1d(d_inner, d, 1)
        self.layer_norm = _LayerNorm(d)

    def forward(self, x):
        o = self.w_2(torch.relu(self.w_1(x.transpose(1, 2)))).transpose(2, 1)
        return self.layer_norm(o + x)


class _TransformerEncoder(nn.Module):
    def __init__(self, d_model, d_inner, n_head, d_k, d_v):
        super().__init__()
        self.slf_attn = _MHA(n_head, d_model, d_k, d_v)
        self.pos_ffn = _PosFFN(d_model, d_inner)

    def forward(self, x):
        return self.pos_ffn(self.slf_attn(x))


class TransformerHead(nn.Module):
    """rnn_clf.Transformer (rnn_clf.py:166-203): one self-attention block over the [T, B, 39] input, its output -- behind an
    F.dropout(.., 0.5) that is active in every mode (rnn_clf.py:184) -- concatenated with the input into a bidirectional GRU,
    every 5th row into a second one, pooled.  ``dropout=False`` leaves both F.dropout calls out.  -> (logits, feat, attn_out)."""
    hir = 5

    def __init__(self):
        super().__init__()
        self.attn_enc = _TransformerEncoder(39, 200, 1, 100, 100)
        self.rnn_enc_1 = _DynEnc(78, 200, 1, dropout=0.2)
        self.rnn_enc_2 = _DynEnc(200, 200, 1)
        self.out = nn.Linear(400, 20)
        self.hidden_size = 200

    def forward(self, inp, len0, dropout=True):
        len0 = np.asarray(len0.cpu() if torch.is_tensor(len0) else len0)
        len1 = (len0 + self.hir - 1) // self.hir
        attn_out = self.attn_enc(inp)
        a = torch.nn.functional.dropout(attn_out, 0.5) if dropout else attn_out
        # (the reference concatenates all 200 padded rows; the packed GRU then reads max(len0) of them)
        y = self.rnn_enc_1(torch.cat([inp, a], 2), len0)[:, :, -self.hidden_size:]
        y2 = self.rnn_enc_2(y[0::self.hir], len1)
        out, feat = _pool_head(y2, len1, self.out)
        return (torch.nn.functional.dropout(out, 0.2) if dropout else out), feat, attn_out


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
