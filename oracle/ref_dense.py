"""CPU ORACLE (test infrastructure, NOT product code).

A restatement, in plain PyTorch-CPU ops, of the one hot path this repository
accelerates.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; the product package
(``ed-gated-gcn_amd/``) never does and fails loudly without its HIP library.

Parity is PINNED: ``oracle/make_golden.py`` imports the reference classes in the
build container, runs them on seeded inputs and commits the tensors under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against those fixtures (bit-equal for the dense layer, see the test).

What each function follows (paths relative to the reference checkout):

* ``graph_convolution``      -> ``models/gcn.py:30-45``  (GraphConvolution.forward)
* ``gated_block``            -> ``models/bert_amir5.py:621-640`` (BertAmir55; the
  same block is at ``:512-531`` in BertAmir54)
* ``reset_params_like_train``-> ``train.py:75-84`` (Instructor._reset_params)
"""
import math

import torch


def graph_convolution(text, adj, weight, bias=None):
    """``models/gcn.py:30-45``.

    text [B,T,Din] fp32, adj [B,T,T] any real dtype (non-binary values act as
    edge weights), weight [Din,Dout] (note: in x out, not nn.Linear's layout),
    bias [Dout] or None.  Same op order as the reference: linear first, then
    the dense adjacency product, then the division by (row-sum + 1), then bias.
    """
    adj = adj.float()                                   # gcn.py:33
    hidden = torch.matmul(text, weight)                 # gcn.py:34
    denom = torch.sum(adj, dim=2, keepdim=True) + 1     # gcn.py:35
    output = torch.matmul(adj, hidden) / denom          # gcn.py:41
    if bias is not None:
        return output + bias                            # gcn.py:43
    return output                                       # gcn.py:45


def gated_block(x, adj, gate1, gate2, w1, b1, w2, b2):
    """``models/bert_amir5.py:621-640`` in eval mode (dropout = identity).

    x [B,T,H]; gate1/gate2 [B,H] (the reference materialises them as [B,T,H]
    with ``.repeat(1,T).view(x.shape)``, ``:621-622`` -- a broadcast over
    tokens).  Returns a dict with every tensor the block produces:
    gcn1 (ungated, feeds layer 2), x1, y1, xy, x (gated layer-2 output), out.
    """
    B, T, H = x.shape
    g1 = gate1.repeat(1, T).view(B, T, -1)              # :621
    g2 = gate2.repeat(1, T).view(B, T, -1)              # :622
    gcn1 = graph_convolution(x, adj, w1, b1)            # :626
    gcngate1 = gcn1 * g1                                # :627
    gcngate2 = gcn1 * g2                                # :631
    x1 = torch.max(gcngate1, 1)[0]                      # :635
    y1 = torch.max(gcngate2, 1)[0]                      # :636
    xy = (x1 * y1).sum(1).mean()                        # :638
    x2 = g2 * graph_convolution(gcn1, adj, w2, b2)      # :639
    out = torch.max(x2, dim=1)[0]                       # :640
    return {"gcn1": gcn1, "x1": x1, "y1": y1, "xy": xy, "x": x2, "out": out}


def reset_params_like_train(params, generator=None):
    """``train.py:75-84``: xavier_uniform_ on >=2-D, U(+-1/sqrt(shape[0])) on 1-D."""
    for p in params:
        if p.dim() > 1:
            fan_in, fan_out = p.shape[0], p.shape[1]
            a = math.sqrt(6.0 / (fan_in + fan_out))
            with torch.no_grad():
                p.uniform_(-a, a, generator=generator)
        else:
            stdv = 1.0 / math.sqrt(p.shape[0])
            with torch.no_grad():
                p.uniform_(-stdv, stdv, generator=generator)
