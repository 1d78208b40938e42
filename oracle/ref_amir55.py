"""CPU ORACLE (test infrastructure, NOT product code): restatement of the classifier that
wraps the gated-GCN block, ``BertAmir55`` (``models/bert_amir5.py:544-650``), in plain PyTorch-CPU.

Pinned by ``tests/golden/amir55_full.npz``: ``oracle/make_golden.py`` runs the REFERENCE class
with parameters drawn by the seeded ``train.py:75-84`` procedure and a seeded encoder stand-in;
``tests/test_oracle_golden.py`` rebuilds this restatement with the same seeds and must reproduce
the reference's logits / xy / kl / scores.  Sub-module names and construction order follow the
reference, so ``state_dict`` keys (and seeded initialisation) coincide.
"""
import math

import torch
import torch.nn as nn

from . import ref_dense


class EncoderStandIn(nn.Module):
    """Seeded hidden states in the pytorch_pretrained_bert form used at
    ``models/bert_amir5.py:591-596``: (list of 12 tensors [B,L,768], pooled [B,768])."""

    def __init__(self, seed):
        super().__init__()
        self.seed = seed

    def forward(self, ids, seg, output_all_encoded_layers=True):
        g = torch.Generator().manual_seed(self.seed)
        B, L = ids.shape
        layers = [(torch.randn(B, L, 768, generator=g) * 0.5).to(ids.device) for _ in range(12)]
        return layers, torch.randn(B, 768, generator=g).to(ids.device)


class _GCParams(nn.Module):
    """Parameter holder with GraphConvolution's layout (``models/gcn.py:18,21``)."""

    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fin, fout))
        self.bias = nn.Parameter(torch.empty(fout))

    def forward(self, text, adj):
        return ref_dense.graph_convolution(text, adj, self.weight, self.bias)


def reset_params(module, gen):
    """``train.py:75-84`` applied to one child: xavier_uniform_ on >=2-D, U(+-1/sqrt(n)) on 1-D."""
    for p in module.parameters():
        if not p.requires_grad:
            continue
        with torch.no_grad():
            if p.dim() > 1:
                a = math.sqrt(6.0 / (p.shape[0] + p.shape[1]))
                p.uniform_(-a, a, generator=gen)
            else:
                s = 1.0 / math.sqrt(p.shape[0])
                p.uniform_(-s, s, generator=gen)


class BertAmir55Oracle(nn.Module):
    """VARIANT "55": ``BertAmir55`` (``bert_amir5.py:544-650``); "54": ``BertAmir54`` (``:434-541``: a two-layer ``dense`` on
    [aspect, out, dropout(pooled)], no anchor row, a Sigmoid in front of ``fc``); "55nogate": ``BertAmir55NoGate``
    (``:654-752``: the two layers without gates and pools, ``xy`` = 0.0).  The other live models of ``train.py:268-282``."""
    VARIANT = "55"

    def __init__(self, bert, polarities_dim, dropout=0.25):
        super().__init__()                                                  # bert_amir5.py:545-571 / :435-467 / :655-683
        self.bert = bert
        self.dropout = nn.Dropout(dropout)
        self.hidden_dim = hd = 128
        self.n_layer = 12
        if self.VARIANT == "54":
            self.dense = nn.Sequential(nn.Linear(2 * 2 * hd + 768, 768), nn.Linear(768, polarities_dim))   # :444-447
        else:
            self.dense = nn.Linear(2 * 2 * hd + 768 * self.n_layer, polarities_dim)
        self.lstm = nn.LSTM(self.n_layer * 768, hd, bidirectional=True, batch_first=True, num_layers=1)
        self.gc1 = _GCParams(2 * hd, 2 * hd)
        self.gc2 = _GCParams(2 * hd, 2 * hd)
        self.gate1 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        self.gate2 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        if self.VARIANT == "54":
            self.fc = nn.Sequential(nn.Sigmoid(), nn.Linear(2 * 2 * hd, polarities_dim))                   # :466-467
        else:
            self.fc = nn.Sequential(nn.Linear(2 * 2 * hd, polarities_dim))

    def seeded_init(self, gen):
        for child in self.children():                                       # train.py:75-84
            if child is not self.bert:
                reset_params(child, gen)

    def forward(self, inputs):
        B = inputs["sentence_length"].shape[0]                              # :579-589
        L = int(inputs["cls_text_sep_length"].max())
        T = int(inputs["sentence_length"].max())
        ids = inputs["cls_text_sep_indices"][:, :L]
        seg = inputs["cls_text_sep_segments_ids"][:, :L]
        transform = inputs["transform"][:, :T, :L]
        anchor = inputs["anchor_index"]
        dist = inputs["dist_to_target"][:, :T]
        adj = inputs["dependency_graph"][:, :T, :T]
        x, pooled = self.bert(ids, seg, output_all_encoded_layers=True)     # :591
        x = torch.cat(x[-self.n_layer:], dim=-1)                            # :596
        x = torch.bmm(transform, x)                                         # :600
        rows = torch.arange(B)
        anchor_rep = self.dropout(x[rows, anchor])                          # :604-608 (masked_select of the anchor row)
        x, _ = self.lstm(x)                                                 # :610
        aspect = x[rows, anchor]                                            # :615-618
        if self.VARIANT == "55nogate":                                      # :736-752: no gates, no pools of layer 1
            gcn1 = ref_dense.graph_convolution(x, adj, self.gc1.weight, self.gc1.bias)
            xg = ref_dense.graph_convolution(gcn1, adj, self.gc2.weight, self.gc2.bias)
            r = {"xy": 0.0, "x": xg, "out": torch.max(xg, dim=1)[0]}
        elif self.training and self.dropout.p > 0:
            # :621-625 in the reference's order: repeat to [B,T,H] FIRST, then dropout (one mask per token)
            gate1 = self.dropout(self.gate1(aspect).repeat(1, T).view(x.shape))
            gate2 = self.dropout(self.gate2(aspect).repeat(1, T).view(x.shape))
            gcn1 = ref_dense.graph_convolution(x, adj, self.gc1.weight, self.gc1.bias)       # :626
            x1 = torch.max(gcn1 * gate1, 1)[0]                                                # :627-635
            y1 = torch.max(gcn1 * gate2, 1)[0]                                                # :631-636
            xg = gate2 * ref_dense.graph_convolution(gcn1, adj, self.gc2.weight, self.gc2.bias)   # :639
            r = {"xy": (x1 * y1).sum(1).mean(), "x": xg, "out": torch.max(xg, dim=1)[0]}      # :638,640
            if self.VARIANT != "54":
                self.dropout(pooled)                                                          # :641
        else:
            gate1 = self.gate1(aspect)                                      # :621-622 (dropout is the identity)
            gate2 = self.gate2(aspect)
            r = ref_dense.gated_block(x, adj, gate1, gate2, self.gc1.weight, self.gc1.bias,
                                      self.gc2.weight, self.gc2.bias)       # :626-640
        if self.VARIANT == "54":
            pooled = self.dropout(pooled)                                   # :531
            out = self.dropout(r["out"])                                    # :532
            logits = self.dense(torch.cat([aspect, out, pooled], dim=1))    # :533
        else:
            out = self.dropout(r["out"])                                    # :642
            logits = self.dense(torch.cat([anchor_rep, aspect, out], dim=1))    # :643
        xg = r["x"]
        output_w = self.fc(torch.cat([xg, aspect[:, None, :].expand(-1, T, -1)], dim=2))   # :645
        scores = (logits[:, None, :] * output_w).sum(2)                     # :646
        kl = (torch.softmax(scores, 1) * torch.softmax(dist.float(), 1)).sum(1).mean()   # :648
        return logits, r["xy"], kl, scores


class BertAmir54Oracle(BertAmir55Oracle):
    VARIANT = "54"


class BertAmir55NoGateOracle(BertAmir55Oracle):
    VARIANT = "55nogate"
