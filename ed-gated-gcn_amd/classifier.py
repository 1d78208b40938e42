"""The classifier around the gated-GCN block: ``BertAmir55`` (``models/bert_amir5.py:544-650``)
with the block of lines 621-640 running on the HIP path.

Sub-module names, shapes and construction order follow the reference, so a reference
``state_dict`` loads unchanged (``bert.*``, ``dense``, ``lstm``, ``gc1``, ``gc2``, ``gate1``,
``gate2``, ``fc``) and ``Instructor._reset_params`` (``train.py:75-84``) initialises it the same
way.  ``forward(inputs) -> (logits, gate_reg, kl_reg, scores)`` like every live model of
``train.py:109``.  The sub-word pooling (``:600``), the gate MLPs (``:562-571``), the block (``:621-640``) and the
``scores`` / ``kl`` head (``:645-648``) run on the HIP path in inference; under autograd the gate MLPs and the
head keep the reference's PyTorch ops (they are trained), and with dropout active the gates are dropped per
token like the reference's (``:621-625``).  BERT, the BiLSTM and ``dense`` stay PyTorch-ROCm.
"""
import os

import torch
import torch.nn as nn

from .csr import tensor_version
from .gated_block import gated_gcn_block, takes_block_path
from .gcn import GraphConvolution
from .heads import gate_mlps, scores_and_kl
from .pooling import subword_pool


class LegacyBertAdapter(nn.Module):
    """``pytorch_pretrained_bert`` call convention (``bert(ids, seg, output_all_encoded_layers=True)
    -> (list of 12 x [B,L,768], pooled)``, ``models/bert_amir5.py:591-593``: the second positional
    argument is token_type_ids, no attention mask is passed) on top of a ``transformers.BertModel``."""

    def __init__(self, hf_bert):
        super().__init__()
        self.model = hf_bert

    def forward(self, input_ids, token_type_ids=None, output_all_encoded_layers=True):
        o = self.model(input_ids=input_ids, token_type_ids=token_type_ids, output_hidden_states=True,
                       return_dict=True)
        layers = list(o.hidden_states[1:])   # the 12 encoder layers (index 0 is the embedding output)
        return (layers if output_all_encoded_layers else layers[-1]), o.pooler_output


class GatedGCNEventDetector(nn.Module):
    """``BertAmir55`` (VARIANT "55").  Subclasses mirror the other live models of ``train.py:268-282`` around the same
    block: ``GatedGCNEventDetector54`` = ``BertAmir54`` (``bert_amir5.py:434``: two-layer ``dense`` on [aspect, out,
    pooled], a Sigmoid in front of ``fc``), ``GCNEventDetectorNoGate`` = ``BertAmir55NoGate`` (``:654``: the two layers
    without gates; ``xy`` = 0.0).  Each loads its reference ``state_dict`` unchanged."""
    VARIANT = "55"

    def __init__(self, bert, opt):
        super().__init__()                                                  # bert_amir5.py:545-571 / :435-467 / :655-683
        self.device = getattr(opt, "device", None)
        self.bert = bert
        self.dropout = nn.Dropout(opt.dropout)
        self.hidden_dim = hd = 128
        self.n_layer = 12
        if self.VARIANT == "54":
            self.dense = nn.Sequential(nn.Linear(2 * 2 * hd + 768, 768), nn.Linear(768, opt.polarities_dim))   # :444-447
        else:
            self.dense = nn.Linear(2 * 2 * hd + 768 * self.n_layer, opt.polarities_dim)
        self.lstm = nn.LSTM(self.n_layer * 768, hd, bidirectional=True, batch_first=True, num_layers=1)
        self.gc1 = GraphConvolution(2 * hd, 2 * hd, opt)
        self.gc2 = GraphConvolution(2 * hd, 2 * hd, opt)
        self.gate1 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        self.gate2 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        if self.VARIANT == "54":
            self.fc = nn.Sequential(nn.Sigmoid(), nn.Linear(2 * 2 * hd, opt.polarities_dim))                  # :466-467
        else:
            self.fc = nn.Sequential(nn.Linear(2 * 2 * hd, opt.polarities_dim))
        # no precision asked for (opt.ggcn_precision / GGCN_PRECISION): inference picks the faster "f16mx8" whenever the
        # weights PROVE its fp16 range sufficient for this model (_proved_precision), "bf16x3" otherwise
        self._auto_precision = getattr(opt, "ggcn_precision", None) is None and "GGCN_PRECISION" not in os.environ
        self._assigned = (self.gc1.precision, self.gc2.precision)   # what _assign / the constructor last put there (_auto)
        self._proved = None
        # train.py:227 keeps the logits of an evaluation batch and nothing else; with opt.ggcn_eval_logits_only = True the
        # inference forward computes only what they need -- `out` (bert_amir5.py:640,643): the W12 tiles of the block, no
        # [B,T,H] store of x, no regulariser, no scores / kl head -- and returns (logits, None, None, None)
        self.eval_logits_only = bool(getattr(opt, "ggcn_eval_logits_only", False))

    F16_RANGE_MARGIN = 32752.0   # half of fp16's largest finite value
    F16MX8_WINDOW = 448.0        # include/ggcn.h GGCN_RANGE_WINDOW: beyond it an activation's fp8 correction saturates

    def _auto(self):
        """Still choosing the layers' precision ourselves?  A precision assigned to gc1 / gc2 after construction
        (``model.gc1.precision = "fp32"``) is the user's explicit choice from then on, like opt.ggcn_precision."""
        if self._auto_precision and (self.gc1.precision, self.gc2.precision) != self._assigned:
            self._auto_precision = False
        return self._auto_precision

    def _assign(self, precision):
        self.gc1.precision = self.gc2.precision = precision
        self._assigned = (precision, precision)

    def _proved_precision(self, csr, x=None):
        """"f16mx8" when every value that kernel rounds to fp16 is bounded below ``F16_RANGE_MARGIN`` by the weights alone
        AND every activation an f16mx8 main loop splits stays inside its accuracy window (``F16MX8_WINDOW``: the sticky
        range flag trips beyond it, ``range_guard`` raises) -- else "bf16x3".  The block's input is |x| < 1; where the block
        does not run as one launch (graphs of more than 32 nodes: LitBank, ACE cased) gc2's main loop splits gcn1, so its
        bound ``m1`` must stay inside the window too.  The block's input is the BiLSTM output (``:610``), |x| < 1 (o * tanh(c)); with a 0/1 adjacency a
        layer's output is a mean of hidden rows plus the bias (``gcn.py:35,41,43``), so
        ``|x.W1| <= c1 = colsum|W1|``, ``|gcn1| <= m1 = max(c1 + |b1|)``, ``|gcn1.W2| <= m1 * colsum|W2|``, and for the one-launch
        block ``|x.W12| + |mid| <= colsum|W1.W2| + |b1.W2|``.  One host read per weight update (cached on the parameters'
        version counters); weighted adjacencies and inference tensors (no version counter) keep "bf16x3"."""
        if not csr.is_binary:
            return "bf16x3"
        ps = (self.gc1.weight, self.gc1.bias, self.gc2.weight, self.gc2.bias)
        key = tuple(None if q is None else (q.data_ptr(), tensor_version(q)) for q in ps)
        if any(k is not None and k[1] is None for k in key):
            return "bf16x3"
        if self._proved is None or self._proved[0] != key:
            with torch.no_grad():
                w1, w2 = self.gc1.weight.detach().float(), self.gc2.weight.detach().float()
                b1 = torch.zeros_like(w1[0]) if self.gc1.bias is None else self.gc1.bias.detach().float()
                c1 = w1.abs().sum(0)
                m1 = (c1 + b1.abs()).max()
                w12 = w1 @ w2
                bound = torch.stack([m1.new_tensor(1.0), m1, m1 * w2.abs().sum(0).max(),
                                     (w12.abs().sum(0) + (b1 @ w2).abs()).max()]).max()
                ok = bool(torch.isfinite(bound)) and float(bound) < self.F16_RANGE_MARGIN
            self._proved = (key, ok, float(m1))
        _, ok, m1 = self._proved
        if ok and not (m1 <= self.F16MX8_WINDOW):
            # gcn1 may leave the window: fine only where no main loop ever splits it -- the one-launch block
            before = (self.gc1.precision, self.gc2.precision)
            self.gc1.precision = self.gc2.precision = "f16mx8"
            try:
                ok = x is not None and takes_block_path(x, csr, self.gc1, self.gc2)
            finally:
                self.gc1.precision, self.gc2.precision = before
        return "f16mx8" if ok else "bf16x3"

    def forward(self, inputs):
        B = inputs["sentence_length"].shape[0]                              # :579-589
        L = int(inputs["cls_text_sep_length"].max())                        # one host sync, as :580-581
        T = int(inputs["sentence_length"].max())
        ids = inputs["cls_text_sep_indices"][:, :L]
        seg = inputs["cls_text_sep_segments_ids"][:, :L]
        transform = inputs["transform"][:, :T, :L]
        anchor = inputs["anchor_index"]
        dist = inputs["dist_to_target"][:, :T]
        adj = inputs["dependency_graph"]                                    # :589 dense slice, or a BatchedCSR
        if isinstance(adj, torch.Tensor):
            adj = adj[:, :T, :T]
        x, pooled = self.bert(ids, seg, output_all_encoded_layers=True)     # :591
        x = torch.cat(x[-self.n_layer:], dim=-1)                            # :596
        x = subword_pool(transform.float(), x)                              # :600 on the HIP path (non-zeros only)
        rows = torch.arange(B, device=x.device)
        anchor_rep = self.dropout(x[rows, anchor])                          # :604-608: the anchor token's row
        x, _ = self.lstm(x)                                                 # :610
        aspect = x[rows, anchor]                                            # :615-618
        x = x.contiguous()
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        dropping = self.training and self.dropout.p > 0     # the reference drops whenever the module is in training mode
        v54, nogate = self.VARIANT == "54", self.VARIANT == "55nogate"
        if not grad and not dropping:
            # ---- inference: everything from `aspect` to `scores` on the HIP path, gates kept [B,H] ----
            if self._auto():
                if isinstance(adj, torch.Tensor):
                    adj = self.gc1._as_csr(adj, x)   # (the block would convert it anyway; identity-cached)
                self._assign(self._proved_precision(adj, x))
            want = ("out",) if self.eval_logits_only else None     # train.py:227: logits only -> only `out` of the block
            if nogate:   # :736-752: gc2(gc1(x)) and its max-pool -- the block with unit gates (one launch for T <= 32)
                ones = x.new_ones(B, 2 * self.hidden_dim)
                r = gated_gcn_block(x, adj, ones, ones, self.gc1, self.gc2, want=want)
                xy = 0.0
            else:
                gate1, gate2 = gate_mlps(aspect.contiguous(), self.gate1, self.gate2)     # :562-571,621-622, one launch
                r = gated_gcn_block(x, adj, gate1, gate2, self.gc1, self.gc2, want=want)  # :626-640, one launch (T <= 32)
                xy = r["xy"]
            if self.eval_logits_only:
                cat = [aspect, r["out"], pooled] if v54 else [anchor_rep, aspect, r["out"]]
                return self.dense(torch.cat(cat, dim=1)), None, None, None               # :533 / :643
            if v54:      # :531-536: dense on [aspect, out, pooled]; fc = Linear o Sigmoid, and sigmoid(cat) = cat(sigmoid)
                logits = self.dense(torch.cat([aspect, r["out"], pooled], dim=1))
                scores, kl = scores_and_kl(torch.sigmoid(r["x"]), torch.sigmoid(aspect), logits, self.fc[1], dist)
            else:
                logits = self.dense(torch.cat([anchor_rep, aspect, r["out"]], dim=1))     # :642-643 (dropout = identity)
                scores, kl = scores_and_kl(r["x"], aspect, logits, self.fc[0], dist)      # :645-648, one launch
            return logits, xy, kl, scores
        if self._auto():
            self._assign("bf16x3")   # training: the full-range default
        csr = adj if not isinstance(adj, torch.Tensor) else self.gc1._as_csr(adj, x)
        if nogate:
            gcn1 = self.gc1(x, csr)                                                    # :736
            xg, out, _ = self.gc2.forward_gated(gcn1, csr, want_pool_a=True)           # :748-749
            xy = 0.0
        elif dropping and self.gc1.takes_dropout_path(x, csr) and self.gc2.takes_dropout_path(x, csr):
            # ---- training with dropout on the one-launch path (graphs of <= 256 nodes): the reference drops entries of the REPEATED [B,T,H] gates
            # (:621-625), one draw per token and feature.  The layers draw those keep factors in their own epilogues from a
            # counter-based hash of (seed, element) -- stream 1 for gate1, stream 2 for gate2 in BOTH layers, as the
            # reference's one dropped copy of gate2 serves :631 and :639 -- and again in the backward pass: nothing of
            # size [B,T,H] is materialised for the gates, and the pools stay in the layer launches. ----
            p = float(self.dropout.p)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # CPU generator: follows torch.manual_seed, no device sync
            g1, g2 = self.gate1(aspect).contiguous(), self.gate2(aspect).contiguous()
            gcn1, x1, y1 = self.gc1.forward_gated(x, csr, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True,
                                                  dropout=(p, seed, (0, 1, 2)))          # :626-636
            xy = (x1 * y1).sum(1).mean()                                               # :638
            xg, out, _ = self.gc2.forward_gated(gcn1, csr, store_gate=g2, pool_gate_a=g2, want_pool_a=True,
                                                dropout=(p, seed, (2, 2, 0)))            # :639-640
            if pooled is not None and not v54:
                self.dropout(pooled)                                                   # :641 (unused; keeps the RNG stream)
        elif dropping:
            # ---- a layer off the one-launch path (weighted adjacency, > 256 nodes, ...): gating, dropout and the pools as the reference's own
            # ops around the two HIP layers (three [B,T,H] temporaries) ----
            gate1 = self.dropout(self.gate1(aspect)[:, None, :].expand(-1, T, -1))     # :621-624 (repeat, then dropout)
            gate2 = self.dropout(self.gate2(aspect)[:, None, :].expand(-1, T, -1))
            gcn1 = self.gc1(x, adj)                                                    # :626
            x1 = torch.max(gcn1 * gate1, 1)[0]                                         # :627-635
            y1 = torch.max(gcn1 * gate2, 1)[0]                                         # :631-636
            xy = (x1 * y1).sum(1).mean()                                               # :638
            xg = gate2 * self.gc2(gcn1, adj)                                           # :639
            out = torch.max(xg, dim=1)[0]                                              # :640
            if pooled is not None and not v54:
                self.dropout(pooled)                                                   # :641 (unused; keeps the RNG stream)
        else:
            gate1 = self.gate1(aspect)                                                 # dropout is the identity here
            gate2 = self.gate2(aspect)
            r = gated_gcn_block(x, adj, gate1.contiguous(), gate2.contiguous(), self.gc1, self.gc2)   # :626-640
            xy, xg, out = r["xy"], r["x"], r["out"]
        if v54:
            pooled_d = self.dropout(pooled)                                 # :531
            out = self.dropout(out)                                         # :532
            logits = self.dense(torch.cat([aspect, out, pooled_d], dim=1))  # :533
        else:
            out = self.dropout(out)                                         # :642
            logits = self.dense(torch.cat([anchor_rep, aspect, out], dim=1))    # :643
        output_w = self.fc(torch.cat([xg, aspect[:, None, :].expand(-1, T, -1)], dim=2))   # :645
        scores = (logits[:, None, :] * output_w).sum(2)                     # :646
        kl = (torch.softmax(scores, 1) * torch.softmax(dist.float(), 1)).sum(1).mean()          # :648
        return logits, xy, kl, scores


class GatedGCNEventDetector54(GatedGCNEventDetector):
    """``BertAmir54`` (``models/bert_amir5.py:434-541``)."""
    VARIANT = "54"


class GCNEventDetectorNoGate(GatedGCNEventDetector):
    """``BertAmir55NoGate`` (``models/bert_amir5.py:654-752``)."""
    VARIANT = "55nogate"
