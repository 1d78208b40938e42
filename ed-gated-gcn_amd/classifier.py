"""The classifier around the gated-GCN block: ``BertAmir55`` (``models/bert_amir5.py:544-650``)
with the block of lines 621-640 running on the HIP path.

Sub-module names, shapes and construction order follow the reference, so a reference
``state_dict`` loads unchanged (``bert.*``, ``dense``, ``lstm``, ``gc1``, ``gc2``, ``gate1``,
``gate2``, ``fc``) and ``Instructor._reset_params`` (``train.py:75-84``) initialises it the same
way.  ``forward(inputs) -> (logits, gate_reg, kl_reg, scores)`` like every live model of
``train.py:109``.  The sub-word pooling (``:600``) and the block (``:621-640``) run on the HIP path;
BERT, the BiLSTM and the small heads stay PyTorch-ROCm.
"""
import torch
import torch.nn as nn

from .gated_block import gated_gcn_block
from .gcn import GraphConvolution
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
    def __init__(self, bert, opt):
        super().__init__()                                                  # bert_amir5.py:545-571
        self.device = getattr(opt, "device", None)
        self.bert = bert
        self.dropout = nn.Dropout(opt.dropout)
        self.hidden_dim = hd = 128
        self.n_layer = 12
        self.dense = nn.Linear(2 * 2 * hd + 768 * self.n_layer, opt.polarities_dim)
        self.lstm = nn.LSTM(self.n_layer * 768, hd, bidirectional=True, batch_first=True, num_layers=1)
        self.gc1 = GraphConvolution(2 * hd, 2 * hd, opt)
        self.gc2 = GraphConvolution(2 * hd, 2 * hd, opt)
        self.gate1 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        self.gate2 = nn.Sequential(nn.Sigmoid(), nn.Linear(hd * 2, hd * 2), nn.Sigmoid(),
                                   nn.Linear(hd * 2, hd * 2), nn.Sigmoid())
        self.fc = nn.Sequential(nn.Linear(2 * 2 * hd, opt.polarities_dim))

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
        gate1 = self.dropout(self.gate1(aspect))                            # :621-625, kept [B,H] (no repeat)
        gate2 = self.dropout(self.gate2(aspect))
        r = gated_gcn_block(x.contiguous(), adj, gate1.contiguous(), gate2.contiguous(),
                            self.gc1, self.gc2)                             # :626-640 on the HIP path
        out = self.dropout(r["out"])                                        # :642
        logits = self.dense(torch.cat([anchor_rep, aspect, out], dim=1))    # :643
        output_w = self.fc(torch.cat([r["x"], aspect[:, None, :].expand(-1, T, -1)], dim=2))   # :645
        scores = (logits[:, None, :] * output_w).sum(2)                     # :646
        kl = (torch.softmax(scores, 1) * torch.softmax(dist.float(), 1)).sum(1).mean()          # :648
        return logits, r["xy"], kl, scores
