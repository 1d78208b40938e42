"""The gated block of the BERT+GCN classifiers: ``models/bert_amir5.py:621-640``.

Reference (BertAmir55; identical in BertAmir54 ``:512-531``)::

    gate1 = self.gate1(aspect).repeat(1, T).view(x.shape)     # [B,H] -> [B,T,H]
    gate2 = self.gate2(aspect).repeat(1, T).view(x.shape)
    gcn1 = self.gc1(x, adj)
    x1 = max_t(gcn1 * gate1);  y1 = max_t(gcn1 * gate2)
    xy = (x1 * y1).sum(1).mean()
    x = gate2 * self.gc2(gcn1, adj);  out = max_t(x)

Here the gates stay ``[B,H]``; each layer is one linear + one aggregation launch that
also applies the gate and the max over tokens, and the adjacency is converted to CSR
once for both layers.
"""
import torch

from . import _capi
from .csr import BatchedCSR


def gate_overlap(x1, y1):
    """``xy = (x1 * y1).sum(1).mean()`` (``bert_amir5.py:638``) as one device scalar."""
    lib = _capi.load_library()
    B, F = x1.shape
    dev = x1.device
    with torch.cuda.device(dev):
        xy = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.ggcn_overlap_workspace_bytes(B), dtype=torch.uint8, device=dev)
        _capi.check(lib.ggcn_gate_overlap(_capi.ptr(x1), _capi.ptr(y1), B, F, _capi.ptr(xy), _capi.ptr(ws),
                                          _capi.stream_of(dev)), "ggcn_gate_overlap")
    return xy


def gated_gcn_block(x, adj, gate1, gate2, gc1, gc2):
    """x [B,T,H] fp32, adj dense [B,T,T] or BatchedCSR, gate1/gate2 [B,H], gc1/gc2 GraphConvolution.

    Returns the tensors the reference block produces:
    ``gcn1`` (ungated, feeds layer 2), ``x1``, ``y1``, ``xy``, ``x`` (gated layer-2 output), ``out``.
    """
    if x.shape[0] == 0:   # empty batch: what the reference's ops give on empty tensors (the mean of nothing is nan)
        gc1._check(x)
        B, T, F = 0, x.shape[1], gc2.out_features
        z2 = x.new_zeros((0, F), dtype=torch.float32)
        return {"gcn1": x.new_zeros((0, T, gc1.out_features)), "x1": x.new_zeros((0, gc1.out_features)),
                "y1": x.new_zeros((0, gc1.out_features)), "xy": x.new_full((), float("nan")),
                "x": x.new_zeros((0, T, F)), "out": z2}
    csr = adj if isinstance(adj, BatchedCSR) else gc1._as_csr(adj, x)
    training = torch.is_grad_enabled() and (gc1._needs_grad(x, gate1, gate2) or gc2._needs_grad(x, gate2))
    if (not training and gc1.takes_fused_path(x, csr) and gc2.takes_fused_path(x, csr)
            and gc1.out_features == gc2.out_features):
        # two launches in all: layer 1 leaves its share of sum_f x1*y1 per (graph, 64 columns), layer 2's
        # launch adds them up before it starts on its own tiles (:638 costs no launch of its own)
        B, F = x.shape[0], gc1.out_features
        part = torch.empty(B, (F + 63) // 64, dtype=torch.float32, device=x.device)
        xy = torch.empty((), dtype=torch.float32, device=x.device)
        gcn1, x1, y1 = gc1.forward_gated(x, csr, store_gate=None, pool_gate_a=gate1, pool_gate_b=gate2,
                                         want_pool_a=True, want_pool_b=True, overlap_partial=part)   # :626-636
        x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2, want_pool_a=True,
                                       overlap_reduce=(part, xy))                                    # :638-640
        return {"gcn1": gcn1, "x1": x1, "y1": y1, "xy": xy, "x": x2, "out": out}
    gcn1, x1, y1 = gc1.forward_gated(x, csr, store_gate=None, pool_gate_a=gate1, pool_gate_b=gate2,
                                     want_pool_a=True, want_pool_b=True)           # :626-636
    if torch.is_grad_enabled() and (x1.requires_grad or y1.requires_grad):
        xy = (x1 * y1).sum(1).mean()   # differentiable form of :638 (the regulariser is trained on)
    else:
        xy = gate_overlap(x1, y1)                                                  # :638
    x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2,
                                   want_pool_a=True)                               # :639-640
    return {"gcn1": gcn1, "x1": x1, "y1": y1, "xy": xy, "x": x2, "out": out}
