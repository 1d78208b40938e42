"""MI355X-native gated graph convolution: drop-in for the GCN hot path of laiviet/ed-gated-gcn.

The directory carries the reference's (hyphenated) name; import it as
``ed_gated_gcn_amd`` (the one-file alias at the repository root).

Public surface (mirrors the reference interface for this path):

* ``GraphConvolution``  -- ``models/gcn.py:9-45``: same constructor, parameters,
  ``state_dict`` and ``forward(text, adj)``.
* ``gated_gcn_block``   -- ``models/bert_amir5.py:621-640``: gate -> gc1 -> gate ->
  gc2 -> gate -> max-pool, never materialising the [B,T,H] gates.
* ``BatchedCSR``        -- many sentence graphs as one block-diagonal CSR.
* ``subword_pool``      -- ``models/bert_amir5.py:600``: ``bmm(transform, x)`` on the non-zeros only.

All compute is in ``libggcn_hip.so`` (hand-written HIP for gfx950, C ABI in
``include/ggcn.h``).  There is no CPU or PyTorch fallback: without the library or
a GPU tensor every entry point raises.
"""
from ._capi import lib_path, load_library  # noqa: F401
from .csr import BatchedCSR  # noqa: F401
from .gcn import GraphConvolution  # noqa: F401
from .gated_block import gated_gcn_block  # noqa: F401
from .pooling import subword_pool  # noqa: F401
from .heads import dense_head, gate_mlps, scores_and_kl  # noqa: F401
from .classifier import GatedGCNEventDetector, GatedGCNEventDetector54, GCNEventDetectorNoGate, LegacyBertAdapter  # noqa: F401

__all__ = ["GraphConvolution", "gated_gcn_block", "BatchedCSR", "subword_pool", "gate_mlps", "scores_and_kl", "GatedGCNEventDetector", "GatedGCNEventDetector54", "GCNEventDetectorNoGate", "LegacyBertAdapter",
           "load_library", "lib_path"]
