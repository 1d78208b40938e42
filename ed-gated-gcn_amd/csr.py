"""Batched (block-diagonal) CSR: many small sentence graphs as ONE sparse matrix.

The reference ships every sentence's adjacency dense, ``float32 [B,ORI_ML,ORI_ML]``
(``data_utils.py:376,394``), slices it to ``[:, :T, :T]`` (``models/bert_amir5.py:589``)
and multiplies by it densely (``models/gcn.py:41``).  Here the batch is one CSR over
N = B*T nodes with GLOBAL node ids, so a single kernel launch covers the whole
batch.  Padding rows keep their identity self-loop.
"""
import weakref

import torch

from . import _capi

_ADJ_DTYPES = {
    torch.float32: 0, torch.uint8: 1, torch.bool: 1, torch.int32: 2, torch.int64: 3,
    torch.float64: 4, torch.float16: 5,
}


# Conversions of the dense adjacency tensors most recently handed to forward(text, adj), keyed by the tensor
# OBJECT (weak reference) and its version counter: gc1 and gc2 of one classifier forward receive the same
# `adj` (models/bert_amir5.py:589,626,639) and share one conversion -- and one read-back of the weighted flag.
_RECENT = []
_RECENT_MAX = 2   # gc1 and gc2 of ONE forward share a conversion (promised-binary and flag-checked forms of the same
                  # tensor are two entries); older batches must not stay pinned: a cached BatchedCSR keeps its dense
                  # tensor alive (B*T*T floats per entry)
MASK_MAX_T = 256   # include/ggcn.h GGCN_MASK_MAX_T: largest graph the row-mask (one-launch layer) path takes
MASKS_ONLY_MAX_T = 128   # from_dense: up to here the one-launch layer is the default consumer, the CSR arrays are made on demand


def tensor_version(t):
    """The autograd version counter of ``t``, or None for an inference tensor (``torch.inference_mode``): those
    track no version, so nothing keyed on it -- the identity cache, the staleness check -- applies to them."""
    try:
        return None if t.is_inference() else t._version
    except RuntimeError:
        return None


def cached_from_dense(adj, binary=None):
    """``BatchedCSR.from_dense`` with a small identity-keyed cache (see ``_RECENT``)."""
    ver = tensor_version(adj)
    if ver is None:   # no version counter: an in-place edit could not be noticed, so nothing is cached
        return BatchedCSR.from_dense(adj, binary=binary)
    for ref, v, want, csr in _RECENT:
        if ref() is adj and v == ver and want == binary:
            return csr
    csr = BatchedCSR.from_dense(adj, binary=binary)
    _RECENT.insert(0, (weakref.ref(adj), ver, binary, csr))
    del _RECENT[_RECENT_MAX:]
    return csr


class BatchedCSR:
    """rowptr int32[N+1], colidx int32[cap], vals fp32[cap] or None (binary adjacency),
    rowmask uint32-as-int32[N * ceil(T/32)] or None (T <= MASK_MAX_T = 256: bit j%32 of word j/32 of node i =
    edge i<-j), on one GPU.

    Built from a dense adjacency with T <= 128 only the row masks are computed up front (one
    kernel: all the fused layer needs); the CSR arrays are materialised on first access."""

    __slots__ = ("_rowptr", "_colidx", "_vals", "rowmask", "B", "T", "nnz", "is_binary",
                 "_dense", "_dense_version", "_t", "_inv", "_graph_ops", "_graph_ops2", "_graph_ops_t", "_edge_lists", "_graph_ops_w", "__weakref__")

    def __init__(self, rowptr, colidx, vals, B, T, nnz=None, rowmask=None):
        self._rowptr, self._colidx, self._vals, self.rowmask = rowptr, colidx, vals, rowmask
        self.B, self.T, self.nnz = int(B), int(T), nnz
        self.is_binary = vals is None
        self._dense = None   # the dense tensor this CSR was built from (lazy arrays, transposed())
        self._dense_version = None
        self._t = None       # cached CSR of the transposed adjacency (backward pass)
        self._inv = None     # cached 1/(rowsum+1) per node
        self._graph_ops = None   # cached ggcn_graph_operands blocks (T <= 32)
        self._graph_ops2 = None  # cached ggcn_graph_operands2 blocks per plane type (the one-launch block)
        self._graph_ops_t = None  # cached ggcn_graph_operands blocks of the TRANSPOSED row masks (the MFMA backward)
        self._edge_lists = None   # cached ggcn_graph_edge_lists blocks (graphs of 129..256 nodes: the eight-wavefront layer)
        self._graph_ops_w = None  # cached ggcn_graph_operands_weighted blocks per plane type (real-valued adjacency, <= 32 nodes)

    @property
    def graph_ops(self):
        """uint8 [B * GGCN_GRAPH_OPS_BYTES] or None: what the one-launch layer / block reads per graph of <= 32 nodes
        (adjacency in MFMA operand order + reciprocal denominators), built from the row masks on first use."""
        if self._graph_ops is None and self.rowmask is not None and self.T <= 32 and self.rowmask.is_cuda:
            lib = _capi.load_library()
            dev = self.rowmask.device
            ops = torch.empty(lib.ggcn_graph_operands_bytes(self.B), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.ggcn_graph_operands(_capi.ptr(self.rowmask), self.B, self.T, _capi.ptr(ops),
                                                    _capi.stream_of(dev)), "ggcn_graph_operands")
            self._graph_ops = ops
        return self._graph_ops

    @property
    def edge_lists(self):
        """``ggcn_graph_edge_lists`` blocks (graphs of 129..256 nodes with row masks) or None: the per-row edge lists the
        eight-wavefront one-launch layer walks, made once per adjacency tensor instead of by every workgroup of every launch."""
        if self._edge_lists is None and self.rowmask is not None and 128 < self.T <= 256 and self.rowmask.is_cuda:
            lib = _capi.load_library()
            dev = self.rowmask.device
            lists = torch.empty(lib.ggcn_graph_edge_lists_bytes(self.B), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.ggcn_graph_edge_lists(_capi.ptr(self.rowmask), self.B, self.T, _capi.ptr(lists), _capi.stream_of(dev)),
                            "ggcn_graph_edge_lists")
            self._edge_lists = lists
        return self._edge_lists

    @property
    def graph_ops_t(self):
        """``ggcn_graph_operands`` blocks of the transposed adjacency (graphs of <= 32 nodes): the A^T operand of the backward's
        ``dH = A^T . D . dY`` on the matrix cores (``ggcn_gate_pool_backward_mma``); built from the row masks on first use."""
        if self._graph_ops_t is None and self.rowmask is not None and self.T <= 32 and self.rowmask.is_cuda:
            lib = _capi.load_library()
            dev = self.rowmask.device
            mt = torch.empty_like(self.rowmask)
            ops = torch.empty(lib.ggcn_graph_operands_bytes(self.B), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                st = _capi.stream_of(dev)
                _capi.check(lib.ggcn_rowmask_transpose(_capi.ptr(self.rowmask), self.B, self.T, _capi.ptr(mt), st), "ggcn_rowmask_transpose")
                _capi.check(lib.ggcn_graph_operands(_capi.ptr(mt), self.B, self.T, _capi.ptr(ops), st), "ggcn_graph_operands")
            self._graph_ops_t = ops
        return self._graph_ops_t

    def graph_ops2(self, plane):
        """uint8 [B * GGCN_GRAPH_OPS2_BYTES]: the block's second layer as ONE operand per graph -- (D.A)^2 in the plane type
        of the launch (0 = bf16 pairs for "bf16x3", 1 = fp16 pairs for "f16mx8" / "f16mx6") -- built from the row masks on
        first use (``ggcn_graph_operands2``), one per plane type."""
        store = self._graph_ops2
        if store is None:
            store = self._graph_ops2 = {}
        if plane not in store:
            if self.rowmask is None or self.T > 32 or not self.rowmask.is_cuda:
                return None
            lib = _capi.load_library()
            dev = self.rowmask.device
            ops = torch.empty(lib.ggcn_graph_operands2_bytes(self.B), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.ggcn_graph_operands2(_capi.ptr(self.rowmask), self.B, self.T, plane, _capi.ptr(ops),
                                                     _capi.stream_of(dev)), "ggcn_graph_operands2")
            store[plane] = ops
        return store[plane]

    def graph_ops_weighted(self, plane):
        """uint8 [B * GGCN_GRAPH_OPS2_BYTES] or None: a REAL-valued adjacency of graphs of <= 32 nodes as the one-launch
        layer's operand (``ggcn_graph_operands_weighted``: D.A_w as hi / lo parts in the plane type, 0 = bf16 pairs, 1 = fp16
        pairs), built from the CSR arrays on first use.  None when an entry does not fit the plane type (one read-back of
        the builder's flag per adjacency and plane type): that adjacency keeps linear + aggregate."""
        store = self._graph_ops_w
        if store is None:
            store = self._graph_ops_w = {}
        if plane not in store:
            if self.T > 32 or not self.rowptr.is_cuda:
                return None
            lib = _capi.load_library()
            dev = self.rowptr.device
            ops = torch.empty(lib.ggcn_graph_operands2_bytes(self.B), dtype=torch.uint8, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.ggcn_graph_operands_weighted(_capi.ptr(self.rowptr), _capi.ptr(self.colidx), _capi.ptr(self.vals),
                                                             self.B, self.T, plane, _capi.ptr(ops), _capi.ptr(flag),
                                                             _capi.stream_of(dev)), "ggcn_graph_operands_weighted")
            store[plane] = None if int(flag.item()) else ops
        return store[plane]

    @property
    def device(self):
        return self.rowmask.device if self._rowptr is None else self._rowptr.device

    @property
    def n_nodes(self):
        return self.B * self.T

    @property
    def rowptr(self):
        self._materialize()
        return self._rowptr

    @property
    def colidx(self):
        self._materialize()
        return self._colidx

    @property
    def vals(self):
        self._materialize()
        return self._vals

    def _source(self):
        """The dense adjacency this CSR was built from, for the lazily built parts (CSR arrays, transposed CSR).
        It must not have been modified in place since: the row masks already in use would no longer match."""
        adj = self._dense
        if adj is not None and self._dense_version is not None and tensor_version(adj) != self._dense_version:
            raise RuntimeError("the dense adjacency was modified in place after its BatchedCSR was built "
                               "(version %s -> %s): build a new BatchedCSR" % (self._dense_version, tensor_version(adj)))
        return adj

    def _materialize(self):
        if self._rowptr is not None:
            return
        adj = self._source()
        lib = _capi.load_library()
        B, T = self.B, self.T
        dev = adj.device
        n, cap = B * T, B * T * T
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        colidx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = None if self.is_binary else torch.empty(cap, dtype=torch.float32, device=dev)
        ws = torch.empty(max(1, lib.ggcn_csr_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        sb, sr, sc = adj.stride()
        with torch.cuda.device(dev):
            rc = lib.ggcn_csr_from_dense(_capi.ptr(adj), _ADJ_DTYPES[adj.dtype], B, T, sb, sr, sc,
                                         _capi.ptr(rowptr), _capi.ptr(colidx), _capi.ptr(vals), cap,
                                         None, None, _capi.ptr(ws), _capi.stream_of(dev))
        _capi.check(rc, "ggcn_csr_from_dense")
        self._rowptr, self._colidx, self._vals = rowptr, colidx, vals

    @classmethod
    def from_dense(cls, adj, binary=None):
        """Device-side conversion of a dense [B,T,T] adjacency (any real dtype, any strides).

        ``binary=True`` promises a 0/1 adjacency (no value array, no host sync; the kernels use
        deg+1 as the denominator, ``models/gcn.py:35``); ``binary=False`` keeps the values as edge
        weights; ``binary=None`` (default) lets the device decide: the builder raises a flag when
        some non-zero differs from 1 and this call reads that one int back (one stream sync -- the
        reference's forward syncs too, ``bert_amir5.py:580``).  colidx/vals are sized for the worst
        case B*T*T; ``nnz`` stays unknown (None)."""
        if not isinstance(adj, torch.Tensor) or adj.dim() != 3 or adj.shape[1] != adj.shape[2]:
            raise RuntimeError("adj must be a [B,T,T] tensor, got %r" % (getattr(adj, "shape", None),))
        if not adj.is_cuda:
            raise RuntimeError("adj must live on the GPU (no CPU path exists in this package)")
        if adj.dtype not in _ADJ_DTYPES:
            raise TypeError("unsupported adjacency dtype %s" % adj.dtype)
        lib = _capi.load_library()
        B, T, _ = adj.shape
        dev = adj.device
        n = B * T
        flags = torch.empty(1, dtype=torch.int32, device=dev) if binary is None else None
        sb, sr, sc = adj.stride()
        out = cls(None, None, None, B, T)
        out._dense, out._dense_version = adj, tensor_version(adj)
        if T <= MASKS_ONLY_MAX_T:
            out.rowmask = torch.empty(n * ((T + 31) // 32), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                rc = lib.ggcn_rowmask_from_dense(_capi.ptr(adj), _ADJ_DTYPES[adj.dtype], B, T, sb, sr, sc,
                                                 _capi.ptr(out.rowmask), _capi.ptr(flags), _capi.stream_of(dev))
            _capi.check(rc, "ggcn_rowmask_from_dense")
            out.is_binary = bool(binary) if binary is not None else not (int(flags.item()) & _capi.FLAG_WEIGHTED)
            return out
        cap = n * T
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        colidx = torch.empty(cap, dtype=torch.int32, device=dev)
        vals = None if binary is True else torch.empty(cap, dtype=torch.float32, device=dev)
        ws = torch.empty(max(1, lib.ggcn_csr_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        if T <= MASK_MAX_T:   # 129..256: the same pass leaves the row masks for the 256-row graph slot (fused_max_t = 256)
            out.rowmask = torch.empty(n * ((T + 31) // 32), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.ggcn_csr_from_dense(_capi.ptr(adj), _ADJ_DTYPES[adj.dtype], B, T, sb, sr, sc,
                                         _capi.ptr(rowptr), _capi.ptr(colidx), _capi.ptr(vals), cap,
                                         _capi.ptr(out.rowmask), _capi.ptr(flags), _capi.ptr(ws), _capi.stream_of(dev))
        _capi.check(rc, "ggcn_csr_from_dense")
        if binary is None and not (int(flags.item()) & _capi.FLAG_WEIGHTED):
            vals = None
        out._rowptr, out._colidx, out._vals = rowptr, colidx, vals
        out.is_binary = vals is None
        return out

    @classmethod
    def from_arrays(cls, rowptr, colidx, B, T, device, vals=None):
        """Upload a CSR built on the host (numpy int32 arrays, global node ids)."""
        import numpy as np
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        if rowptr.shape[0] != B * T + 1:
            raise RuntimeError("rowptr must have B*T+1 = %d entries, got %d" % (B * T + 1, rowptr.shape[0]))
        if int(rowptr[-1]) != colidx.shape[0]:
            raise RuntimeError("rowptr[-1]=%d != len(colidx)=%d" % (int(rowptr[-1]), colidx.shape[0]))
        if colidx.size and (colidx.min() < 0 or colidx.max() >= B * T):
            raise RuntimeError("colidx out of range")
        rows = np.repeat(np.arange(B * T, dtype=np.int64), np.diff(rowptr))
        if np.any(rows // T != colidx // T):
            raise RuntimeError("an edge crosses two graphs: the batched CSR must be block-diagonal")
        v = None
        if vals is not None and not np.all(np.asarray(vals) == 1):
            v = torch.from_numpy(np.ascontiguousarray(vals, dtype=np.float32)).to(device)
        mask = None
        if T <= MASK_MAX_T:  # 0/1 adjacency words for the fused layer kernel: ceil(T/32) words per node
            W = (T + 31) // 32
            m = np.zeros(B * T * W, dtype=np.uint32)
            local = colidx.astype(np.int64) % T
            np.bitwise_or.at(m, rows * W + local // 32, (np.uint32(1) << (local % 32).astype(np.uint32)))
            mask = torch.from_numpy(m.view(np.int32)).to(device)
        out = cls(torch.from_numpy(rowptr).to(device), torch.from_numpy(colidx).to(device), v, B, T,
                  nnz=int(colidx.shape[0]), rowmask=mask)
        return out

    def transposed(self):
        """CSR of the transposed adjacency (rows = source nodes), cached: the backward pass applies
        A^T.  Built from the dense tensor by swapping its strides, or from the CSR arrays by ``ggcn_csr_transpose``."""
        if self._t is None:
            if self._dense is not None:
                self._t = BatchedCSR.from_dense(self._source().transpose(1, 2), binary=self.is_binary)
            elif self._rowptr is not None and self.nnz is not None:
                # built from arrays (GraphBatcher.collate, from_arrays): transposed on the device, graph by graph
                lib = _capi.load_library()
                dev = self.device
                n = self.n_nodes
                rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
                ci = torch.empty(max(1, self.nnz), dtype=torch.int32, device=dev)
                va = None if self._vals is None else torch.empty(max(1, self.nnz), dtype=torch.float32, device=dev)
                ws = torch.empty(n, dtype=torch.int32, device=dev)
                with torch.cuda.device(dev):
                    _capi.check(lib.ggcn_csr_transpose(_capi.ptr(self._rowptr), _capi.ptr(self._colidx), _capi.ptr(self._vals),
                                                       self.B, self.T, _capi.ptr(rp), _capi.ptr(ci), _capi.ptr(va),
                                                       _capi.ptr(ws), _capi.stream_of(dev)), "ggcn_csr_transpose")
                self._t = BatchedCSR(rp, ci, va, self.B, self.T, nnz=self.nnz)
            else:
                raise RuntimeError("this BatchedCSR was assembled by hand: no source to transpose from")
            self._t._t = self
        return self._t

    def inv_denominators(self):
        """fp32 [N]: 1 / (rowsum(adj) + 1) (``models/gcn.py:35``), cached."""
        if self._inv is None:
            lib = _capi.load_library()
            dev = self.device
            inv = torch.empty(self.n_nodes, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.ggcn_inv_denominators(_capi.ptr(self.rowptr), _capi.ptr(self.vals), self.n_nodes,
                                                      _capi.ptr(inv), _capi.stream_of(dev)), "ggcn_inv_denominators")
            self._inv = inv
        return self._inv
