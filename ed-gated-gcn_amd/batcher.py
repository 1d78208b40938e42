"""Per-sample CSR cache + batch collation: the "wire format" into the layer.

The reference keeps every sentence's adjacency dense (``int ndarray [ORI_ML,ORI_ML]``,
``graph.py:66-74``), ships it as ``float32 [B,ORI_ML,ORI_ML]`` to the device on EVERY batch
(``data_utils.py:376,394``; ``train.py:108``; 3.8 KB/sample at ORI_ML=31, 213 KB at 231) and
slices ``[:, :T, :T]`` with T = the batch's longest sentence (``models/bert_amir5.py:581,589``).
The adjacency of a sample never changes, so here it is converted ONCE at dataset-load time into a
compact per-sample edge list; a batch is then a concatenation (O(nnz) host work, 4 B per edge +
4 B per node over PCIe) that yields the same BatchedCSR -- rowptr, colidx, row masks -- the
device-side builder produces from the dense tensor.
"""
import numpy as np

from .csr import BatchedCSR


class SampleGraph:
    """Edges of one sentence graph: ``rows[e] <- cols[e]`` (local node ids), sorted by (row, col)."""

    __slots__ = ("rows", "cols", "vals", "n")

    def __init__(self, dense):
        dense = np.asarray(dense)
        if dense.ndim != 2 or dense.shape[0] != dense.shape[1]:
            raise ValueError("a sample adjacency must be square, got %r" % (dense.shape,))
        r, c = np.nonzero(dense)  # row-major order = sorted by (row, col)
        self.rows = r.astype(np.int32)
        self.cols = c.astype(np.int32)
        v = dense[r, c].astype(np.float32)
        self.vals = None if np.all(v == 1.0) else v
        self.n = int(dense.shape[0])


class GraphBatcher:
    """Cache of SampleGraph keyed by sample id; ``collate`` builds one batch."""

    def __init__(self):
        self._cache = {}

    def add(self, sample_id, dense):
        self._cache[sample_id] = SampleGraph(dense)

    def __len__(self):
        return len(self._cache)

    def collate(self, sample_ids, T, device):
        """BatchedCSR of ``adj[:, :T, :T]`` for the given samples (same result as
        ``BatchedCSR.from_dense`` on the stacked dense slice)."""
        B = len(sample_ids)
        rows_l, cols_l, vals_l = [], [], []
        weighted = False
        for b, sid in enumerate(sample_ids):
            g = self._cache[sid]
            if T > g.n:
                raise ValueError("T=%d exceeds the stored adjacency size %d of sample %r" % (T, g.n, sid))
            keep = (g.rows < T) & (g.cols < T)
            rows_l.append(g.rows[keep].astype(np.int64) + b * T)
            cols_l.append(g.cols[keep].astype(np.int64) + b * T)
            if g.vals is not None:
                weighted = True
                vals_l.append(g.vals[keep])
            else:
                vals_l.append(None)
        rows = np.concatenate(rows_l) if rows_l else np.zeros(0, np.int64)
        cols = np.concatenate(cols_l) if cols_l else np.zeros(0, np.int64)
        rowptr = np.zeros(B * T + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=B * T), out=rowptr[1:])
        vals = None
        if weighted:
            vals = np.concatenate([v if v is not None else np.ones(len(r), np.float32)
                                   for v, r in zip(vals_l, rows_l)])
        return BatchedCSR.from_arrays(rowptr, cols.astype(np.int32), B, T, device, vals=vals)
