"""Synthetic dependency-graph batches for the benchmark and the parity tests (host side, numpy).

Shape follows the reference's adjacency contract (``graph.py:66-74``): per sentence a
symmetric 0/1 matrix with self loops; here a random tree (``parent(i) ~ U{0..i-1}``) plus
distinct random extra edges until nnz = round(deg*T) exactly.  Seed 14181 is the
reference's default seed (``train.py:307``).
"""
import numpy as np

SEED = 14181


def dependency_batch(B, T, deg=4.0, seed=SEED, lengths=None):
    """Dense uint8 adjacency [B,T,T].

    ``lengths`` (int array [B], optional): rows/cols >= lengths[b] are padding and keep only
    their identity self loop (``graph.py:66`` starts from ``eye(ORI_ML)``); the tree and the
    extra edges then live in the first lengths[b] nodes (nnz = round(deg*len) + padding)."""
    rng = np.random.default_rng(seed)
    adj = np.zeros((B, T, T), dtype=np.uint8)
    idx = np.arange(T)
    adj[:, idx, idx] = 1
    L = np.full(B, T, dtype=np.int64) if lengths is None else np.asarray(lengths, dtype=np.int64)
    bi = np.arange(B)
    for i in range(1, T):
        p = np.floor(rng.random(B) * i).astype(np.int64)
        on = i < L
        adj[bi[on], i, p[on]] = 1
        adj[bi[on], p[on], i] = 1
    # extra undirected edges: pick the n_extra smallest random keys among the free upper-triangle pairs
    iu, ju = np.triu_indices(T, k=1)
    chunk = max(1, (1 << 24) // max(1, iu.size))
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        keys = rng.random((b1 - b0, iu.size))
        taken = adj[b0:b1, iu, ju] != 0
        outside = (ju[None, :] >= L[b0:b1, None])
        keys[taken | outside] = np.inf
        target = np.rint(deg * L[b0:b1]).astype(np.int64)
        n_extra = np.maximum(0, (target - L[b0:b1] - 2 * np.maximum(L[b0:b1] - 1, 0)) // 2)
        free = np.sum(np.isfinite(keys), axis=1)
        n_extra = np.minimum(n_extra, free)
        order = np.argsort(keys, axis=1)
        for r in range(b1 - b0):
            sel = order[r, :n_extra[r]]
            adj[b0 + r, iu[sel], ju[sel]] = 1
            adj[b0 + r, ju[sel], iu[sel]] = 1
    return adj


def csr_from_dense_host(adj):
    """Host batched CSR (global node ids) of a dense [B,T,T] array: rowptr, colidx, vals."""
    B, T, _ = adj.shape
    flat = adj.reshape(B * T, T)
    r, c = np.nonzero(flat)
    counts = np.bincount(r, minlength=B * T)
    rowptr = np.zeros(B * T + 1, dtype=np.int32)
    np.cumsum(counts, out=rowptr[1:])
    colidx = (c + (r // T) * T).astype(np.int32)
    vals = flat[r, c].astype(np.float32)
    return rowptr, colidx, vals


def layer_params(K, F, seed=SEED):
    """weight [K,F] xavier-uniform, bias [F] ~ U(+-1/sqrt(F)) -- ``train.py:75-84``."""
    rng = np.random.default_rng(seed)
    a = np.sqrt(6.0 / (K + F))
    w = rng.uniform(-a, a, size=(K, F)).astype(np.float32)
    s = 1.0 / np.sqrt(F)
    b = rng.uniform(-s, s, size=(F,)).astype(np.float32)
    return w, b


def algorithmic_bytes_per_layer(B, T, F, nnz, n_gates=1, s=4):
    """features in+out once, CSR index arrays, per-graph gate(s), weights."""
    N = B * T
    return 2 * s * N * F + 4 * (N + 1) + 4 * nnz + n_gates * s * B * F + 4 * F * F + 4 * F
