"""ctypes loader for the C restatement (oracle/csr_ref.c).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libggcn_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_csr_from_dense.restype = ctypes.c_int64
        L.oracle_gate_overlap.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def csr_from_dense(adj):
    adj = np.ascontiguousarray(adj, dtype=np.float32)
    B, T, _ = adj.shape
    rowptr = np.zeros(B * T + 1, dtype=np.int32)
    colidx = np.zeros(B * T * T, dtype=np.int32)
    vals = np.zeros(B * T * T, dtype=np.float32)
    nnz = lib().oracle_csr_from_dense(_p(adj), B, T, _p(rowptr), _p(colidx), _p(vals))
    return rowptr, colidx[:nnz].copy(), vals[:nnz].copy()


def gcn_layer_csr(X, W, bias, rowptr, colidx, vals=None):
    X = np.ascontiguousarray(X, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    N, K = X.shape
    F = W.shape[1]
    out = np.empty((N, F), dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    v = None if vals is None else np.ascontiguousarray(vals, dtype=np.float32)
    rc = lib().oracle_gcn_layer_csr(_p(X), _p(W), _p(b), _p(rowptr), _p(colidx), _p(v),
                                    ctypes.c_int64(N), K, F, _p(out))
    if rc:
        raise MemoryError("oracle_gcn_layer_csr")
    return out


def gate_pool(y, gate, B, T):
    y = np.ascontiguousarray(y, dtype=np.float32)
    F = y.shape[-1]
    g = None if gate is None else np.ascontiguousarray(gate, dtype=np.float32)
    gated = np.empty((B * T, F), dtype=np.float32)
    pooled = np.empty((B, F), dtype=np.float32)
    lib().oracle_gate_pool(_p(y), _p(g), B, T, F, _p(gated), _p(pooled))
    return gated, pooled


def gate_overlap(x1, y1):
    x1 = np.ascontiguousarray(x1, dtype=np.float32)
    y1 = np.ascontiguousarray(y1, dtype=np.float32)
    return float(lib().oracle_gate_overlap(_p(x1), _p(y1), x1.shape[0], x1.shape[1]))
