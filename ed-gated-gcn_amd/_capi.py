"""ctypes binding of libggcn_hip.so (C ABI: include/ggcn.h).

This is the binding a maintainer of the reference would add next to
``models/gcn.py``: plain pointers, sizes and a stream handle --
no torch types cross the boundary.  Missing library => ImportError-like
RuntimeError, never a silent fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i32, c_i64, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes): exactly the prototypes of include/ggcn.h
PROTOTYPES = {
    "ggcn_abi_version": (c_i32, []),
    "ggcn_has_f16mx6": (c_i32, []),
    "ggcn_last_error": (ctypes.c_char_p, []),
    "ggcn_csr_workspace_bytes": (c_sz, [c_i64]),
    "ggcn_csr_from_dense": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp,
                                    c_i64, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_rowmask_from_dense": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "ggcn_csr_transpose": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_csr_rowmask": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_graph_edge_lists_bytes": (c_sz, [c_i32]),
    "ggcn_graph_edge_lists": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_graph_operands_bytes": (c_sz, [c_i32]),
    "ggcn_graph_operands": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_graph_operands2_bytes": (c_sz, [c_i32]),
    "ggcn_graph_operands2": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_graph_operands_weighted": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ggcn_layer_fused_weighted": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "ggcn_layer_fused": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                 c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "ggcn_layer_fused_prebias": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                         c_vp, c_i64, c_vp, c_vp, c_i32, c_vp]),
    "ggcn_block_fused": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                 c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "ggcn_overlap_reduce": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_transpose": (c_i32, [c_vp, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "ggcn_gate_mlp": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_scores_head": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32,
                                 c_i32, c_i32, c_vp, c_i64, c_vp, c_vp]),
    "ggcn_absmax": (c_i32, [c_vp, c_i32, c_i64, c_i64, c_i32, c_vp, c_vp]),
    "ggcn_range_flag": (c_i32, [c_vp, c_i32, c_vp]),
    "ggcn_debug_poison_lds": (c_i32, [ctypes.c_uint32, c_vp]),
    "ggcn_lab_block_fused8": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp,
                                      c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_debug_mfma_calibrate": (c_i32, [c_i32, c_i32, c_vp, c_vp, c_vp]),
    "ggcn_debug_block_fused_stamped": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                               c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "ggcn_subword_pool": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64,
                                  c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ggcn_weight_pack_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "ggcn_weight_pack": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_linear": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "ggcn_aggregate": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                               c_vp, c_i64, c_vp, c_vp, c_vp]),
    "ggcn_gate_pool_backward": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32,
                                        c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_gate_pool_backward_drop": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32,
                                             c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_uint64, c_i32, c_i32, c_i32, c_vp]),
    "ggcn_gate_pool_backward_agg": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                            c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, ctypes.c_float, ctypes.c_uint64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_rowmask_transpose": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_gate_pool_backward_mma": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                            c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ggcn_linear_scaled": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "ggcn_layer_fused_drop": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                      c_vp, c_i64, c_vp, c_vp, c_i32, ctypes.c_float, ctypes.c_uint64, c_i32, c_i32, c_i32, c_vp]),
    "ggcn_dropout_mask": (c_i32, [c_i64, c_i32, ctypes.c_float, ctypes.c_uint64, c_i32, c_vp, c_vp]),
    "ggcn_colsum_workspace_bytes": (c_sz, [c_i32]),
    "ggcn_colsum": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "ggcn_dweight_workspace_bytes": (c_sz, [c_i64, c_i32, c_i32, c_i32]),
    "ggcn_dweight": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "ggcn_inv_denominators": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "ggcn_aggregate_t": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "ggcn_linear_h": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "ggcn_aggregate_h": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                 c_vp, c_i64, c_vp, c_vp, c_vp]),
    "ggcn_layer_fused_h": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp,
                                   c_vp, c_i64, c_vp, c_vp, c_vp]),
    "ggcn_block_fused_form": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "ggcn_dense_head": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "ggcn_dense_head_signal": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "ggcn_overlap_workspace_bytes": (c_sz, [c_i32]),
    "ggcn_gate_overlap": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
}

ABI_VERSION = 13
PREC = {"bf16x3": 0, "fp32": 1, "f16mx8": 2, "f16": 3, "f16mx6": 4}
PACKED = ("bf16x3", "f16mx8", "f16", "f16mx6")  # precisions whose linear reads a ggcn_weight_pack image ("f16": half features only)
FLAG_WEIGHTED = 1
RANGE_OVERFLOW, RANGE_WINDOW, RANGE_HIDDEN = 1, 2, 4   # include/ggcn.h ggcn_range_bits


def has_f16mx6():
    """True when libggcn_hip.so was built with the experimental fp6-correction kernel (``make F16MX6=1``)."""
    return bool(load_library().ggcn_has_f16mx6())


def lib_path():
    return os.environ.get("GGCN_LIB", os.path.join(_HERE, "libggcn_hip.so"))


def load_library():
    """Load libggcn_hip.so and bind every prototype; raise if it is absent or stale."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libggcn_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C ed-gated-gcn_amd/csrc` (no CPU fallback exists)" % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.ggcn_abi_version() != ABI_VERSION:
        raise RuntimeError("libggcn_hip.so ABI %d != binding ABI %d" % (lib.ggcn_abi_version(), ABI_VERSION))
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load_library().ggcn_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
