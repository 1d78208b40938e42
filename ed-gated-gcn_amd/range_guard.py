"""Lazy surfacing of the library's sticky f16mx8 range flag (``ggcn_range_flag``, ``include/ggcn.h``).

``precision="f16mx8"`` (the default) needs finite inputs below fp16's largest value, 65504; the reference's fp32
matmul (``models/gcn.py:34``) has no such limit.  Every f16mx8 main loop records a violation in a sticky per-device flag
at no extra launch.  This module makes the Python side notice WITHOUT a device synchronisation on the forward path: every
``POLL_EVERY``-th forward that ran such kernels enqueues "OR the flag into a 4-byte device word, clear it, copy the word to
pinned host memory, record an event" behind its kernels; a later forward whose poll finds that event complete reads the
host word and raises if it is set.  ``check(device)`` does the same synchronously (one read-back), for tests and for code
that wants the verdict now.
"""
import torch

from . import _capi

POLL_EVERY = 16
_STATE = {}

MESSAGE = ("precision='f16mx8' met a value outside its range (|v| >= 65504 or infinite) in an earlier launch on %s: its "
           "results from that launch on are saturated; use precision='bf16x3' (full fp32 range) for this data "
           "(GraphConvolution(..., opt) with opt.ggcn_precision = 'bf16x3', or GGCN_PRECISION=bf16x3)")


def _state(dev):
    st = _STATE.get(dev.index)
    if st is None:
        st = {"flag": torch.zeros(1, dtype=torch.int32, device=dev), "host": torch.zeros(1, dtype=torch.int32).pin_memory(),
              "event": torch.cuda.Event(), "pending": False, "calls": 0}
        _STATE[dev.index] = st
    return st


def _verdict(st, dev):
    if int(st["host"][0]) != 0:
        st["host"].zero_()
        st["flag"].zero_()
        raise RuntimeError(MESSAGE % (dev,))


def before(dev):
    """Start of a forward: has an earlier snapshot of the flag reached the host?  (No device synchronisation.)"""
    st = _STATE.get(dev.index)
    if st is not None and st["pending"] and st["event"].query():
        st["pending"] = False
        _verdict(st, dev)


def _snapshot(st, dev):
    lib = _capi.load_library()
    with torch.cuda.device(dev):
        _capi.check(lib.ggcn_range_flag(_capi.ptr(st["flag"]), 1, _capi.stream_of(dev)), "ggcn_range_flag")


def after(dev):
    """End of a forward that ran f16mx8 kernels: every POLL_EVERY-th one leaves a snapshot of the flag on its way."""
    if torch.cuda.is_current_stream_capturing():
        return
    st = _state(dev)
    st["calls"] += 1
    if st["pending"] or st["calls"] % POLL_EVERY:
        return
    _snapshot(st, dev)
    with torch.cuda.device(dev):
        st["host"].copy_(st["flag"], non_blocking=True)
        st["event"].record()
    st["pending"] = True


def check(dev):
    """Synchronous verdict: raises RuntimeError if any f16mx8 launch on `dev` since the last report met a value outside
    fp16's range; one device read-back."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError("range_guard.check needs a GPU device (got %s): the flag lives in libggcn_hip.so's device memory" % (dev,))
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _state(dev)
    _snapshot(st, dev)
    st["pending"] = False
    st["host"].copy_(st["flag"])   # blocking copy into pinned memory
    _verdict(st, dev)
