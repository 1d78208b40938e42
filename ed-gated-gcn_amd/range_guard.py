"""Lazy surfacing of the library's sticky f16mx8 range flag (``ggcn_range_flag``, ``include/ggcn.h``).

``precision="f16mx8"`` (the default) keeps the reference's fp32 accuracy (1e-4 gate) for activations of |x| <= 448 and
hidden values below fp16's largest value, 65504; the reference's fp32 matmul (``models/gcn.py:34``) has no such limits.  Every f16mx8 main loop records a violation in a sticky per-device flag
at no extra launch.  This module makes the Python side notice WITHOUT a device synchronisation on the forward path: the
first and then every ``POLL_EVERY``-th forward that ran such kernels enqueues "OR the flag into a 4-byte device word, clear it, copy the word to
pinned host memory, record an event" behind its kernels; a later forward whose poll finds that event complete reads the
host word and raises if it is set.  ``check(device)`` does the same synchronously (one read-back), for tests and for code
that wants the verdict now.
"""
import atexit
import sys

import torch

from . import _capi

POLL_EVERY = 16
_STATE = {}

WHAT = {
    _capi.RANGE_OVERFLOW: "a value reached fp16's range (|v| >= 65504 or infinite): results from that launch on are saturated",
    _capi.RANGE_WINDOW: "an activation left |x| <= 448, the window in which the fp8 correction terms are exact to their "
                        "format: results from that launch on have plain fp16 accuracy (2^-12 relative), outside the 1e-4 "
                        "parity with the reference's fp32 matmul (models/gcn.py:34)",
    _capi.RANGE_HIDDEN: "the one-launch layer could not rule out an fp16 overflow of its hidden values (max|x| * max_f "
                        "sum_k |w[k,f]| reached 65504): NaN outputs are possible",
}
MESSAGE = ("precision='f16mx8' on %s, in an earlier launch: %s.  Use precision='bf16x3' (full fp32 range) for this data "
           "(GraphConvolution(..., opt) with opt.ggcn_precision = 'bf16x3', or GGCN_PRECISION=bf16x3)")


def _state(dev):
    st = _STATE.get(dev.index)
    if st is None:
        st = {"flag": torch.zeros(1, dtype=torch.int32, device=dev), "host": torch.zeros(1, dtype=torch.int32).pin_memory(),
              "event": torch.cuda.Event(), "pending": False, "calls": 0}
        _STATE[dev.index] = st
    return st


def _verdict(st, dev):
    bits = int(st["host"][0])
    if bits != 0:
        st["host"].zero_()
        st["flag"].zero_()
        raise RuntimeError(MESSAGE % (dev, "; ".join(text for bit, text in WHAT.items() if bits & bit) or "flag %d" % bits))


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
    if st["pending"] or (st["calls"] != 1 and st["calls"] % POLL_EVERY):   # the first guarded forward, then every POLL_EVERY-th
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


def _at_exit():
    """A process that ends before a polled snapshot was read (fewer forwards than it takes, or none after the violation)
    still gets told: one read-back per device that ran guarded forwards, a line on stderr instead of an exception."""
    for idx in list(_STATE):
        try:
            check(torch.device("cuda", idx))
        except RuntimeError as e:
            print("ed-gated-gcn_amd: %s" % (e,), file=sys.stderr)
        except Exception:   # noqa: BLE001 -- the device may already be gone at interpreter exit
            pass


atexit.register(_at_exit)
