"""Sub-word -> word pooling on the HIP path: ``x = torch.bmm(transform, x)``
(``models/bert_amir5.py:600``; the transform of ``data_utils.py:749-766`` holds ``1/l`` on the ``l``
sub-word positions of each word and zeros elsewhere, x is the 9216-wide concatenation of the 12
BERT layers, ``bert_amir5.py:596``).  ``ggcn_subword_pool`` multiplies only the non-zeros and reads
each selected row of x once; the backward ``dX = transformᵀ · dY`` is the same kernel with the
strides of ``transform`` swapped (``train.py:120`` back-propagates into BERT through this step).
No CPU fallback: CPU tensors raise.
"""
import torch

from . import _capi


def _check(name, t, dims):
    if not isinstance(t, torch.Tensor) or t.dim() != dims:
        raise TypeError("%s must be a %d-d tensor" % (name, dims))
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32, got %s" % (name, t.dtype))


def _launch(a, x, swap):
    """y[b] = a[b] @ x[b] (swap=False) or a[b].T @ x[b] (swap=True); a strided, x/y row-contiguous."""
    lib = _capi.load_library()
    B, R, C = a.shape
    sb, sr, sc = a.stride()
    if swap:
        R, C, sr, sc = C, R, sc, sr
    if x.stride(2) != 1:
        x = x.contiguous()
    D = x.shape[2]
    y = torch.empty(B, R, D, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = _capi.stream_of(x.device)
        _capi.check(lib.ggcn_subword_pool(_capi.ptr(a), sb, sr, sc, _capi.ptr(x), x.stride(0), x.stride(1),
                                          _capi.ptr(y), y.stride(0), y.stride(1), B, R, C, D, st),
                    "ggcn_subword_pool")
    return y


class _SubwordPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, transform, x):
        ctx.save_for_backward(transform)
        return _launch(transform, x, False)

    @staticmethod
    def backward(ctx, dy):
        (transform,) = ctx.saved_tensors
        dx = _launch(transform, dy.contiguous(), True) if ctx.needs_input_grad[1] else None
        return None, dx    # the transform is data (data_utils.py:749-766), never a parameter


def subword_pool(transform, x):
    """``torch.bmm(transform, x)`` for a sparse ``transform [B,T,L]`` (any strides) and ``x [B,L,D]``."""
    _check("transform", transform, 3)
    _check("x", x, 3)
    if transform.shape[0] != x.shape[0] or transform.shape[2] != x.shape[1]:
        raise RuntimeError("transform %s does not match x %s" % (tuple(transform.shape), tuple(x.shape)))
    if transform.device != x.device:
        raise RuntimeError("transform and x are on different devices")
    if transform.requires_grad:
        raise RuntimeError("subword_pool does not differentiate with respect to the transform")
    return _SubwordPool.apply(transform, x)
