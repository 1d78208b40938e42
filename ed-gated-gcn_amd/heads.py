"""The small pieces of ``BertAmir55.forward`` on either side of the gated block, each as one HIP launch.

* ``gate_mlps``   -- ``models/bert_amir5.py:562-571,621-622``: both gate MLPs from ``aspect`` (the ``[B,T,H]``
  repeat of ``:621-622`` is never made);
* ``scores_and_kl`` -- ``models/bert_amir5.py:645-648``: the ``fc`` / ``scores`` / ``kl`` head without its
  ``[B,T,2H]`` concat and ``[B,T,C]`` product.

Inference only (``torch.no_grad`` or no parameter needing a gradient): under autograd the classifier keeps the
reference's PyTorch ops for these ~0.1 GFLOP pieces, so ``train.py:115-121`` trains them unchanged.
"""
import torch

from . import _capi
from .csr import tensor_version


def _transposed(linear, lib, st):
    """[in,out] copy of an nn.Linear weight, rebuilt when the weight changes."""
    w = linear.weight
    key = (w.data_ptr(), tensor_version(w), w.device)
    cached = getattr(linear, "_ggcn_wt", None)
    if cached is None or cached[0] != key:
        wc = w.detach()
        if not wc.is_contiguous():
            wc = wc.contiguous()
        wt = torch.empty(wc.shape[1], wc.shape[0], dtype=torch.float32, device=w.device)
        _capi.check(lib.ggcn_transpose(_capi.ptr(wc), wc.shape[0], wc.shape[1], wc.stride(0), _capi.ptr(wt), st),
                    "ggcn_transpose")
        cached = (key, wt)
        linear._ggcn_wt = cached
    return cached[1]


def _gate_linears(seq):
    """The two nn.Linear of ``Sequential(Sigmoid, Linear, Sigmoid, Linear, Sigmoid)`` (``bert_amir5.py:562-571``)."""
    mods = list(seq)
    ok = (len(mods) == 5 and all(isinstance(mods[i], torch.nn.Sigmoid) for i in (0, 2, 4))
          and all(isinstance(mods[i], torch.nn.Linear) for i in (1, 3)))
    if not ok:
        raise RuntimeError("gate MLP must be Sequential(Sigmoid, Linear, Sigmoid, Linear, Sigmoid) (bert_amir5.py:562-571)")
    return mods[1], mods[3]


def gate_mlps(aspect, gate1_seq, gate2_seq):
    """``gate1(aspect), gate2(aspect)`` -> two contiguous ``[B,H]`` float32 tensors, one launch."""
    if not (aspect.is_cuda and aspect.dtype == torch.float32 and aspect.dim() == 2):
        raise RuntimeError("aspect must be a float32 [B,H] GPU tensor (no CPU path exists)")
    lib = _capi.load_library()
    B, H = aspect.shape
    la1, la2 = _gate_linears(gate1_seq)
    lb1, lb2 = _gate_linears(gate2_seq)
    for lin in (la1, la2, lb1, lb2):
        if lin.in_features != H or lin.out_features != H:
            raise RuntimeError("gate linears must be %d -> %d" % (H, H))
    a = aspect if aspect.stride(1) == 1 else aspect.contiguous()
    dev = aspect.device
    with torch.cuda.device(dev):
        st = _capi.stream_of(dev)
        g1 = torch.empty(B, H, dtype=torch.float32, device=dev)
        g2 = torch.empty(B, H, dtype=torch.float32, device=dev)
        bias = lambda l: None if l.bias is None else l.bias.detach()   # noqa: E731
        _capi.check(lib.ggcn_gate_mlp(_capi.ptr(a), a.stride(0), B, H,
                                      _capi.ptr(_transposed(la1, lib, st)), _capi.ptr(bias(la1)),
                                      _capi.ptr(_transposed(la2, lib, st)), _capi.ptr(bias(la2)), _capi.ptr(g1),
                                      _capi.ptr(_transposed(lb1, lib, st)), _capi.ptr(bias(lb1)),
                                      _capi.ptr(_transposed(lb2, lib, st)), _capi.ptr(bias(lb2)), _capi.ptr(g2), st),
                    "ggcn_gate_mlp")
    return g1, g2


def scores_and_kl(x, aspect, logits, fc_linear, dist):
    """``models/bert_amir5.py:645-648``: ``(scores [B,T], kl scalar)`` from the block's gated output ``x [B,T,H]``."""
    for name, t in (("x", x), ("aspect", aspect), ("logits", logits)):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError("%s must be a float32 GPU tensor (no CPU path exists)" % name)
    lib = _capi.load_library()
    B, T, H = x.shape
    C = logits.shape[1]
    if fc_linear.in_features != 2 * H or fc_linear.out_features != C:
        raise RuntimeError("fc must be Linear(%d, %d)" % (2 * H, C))
    x2 = x.reshape(B * T, H)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    a = aspect if aspect.stride(1) == 1 else aspect.contiguous()
    lg = logits if logits.stride(1) == 1 else logits.contiguous()
    d = dist.float()                                   # :648 `dist_to_target.float()`
    if d.stride(1) != 1:
        d = d.contiguous()
    w = fc_linear.weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    fb = None if fc_linear.bias is None else fc_linear.bias.detach()
    dev = x.device
    with torch.cuda.device(dev):
        st = _capi.stream_of(dev)
        scores = torch.empty(B, T, dtype=torch.float32, device=dev)
        part = torch.empty(B, dtype=torch.float32, device=dev)
        kl = torch.empty((), dtype=torch.float32, device=dev)
        _capi.check(lib.ggcn_scores_head(_capi.ptr(x2), x2.stride(0), _capi.ptr(a), a.stride(0), _capi.ptr(lg),
                                         lg.stride(0), _capi.ptr(w), w.stride(0), _capi.ptr(fb), _capi.ptr(d),
                                         d.stride(0), B, T, H, C, _capi.ptr(scores), T, _capi.ptr(part), st),
                    "ggcn_scores_head")
        _capi.check(lib.ggcn_overlap_reduce(_capi.ptr(part), B, 1, _capi.ptr(kl), st), "ggcn_overlap_reduce")
    return scores, kl


def dense_head(pooled, wt, bias=None, partials=None, f_block=None, signal=None):
    """``logits = pooled @ wt (+ bias)`` -- the share of ``bert_amir5.py:643``'s ``dense`` that reads the block's pooled
    output -- as ONE launch; with ``partials`` (the regulariser's per-graph partial sums of the one-launch block,
    ``bert_amir5.py:638``) the same launch also finishes ``xy``.  ``wt`` is ``[H, C]`` (the ``nn.Linear`` weight slice
    transposed), ``C <= 64``.  Returns ``logits`` or ``(logits, xy)``.  A row's logits are the same bits whatever batch or
    shard the row sits in.  ``signal`` (two zeroed int32 words on the GPU): the launch counts itself done in ``signal[1]``
    (``ggcn_dense_head_signal``), so that another stream can be gated on it without an event (``shard.PooledGather``)."""
    for name, t in (("pooled", pooled), ("wt", wt)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError("%s must be a float32 2-D GPU tensor with unit column stride (no CPU path exists)" % name)
    B, H = pooled.shape
    if wt.shape[0] != H:
        raise RuntimeError("wt must be [%d, C], got %s" % (H, tuple(wt.shape)))
    C = wt.shape[1]
    if bias is not None and not (bias.is_cuda and bias.dtype == torch.float32 and tuple(bias.shape) == (C,) and bias.is_contiguous()):
        raise RuntimeError("bias must be a contiguous float32 [%d] GPU tensor" % C)
    lib = _capi.load_library()
    dev = pooled.device
    with torch.cuda.device(dev):
        logits = torch.empty(B, C, dtype=torch.float32, device=dev)
        xy = torch.empty((), dtype=torch.float32, device=dev) if partials is not None else None
        if signal is not None:
            if not (signal.is_cuda and signal.dtype == torch.int32 and signal.numel() >= 2 and signal.is_contiguous()):
                raise RuntimeError("signal must be a contiguous int32 GPU tensor of two words")
            _capi.check(lib.ggcn_dense_head_signal(_capi.ptr(pooled), pooled.stride(0), _capi.ptr(wt), wt.stride(0), _capi.ptr(bias), B, H, C,
                                                   _capi.ptr(logits), C, _capi.ptr(partials), int(f_block or 0), _capi.ptr(xy),
                                                   _capi.ptr(signal), _capi.stream_of(dev)), "ggcn_dense_head_signal")
        else:
            _capi.check(lib.ggcn_dense_head(_capi.ptr(pooled), pooled.stride(0), _capi.ptr(wt), wt.stride(0), _capi.ptr(bias), B, H, C,
                                            _capi.ptr(logits), C, _capi.ptr(partials), int(f_block or 0), _capi.ptr(xy),
                                            _capi.stream_of(dev)), "ggcn_dense_head")
    return logits if partials is None else (logits, xy)
