"""The gated block of the BERT+GCN classifiers: ``models/bert_amir5.py:621-640``.

Reference (BertAmir55; identical in BertAmir54 ``:512-531``)::

    gate1 = self.gate1(aspect).repeat(1, T).view(x.shape)     # [B,H] -> [B,T,H]
    gate2 = self.gate2(aspect).repeat(1, T).view(x.shape)
    gcn1 = self.gc1(x, adj)
    x1 = max_t(gcn1 * gate1);  y1 = max_t(gcn1 * gate2)
    xy = (x1 * y1).sum(1).mean()
    x = gate2 * self.gc2(gcn1, adj);  out = max_t(x)

Here the gates stay ``[B,H]`` and the adjacency is converted once for both layers.  Inference on
graphs of <= 32 tokens runs the WHOLE block as one launch (``ggcn_block_fused``: no non-linearity sits
between gc1 and gc2, so gc2(gc1(x)) is a product of x with the folded weight W1.W2 and gcn1 never
touches HBM); otherwise each layer is one fused launch, or one linear + one aggregation launch that
also applies the gate and the max over tokens.
"""
import torch

from . import _capi, range_guard
from .csr import BatchedCSR, tensor_version


def gate_overlap(x1, y1):
    """``xy = (x1 * y1).sum(1).mean()`` (``bert_amir5.py:638``) as one device scalar."""
    lib = _capi.load_library()
    B, F = x1.shape
    dev = x1.device
    with torch.cuda.device(dev):
        xy = torch.empty((), dtype=torch.float32, device=dev)
        ws = torch.empty(lib.ggcn_overlap_workspace_bytes(B), dtype=torch.uint8, device=dev)
        _capi.check(lib.ggcn_gate_overlap(_capi.ptr(x1), _capi.ptr(y1), B, F, _capi.ptr(xy), _capi.ptr(ws),
                                          _capi.stream_of(dev)), "ggcn_gate_overlap")
    return xy


def _block_operands(gc1, gc2, lib, st, precision=None):
    """Operands of the one-launch block (``ggcn_block_fused``), rebuilt only when a parameter changes:

    * packed ``W1`` (gc1's own image), packed ``W12 = W1 . W2`` and ``mid = W2^T . b1``.

    ``bert_amir5.py:626,639`` feed gc2 with the UNGATED gcn1 and ``gcn.py:30-45`` applies no non-linearity,
    so ``gc2(gc1(x)) = D.A.(D.A.(x.W12) + mid) + b2 = (D.A)^2.(x.W12) + rowsum(D.A).mid + b2`` (the launch applies the
    graph's precomputed ``(D.A)^2``: ``BatchedCSR.graph_ops2``).  W12 and mid come from the library's exact-fp32 MFMA
    linear (a k-ordered fp32 FMA chain): they are parameters folded once per weight update, not activations."""
    w1, w2, b1 = gc1.weight, gc2.weight, gc1.bias
    precision = precision or gc1.precision
    prec = _capi.PREC[precision]
    key = (w1.data_ptr(), tensor_version(w1), w2.data_ptr(), tensor_version(w2), None if b1 is None else (b1.data_ptr(), tensor_version(b1)),
           w1.device, prec)
    store = getattr(gc2, "_block_ops", None)
    if not isinstance(store, dict):
        store = gc2._block_ops = {}   # one entry per precision (like _packed_weight): an eval / train switch of the classifier folds nothing again
    cached = store.get(prec)
    if cached is None or cached[0] != key:
        K, F1, F2 = gc1.in_features, gc1.out_features, gc2.out_features
        dev = w1.device
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        w12 = torch.empty(K, F2, dtype=torch.float32, device=dev)
        _capi.check(lib.ggcn_linear(_capi.ptr(w1c), F1, _capi.ptr(w2c), F2, None, _capi.ptr(w12), F2, K, F1, F2,
                                    _capi.PREC["fp32"], st), "ggcn_linear(W1.W2)")
        mid = torch.zeros(F2, dtype=torch.float32, device=dev)
        if b1 is not None:
            b1c = b1.detach().contiguous()
            _capi.check(lib.ggcn_linear(_capi.ptr(b1c), F1, _capi.ptr(w2c), F2, None, _capi.ptr(mid), F2, 1, F1, F2,
                                        _capi.PREC["fp32"], st), "ggcn_linear(b1.W2)")
        pack12 = torch.empty(lib.ggcn_weight_pack_bytes(K, F2, prec), dtype=torch.uint8, device=dev)
        _capi.check(lib.ggcn_weight_pack(_capi.ptr(w12), F2, K, F2, prec, 0, _capi.ptr(pack12), st), "ggcn_weight_pack(W12)")
        cached = (key, pack12, mid)
        store[prec] = cached
    return gc1._packed_weight(lib, st, precision=precision), cached[1], cached[2]


def takes_folded_eval_path(x, csr, gc1, gc2):
    """True when an evaluation that needs only ``out`` / ``x`` of graphs of 33..256 nodes runs WITHOUT the W1 product:
    ``Z = D.A.X`` (the aggregation kernel on the features), then ONE one-launch layer on Z with the folded weight
    ``W12 = W1.W2`` and ``mid = W2^T.b1`` added before its aggregation (``ggcn_layer_fused_prebias``):
    ``gc2(gc1(X)) = D.A.(Z.W12 + 1.mid^T) + b2`` (``bert_amir5.py:626,639``: no non-linearity between the layers)."""
    return (32 < csr.T <= 256 and gc1.takes_fused_path(x, csr) and gc2.takes_fused_path(x, csr) and gc1.precision == gc2.precision
            and gc1.precision in ("f16mx8", "bf16x3") and gc1.out_features == gc2.in_features
            and gc1.in_features == gc1.out_features == gc2.out_features)


def _folded_eval(x, csr, gate2, gc1, gc2, want_x):
    """``(x or None, out)`` of ``bert_amir5.py:639-640`` through the folded weight, two launches, no product with W1."""
    gc1._check(x)
    range_guard.before(x.device)
    lib = _capi.load_library()
    B, T, K = x.shape
    F = gc2.out_features
    dev = x.device
    x2d = x.reshape(B * T, K)
    if x2d.stride(1) != 1:
        x2d = x2d.contiguous()
    if not (isinstance(gate2, torch.Tensor) and gate2.is_cuda and gate2.dtype == torch.float32 and tuple(gate2.shape) == (B, F)
            and gate2.is_contiguous()):
        raise RuntimeError("gate2 must be a contiguous float32 [B,F]=[%d,%d] GPU tensor" % (B, F))
    with torch.cuda.device(dev):
        st = _capi.stream_of(dev)
        prec = gc1.precision
        _, pack12, mid = _block_operands(gc1, gc2, lib, st, precision=prec)
        z = torch.empty(B * T, K, dtype=torch.float32, device=dev)
        # Z = D.A.X: gcn.py:35,41 applied to the features themselves (no bias, no gate, no pool)
        _capi.check(lib.ggcn_aggregate(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(csr.rowptr), _capi.ptr(csr.colidx), _capi.ptr(csr.vals),
                                       None, B, T, K, None, None, None, _capi.ptr(z), K, None, None, st), "ggcn_aggregate(D.A.X)")
        xo = torch.empty(B * T, F, dtype=torch.float32, device=dev) if want_x else None
        out = torch.empty(B, F, dtype=torch.float32, device=dev)
        b2 = None if gc2.bias is None else gc2.bias.detach()
        _capi.check(lib.ggcn_layer_fused_prebias(_capi.ptr(z), K, _capi.ptr(pack12), _capi.ptr(csr.rowmask), _capi.ptr(b2), _capi.ptr(mid),
                                                 B, T, K, F, _capi.ptr(gate2), _capi.ptr(gate2), None, _capi.ptr(xo), F,
                                                 _capi.ptr(out), None, _capi.PREC[prec], st), "ggcn_layer_fused_prebias")
    if prec == "f16mx8":
        range_guard.after(x.device)
    return (None if xo is None else xo.view(B, T, F)), out


def takes_block_path(x, csr, gc1, gc2):
    """True when the inference block runs as ONE launch: both layers on the one-launch layer path with the
    same arithmetic, and gc1's output width = gc2's (the reference's blocks are square, bert_amir5.py:559-560)."""
    return (csr.T <= 32 and gc1.takes_fused_path(x, csr) and gc2.takes_fused_path(x, csr) and gc1.precision == gc2.precision
            and gc1.out_features == gc2.in_features and gc1.out_features == gc2.out_features)


BLOCK_OUTPUTS = ("x1", "y1", "xy", "x", "out")


def gated_gcn_block(x, adj, gate1, gate2, gc1, gc2, want_gcn1=False, one_launch=True, want=None, dense_head=None):
    """x [B,T,H] fp32, adj dense [B,T,T] or BatchedCSR, gate1/gate2 [B,H], gc1/gc2 GraphConvolution.

    Returns the tensors the reference block produces: ``x1``, ``y1``, ``xy``, ``x`` (gated layer-2 output),
    ``out``, and ``gcn1`` (the ungated layer-1 output).  Nothing downstream of ``bert_amir5.py:640`` reads
    gcn1, so in inference it is only produced on request (``want_gcn1=True``; ``None`` otherwise) -- the
    one-launch block never writes it to HBM unless asked.  Under autograd (training) the two layers run as
    two launches and gcn1 is always returned.

    ``want`` (inference only; default: all of ``BLOCK_OUTPUTS``) names the outputs the caller will read; the others come back
    as ``None`` and what only they need is not computed.  The reference's evaluation keeps the logits alone
    (``train.py:227``) and those need just ``out`` (``bert_amir5.py:640,643``): ``want=("out",)`` launches only the W12 column
    tiles of the one-launch block -- half of the matrix work, no ``[B,T,H]`` store of ``x``, no regulariser -- and ``out`` is the
    full block's bit for bit (same tiles, same arithmetic).

    ``dense_head=(wt [H,C], bias or None)`` (inference): also returns ``logits = out @ wt (+ bias)``, the share of the
    classifier's ``dense`` that reads the block's output (``bert_amir5.py:643``) -- on the one-launch path in the SAME small
    launch that finishes ``xy`` (``ggcn_dense_head``), so the block + head are two launches in all."""
    if dense_head is not None:
        from .heads import dense_head as _dense_head
        if torch.is_grad_enabled() and (gc1._needs_grad(x, gate1, gate2) or gc2._needs_grad(x, gate2)):
            raise RuntimeError("dense_head= is an inference feature; under autograd apply the classifier's own dense layer")
        r = _gated_gcn_block(x, adj, gate1, gate2, gc1, gc2, want_gcn1, one_launch, want, _defer_xy=True)
        part = r.pop("_xy_partials", None)
        sig = dense_head[2] if len(dense_head) > 2 else None   # (wt, bias[, signal words]): see heads.dense_head
        if r["out"] is None:
            raise RuntimeError("dense_head= needs `out` among want=")
        if r["out"].shape[0] == 0:
            r["logits"] = r["out"].new_zeros((0, dense_head[0].shape[1]))
            return r
        if part is not None:
            r["logits"], r["xy"] = _dense_head(r["out"], dense_head[0], dense_head[1], partials=part, f_block=gc2.out_features, signal=sig)
        else:
            r["logits"] = _dense_head(r["out"], dense_head[0], dense_head[1], signal=sig)
        return r
    return _gated_gcn_block(x, adj, gate1, gate2, gc1, gc2, want_gcn1, one_launch, want)


def _gated_gcn_block(x, adj, gate1, gate2, gc1, gc2, want_gcn1=False, one_launch=True, want=None, _defer_xy=False):
    """gated_gcn_block proper.  _defer_xy: on the one-launch path leave the regulariser's partial sums under "_xy_partials"
    instead of launching ggcn_overlap_reduce (the caller's dense head finishes them)."""
    want = BLOCK_OUTPUTS if want is None else tuple(want)
    bad = [k for k in want if k not in BLOCK_OUTPUTS]
    if bad:
        raise ValueError("want: unknown output(s) %s (choose from %s; gcn1 has its own flag want_gcn1)" % (bad, list(BLOCK_OUTPUTS)))
    w_l1 = any(k in want for k in ("x1", "y1", "xy"))   # anything of bert_amir5.py:627-638
    w_x = "x" in want
    if x.shape[0] == 0:   # empty batch: what the reference's ops give on empty tensors (the mean of nothing is nan)
        gc1._check(x)
        B, T, F = 0, x.shape[1], gc2.out_features
        z2 = x.new_zeros((0, F), dtype=torch.float32)
        return {"gcn1": x.new_zeros((0, T, gc1.out_features)), "x1": x.new_zeros((0, gc1.out_features)),
                "y1": x.new_zeros((0, gc1.out_features)), "xy": x.new_full((), float("nan")),
                "x": x.new_zeros((0, T, F)), "out": z2}
    csr = adj if isinstance(adj, BatchedCSR) else gc1._as_csr(adj, x)
    training = torch.is_grad_enabled() and (gc1._needs_grad(x, gate1, gate2) or gc2._needs_grad(x, gate2))
    if training and set(want) != set(BLOCK_OUTPUTS):
        raise RuntimeError("want= selects outputs of the inference block; under autograd every output is produced")

    def pick(r):   # outputs the caller did not ask for are not handed out (whether or not a path had to compute them)
        for k in BLOCK_OUTPUTS:
            if k not in want:
                r[k] = None
        return r
    if not training and one_launch and takes_block_path(x, csr, gc1, gc2):
        # ---- ONE launch for :626-640 (+ one 1-block launch that finishes :638) ----
        gc1._check(x)
        range_guard.before(x.device)   # the lazy f16mx8 range report (a violation of an EARLIER launch raises here)
        lib = _capi.load_library()
        B, T, K = x.shape
        F = gc2.out_features
        dev = x.device
        x2d = x.reshape(B * T, K)
        if x2d.stride(1) != 1:
            x2d = x2d.contiguous()
        for name, g in (("gate1", gate1), ("gate2", gate2)):
            if g is None and name == "gate1" and not (w_l1 or want_gcn1):
                continue   # the eval form never reads gate1
            if not (isinstance(g, torch.Tensor) and g.is_cuda and g.dtype == torch.float32
                    and tuple(g.shape) == (B, F) and g.is_contiguous()):
                raise RuntimeError("%s must be a contiguous float32 [B,F]=[%d,%d] GPU tensor" % (name, B, F))
        with torch.cuda.device(dev):
            st = _capi.stream_of(dev)
            kprec = gc1.kernel_precision(x2d, csr)   # "f16mx6" where the fp6 kernel takes the shape, else "f16mx8"
            pack1, pack12, mid = _block_operands(gc1, gc2, lib, st, precision=kprec)
            layer1 = w_l1 or want_gcn1   # False: the eval form -- only the W12 column tiles are launched
            gcn1 = torch.empty(B * T, F, dtype=torch.float32, device=dev) if want_gcn1 else None
            xo = torch.empty(B * T, F, dtype=torch.float32, device=dev) if w_x else None
            x1 = torch.empty(B, F, dtype=torch.float32, device=dev) if layer1 else None
            y1 = torch.empty(B, F, dtype=torch.float32, device=dev) if layer1 else None
            out = torch.empty(B, F, dtype=torch.float32, device=dev)
            part = torch.empty(B, (F + 63) // 64, dtype=torch.float32, device=dev) if "xy" in want else None
            xy = torch.empty((), dtype=torch.float32, device=dev) if "xy" in want else None
            b1 = None if gc1.bias is None else gc1.bias.detach()
            b2 = None if gc2.bias is None else gc2.bias.detach()
            _capi.check(lib.ggcn_block_fused(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack1), _capi.ptr(pack12),
                                             _capi.ptr(csr.graph_ops), _capi.ptr(csr.graph_ops2(0 if kprec == "bf16x3" else 1)),
                                             _capi.ptr(b1), _capi.ptr(mid), _capi.ptr(b2),
                                             B, T, K, F, _capi.ptr(gate1 if layer1 else None), _capi.ptr(gate2), _capi.ptr(gcn1), F,
                                             _capi.ptr(xo), F, _capi.ptr(x1), _capi.ptr(y1), _capi.ptr(out),
                                             _capi.ptr(part), _capi.PREC[kprec], st), "ggcn_block_fused")
            if part is not None and not _defer_xy:
                _capi.check(lib.ggcn_overlap_reduce(_capi.ptr(part), B, F, _capi.ptr(xy), st), "ggcn_overlap_reduce")
        if kprec in ("f16mx8", "f16mx6"):
            range_guard.after(x.device)
        r = pick({"gcn1": None if gcn1 is None else gcn1.view(B, T, F), "x1": x1, "y1": y1, "xy": xy,
                  "x": None if xo is None else xo.view(B, T, F), "out": out})
        if _defer_xy and part is not None:
            r["_xy_partials"] = part
        return r
    if (not training and gc1.takes_fused_path(x, csr) and gc2.takes_fused_path(x, csr)
            and gc1.out_features == gc2.out_features):
        # two launches in all: layer 1 leaves its share of sum_f x1*y1 per (graph, 64 columns), layer 2's
        # launch adds them up before it starts on its own tiles (:638 costs no launch of its own)
        B, F = x.shape[0], gc1.out_features
        if not w_l1 and not want_gcn1 and one_launch and takes_folded_eval_path(x, csr, gc1, gc2):
            x2, out = _folded_eval(x, csr, gate2, gc1, gc2, w_x)      # the eval form of 33..256-node graphs: no product with W1
            return pick({"gcn1": None, "x1": None, "y1": None, "xy": None, "x": x2, "out": out})
        if not w_l1:   # gc2 needs gcn1 itself, nothing else of layer 1: no pools, no regulariser
            gcn1, _, _ = gc1.forward_gated(x, csr)
            x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2, want_out=w_x, want_pool_a=True)
            return pick({"gcn1": gcn1 if want_gcn1 else None, "x1": None, "y1": None, "xy": None, "x": x2, "out": out})
        part = torch.empty(B, (F + 63) // 64, dtype=torch.float32, device=x.device)
        xy = torch.empty((), dtype=torch.float32, device=x.device)
        gcn1, x1, y1 = gc1.forward_gated(x, csr, store_gate=None, pool_gate_a=gate1, pool_gate_b=gate2,
                                         want_pool_a=True, want_pool_b=True, overlap_partial=part)   # :626-636
        x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2, want_pool_a=True,
                                       want_out=w_x, overlap_reduce=(part, xy))        # :638-640
        return pick({"gcn1": gcn1, "x1": x1, "y1": y1, "xy": xy, "x": x2, "out": out})
    if not training and not w_l1:
        gcn1, _, _ = gc1.forward_gated(x, csr)
        x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2, want_out=w_x, want_pool_a=True)
        return pick({"gcn1": gcn1 if want_gcn1 else None, "x1": None, "y1": None, "xy": None, "x": x2, "out": out})
    gcn1, x1, y1 = gc1.forward_gated(x, csr, store_gate=None, pool_gate_a=gate1, pool_gate_b=gate2,
                                     want_pool_a=True, want_pool_b=True)           # :626-636
    if torch.is_grad_enabled() and (x1.requires_grad or y1.requires_grad):
        xy = (x1 * y1).sum(1).mean()   # differentiable form of :638 (the regulariser is trained on)
    else:
        xy = gate_overlap(x1, y1) if "xy" in want else None                        # :638
    x2, out, _ = gc2.forward_gated(gcn1, csr, store_gate=gate2, pool_gate_a=gate2,
                                   want_out=training or w_x, want_pool_a=True)   # :639-640
    return pick({"gcn1": gcn1, "x1": x1, "y1": y1, "xy": xy, "x": x2, "out": out})
