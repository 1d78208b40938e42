"""``GraphConvolution`` -- drop-in for ``models/gcn.py:9-45`` running on MI355X.

Same constructor ``(in_features, out_features, opt, bias=True)`` (``gcn.py:14``; ``opt``
is accepted and, as in the reference, not needed), same parameters ``weight [in,out]``
and ``bias [out]`` (``gcn.py:18,21``; left uninitialised like the reference --
``train.py:75-84`` initialises them), same ``forward(text, adj) -> [B,T,out]``
(``gcn.py:30-45``).  Underneath: dense adjacency -> batched CSR -> MFMA linear ->
one-wavefront-per-node gated aggregation, all in libggcn_hip.so.

Extras that the classifier block uses (``models/bert_amir5.py:621-640``):
``forward_gated`` fuses the per-sentence gate and the max-pool over tokens into the
aggregation pass, taking the gate as ``[B,H]`` instead of a materialised ``[B,T,H]``.
"""
import os

import torch
import torch.nn as nn

from . import _capi, range_guard
from .csr import BatchedCSR, cached_from_dense, tensor_version


def _require_gpu_f32(name, t, allow_half=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: this layer only runs on the GPU (libggcn_hip.so); "
                           "there is no CPU fallback" % (name, t.device))
    if t.dtype != torch.float32 and not (allow_half and t.dtype == torch.float16):
        raise RuntimeError("%s must be float32%s, got %s" % (name, " or float16" if allow_half else "", t.dtype))


class _GatedLayerFunction(torch.autograd.Function):
    """One gated layer under autograd (``train.py:115-121`` trains through gc1/gc2 and the gates).

    Forward = the inference kernels (fused layer or linear + aggregate, gate and max-pool in the
    epilogue).  Backward, for y = D.A.(X.W) + b, out = y*sg, pa = max_t y*ga, pb = max_t y*gb:

        dY, d_sg, d_ga, d_gb   HIP, one pass over the stored output (gate_pool_backward.hip)
        dH = A^T.(D.dY)        HIP, one wavefront per SOURCE node on the transposed CSR
        dX = dH.W^T            HIP bf16x3 MFMA linear on the packed W^T
        dW = X^T.dH            HIP split-K: bf16x3 main loop on X^T and packed dH (dweight_bx3.hip), or the
                               exact-fp32 MFMA form for precision 'fp32' (dweight_fp32.hip)
        db = sum_rows dY       HIP: per-graph sums from the gate/pool pass + ggcn_colsum
    """

    @staticmethod
    def forward(ctx, text, weight, bias, store_gate, gate_a, gate_b, layer, csr, want_pa, want_pb, dropout=None):
        with torch.no_grad():
            out, pa, pb = layer.forward_gated(text, csr, store_gate=store_gate, pool_gate_a=gate_a,
                                              pool_gate_b=gate_b, want_out=True, want_pool_a=want_pa,
                                              want_pool_b=want_pb, _internal=True, dropout=dropout)
        ctx.layer, ctx.csr, ctx.dropout = layer, csr, dropout
        ctx.save_for_backward(text, weight, out, store_gate, gate_a, gate_b)
        ctx.has_bias = bias is not None
        # an output the loss does not use arrives as None, not as a tensor of zeros: the [B,T,F] `out` of a layer whose pools alone
        # are used would otherwise cost a 400 MB fill AND a 400 MB read per step (every backward kernel takes d_out = NULL)
        ctx.set_materialize_grads(False)
        return out, pa, pb

    @staticmethod
    def backward(ctx, d_out, d_pa, d_pb):
        text, weight, out, store_gate, gate_a, gate_b = ctx.saved_tensors
        layer, csr = ctx.layer, ctx.csr
        if d_out is None and d_pa is None and d_pb is None:
            return (None,) * 11
        lib = _capi.load_library()
        B, T, K = text.shape
        F = layer.out_features
        dev = text.device

        def f32c(t, shape):
            if t is None:
                return None
            t = t.reshape(shape)
            return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()

        d_out2, d_pa, d_pb = f32c(d_out, (B * T, F)), f32c(d_pa, (B, F)), f32c(d_pb, (B, F))
        out2 = out.reshape(B * T, F)
        with torch.cuda.device(dev):
            st = _capi.stream_of(dev)
            need = ctx.needs_input_grad
            d_sg = torch.empty(B, F, dtype=torch.float32, device=dev) if (store_gate is not None and need[3]) else None
            d_ga = torch.empty(B, F, dtype=torch.float32, device=dev) if (gate_a is not None and need[4] and d_pa is not None) else None
            d_gb = torch.empty(B, F, dtype=torch.float32, device=dev) if (gate_b is not None and need[5] and d_pb is not None) else None
            d_bsum = torch.empty(B, F, dtype=torch.float32, device=dev) if (ctx.has_bias and need[2]) else None
            dh = torch.empty(B * T, F, dtype=torch.float32, device=dev)
            # graphs of up to 32 nodes with a 0/1 adjacency: gate / pool backward AND the transposed aggregation in one launch
            # (dY is consumed by nothing else: it never reaches memory)
            # (its 16-byte accesses need every operand 16-byte aligned: a contiguous view at an odd storage offset takes the two calls)
            one_pass = (T <= 32 and F % 4 == 0 and csr.is_binary and csr.rowmask is not None and csr.rowmask.is_cuda
                        and os.environ.get("GGCN_BACKWARD_TWO_PASS", "0") != "1"
                        and all(t is None or t.data_ptr() % 16 == 0
                                for t in (out2, store_gate, gate_a, gate_b, d_out2, d_pa, d_pb, dh, d_sg, d_ga, d_gb, d_bsum)))
            # dX on the two-unit f16mx8 product (ggcn_linear_scaled): the launch that makes dH also leaves max |dH|, from which the
            # linear derives a power-of-two scale on the device -- gradients have no range contract of their own
            dx = None
            # without gate dropout the backward runs on the matrix cores: dH_g = A_g^T . (D.dY_g) as an MFMA chain (ggcn_gate_pool_backward_mma)
            mma = (one_pass and ctx.dropout is None and os.environ.get("GGCN_BACKWARD_SCALAR", "0") != "1"
                   and csr.graph_ops is not None and csr.graph_ops_t is not None)
            # (the scaled linear wants its reduction length F % 32 == 0 and 16-byte rows; the scalar launch hands max |dH| over for
            # whole wavefronts of columns only)
            scaled_dx = (one_pass and need[0] and layer.precision == "f16mx8" and K % 4 == 0 and F % 32 == 0 and (mma or F % 256 == 0)
                         and os.environ.get("GGCN_DX_PRECISION", "f16mx8") == "f16mx8")
            dh_amax = torch.zeros(1, dtype=torch.float32, device=dev) if scaled_dx else None
            if mma:
                _capi.check(lib.ggcn_gate_pool_backward_mma(
                    _capi.ptr(out2), F, _capi.ptr(store_gate), _capi.ptr(gate_a), _capi.ptr(gate_b),
                    _capi.ptr(d_out2), F, _capi.ptr(d_pa), _capi.ptr(d_pb), _capi.ptr(csr.graph_ops), _capi.ptr(csr.graph_ops_t), B, T, F,
                    _capi.ptr(dh), F, _capi.ptr(d_sg), _capi.ptr(d_ga), _capi.ptr(d_gb), _capi.ptr(d_bsum), _capi.ptr(dh_amax), st),
                    "ggcn_gate_pool_backward_mma")
            elif one_pass:
                dp, dseed, (ss, sa, sb) = ctx.dropout if ctx.dropout is not None else (0.0, 0, (0, 0, 0))
                _capi.check(lib.ggcn_gate_pool_backward_agg(
                    _capi.ptr(out2), F, _capi.ptr(store_gate), _capi.ptr(gate_a), _capi.ptr(gate_b),
                    _capi.ptr(d_out2), F, _capi.ptr(d_pa), _capi.ptr(d_pb), _capi.ptr(csr.rowmask), B, T, F, _capi.ptr(dh), F,
                    _capi.ptr(d_sg), _capi.ptr(d_ga), _capi.ptr(d_gb), _capi.ptr(d_bsum), float(dp), int(dseed), ss, sa, sb,
                    _capi.ptr(dh_amax), st), "ggcn_gate_pool_backward_agg")
            elif ctx.dropout is None:
                dy = torch.empty(B * T, F, dtype=torch.float32, device=dev)
                _capi.check(lib.ggcn_gate_pool_backward(
                    _capi.ptr(out2), F, _capi.ptr(store_gate), _capi.ptr(gate_a), _capi.ptr(gate_b),
                    _capi.ptr(d_out2), F, _capi.ptr(d_pa), _capi.ptr(d_pb), B, T, F, _capi.ptr(dy), F,
                    _capi.ptr(d_sg), _capi.ptr(d_ga), _capi.ptr(d_gb), _capi.ptr(d_bsum), st), "ggcn_gate_pool_backward")
            else:   # the keep factors of the forward launch, drawn again from (seed, element)
                dy = torch.empty(B * T, F, dtype=torch.float32, device=dev)
                dp, dseed, (ss, sa, sb) = ctx.dropout
                _capi.check(lib.ggcn_gate_pool_backward_drop(
                    _capi.ptr(out2), F, _capi.ptr(store_gate), _capi.ptr(gate_a), _capi.ptr(gate_b),
                    _capi.ptr(d_out2), F, _capi.ptr(d_pa), _capi.ptr(d_pb), B, T, F, _capi.ptr(dy), F,
                    _capi.ptr(d_sg), _capi.ptr(d_ga), _capi.ptr(d_gb), _capi.ptr(d_bsum), float(dp), int(dseed), ss, sa, sb, st),
                    "ggcn_gate_pool_backward_drop")
            if not one_pass:
                csr_t = csr.transposed()
                inv = csr.inv_denominators()
                _capi.check(lib.ggcn_aggregate_t(_capi.ptr(dy), F, _capi.ptr(csr_t.rowptr), _capi.ptr(csr_t.colidx),
                                                 _capi.ptr(csr_t.vals), _capi.ptr(inv), B, T, F, _capi.ptr(dh), F, st),
                            "ggcn_aggregate_t")
            dw = db = None
            if need[0]:
                dx = torch.empty(B * T, K, dtype=torch.float32, device=dev)
                if scaled_dx:
                    pack_t = layer._packed_weight(lib, st, transposed=True, precision="f16mx8")
                    _capi.check(lib.ggcn_linear_scaled(_capi.ptr(dh), F, _capi.ptr(pack_t), _capi.ptr(dx), K, B * T, F, K,
                                                       _capi.ptr(dh_amax), st), "ggcn_linear_scaled(dX)")
                elif layer.precision in _capi.PACKED:
                    # gradients can be far below fp16's range (f16mx8 would flush them): dX always takes
                    # the bf16x3 linear, which keeps the fp32 exponent range
                    pack_t = layer._packed_weight(lib, st, transposed=True)
                    _capi.check(lib.ggcn_linear(_capi.ptr(dh), F, None, 0, _capi.ptr(pack_t), _capi.ptr(dx), K,
                                                B * T, F, K, _capi.PREC["bf16x3"], st), "ggcn_linear(dX)")
                else:   # exact-fp32 mode: W^T as a plain matrix, transposed once per weight update
                    key = (weight.data_ptr(), tensor_version(weight), weight.device)
                    if getattr(layer, "_wt_key", None) != key:
                        layer._wt, layer._wt_key = weight.detach().t().contiguous(), key
                    wt = layer._wt
                    _capi.check(lib.ggcn_linear(_capi.ptr(dh), F, _capi.ptr(wt), K, None, _capi.ptr(dx), K,
                                                B * T, F, K, _capi.PREC["fp32"], st), "ggcn_linear(dX)")
                dx = dx.view(B, T, K)
            if need[1]:
                x2d = text.reshape(B * T, K)
                if x2d.stride(1) != 1:
                    x2d = x2d.contiguous()
                # split-precision layers: bf16x3 on the forward's main loop (fp32 exponent range, ~1e-5);
                # precision "fp32": the exact fp32 MFMA form, which wants 16-byte aligned rows
                prec = "bf16x3" if layer.precision in _capi.PACKED else "fp32"
                dw = torch.empty(K, F, dtype=torch.float32, device=dev)
                ws = torch.empty(lib.ggcn_dweight_workspace_bytes(B * T, K, F, _capi.PREC[prec]),
                                 dtype=torch.uint8, device=dev)
                _capi.check(lib.ggcn_dweight(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(dh), F, B * T, K, F,
                                             _capi.ptr(dw), F, _capi.PREC[prec], _capi.ptr(ws), st), "ggcn_dweight")
            if d_bsum is not None:   # db = sum_rows dY: per-graph sums from the pass above, added over the graphs
                db = torch.empty(F, dtype=torch.float32, device=dev)
                ws = torch.empty(lib.ggcn_colsum_workspace_bytes(F), dtype=torch.uint8, device=dev)
                _capi.check(lib.ggcn_colsum(_capi.ptr(d_bsum), F, B, F, _capi.ptr(db), _capi.ptr(ws), st), "ggcn_colsum")
        return dx, dw, db, d_sg, d_ga, d_gb, None, None, None, None, None


class GraphConvolution(nn.Module):
    """Mean-normalised GCN layer: ``(adj @ (text @ W)) / (rowsum(adj) + 1) + b``."""

    def __init__(self, in_features, out_features, opt=None, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features, dtype=torch.float32))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features, dtype=torch.float32))
        else:
            self.register_parameter("bias", None)
        # arithmetic of the dense linear: "bf16x3" (3 bf16 MFMAs per product, ~1e-5 abs, the whole fp32 exponent range),
        # "f16mx8" (fp16 MFMA + one block-scaled fp8 correction MFMA) or "fp32" (exact fp32 MFMA).
        # Default "f16mx8" (26 % faster than "bf16x3", the arithmetic bench.py's headline is measured in).  It meets the
        # 1e-4 parity gate for activations of |x| <= 448 and hidden values below 65504, where the reference's fp32 matmul
        # has no limits -- so the kernels watch exactly that: a sticky device flag (overflow / accuracy window / hidden
        # bound, include/ggcn.h ggcn_range_bits), reported lazily without a device synchronisation (range_guard: after the
        # first forward and every 16th; check_range() asks now).  opt.ggcn_precision / GGCN_PRECISION = "bf16x3" lifts the limits.
        self.precision = getattr(opt, "ggcn_precision", None) or os.environ.get("GGCN_PRECISION", "f16mx8")
        # one-launch layer (fused_layer.hip) when the batch allows it: T <= fused_max_t, binary adjacency, split
        # precision.  128 by default; graphs of 193..256 nodes (ACE cased's ORI_ML = 231) take the eight-wavefront
        # form on their own (takes_fused_path); set 256 to send graphs of 129..192 nodes there as well.
        self.fused = bool(getattr(opt, "ggcn_fused", True)) and os.environ.get("GGCN_FUSED", "1") != "0"
        self.fused_max_t = int(getattr(opt, "ggcn_fused_max_t", None) or os.environ.get("GGCN_FUSED_MAX_T", "128"))
        # dense adjacency handed to forward(): None = let the device detect edge weights (one 4-byte
        # read-back per conversion), True = promise 0/1 entries like the reference's (graph.py:66-74)
        # and stay sync-free, False = always keep the values
        self.binary_adj = getattr(opt, "ggcn_binary_adj", None)
        self._pack = None
        self._pack_key = None

    def extra_repr(self):
        return "in_features=%d, out_features=%d, bias=%s, precision=%s" % (
            self.in_features, self.out_features, self.bias is not None, self.precision)

    # -- weight image for the split-precision linears, rebuilt only when the weight changes ----------
    def _packed_weight(self, lib, stream, transposed=False, precision=None):
        """MFMA-order image of W (forward) or W^T (backward's dX) for `precision` (default: the layer's), rebuilt when W
        changes; one image per precision is kept ("f16mx6" layers also need the "f16mx8" image for the shapes the
        fp6 kernel does not take)."""
        w = self.weight
        # the transposed image serves the backward's dX linear: bf16x3 (full range) unless the caller names f16mx8 (the scaled form)
        name = precision or (self.precision if not transposed else "bf16x3")
        name = name if (name in _capi.PACKED and (not transposed or name == "f16mx8")) else "bf16x3"
        prec = _capi.PREC[name]
        key = (w.data_ptr(), tensor_version(w), w.device)
        slot = (name, bool(transposed))
        if not isinstance(self._pack, dict):
            self._pack, self._pack_key = {}, {}
        if self._pack.get(slot) is None or self._pack_key.get(slot) != key:
            K, F = (self.out_features, self.in_features) if transposed else (self.in_features, self.out_features)
            pack = torch.empty(lib.ggcn_weight_pack_bytes(K, F, prec), dtype=torch.uint8, device=w.device)
            wc = w.detach()
            if not wc.is_contiguous():
                wc = wc.contiguous()
            _capi.check(lib.ggcn_weight_pack(_capi.ptr(wc), self.out_features, K, F, prec, 1 if transposed else 0,
                                             _capi.ptr(pack), stream), "ggcn_weight_pack")
            self._pack[slot], self._pack_key[slot] = pack, key
        return self._pack[slot]

    def kernel_precision(self, x2d=None, csr=None):
        """The arithmetic a launch really uses: "f16mx6" is taken by the one-launch layer / block for graphs of <= 32
        nodes with K % 32 == 0 and 16-byte aligned fp32 rows; every other shape of such a layer runs "f16mx8" (the same
        scheme with fp8 corrections: same accuracy class, its own weight image)."""
        if self.precision != "f16mx6":
            return self.precision
        ok = (csr is not None and csr.T <= 32 and x2d is not None and x2d.dtype == torch.float32
              and self.in_features % 32 == 0 and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0 and self.fused)
        return "f16mx6" if ok else "f16mx8"

    def _as_csr(self, adj, text):
        if isinstance(adj, BatchedCSR):
            if adj.B != text.shape[0] or adj.T != text.shape[1]:
                raise RuntimeError("CSR is for B=%d,T=%d but text is %s" % (adj.B, adj.T, tuple(text.shape)))
            if adj.device != text.device:
                raise RuntimeError("CSR and text are on different devices")
            return adj
        if not isinstance(adj, torch.Tensor):
            raise TypeError("adj must be a [B,T,T] tensor or a BatchedCSR")
        if adj.dim() != 3 or adj.shape[0] != text.shape[0] or adj.shape[1] != text.shape[1] \
                or adj.shape[2] != text.shape[1]:
            raise RuntimeError("adj %s does not match text %s" % (tuple(adj.shape), tuple(text.shape)))
        if adj.device != text.device:
            raise RuntimeError("adj and text are on different devices")
        return cached_from_dense(adj, binary=self.binary_adj)  # gcn.py:33 accepts any real dtype

    def _check(self, text):
        # float16 features (BASELINE configs[3]) are an extension: the reference itself raises a
        # dtype mismatch for half inputs.  Weights, bias, gates stay float32.
        _require_gpu_f32("text", text, allow_half=True)
        if self.precision == "f16" and text.dtype != torch.float16:
            raise RuntimeError("precision='f16' (plain fp16 MFMA) is for float16 features only; float32 features take "
                               "'bf16x3', 'f16mx8' or 'fp32'")
        if text.dtype == torch.float16 and self.precision not in _capi.PACKED:
            raise RuntimeError("float16 features need precision='bf16x3' or 'f16mx8' (the exact-fp32 linear is "
                               "fp32 only)")
        if text.dim() != 3 or text.shape[2] != self.in_features:
            raise RuntimeError("text must be [B,T,%d], got %s" % (self.in_features, tuple(text.shape)))
        if self.weight.device != text.device:
            raise RuntimeError("weight is on %s but text is on %s" % (self.weight.device, text.device))
        if self.precision == "f16mx6" and not _capi.has_f16mx6():
            raise RuntimeError("precision='f16mx6' is an experiment this libggcn_hip.so was built without (make -C "
                               "ed-gated-gcn_amd/csrc F16MX6=1); use 'f16mx8'")
        if self.precision not in _capi.PREC:
            raise RuntimeError("unknown precision %r (use 'bf16x3', 'f16mx8', 'f16mx6', 'fp32', or 'f16' for float16 features)"
                               % (self.precision,))

    def validate_range(self, text=None):
        """On-demand range check for ``precision='f16mx8'`` (one pass over the data, one read-back: NOT part of
        ``forward``).  Returns ``{"text_absmax", "weight_absmax"}``; raises ``RuntimeError`` when the fp16 range
        (|v| < 65504, finite) is exceeded -- where f16mx8 would saturate silently (f16mx8_core.h)."""
        lib = _capi.load_library()
        rep = {}
        for name, t in (("weight", self.weight.detach()), ("text", text)):
            if t is None:
                continue
            _require_gpu_f32(name, t, allow_half=True)
            t2 = t.reshape(-1, t.shape[-1])
            if t2.stride(1) != 1:
                t2 = t2.contiguous()
            out = torch.empty(2, dtype=torch.float32, device=t.device)
            with torch.cuda.device(t.device):
                _capi.check(lib.ggcn_absmax(_capi.ptr(t2), 1 if t2.dtype == torch.float16 else 0, t2.stride(0),
                                            t2.shape[0], t2.shape[1], _capi.ptr(out), _capi.stream_of(t.device)),
                            "ggcn_absmax")
            amax, bad = out.tolist()
            rep[name + "_absmax"] = amax
            if bad or amax >= 65504.0:
                raise RuntimeError("%s is outside the range of precision='f16mx8' (max |v| = %g%s; needs finite "
                                   "|v| < 65504): use precision='bf16x3'" % (name, amax, ", non-finite entries" if bad else ""))
        return rep

    def linear(self, x2d):
        """``hidden = text @ W`` (``gcn.py:34``) on [N,in] -> [N,out]."""
        lib = _capi.load_library()
        dev = x2d.device
        with torch.cuda.device(dev):
            st = _capi.stream_of(dev)
            y = torch.empty(x2d.shape[0], self.out_features, dtype=x2d.dtype, device=dev)
            kprec = self.kernel_precision()   # "f16mx6" has no stand-alone linear: f16mx8
            if x2d.dtype == torch.float16:
                pack = self._packed_weight(lib, st, precision=kprec)  # noqa
                _capi.check(lib.ggcn_linear_h(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack), _capi.ptr(y),
                                              y.stride(0), x2d.shape[0], self.in_features, self.out_features,
                                              _capi.PREC[kprec], st),
                            "ggcn_linear_h")
                return y
            w = self.weight.detach()
            if not w.is_contiguous():
                w = w.contiguous()
            pack = self._packed_weight(lib, st, precision=kprec) if kprec in _capi.PACKED else None
            _capi.check(lib.ggcn_linear(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(w), w.stride(0),
                                        _capi.ptr(pack), _capi.ptr(y), y.stride(0), x2d.shape[0],
                                        self.in_features, self.out_features, _capi.PREC[kprec], st),
                        "ggcn_linear")
        return y

    def _needs_grad(self, text, *gates):
        return torch.is_grad_enabled() and (text.requires_grad or self.weight.requires_grad
                                            or (self.bias is not None and self.bias.requires_grad)
                                            or any(g is not None and g.requires_grad for g in gates))

    WIDE_AUTO_MIN_T = 193     # graphs of 193..256 nodes fill >= 75 % of the 256-row slot of the eight-wavefront kernel
    WIDE_AUTO_MIN_T_FULL = {"f16mx8": 129, "bf16x3": 161}   # shorter graphs: only batches that fill whole rounds of workgroups (one per CU)
    WIDE_AUTO_FILL = 0.9

    def takes_fused_path(self, text, csr):
        """True when ``forward_gated`` will run as ONE launch (``ggcn_layer_fused``): graphs of <= ``fused_max_t``
        nodes (row masks exist up to 256), 0/1 adjacency, float32 features, a split-precision linear.  Beyond
        ``fused_max_t`` (128 by default), graphs of 193..256 nodes (ACE cased: ``ORI_ML = 231``, ``constant.py:267``) take
        the eight-wavefront form (``layer_fused_wide8_kernel``: one workgroup per graph x 256 columns) on their own: it
        wins over linear + aggregate by 15 % on large batches (512 x 231 x 768: 418 vs 492 us) and ties on small ones
        (128 x 231 x 768: 128 vs 134 us).  Shorter graphs leave part of the 256-row slot empty: the second row group runs
        a main loop compiled for its 1-3 live 32-row blocks (f16mx8) and both row groups share the epilogue's row steps,
        which wins when the workgroups fill whole rounds (512 x 129 x 768: 296 vs 302 us, 512 x 160: 324 vs 346, 512 x 192:
        349 vs 416) and loses on a fraction of a round (128 x 129 x 768: 89 vs 79 us, 128 x 160: 95 vs 90) -- those keep the two
        launches (``tools/wide_timing.py``).  The choice depends on the batch size and the device's CU count, and the two
        paths sum in different orders (both inside the parity gate): ``fused_max_t = 256`` (always one launch) or
        ``fused = False`` (never) pin it where bit-reproducibility across batch sizes matters."""
        if not (self.fused and self.precision in _capi.PACKED and csr.rowmask is not None and csr.is_binary
                and text.dtype == torch.float32):
            return False
        if csr.T <= self.fused_max_t:
            return True
        if self.fused_max_t < 128 or csr.T > 256:
            return False
        if csr.T >= self.WIDE_AUTO_MIN_T:
            return True
        if csr.T < self.WIDE_AUTO_MIN_T_FULL.get(self.precision, 161) or not text.is_cuda:
            return False
        wgs = text.shape[0] * ((self.out_features + 255) // 256)
        cus = torch.cuda.get_device_properties(text.device).multi_processor_count
        rounds = -(-wgs // cus)
        return rounds >= 2 and wgs >= self.WIDE_AUTO_FILL * rounds * cus

    def takes_weighted_path(self, text, csr):
        """True when ``forward_gated`` (inference) will run a REAL-valued adjacency (``gcn.py:33`` accepts any ``adj``) as ONE
        launch (``ggcn_layer_fused_weighted``): graphs of <= 32 nodes, float32 features, a split-precision linear, every entry
        of D.A_w inside the plane type (``BatchedCSR.graph_ops_weighted``).  Anything else: linear + aggregate."""
        if not (self.fused and self.precision in _capi.PACKED and not csr.is_binary and csr.T <= 32 and self.fused_max_t >= 32
                and text.dtype == torch.float32 and text.is_cuda):
            return False
        return csr.graph_ops_weighted(0 if self.precision == "bf16x3" else 1) is not None

    LONG_MAX_T = 512   # include/ggcn.h GGCN_LONG_MAX_T

    def takes_long_path(self, text, csr):
        """True when ``forward_gated`` will run as ONE launch of ``ggcn_layer_fused_h``: half features with
        ``precision="f16"``, graphs of 129..512 nodes (shorter ones leave most of the 512 row slots empty and stay
        with linear + aggregate), K % 64 == 0, F % 8 == 0."""
        return (self.fused and self.precision == "f16" and text.dtype == torch.float16
                and 128 < csr.T <= self.LONG_MAX_T and self.in_features % 64 == 0 and self.out_features % 8 == 0)

    def takes_dropout_path(self, text, csr):
        """True when the gates' training-mode dropout (``bert_amir5.py:621-625``) can be drawn inside the layer launch:
        every one-launch form (graphs of <= 256 nodes on the fused path), element index below 2^32."""
        return self.takes_fused_path(text, csr) and text.shape[0] * text.shape[1] * self.out_features < 2 ** 32

    def forward_gated(self, text, adj, store_gate=None, pool_gate_a=None, pool_gate_b=None,
                      want_out=True, want_pool_a=False, want_pool_b=False, _internal=False,
                      overlap_partial=None, overlap_reduce=None, dropout=None):
        """``_forward_gated`` between the two halves of the lazy f16mx8 range report (``range_guard``: no device
        synchronisation; a violation of an EARLIER launch raises here)."""
        guarded = (not _internal and self.precision in ("f16mx8", "f16mx6") and isinstance(text, torch.Tensor) and text.is_cuda)
        if guarded:
            range_guard.before(text.device)
        r = self._forward_gated(text, adj, store_gate, pool_gate_a, pool_gate_b, want_out, want_pool_a, want_pool_b, _internal,
                                overlap_partial, overlap_reduce, dropout)
        if guarded:
            range_guard.after(text.device)
        return r

    def check_range(self):
        """Synchronous verdict of the sticky f16mx8 range flag for this layer's device (one read-back): raises if an
        f16mx8 launch since the last report met |v| >= 65504 or an infinity, an activation beyond the accuracy window
        (|x| > 448), or weights and activations whose hidden values the one-launch layer cannot bound below 65504."""
        range_guard.check(self.weight.device)

    def _forward_gated(self, text, adj, store_gate=None, pool_gate_a=None, pool_gate_b=None,
                       want_out=True, want_pool_a=False, want_pool_b=False, _internal=False,
                       overlap_partial=None, overlap_reduce=None, dropout=None):
        """Layer + gate + max-pool in one aggregation pass.

        Returns ``(out [B,T,F] or None, pool_a [B,F] or None, pool_b [B,F] or None)`` with
        ``out = y * store_gate`` and ``pool_x = max_t (y * pool_gate_x)``, ``y`` being the
        plain layer output.  Gates are ``[B,F]`` (broadcast over tokens).

        One-launch path only (``takes_fused_path``): ``overlap_partial`` (float32 ``[B, ceil(F/64)]``)
        receives this layer's share of ``sum_f pool_a*pool_b``; ``overlap_reduce=(partials, xy)`` makes
        this launch reduce the partials an earlier launch wrote into the scalar ``xy``
        (``bert_amir5.py:638`` without its own launches).

        ``dropout=(p, seed, (stream_store, stream_a, stream_b))`` (one-launch path: ``takes_dropout_path``): the three gates
        are dropped per (token, feature) like the reference's repeated ``[B,T,H]`` gates (``bert_amir5.py:621-625``);
        stream 0 = not dropped, 1 / 2 = the two independent Bernoulli streams of ``seed`` (``include/ggcn.h``)."""
        self._check(text)
        if text.shape[0] == 0:   # an empty batch is a valid input of the reference (gcn.py:30-45): empty outputs
            B, T, F = 0, text.shape[1], self.out_features
            z = text.new_zeros((0, T, F))
            return ((z if want_out else None), (text.new_zeros((0, F), dtype=torch.float32) if want_pool_a else None),
                    (text.new_zeros((0, F), dtype=torch.float32) if want_pool_b else None))
        csr = self._as_csr(adj, text)
        if not _internal and self._needs_grad(text, store_gate, pool_gate_a, pool_gate_b):
            # training: the same kernels, wrapped in an autograd Function with a HIP backward
            if text.dtype != torch.float32:
                raise RuntimeError("training through the HIP layer needs float32 features")
            if dropout is not None and not self.takes_dropout_path(text, csr):
                raise RuntimeError("dropout= needs the one-launch layer (takes_dropout_path: takes_fused_path and B*T*F < 2^32)")
            out, pa, pb = _GatedLayerFunction.apply(text, self.weight, self.bias, store_gate, pool_gate_a,
                                                    pool_gate_b, self, csr, want_pool_a, want_pool_b, dropout)
            return (out if want_out else None), pa, pb
        lib = _capi.load_library()
        B, T, _ = text.shape
        F = self.out_features
        dev = text.device
        x2d = text.reshape(B * T, self.in_features)
        if x2d.stride(1) != 1:
            x2d = x2d.contiguous()
        for name, g in (("store_gate", store_gate), ("pool_gate_a", pool_gate_a), ("pool_gate_b", pool_gate_b)):
            if g is not None:
                _require_gpu_f32(name, g)
                if tuple(g.shape) != (B, F) or not g.is_contiguous():
                    raise RuntimeError("%s must be a contiguous [B,F]=[%d,%d] tensor, got %s"
                                       % (name, B, F, tuple(g.shape)))
        half = text.dtype == torch.float16
        use_fused = self.takes_fused_path(text, csr)
        if dropout is not None and not (use_fused and self.takes_dropout_path(text, csr)):
            raise RuntimeError("dropout= needs the one-launch layer (takes_dropout_path: takes_fused_path and B*T*F < 2^32)")
        if (overlap_partial is not None or overlap_reduce is not None) and not use_fused:
            raise RuntimeError("overlap_partial / overlap_reduce need the one-launch layer (takes_fused_path)")
        use_long = ((not use_fused) and self.takes_long_path(text, csr) and x2d.data_ptr() % 16 == 0
                    and x2d.stride(0) % 8 == 0)   # ggcn_layer_fused_h wants 16-byte aligned rows; other views: linear_h + aggregate_h
        use_weighted = (not use_fused) and dropout is None and self.takes_weighted_path(text, csr)
        hidden = None if (use_fused or use_long or use_weighted) else self.linear(x2d)
        with torch.cuda.device(dev):
            st = _capi.stream_of(dev)
            out = torch.empty(B * T, F, dtype=text.dtype, device=dev) if want_out else None
            pa = torch.empty(B, F, dtype=torch.float32, device=dev) if want_pool_a else None
            pb = torch.empty(B, F, dtype=torch.float32, device=dev) if want_pool_b else None
            bias = None if self.bias is None else self.bias.detach()
            if use_fused and dropout is not None:
                kprec = "f16mx8" if self.precision == "f16mx6" else self.precision   # the fp6 kernel has no dropout epilogue
                pack = self._packed_weight(lib, st, precision=kprec)
                dp, dseed, (ss, sa, sb) = dropout
                _capi.check(lib.ggcn_layer_fused_drop(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack), _capi.ptr(csr.rowmask), _capi.ptr(csr.graph_ops),
                                                      _capi.ptr(bias), B, T, self.in_features, F, _capi.ptr(store_gate),
                                                      _capi.ptr(pool_gate_a), _capi.ptr(pool_gate_b), _capi.ptr(out), F,
                                                      _capi.ptr(pa), _capi.ptr(pb), _capi.PREC[kprec], float(dp), int(dseed),
                                                      ss, sa, sb, st), "ggcn_layer_fused_drop")
                return (None if out is None else out.view(B, T, F)), pa, pb
            if use_fused:
                kprec = self.kernel_precision(x2d, csr)
                pack = self._packed_weight(lib, st, precision=kprec)
                _capi.check(lib.ggcn_layer_fused(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack),
                                                 _capi.ptr(csr.rowmask), _capi.ptr(csr.graph_ops if csr.T <= 32 else (csr.edge_lists if os.environ.get("GGCN_EDGE_LISTS", "1") != "0" else None)), _capi.ptr(bias), B, T,
                                                 self.in_features, F, _capi.ptr(store_gate),
                                                 _capi.ptr(pool_gate_a), _capi.ptr(pool_gate_b), _capi.ptr(out),
                                                 F, _capi.ptr(pa), _capi.ptr(pb), _capi.ptr(overlap_partial),
                                                 _capi.ptr(overlap_reduce[0]) if overlap_reduce else None,
                                                 _capi.ptr(overlap_reduce[1]) if overlap_reduce else None,
                                                 _capi.PREC[kprec], st),
                            "ggcn_layer_fused")
                return (None if out is None else out.view(B, T, F)), pa, pb
            if use_weighted:   # real-valued adjacency, graphs of <= 32 nodes: one launch on D.A_w operand blocks
                kprec = "bf16x3" if self.precision == "bf16x3" else "f16mx8"
                pack = self._packed_weight(lib, st, precision=kprec)
                zmid = getattr(self, "_zero_mid", None)
                if zmid is None or zmid.device != dev or zmid.numel() < F:
                    zmid = self._zero_mid = torch.zeros(F, dtype=torch.float32, device=dev)
                _capi.check(lib.ggcn_layer_fused_weighted(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack),
                                                          _capi.ptr(csr.graph_ops_weighted(0 if kprec == "bf16x3" else 1)),
                                                          _capi.ptr(bias), _capi.ptr(zmid), B, T, self.in_features, F,
                                                          _capi.ptr(store_gate), _capi.ptr(pool_gate_a), _capi.ptr(pool_gate_b),
                                                          _capi.ptr(out), F, _capi.ptr(pa), _capi.ptr(pb), None, None, None,
                                                          _capi.PREC[kprec], st), "ggcn_layer_fused_weighted")
                return (None if out is None else out.view(B, T, F)), pa, pb
            if use_long:   # long fp16 graphs (BASELINE configs[3]): linear + aggregation in one launch, hidden stays in LDS
                pack = self._packed_weight(lib, st)
                _capi.check(lib.ggcn_layer_fused_h(_capi.ptr(x2d), x2d.stride(0), _capi.ptr(pack), _capi.ptr(csr.rowptr),
                                                   _capi.ptr(csr.colidx), _capi.ptr(csr.vals), _capi.ptr(bias), B, T,
                                                   self.in_features, F, _capi.ptr(store_gate), _capi.ptr(pool_gate_a),
                                                   _capi.ptr(pool_gate_b), _capi.ptr(out), F, _capi.ptr(pa), _capi.ptr(pb), st),
                            "ggcn_layer_fused_h")
                return (None if out is None else out.view(B, T, F)), pa, pb
            agg = lib.ggcn_aggregate_h if half else lib.ggcn_aggregate
            _capi.check(agg(_capi.ptr(hidden), hidden.stride(0), _capi.ptr(csr.rowptr),
                                           _capi.ptr(csr.colidx), _capi.ptr(csr.vals), _capi.ptr(bias),
                                           B, T, F, _capi.ptr(store_gate), _capi.ptr(pool_gate_a),
                                           _capi.ptr(pool_gate_b), _capi.ptr(out), F, _capi.ptr(pa),
                                           _capi.ptr(pb), st), "ggcn_aggregate")
        return (None if out is None else out.view(B, T, F)), pa, pb

    def forward(self, text, adj):
        """``models/gcn.py:30-45``; ``adj`` is the reference's dense [B,T,T] (or a BatchedCSR)."""
        out, _, _ = self.forward_gated(text, adj)
        return out
