#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

ORACLE TOOLING (test infrastructure, not product code).  Runs only where
``/root/reference`` exists (never on the GPU box).  The fixtures it writes hold
tensors only: seeded inputs and the outputs the reference's own classes
produced for them.

The reference's ``models/gcn.py:3-4`` and ``models/bert_amir5.py:3-4`` import a
``layers`` package that the repository does not ship; the imported names are
unused by the classes exercised here, so empty stand-in modules are registered
in ``sys.modules`` before the import.  BERT weights cannot be
fetched offline, so ``BertAmir55`` is driven by a local encoder
stand-in that returns seeded hidden states in the old
``(list of 12 x [B,L,768], pooled)`` form (``models/bert_amir5.py:591-596``).

Fixtures:
  G1  gcn_config1.npz      GraphConvolution.forward, B=1,T=32,H=300, seed 14181
  G2  gcn_sweep.npz        same layer over H in {64,200,256,768}, T in {5,31,32},
                           B in {1,4}; padded (identity-only) rows; bias=False;
                           bool / int64 / float32 / weighted adjacency
  G3  amir55_block.npz     BertAmir55 in eval(): LSTM output, adj, both gates,
                           gc1/gc2 outputs, xy, logits, kl, scores + the gate
                           MLP / GCN parameters
  G4  amir55_full.npz      the whole BertAmir55: inputs, seeds, reference outputs
  G5  amir54_full.npz      BertAmir54 (bert_amir5.py:434) on the same inputs
  G6  amir55nogate_full.npz  BertAmir55NoGate (bert_amir5.py:654) on the same inputs

Usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import math
import os
import sys
import types

import numpy as np
import torch

SEED = 14181  # train.py:307


def _import_reference(ref_root):
    for name, attr in (("layers", None),
                       ("layers.squeeze_embedding", "SqueezeEmbedding"),
                       ("layers.dynamic_rnn", "DynamicLSTM")):
        m = types.ModuleType(name)
        if attr:
            setattr(m, attr, type(attr, (), {}))
        sys.modules.setdefault(name, m)
    sys.path.insert(0, ref_root)
    from models.gcn import GraphConvolution          # noqa: E402
    from models.bert_amir5 import BertAmir55, BertAmir54, BertAmir55NoGate         # noqa: E402
    return GraphConvolution, BertAmir55, BertAmir54, BertAmir55NoGate


def _reset_params(module, gen):
    """train.py:75-84 applied to one child module."""
    for p in module.parameters():
        if not p.requires_grad:
            continue
        with torch.no_grad():
            if p.dim() > 1:
                # torch.nn.init.xavier_uniform_ (the default --initializer)
                fan_out, fan_in = p.shape[0], p.shape[1]
                a = math.sqrt(6.0 / (fan_in + fan_out))
                p.uniform_(-a, a, generator=gen)
            else:
                s = 1.0 / math.sqrt(p.shape[0])
                p.uniform_(-s, s, generator=gen)


def _dep_adj(rng, B, T, lengths=None, extra=0.08):
    """Symmetric 0/1 adjacency with self loops, identity on padding rows
    (graph.py:66-74: eye(ORI_ML) + symmetric edges)."""
    adj = np.zeros((B, T, T), dtype=np.int64)
    for b in range(B):
        n = T if lengths is None else int(lengths[b])
        a = np.eye(T, dtype=np.int64)
        for i in range(1, n):
            p = int(rng.integers(0, i))
            a[i, p] = a[p, i] = 1
        for _ in range(int(extra * n * n)):
            i, j = (int(v) for v in rng.integers(0, n, size=2))
            a[i, j] = a[j, i] = 1
        adj[b] = a
    return adj


def make_g1(GraphConvolution, out_dir):
    gen = torch.Generator().manual_seed(SEED)
    rng = np.random.default_rng(SEED)
    B, T, H = 1, 32, 300
    layer = GraphConvolution(H, H, opt=None)
    _reset_params(layer, gen)
    text = torch.randn(B, T, H, generator=gen)
    adj = torch.from_numpy(_dep_adj(rng, B, T)).float()
    with torch.no_grad():
        out = layer(text, adj)
    np.savez_compressed(os.path.join(out_dir, "gcn_config1.npz"),
                        text=text.numpy(), adj=adj.numpy().astype(np.uint8),
                        weight=layer.weight.detach().numpy(),
                        bias=layer.bias.detach().numpy(), out=out.numpy())


def make_g2(GraphConvolution, out_dir):
    gen = torch.Generator().manual_seed(SEED + 1)
    rng = np.random.default_rng(SEED + 1)
    blob = {}
    cases = []
    idx = 0
    for (B, T, Hin, Hout, use_bias, adj_kind, padded) in [
            (1, 5, 64, 64, True, "float", False),
            (4, 31, 200, 200, True, "float", True),
            (4, 32, 256, 256, True, "bool", True),
            (1, 32, 768, 768, True, "float", False),
            (4, 31, 256, 256, False, "int", True),
            (4, 32, 64, 200, True, "weighted", True),
            (1, 31, 200, 64, False, "weighted", False),
            (4, 5, 768, 64, True, "int", True)]:
        layer = GraphConvolution(Hin, Hout, opt=None, bias=use_bias)
        _reset_params(layer, gen)
        text = torch.randn(B, T, Hin, generator=gen)
        lengths = rng.integers(max(2, T // 4), T + 1, size=B) if padded else None
        a = _dep_adj(rng, B, T, lengths)
        if adj_kind == "bool":
            adj = torch.from_numpy(a.astype(bool))
        elif adj_kind == "int":
            adj = torch.from_numpy(a)
        elif adj_kind == "float":
            adj = torch.from_numpy(a).float()
        else:  # non-binary adjacency acts as edge weights (gcn.py:33,41)
            w = rng.uniform(0.25, 2.0, size=a.shape).astype(np.float32)
            adj = torch.from_numpy(a.astype(np.float32) * w)
        with torch.no_grad():
            out = layer(text, adj)
        k = "c%d_" % idx
        blob[k + "text"] = text.numpy()
        blob[k + "adj"] = adj.numpy()
        blob[k + "weight"] = layer.weight.detach().numpy()
        if use_bias:
            blob[k + "bias"] = layer.bias.detach().numpy()
        blob[k + "out"] = out.numpy()
        cases.append("%d,%d,%d,%d,%d,%s,%d" % (B, T, Hin, Hout, int(use_bias), adj_kind, int(padded)))
        idx += 1
    blob["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(out_dir, "gcn_sweep.npz"), **blob)


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.ref_amir55 import EncoderStandIn as _EncoderStandIn  # noqa: E402  (seeded encoder stand-in)


def make_g3(BertAmir55, out_dir):
    gen = torch.Generator().manual_seed(SEED + 2)
    rng = np.random.default_rng(SEED + 2)
    B, ORI_ML, BERT_ML, NCLS = 4, 31, 48, 34          # constant.py:230-240 (ACE34)
    opt = types.SimpleNamespace(device="cpu", dropout=0.25, polarities_dim=NCLS)
    model = BertAmir55(_EncoderStandIn(SEED + 3), opt)
    for child in model.children():                     # train.py:75-84
        if not isinstance(child, _EncoderStandIn):
            _reset_params(child, gen)
    model.eval()

    sent_len = np.array([31, 17, 9, 24])
    bert_len = np.minimum(sent_len + 6, BERT_ML)
    adj = _dep_adj(rng, B, ORI_ML, sent_len).astype(np.float32)
    transform = np.zeros((B, ORI_ML, BERT_ML), dtype=np.float32)
    for b in range(B):                                 # data_utils.py:749-766 shape
        for t in range(int(sent_len[b])):
            transform[b, t, 1 + t] = 1.0
    inputs = {
        "sentence_length": torch.from_numpy(sent_len),
        "cls_text_sep_length": torch.from_numpy(bert_len),
        "cls_text_sep_indices": torch.zeros(B, BERT_ML, dtype=torch.long),
        "cls_text_sep_segments_ids": torch.zeros(B, BERT_ML, dtype=torch.long),
        "transform": torch.from_numpy(transform),
        "anchor_index": torch.from_numpy(np.array([3, 0, 8, 11])),
        "dist_to_target": torch.from_numpy(rng.integers(0, 6, size=(B, ORI_ML))),
        "dependency_graph": torch.from_numpy(adj),
    }

    cap = {}

    def hook(name, take_input=False):
        def fn(mod, inp, out):
            if take_input:
                cap[name + "_in"] = [t.detach().clone() for t in inp]
            cap[name] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return fn

    model.lstm.register_forward_hook(hook("lstm"))
    model.gate1.register_forward_hook(hook("gate1", True))
    model.gate2.register_forward_hook(hook("gate2"))
    model.gc1.register_forward_hook(hook("gc1", True))
    model.gc2.register_forward_hook(hook("gc2", True))
    model.dense.register_forward_hook(hook("dense", True))
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)

    H = 256
    T = int(sent_len.max())
    assert torch.equal(cap["gc1_in"][0], cap["lstm"])
    assert torch.equal(cap["gc2_in"][0], cap["gc1"])
    blob = {
        "lstm_out": cap["lstm"].numpy(),                       # x fed to gc1 (:626)
        "adj": cap["gc1_in"][1].numpy().astype(np.uint8),      # adj[:, :T, :T] (:589)
        "aspect": cap["gate1_in"][0].numpy(),                  # (:618)
        "gate1": cap["gate1"].numpy(), "gate2": cap["gate2"].numpy(),   # [B,H] (:621-622)
        "gcn1": cap["gc1"].numpy(),                            # (:626)
        "gc2_out": cap["gc2"].numpy(),                         # ungated gc2(gcn1, adj) (:639)
        "out": cap["dense_in"][0][:, -H:].numpy(),             # max_t(gate2*gc2) (:640,643)
        "xy": xy.numpy(), "kl": kl.numpy(),
        "logits": logits.numpy(), "scores": scores.numpy(),
        "T": np.array(T),
    }
    sd = model.state_dict()
    for k in ("gc1.weight", "gc1.bias", "gc2.weight", "gc2.bias",
              "gate1.1.weight", "gate1.1.bias", "gate1.3.weight", "gate1.3.bias",
              "gate2.1.weight", "gate2.1.bias", "gate2.3.weight", "gate2.3.bias",
              "fc.0.weight", "fc.0.bias"):
        blob["p_" + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(out_dir, "amir55_block.npz"), **blob)
    # G4: the whole classifier.  Inputs + the reference's outputs + the seeds: the parameters are
    # re-drawn from the seed on the test side (40 MB of LSTM weights do not belong in a fixture).
    full = {"seed_params": np.array(SEED + 2), "seed_encoder": np.array(SEED + 3),
            "n_class": np.array(NCLS), "logits": logits.numpy(), "xy": xy.numpy(), "kl": kl.numpy(),
            "scores": scores.numpy()}
    for k, v in inputs.items():
        full["in_" + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, "amir55_full.npz"), **full)
    return inputs


def make_g5(cls, name, inputs, out_dir):
    """G5 / G6: the other live classifiers of train.py:268-282 (BertAmir54, BertAmir55NoGate) on G4's inputs: the seeds and
    the reference's outputs (parameters are re-drawn from the seed on the test side)."""
    gen = torch.Generator().manual_seed(SEED + 4)
    NCLS = 34
    opt = types.SimpleNamespace(device="cpu", dropout=0.25, polarities_dim=NCLS)
    model = cls(_EncoderStandIn(SEED + 3), opt)
    for child in model.children():                     # train.py:75-84
        if not isinstance(child, _EncoderStandIn):
            _reset_params(child, gen)
    model.eval()
    with torch.no_grad():
        logits, xy, kl, scores = model(inputs)
    full = {"seed_params": np.array(SEED + 4), "seed_encoder": np.array(SEED + 3), "n_class": np.array(NCLS),
            "logits": logits.numpy(), "xy": np.array(float(xy), dtype=np.float32), "kl": kl.numpy(), "scores": scores.numpy()}
    for k, v in inputs.items():
        full["in_" + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, name), **full)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                   "..", "tests", "golden"))
    args = ap.parse_args()
    torch.set_num_threads(1)   # one thread: fixtures do not depend on the host's core count
    os.makedirs(args.out, exist_ok=True)
    GraphConvolution, BertAmir55, BertAmir54, BertAmir55NoGate = _import_reference(args.ref)
    make_g1(GraphConvolution, args.out)
    make_g2(GraphConvolution, args.out)
    inputs = make_g3(BertAmir55, args.out)
    make_g5(BertAmir54, "amir54_full.npz", inputs, args.out)
    make_g5(BertAmir55NoGate, "amir55nogate_full.npz", inputs, args.out)
    for f in sorted(os.listdir(args.out)):
        print(f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    main()
