#!/usr/bin/env python3
"""Random small-graph batches (1..32 nodes) through gated_gcn_block -- the one-launch block where it applies, one launch per
layer or linear + aggregate otherwise -- against the oracle's block, in every split precision.  Development tool.
usage: fuzz_block.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    T = int(rng.integers(1, 33))
    B = int(rng.integers(1, 40))
    H = int(rng.choice([32, 64, 96, 100, 128, 256, 260]))
    prec = str(rng.choice(["f16mx8", "f16mx6", "bf16x3"] if pkg._capi.has_f16mx6() else ["f16mx8", "bf16x3"]))
    lens = np.array([T] + [int(v) for v in rng.integers(1, T + 1, size=B - 1)])
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=int(rng.integers(1 << 30)), lengths=lens).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
    w1, b1 = synth.layer_params(H, H, seed=int(rng.integers(1 << 30)))
    w2, b2 = synth.layer_params(H, H, seed=int(rng.integers(1 << 30)))
    g1 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, H)).astype(np.float32)))
    def layer(w, b):
        m = pkg.GraphConvolution(H, H, None).to(dev)
        m.precision = prec
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        return m.eval()
    try:
        l1, l2 = layer(w1, b1), layer(w2, b2)
        with torch.no_grad():
            r = pkg.gated_gcn_block(x.to(dev), torch.from_numpy(adj).to(dev), g1.to(dev), g2.to(dev), l1, l2, want_gcn1=bool(rng.integers(0, 2)))
        ref = ref_dense.gated_block(x, torch.from_numpy(adj), g1, g2, torch.from_numpy(w1), torch.from_numpy(b1), torch.from_numpy(w2), torch.from_numpy(b2))
        errs = {}
        for k in ("x", "out", "x1", "y1"):
            scale = max(1.0, float(ref[k].abs().max()))
            errs[k] = float((r[k].cpu() - ref[k]).abs().max()) / scale
        errs["xy"] = abs(float(r["xy"]) - float(ref["xy"])) / max(1.0, abs(float(ref["xy"])))
        ok = all(np.isfinite(e) and e <= 2e-4 for e in errs.values())
    except Exception as e:   # noqa: BLE001
        ok, errs = False, {"exc": repr(e)[:200]}
    if not ok:
        bad += 1
        print("FAIL", dict(B=B, T=T, H=H, prec=prec), errs)
print("cases %d, failures %d" % (cases, bad))
