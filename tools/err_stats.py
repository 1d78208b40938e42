#!/usr/bin/env python3
"""Max/rms error of every GPU mode against a float64 evaluation of the reference algorithm
(config-2 statistics, 256 graphs).  Development tool."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense

dev = torch.device("cuda:0")
B, T, H = 256, 32, 768
adj = synth.dependency_batch(B, T, 4.0)
gen = torch.Generator().manual_seed(synth.SEED)
x = torch.randn(B, T, H, generator=gen)
g1 = torch.sigmoid(torch.randn(B, H, generator=gen)); g2 = torch.sigmoid(torch.randn(B, H, generator=gen))
w1, b1 = synth.layer_params(H, H, seed=1); w2, b2 = synth.layer_params(H, H, seed=2)
t = torch.from_numpy
def block64(x, a, g1, g2, w1, b1, w2, b2):
    den = a.sum(2, keepdim=True) + 1
    gcn1 = (a @ (x @ w1)) / den + b1
    x2 = g2[:, None, :] * ((a @ (gcn1 @ w2)) / den + b2)
    return {"gcn1": gcn1, "x": x2, "out": x2.max(1)[0]}
ref = block64(x.double(), t(adj).double(), g1.double(), g2.double(), t(w1).double(), t(b1).double(),
              t(w2).double(), t(b2).double())
f32 = ref_dense.gated_block(x, t(adj).float(), g1, g2, t(w1), t(b1), t(w2), t(b2))
print("%-22s %s" % ("torch-cpu fp32", "  ".join("%s %.2e" % (k, float((f32[k].double() - ref[k]).abs().max())) for k in ("gcn1", "x", "out"))))
for prec, fused in (("fp32", False), ("bf16x3", False), ("bf16x3", True), ("bf16x3", "block"), ("f16mx8", False),
                    ("f16mx8", True), ("f16mx8", "block")):
    ls = []
    for w, b in ((w1, b1), (w2, b2)):
        m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = prec; m.fused = bool(fused)
        with torch.no_grad():
            m.weight.copy_(t(w)); m.bias.copy_(t(b))
        ls.append(m)
    with torch.no_grad():
        r = pkg.gated_gcn_block(x.to(dev), t(adj).to(dev), g1.to(dev), g2.to(dev), *ls, want_gcn1=True,
                                one_launch=(fused == "block"))
    print("%-22s %s" % (prec + ("-block" if fused == "block" else "-fused" if fused else ""), "  ".join(
        "%s max %.2e rms %.2e" % (k, float((r[k].cpu().double() - ref[k]).abs().max()),
                                  float((r[k].cpu().double() - ref[k]).pow(2).mean().sqrt())) for k in ("gcn1", "x", "out"))))
