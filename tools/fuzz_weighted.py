#!/usr/bin/env python3
"""Random real-valued adjacencies (graphs of 1..32 nodes) through the one-launch weighted layer and the gated block against
the oracle (development tool; usage: fuzz_weighted.py [cases] [seed])."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = torch.from_numpy
bad = one_launch = 0
for it in range(cases):
    T = int(rng.integers(1, 33)); B = int(rng.integers(1, 70))
    K = int(rng.choice([8, 20, 32, 64, 100, 256, 768])); F = int(rng.choice([4, 12, 34, 64, 100, 256, 300, 768]))
    prec = str(rng.choice(["f16mx8", "bf16x3"]))
    lens = rng.integers(1, T + 1, size=B)
    adj = synth.dependency_batch(B, T, float(min(T, rng.uniform(1.0, 6.0))), seed=int(rng.integers(1 << 30)), lengths=lens).astype(np.float32)
    kind = int(rng.integers(0, 3))   # 0 positive weights, 1 mixed signs (row sums stay positive), 2 wide dynamic range
    wts = rng.uniform(0.05, 2.0, size=adj.shape)
    if kind == 1: wts *= np.where(rng.random(adj.shape) < 0.25, -0.2, 1.0)
    if kind == 2: wts = np.exp(rng.uniform(-9.0, 3.0, size=adj.shape))
    adj = (adj * wts).astype(np.float32)
    x = rng.standard_normal((B, T, K)).astype(np.float32)
    (w1, b1), (w2, b2) = synth.layer_params(K, F, seed=it), synth.layer_params(F, F, seed=it + 1)
    g1 = torch.sigmoid(t(rng.standard_normal((B, F)).astype(np.float32))); g2 = torch.sigmoid(t(rng.standard_normal((B, F)).astype(np.float32)))
    def layer(w, b):
        m = pkg.GraphConvolution(w.shape[0], w.shape[1], None).to(dev); m.precision = prec
        with torch.no_grad(): m.weight.copy_(t(w)); m.bias.copy_(t(b))
        return m.eval()
    l1, l2 = layer(w1, b1), layer(w2, b2)
    xd, ad = t(x).to(dev), t(adj).to(dev)
    csr = pkg.BatchedCSR.from_dense(ad)
    took = (not csr.is_binary) and l1.takes_weighted_path(xd, csr)
    one_launch += bool(took)
    with torch.no_grad():
        r = pkg.gated_gcn_block(xd, csr, g1.to(dev), g2.to(dev), l1, l2, want_gcn1=True)
    ref = ref_dense.gated_block(t(x), t(adj), g1, g2, t(w1), t(b1), t(w2), t(b2))
    worst = 0.0
    for k in ("gcn1", "x1", "y1", "x", "out"):
        scale = max(1.0, float(ref[k].abs().max()))
        worst = max(worst, float((r[k].cpu() - ref[k]).abs().max()) / (1e-4 * scale))
    worst = max(worst, abs(float(r["xy"]) - float(ref["xy"])) / (1e-4 * max(1.0, abs(float(ref["xy"])))))
    if not worst <= 1.0:
        bad += 1
        print("FAIL case %d B=%d T=%d K=%d F=%d %s kind=%d one_launch=%s worst ratio %.3f" % (it, B, T, K, F, prec, kind, took, worst), flush=True)
print("cases %d, weighted one-launch layers in %d, failures %d" % (cases, one_launch, bad))
