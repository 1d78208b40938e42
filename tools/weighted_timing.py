#!/usr/bin/env python3
"""A real-valued adjacency (gcn.py:33 takes any `adj`) on graphs of <= 32 nodes: the one-launch layer on
ggcn_graph_operands_weighted blocks against linear + aggregate, same process; the 0/1 one-launch layer beside them."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
for B, T, H in ((4096, 32, 768), (512, 32, 768), (1024, 20, 256)):
    adj = synth.dependency_batch(B, T, 3.0).astype(np.float32)
    wadj = adj * np.random.default_rng(0).uniform(0.1, 2.0, size=adj.shape).astype(np.float32)
    x = torch.randn(B, T, H, device=dev); g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = "f16mx8"
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    csr_w = pkg.BatchedCSR.from_dense(torch.from_numpy(wadj).to(dev))
    csr_b = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
    def f(csr, fused):
        m.fused = fused
        return m.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True)
    variants = {"weighted, one launch": (csr_w, True), "weighted, linear + aggregate": (csr_w, False), "0/1, one launch": (csr_b, True)}
    res = {k: [] for k in variants}
    with torch.no_grad():
        for _ in range(50):
            for c, fu in variants.values(): f(c, fu)
        for rnd in range(8):
            for name, (c, fu) in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f(c, fu)
                e1.record(); torch.cuda.synchronize()
                if rnd >= 2: res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
    print("B=%d T=%d H=%d: " % (B, T, H) + "   ".join("%s %.1f us" % (k, statistics.median(v)) for k, v in res.items()), flush=True)
