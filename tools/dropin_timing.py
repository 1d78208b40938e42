#!/usr/bin/env python3
"""What a drop-in user sees: GraphConvolution.forward(text, dense adj) and the gated block with the
reference's dense float32 adjacency, CSR build included.  Development tool."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth

dev = torch.device("cuda:0")
B, T, H = 4096, 32, 768
adj = torch.from_numpy(synth.dependency_batch(B, T, 4.0)).float().to(dev)
x = torch.randn(B, T, H, device=dev)
g1 = torch.sigmoid(torch.randn(B, H, device=dev)); g2 = torch.sigmoid(torch.randn(B, H, device=dev))
ls = []
for s in (1, 2):
    w, b = synth.layer_params(H, H, seed=s)
    m = pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    ls.append(m)

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    return statistics.median(ts)

with torch.no_grad():
    print("csr from dense (auto binary detect, 1 sync): %.1f us" % timeit(lambda: pkg.BatchedCSR.from_dense(adj)))
    print("csr from dense (binary=True, no sync):       %.1f us" % timeit(lambda: pkg.BatchedCSR.from_dense(adj, binary=True)))
    print("layer forward(text, dense adj):               %.1f us" % timeit(lambda: ls[0](x, adj)))
    csr = pkg.BatchedCSR.from_dense(adj)
    print("layer forward(text, csr):                     %.1f us" % timeit(lambda: ls[0](x, csr)))
    print("gated block (dense adj):                      %.1f us" % timeit(lambda: pkg.gated_gcn_block(x, adj, g1, g2, *ls)))
    print("gated block (csr):                            %.1f us" % timeit(lambda: pkg.gated_gcn_block(x, csr, g1, g2, *ls)))
    print("reference ops on GPU (torch, dense):          %.1f us" % timeit(lambda: ((adj @ (x @ ls[0].weight)) / (adj.sum(2, keepdim=True) + 1) + ls[0].bias)))
