#!/usr/bin/env python3
"""Config-2 gated block: one launch (ggcn_block_fused) vs two launches (ggcn_layer_fused x 2), interleaved
in one process (cdna guide §5.4 rule 24).  Development tool.

    python tools/block_timing.py [graphs] [precision] [rounds]
"""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prec = sys.argv[2] if len(sys.argv) > 2 else "f16mx8"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
T, H = 32, 768
dev = torch.device("cuda:0")
adj = synth.dependency_batch(B, T, 4.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
gen = torch.Generator().manual_seed(1)
x = torch.randn(B, T, H, generator=gen).to(dev)
g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
ls = []
for s in (1, 2):
    w, b = synth.layer_params(H, H, seed=s)
    m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = prec
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    ls.append(m.eval())


def run(one, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        e0.record()
        for _ in range(n):
            pkg.gated_gcn_block(x, csr, g1, g2, *ls, one_launch=one)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for one in (True, False):
    run(one, 100)
res = {True: [], False: []}
for r in range(rounds):
    for one in (True, False):
        res[one].append(run(one, 100))
for one in (True, False):
    v = res[one]
    print("%-12s median %.1f us  min %.1f  max %.1f   (%d graphs, %s)" % ("one launch" if one else "two launches",
          statistics.median(v), min(v), max(v), B, prec))
