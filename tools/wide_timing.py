#!/usr/bin/env python3
"""Layer time at LitBank's shape (T = 100, hidden 256, batch 256; constant.py:227, train.py:297) and at a
config-2-sized batch of 100-token graphs, plus ACE cased (T = 231, constant.py:267): one-launch wide-graph layer vs linear + aggregate.  Development tool."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
if os.environ.get("LAB_SHAPES"):   # e.g. LAB_SHAPES="512,231,768;512,256,768"
    SHAPES = tuple(tuple(int(v) for v in t.split(",")) for t in os.environ["LAB_SHAPES"].split(";"))
else:
  SHAPES = ((256, 100, 256), (1024, 100, 768), (2048, 64, 768), (1024, 128, 768), (256, 231, 256), (64, 231, 768), (512, 231, 768), (512, 256, 768))
import numpy as np
def local_batch(B, T, span=6, seed=3):
    """Parse-like arcs: every token's head lies within `span` tokens (most real dependency arcs are short), one long arc
    per sentence -- the 32 x 32 adjacency blocks away from the diagonal are mostly empty."""
    rng = np.random.default_rng(seed)
    adj = np.zeros((B, T, T), dtype=np.uint8); i = np.arange(T); adj[:, i, i] = 1
    for b in range(B):
        head = np.maximum(0, i - rng.integers(1, span + 1, size=T)); head[0] = 0
        adj[b, i, head] = 1; adj[b, head, i] = 1
        a, c = rng.integers(0, T, size=2); adj[b, a, c] = adj[b, c, a] = 1
    return adj
for case in SHAPES + tuple(c + ("local",) for c in SHAPES if c[1] > 128):
    B, T, H = case[:3]
    adj = local_batch(B, T) if len(case) > 3 else synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev)
    g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    res = {}
    for prec in os.environ.get("LAB_PRECS", "f16mx8,bf16x3").split(","):
        for fused in (True, False):
            m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = prec; m.fused = fused; m.fused_max_t = 256
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
                f = lambda: m.forward_gated(x, csr, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True)
                for _ in range(20): f()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): f()
                    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            res[(prec, fused)] = statistics.median(ts)
    print("B=%d T=%d H=%d%s: " % (B, T, H, " (local arcs)" if len(case) > 3 else "") + "  ".join("%s %s %.1f us" % (p, "fused" if f else "unfused", v) for (p, f), v in res.items()))
