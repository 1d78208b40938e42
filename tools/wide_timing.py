#!/usr/bin/env python3
"""Layer time at LitBank's shape (T = 100, hidden 256, batch 256; constant.py:227, train.py:297) and at a
config-2-sized batch of 100-token graphs: one-launch wide-graph layer vs linear + aggregate.  Development tool."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
for B, T, H in ((256, 100, 256), (1024, 100, 768), (2048, 64, 768), (1024, 128, 768)):
    adj = synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev)
    g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    res = {}
    for prec in ("f16mx8", "bf16x3"):
        for fused in (True, False):
            m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = prec; m.fused = fused
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
    print("B=%d T=%d H=%d: " % (B, T, H) + "  ".join("%s %s %.1f us" % (p, "fused" if f else "unfused", v) for (p, f), v in res.items()))
