#!/usr/bin/env python3
"""The eight-wavefront one-launch layer at ACE-cased shapes with the precomputed edge lists (ggcn_graph_edge_lists) and without
(GGCN_EDGE_LISTS=0: every workgroup builds its lists), same process.  Development tool."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
for B, T, H in ((512, 231, 768), (512, 160, 768), (512, 256, 768), (128, 231, 768)):
    adj = synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev); g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = "f16mx8"; m.fused_max_t = 256
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    f = lambda: m.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True)
    res = {"lists": [], "in-kernel": []}
    with torch.no_grad():
        for _ in range(100): f()
        for rnd in range(8):
            for name in res:
                os.environ["GGCN_EDGE_LISTS"] = "1" if name == "lists" else "0"
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                if rnd >= 2: res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
    print("B=%d T=%d H=%d: " % (B, T, H) + "  ".join("%s %.1f us" % (k, statistics.median(v)) for k, v in res.items()), flush=True)
