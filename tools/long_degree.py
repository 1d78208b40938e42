#!/usr/bin/env python3
"""One-launch long-graph layer against linear + aggregate over the average degree (256 x 512 x 1024 fp16): where the
staging area's 4096 ids per graph end.  Development tool."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
B, T, H = 64, 512, 1024
for deg in [float(v) for v in (sys.argv[1:] or ["6", "7.9", "8.5", "10", "14", "17"])]:
    adj = synth.dependency_batch(B, T, deg)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev).half()
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    res = {}
    for fused in (True, False):
        m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = "f16"; m.fused = fused
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
            f = lambda: m.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True)
            for _ in range(60): f()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        res[fused] = statistics.median(ts)
    print("degree %.1f (max edges per graph %d): one launch %.1f us, two launches %.1f us" % (
        deg, int((rp[T::T] - rp[:-1:T]).max()), res[True], res[False]), flush=True)
