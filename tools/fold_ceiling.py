#!/usr/bin/env python3
"""What a folded two-layer block could save for graphs of 33..256 nodes: the folded form's W1 tiles
would skip the [N,F] store of gcn1 and its W12 tiles would read X instead of gcn1 and apply the adjacency twice.  The
ceiling of the saving is therefore  t(layer: store + pools) - t(layer: pools only)  per forward, BEFORE the second
application is paid.  Same process, HIP events, f16mx8.  Development tool."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
SHAPES = ((1024, 100, 768), (2048, 64, 768), (512, 160, 768), (512, 231, 768))
for B, T, H in SHAPES:
    adj = synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev)
    g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    (w1, b1), (w2, b2) = synth.layer_params(H, H, seed=1), synth.layer_params(H, H, seed=2)
    ls = []
    for w, b in ((w1, b1), (w2, b2)):
        m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = "f16mx8"; m.fused_max_t = 256
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        ls.append(m)
    m1, m2 = ls
    forms = {
        "layer1 store+2pools": lambda: m1.forward_gated(x, csr, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True),
        "layer1 2pools only": lambda: m1.forward_gated(x, csr, pool_gate_a=g1, pool_gate_b=g2, want_out=False, want_pool_a=True, want_pool_b=True),
        "layer1 store only": lambda: m1.forward_gated(x, csr),
        "layer2 store+pool": lambda: m2.forward_gated(x, csr, store_gate=g2, pool_gate_a=g2, want_pool_a=True),
        "layer2 pool only": lambda: m2.forward_gated(x, csr, store_gate=g2, pool_gate_a=g2, want_out=False, want_pool_a=True),
        "block (2 launches)": lambda: pkg.gated_gcn_block(x, csr, g1, g2, m1, m2),
        "block want=out": lambda: pkg.gated_gcn_block(x, csr, g1, g2, m1, m2, want=("out",)),
    }
    res = {}
    with torch.no_grad():
        for name, f in forms.items():
            for _ in range(60): f()
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            res[name] = statistics.median(ts)
    print("B=%d T=%d H=%d: " % (B, T, H) + "  ".join("%s %.1f" % kv for kv in res.items()), flush=True)
