#!/usr/bin/env python3
"""The failing case of test_fp16_long_graphs_one_launch in a loop: where do the pools differ?  Development tool."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
dev = torch.device("cuda:0")
B, T, K, F, degree = 9, 129, 192, 128, 4.0
rng = np.random.default_rng(T + K)
lens = np.array([T] + [int(v) for v in rng.integers(T // 3, T + 1, size=B - 1)])
adj = synth.dependency_batch(B, T, min(degree, T), seed=T, lengths=lens).astype(np.float32)
x16 = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half()
g1 = torch.from_numpy(rng.uniform(-1.0, 1.0, (B, F)).astype(np.float32))
g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32)))
w, b = synth.layer_params(K, F, seed=5)
def layer(fused):
    m = pkg.GraphConvolution(K, F).to(dev)
    m.precision = "f16"; m.fused = bool(fused)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    return m.eval()
m = layer(True)
xd, ad = x16.to(dev), torch.from_numpy(adj).to(dev)
csr = pkg.BatchedCSR.from_dense(ad)
kw = dict(store_gate=g2.to(dev), pool_gate_a=g1.to(dev), pool_gate_b=g2.to(dev), want_pool_a=True, want_pool_b=True)
first = None
noise = torch.empty(32 << 20, device=dev)
bad = 0
with torch.no_grad():
    for it in range(300):
        if it % 3 == 0: noise.normal_()
        out, pa, pb = m.forward_gated(xd, csr, **kw)
        torch.cuda.synchronize()
        if first is None: first = (out.clone(), pa.clone(), pb.clone())
        for name, cur, ref in (("out", out, first[0]), ("pa", pa, first[1]), ("pb", pb, first[2])):
            d = torch.nan_to_num((cur.float() - ref.float()).abs(), nan=1e9, posinf=1e9)
            if float(d.max()) > 0:
                bad += 1
                idx = torch.nonzero(d.reshape(d.shape[0], -1) > 0).cpu().numpy()
                print("iter %d %s: %d differ; rows %s cols %s values %s" % (it, name, len(idx), sorted(set(idx[:, 0].tolist()))[:10],
                      sorted(set(idx[:, 1].tolist()))[:16], cur.reshape(cur.shape[0], -1)[idx[0, 0], idx[0, 1]].item()))
print("lens", lens.tolist(), "bad", bad)
