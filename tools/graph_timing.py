import os, sys, statistics, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from ed_gated_gcn_amd.graphs import CapturedGatedBlock
dev = torch.device("cuda:0")
for (B, T, H) in ((256, 31, 256), (4096, 32, 768)):
    adj = synth.dependency_batch(B, T, 3.5, lengths=np.random.default_rng(0).integers(5, T + 1, size=B))
    rp, ci, _ = synth.csr_from_dense_host(adj); csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev); g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    ls = []
    for s in (1, 2):
        w, b = synth.layer_params(H, H, seed=s); m = pkg.GraphConvolution(H, H, None).to(dev)
        with torch.no_grad(): m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        ls.append(m)
    cap = CapturedGatedBlock(x, csr, g1, g2, *ls)
    with torch.no_grad():
        ref = pkg.gated_gcn_block(x, csr, g1, g2, *ls)
    got = cap(x, g1, g2)
    torch.cuda.synchronize()
    assert all(torch.equal(ref[k], got[k]) for k in ("x", "out", "x1", "y1")), "graph replay differs"
    def wall(fn, n=200):
        for _ in range(10): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    with torch.no_grad():
        te = wall(lambda: pkg.gated_gcn_block(x, csr, g1, g2, *ls))
    tg = wall(lambda: cap(x, g1, g2))
    print("B=%d T=%d H=%d: eager %.1f us/step, hipGraph replay %.1f us/step (bit-identical outputs)" % (B, T, H, te, tg))
