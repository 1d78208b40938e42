#!/usr/bin/env python3
"""Large random batches through ggcn_block_fused: the eight-wavefront kernel (what large batches take) against the four-wavefront
kernel (GGCN_BLOCK_FORM=4), bit for bit, and a slice against the oracle.  Development tool: fuzz_block8.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = pkg.load_library()
t = torch.from_numpy
bad = took = 0
for it in range(cases):
    H = int(rng.choice([256, 512, 768]))
    T = int(rng.integers(1, 33))
    need = 6 * 256 // (H // 256)                       # workgroups of four graphs for six rounds
    B = int(rng.choice([4 * need, 4 * need + int(rng.integers(1, 400)), 3 * 256 // (H // 256) * 4]))
    lens = rng.integers(1, T + 1, size=B) if rng.random() < 0.7 else None
    adj = synth.dependency_batch(B, T, float(min(T, rng.uniform(1.0, 5.0))), seed=int(rng.integers(1 << 30)), lengths=lens)
    x = t(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
    g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
    ls = []
    for s in (it, it + 1):
        w, b = synth.layer_params(H, H, seed=s)
        m = pkg.GraphConvolution(H, H, None).to(dev).eval(); m.precision = "f16mx8"
        with torch.no_grad(): m.weight.copy_(t(w)); m.bias.copy_(t(b))
        ls.append((m, w, b))
    csr = pkg.BatchedCSR.from_dense(t(adj).to(dev))
    form = lib.ggcn_block_fused_form(B, T, H, H)
    took += form == 8
    with torch.no_grad():
        r8 = pkg.gated_gcn_block(x, csr, g1, g2, ls[0][0], ls[1][0])
        os.environ["GGCN_BLOCK_FORM"] = "4"
        r4 = pkg.gated_gcn_block(x, csr, g1, g2, ls[0][0], ls[1][0])
        os.environ.pop("GGCN_BLOCK_FORM")
    same = all(torch.equal(r8[k], r4[k]) for k in ("x1", "y1", "x", "out")) and float(r8["xy"]) == float(r4["xy"])
    sl = slice(B - 24, B)
    ref = ref_dense.gated_block(x[sl].cpu(), t(adj[sl]).float(), g1[sl].cpu(), g2[sl].cpu(), t(ls[0][1]), t(ls[0][2]), t(ls[1][1]), t(ls[1][2]))
    worst = max(float((r8[k][sl].cpu() - ref[k]).abs().max()) / (1e-4 * max(1.0, float(ref[k].abs().max()))) for k in ("x1", "y1", "x", "out"))
    ok = same and worst <= 1.0
    bad += not ok
    print("case %d B=%d T=%d H=%d ragged=%s form=%d: bitwise %s, oracle ratio %.3f" % (it, B, T, H, lens is not None, form, same, worst), flush=True)
print("cases %d, eight-wavefront kernel in %d, failures %d" % (cases, took, bad))
