#!/usr/bin/env python3
"""Random shapes through the one-launch layer against linear + aggregate of the same precision (and the oracle on small
cases): graphs of 33..256 nodes, ragged lengths, odd K / F, padded leading dimensions, missing gates / outputs.
Development tool.  usage: fuzz_wide.py [cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    T = int(rng.choice([33, 40, 64, 65, 96, 100, 128, 129, 150, 192, 193, 200, 231, 255, 256]))
    B = int(rng.integers(1, 12))
    K = int(rng.choice([32, 64, 96, 100, 128, 250, 256]))
    F = int(rng.choice([int(v) for v in os.environ["FUZZ_F"].split(",")] if os.environ.get("FUZZ_F") else [32, 60, 64, 100, 128, 256, 260, 300, 512]))
    deg = float(rng.choice([2.0, 4.0, 9.0, 20.0]))
    prec = str(rng.choice(["f16mx8", "bf16x3"]))
    lens = np.array([T] + [int(v) for v in rng.integers(max(1, T // 3), T + 1, size=B - 1)])
    adj = synth.dependency_batch(B, T, min(deg, T), seed=int(rng.integers(1 << 30)), lengths=lens).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32))
    w, b = synth.layer_params(K, F, seed=int(rng.integers(1 << 30)))
    use_bias, use_sg, use_out = bool(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 5))
    g1 = torch.from_numpy(rng.uniform(-1, 1, (B, F)).astype(np.float32)).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    def layer(fused):
        m = pkg.GraphConvolution(K, F, None, bias=use_bias).to(dev)
        m.precision, m.fused, m.fused_max_t = prec, fused, 256
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w))
            if use_bias: m.bias.copy_(torch.from_numpy(b))
        return m.eval()
    pad = int(rng.choice([0, 0, 4, 8]))
    xd = torch.zeros(B, T, K + pad, device=dev)[:, :, :K]
    xd.copy_(x.to(dev))
    ad = torch.from_numpy(adj).to(dev)
    kw = dict(store_gate=g2 if use_sg else None, pool_gate_a=g1, pool_gate_b=g2, want_out=use_out, want_pool_a=True, want_pool_b=True)
    try:
        with torch.no_grad():
            mf, mu = layer(True), layer(False)
            csr = pkg.BatchedCSR.from_dense(ad)
            took = mf.takes_fused_path(xd, csr)
            o1, a1, b1 = mf.forward_gated(xd, csr, **kw)
            o2, a2, b2 = mu.forward_gated(xd, csr, **kw)
        ref = ref_dense.graph_convolution(x, torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b) if use_bias else None)
        scale = max(1.0, float(ref.abs().max()))
        tol = 1e-4 * scale
        errs = []
        if use_out:
            gate = g2.cpu()[:, None, :] if use_sg else 1.0
            errs.append(float((o1.cpu() - ref * gate).abs().max()))
            errs.append(float((o1 - o2).abs().max()))
        errs.append(float((a1.cpu() - (ref * g1.cpu()[:, None, :]).max(dim=1)[0]).abs().max()))
        errs.append(float((b1.cpu() - (ref * g2.cpu()[:, None, :]).max(dim=1)[0]).abs().max()))
        ok = took and all(np.isfinite(e) and e <= tol for e in errs)
    except Exception as e:   # noqa: BLE001
        ok, errs = False, [repr(e)[:200]]
    if not ok:
        bad += 1
        print("FAIL", dict(B=B, T=T, K=K, F=F, deg=deg, prec=prec, bias=use_bias, sg=use_sg, out=use_out, pad=pad), errs)
print("cases %d, failures %d" % (cases, bad))
