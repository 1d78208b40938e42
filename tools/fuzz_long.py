#!/usr/bin/env python3
"""Random shapes through the one-launch long-graph layer (fp16 features, 129..512 nodes): MFMA neighbour sums (default for
unweighted graphs) against the oracle on the fp16-rounded inputs and against linear + aggregate, lane sums
(GGCN_LONG_LANE_SUMS=1) bit for bit against linear + aggregate; degrees from 0 to hubs, ragged lengths, missing bias /
gates / outputs, F with dead columns, padded leading dimensions.  Development tool.  usage: fuzz_long.py [cases] [seed]"""
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
    T = int(rng.choice([129, 130, 159, 160, 161, 200, 255, 256, 257, 300, 333, 400, 480, 481, 511, 512]))
    B = int(rng.integers(1, 11))
    K = 64 * int(rng.integers(1, 7))
    F = 8 * int(rng.integers(1, 49))
    deg = float(rng.choice([0.0, 1.0, 3.0, 6.0, 7.9, 12.0, 30.0]))
    weighted = bool(rng.integers(0, 4) == 0)
    lens = np.array([T] + [int(v) for v in rng.integers(max(1, T // 4), T + 1, size=B - 1)])
    if deg == 0.0:
        adj = np.zeros((B, T, T), dtype=np.float32)
        adj[:, rng.integers(0, T), :] = (rng.random((B, T)) < 0.5)      # one hub row per graph, nothing else
    else:
        adj = synth.dependency_batch(B, T, min(deg, T), seed=int(rng.integers(1 << 30)), lengths=lens).astype(np.float32)
    if weighted:
        adj = adj * rng.uniform(0.25, 2.0, size=adj.shape).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((B, T, K)).astype(np.float32)).half()
    w, b = synth.layer_params(K, F, seed=int(rng.integers(1 << 30)))
    use_bias, use_sg, use_out = bool(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 5))
    g1 = torch.from_numpy(rng.uniform(-1, 1, (B, F)).astype(np.float32)).to(dev)
    g2 = torch.sigmoid(torch.from_numpy(rng.standard_normal((B, F)).astype(np.float32))).to(dev)
    def layer(fused):
        m = pkg.GraphConvolution(K, F, None, bias=use_bias).to(dev)
        m.precision, m.fused = "f16", fused
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w))
            if use_bias: m.bias.copy_(torch.from_numpy(b))
        return m.eval()
    pad = int(rng.choice([0, 0, 8, 64]))
    xd = torch.zeros(B, T, K + pad, device=dev, dtype=torch.float16)[:, :, :K]
    xd.copy_(x.to(dev))
    kw = dict(store_gate=g2 if use_sg else None, pool_gate_a=g1, pool_gate_b=g2, want_out=use_out, want_pool_a=True, want_pool_b=True)
    try:
        with torch.no_grad():
            one, two = layer(True), layer(False)
            csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj).to(dev))
            took = one.takes_long_path(xd, csr)
            o1, a1, b1 = one.forward_gated(xd, csr, **kw)
            o2, a2, b2 = two.forward_gated(xd, csr, **kw)
            os.environ["GGCN_LONG_LANE_SUMS"] = "1"
            o3, a3, b3 = one.forward_gated(xd, csr, **kw)
            os.environ.pop("GGCN_LONG_LANE_SUMS")
        ref = ref_dense.graph_convolution(x.float(), torch.from_numpy(adj), torch.from_numpy(w), torch.from_numpy(b) if use_bias else None)
        scale = max(1.0, float(ref.abs().max()))
        errs = []
        if use_out:
            gate = g2.cpu()[:, None, :] if use_sg else 1.0
            errs.append(float((o1.float().cpu() - ref * gate).abs().max()) / (2e-3 * scale + scale * 2.0 ** -11))
            errs.append(float(((o1.float() - o2.float()).abs() / (2.0 ** -10 * o2.float().abs() + 2.0 ** -24)).max()))
            errs.append(0.0 if torch.equal(o3, o2) else 9.0)
        errs.append(float((a1.cpu() - (ref * g1.cpu()[:, None, :]).max(dim=1)[0]).abs().max()) / (2e-3 * scale))
        errs.append(float((b1.cpu() - (ref * g2.cpu()[:, None, :]).max(dim=1)[0]).abs().max()) / (2e-3 * scale))
        errs.append(float((a1 - a2).abs().max()) / (2e-6 * scale))
        errs.append(float((b1 - b2).abs().max()) / (2e-6 * scale))
        errs.append(0.0 if torch.equal(a3, a2) and torch.equal(b3, b2) else 9.0)
        ok = took and all(np.isfinite(e) and e <= 1.0 for e in errs)
    except Exception as e:   # noqa: BLE001
        ok, errs = False, [repr(e)[:200]]
    if not ok:
        bad += 1
        print("FAIL case %d: B=%d T=%d K=%d F=%d deg=%.1f weighted=%s bias=%s sg=%s out=%s pad=%d  ratios %s" % (
            it, B, T, K, F, deg, weighted, use_bias, use_sg, use_out, pad, errs), flush=True)
    elif it % 25 == 0:
        print("case %d ok (B=%d T=%d K=%d F=%d deg=%.1f weighted=%s) worst ratio %.3f" % (it, B, T, K, F, deg, weighted, max(errs)), flush=True)
print("fuzz_long: %d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
