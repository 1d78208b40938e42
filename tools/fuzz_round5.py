#!/usr/bin/env python3
"""Random cases through this round's new paths against the oracle (development tool; usage: fuzz_round5.py [cases] [seed]):
  * the eval form (want=("out",) / ("x","out")) for graphs of 1..256 nodes -- the W12 tiles alone (<= 32 nodes) or
    D.A.X + one layer launch through W12 (33..256) -- and the dense head riding with xy;
  * the whole block under autograd for graphs of 1..32 nodes, DIRECTED adjacencies included: the matrix-core gate / pool
    backward (A^T from the transposed row masks), the scaled f16mx8 dX, the dW plan -- gradients against torch autograd on
    the oracle's formula with the GPU forward's arg-max rows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
from oracle import ref_dense
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = torch.from_numpy
bad = 0

def layer(H, w, b, prec, train=False):
    m = pkg.GraphConvolution(H, H, None).to(dev)
    m.precision = prec; m.fused_max_t = 256
    with torch.no_grad():
        m.weight.copy_(t(w)); m.bias.copy_(t(b))
    return m.train() if train else m.eval()

for it in range(cases):
    kind = "eval" if it % 2 == 0 else "grad"
    T = int(rng.integers(1, 257)) if kind == "eval" else int(rng.integers(1, 33))
    B = int(rng.integers(1, 12 if T > 64 else 40))
    H = int(rng.choice([32, 64, 96, 128, 256])) if kind == "grad" else int(rng.choice([32, 64, 96, 100, 128, 256, 260]))
    prec = str(rng.choice(["f16mx8", "bf16x3"]))
    lens = np.array([T] + [int(v) for v in rng.integers(1, T + 1, size=B - 1)])
    adj = synth.dependency_batch(B, T, min(4.0, T), seed=int(rng.integers(1 << 30)), lengths=lens).astype(np.float32)
    directed = kind == "grad" and bool(rng.integers(0, 2))
    if directed:
        keep = rng.random(adj.shape) < 0.5
        adj = np.where(np.triu(np.ones((T, T), bool), 1)[None] & keep, 0.0, adj).astype(np.float32)
    x = t(rng.standard_normal((B, T, H)).astype(np.float32))
    w1, b1 = synth.layer_params(H, H, seed=int(rng.integers(1 << 30)))
    w2, b2 = synth.layer_params(H, H, seed=int(rng.integers(1 << 30)))
    g1 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    g2 = torch.sigmoid(t(rng.standard_normal((B, H)).astype(np.float32)))
    errs = {}
    try:
        if kind == "eval":
            l1, l2 = layer(H, w1, b1, prec), layer(H, w2, b2, prec)
            C = int(rng.integers(1, 65))
            wt = (torch.randn(H, C, generator=torch.Generator().manual_seed(it)) / H ** 0.5)
            with torch.no_grad():
                ev = pkg.gated_gcn_block(x.to(dev), t(adj).to(dev), g1.to(dev), g2.to(dev), l1, l2, want=("x", "out") if it % 4 == 0 else ("out",),
                                         dense_head=(wt.to(dev), None))
            ref = ref_dense.gated_block(x, t(adj), g1, g2, t(w1), t(b1), t(w2), t(b2))
            for k in ("out", "x"):
                if ev[k] is not None:
                    errs[k] = float((ev[k].cpu() - ref[k]).abs().max()) / max(1.0, float(ref[k].abs().max()))
            want = ref["out"].double() @ wt.double()
            errs["logits"] = float((ev["logits"].double().cpu() - want).abs().max()) / max(1.0, float(want.abs().max()))
            ok = all(np.isfinite(e) and e <= 2e-4 for e in errs.values())
        else:
            gc1, gc2 = layer(H, w1, b1, prec, True), layer(H, w2, b2, prec, True)
            R1 = t(rng.standard_normal((B, H)).astype(np.float32)); R2 = t(rng.standard_normal((B, T, H)).astype(np.float32))
            loss_of = lambda r, R1, R2: (r["out"] * R1).sum() + 0.1 * (r["x"] * R2).sum() + 0.01 * r["xy"]   # noqa: E731
            xg, g1g, g2g = (v.to(dev).requires_grad_() for v in (x, g1, g2))
            r = pkg.gated_gcn_block(xg, t(adj).to(dev), g1g, g2g, gc1, gc2)
            loss_of(r, R1.to(dev), R2.to(dev)).backward()
            with torch.no_grad():
                i_x1 = (r["gcn1"] * g1g[:, None, :]).argmax(dim=1).cpu(); i_y1 = (r["gcn1"] * g2g[:, None, :]).argmax(dim=1).cpu()
                i_out = r["x"].argmax(dim=1).cpu()
            leaves = [v.clone().requires_grad_() for v in (x, g1, g2, t(w1), t(b1), t(w2), t(b2))]
            lx, lg1, lg2, lw1, lb1, lw2, lb2 = leaves
            a32 = t(adj)
            gcn1 = ref_dense.graph_convolution(lx, a32, lw1, lb1)
            x2 = lg2[:, None, :] * ref_dense.graph_convolution(gcn1, a32, lw2, lb2)
            pick = lambda v, i: v.gather(1, i[:, None, :]).squeeze(1)   # noqa: E731
            rr = {"x": x2, "out": pick(x2, i_out), "xy": (pick(gcn1 * lg1[:, None, :], i_x1) * pick(gcn1 * lg2[:, None, :], i_y1)).sum(1).mean()}
            loss_of(rr, R1, R2).backward()
            got = [xg.grad, g1g.grad, g2g.grad, gc1.weight.grad, gc1.bias.grad, gc2.weight.grad, gc2.bias.grad]
            for name, gv, lv in zip(("x", "gate1", "gate2", "w1", "b1", "w2", "b2"), got, leaves):
                errs[name] = float((gv.cpu() - lv.grad).abs().max()) / max(1e-6, float(lv.grad.abs().max()))
            ok = all(np.isfinite(e) and e <= 1e-3 for e in errs.values())
    except Exception as e:   # noqa: BLE001
        ok, errs = False, {"exc": repr(e)[:300]}
    if not ok:
        bad += 1
        print("FAIL", kind, dict(B=B, T=T, H=H, prec=prec, directed=directed), errs, flush=True)
print("cases %d, failures %d" % (cases, bad))
