#!/usr/bin/env python3
"""Gate / pool backward + transposed aggregation at config 2's shape: the scalar one-launch form (ggcn_gate_pool_backward_agg)
against the matrix-core form (ggcn_gate_pool_backward_mma), same process.  Development tool."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
lib = pkg.load_library(); dev = torch.device("cuda:0")
B, T, F = int(os.environ.get("LAB_GRAPHS", 4096)), int(os.environ.get("LAB_T", 32)), int(os.environ.get("LAB_F", 768))
adj = synth.dependency_batch(B, T, 4.0)
csr = pkg.BatchedCSR.from_dense(torch.from_numpy(adj.astype(np.float32)).to(dev))
out = torch.randn(B * T, F, device=dev); d_out = torch.randn(B * T, F, device=dev)
sg = torch.rand(B, F, device=dev) + 0.1; ga = torch.rand(B, F, device=dev); gb = torch.rand(B, F, device=dev)
d_pa = torch.randn(B, F, device=dev); d_pb = torch.randn(B, F, device=dev)
p, st = _capi.ptr, _capi.stream_of(dev)
dh = {k: torch.empty(B * T, F, device=dev) for k in ("agg", "mma")}
o = [torch.empty(B, F, device=dev) for _ in range(4)]
amax = torch.zeros(1, device=dev)
ops, ops_t = csr.graph_ops, csr.graph_ops_t
def run(k):
    amax.zero_()
    if k == "agg":
        _capi.check(lib.ggcn_gate_pool_backward_agg(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), p(csr.rowmask), B, T, F, p(dh[k]), F,
                                                    p(o[0]), p(o[1]), p(o[2]), p(o[3]), 0.0, 0, 0, 0, 0, p(amax) if F % 256 == 0 else None, st), k)
    else:
        _capi.check(lib.ggcn_gate_pool_backward_mma(p(out), F, p(sg), p(ga), p(gb), p(d_out), F, p(d_pa), p(d_pb), p(ops), p(ops_t), B, T, F, p(dh[k]), F,
                                                    p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(amax), st), k)
for k in dh:
    run(k); torch.cuda.synchronize()
    print(k, "amax", float(amax), "max|dh|", float(dh[k].abs().max()))
d = (dh["agg"] - dh["mma"]).abs()
print("max diff", float(d.max()), "scale", float(dh["agg"].abs().max()))
cm = dh["mma"].abs().max(0)[0]; print("column of max:", int(cm.argmax()), "row of max:", int(dh["mma"].abs().max(1)[0].argmax()) % T)
times = {k: [] for k in dh}
for rnd in range(8):
    for k in dh:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(k)
        e.record(); torch.cuda.synchronize()
        if rnd >= 2: times[k].append(a.elapsed_time(e) / 5 * 1e3)
for k, v in times.items(): print("%-4s median %.1f us (incl. a 4-byte memset)" % (k, statistics.median(v)))
