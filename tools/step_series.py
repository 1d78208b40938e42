#!/usr/bin/env python3
"""Per-launch time series of ggcn_block_fused at config 2 (f16mx8): where do the slow launches sit?  Development tool."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
dev = torch.device("cuda:0")
B, T, H = 4096, 32, 768
adj = synth.dependency_batch(B, T, 4.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
x = torch.randn(B * T, H, device=dev)
w, b = synth.layer_params(H, H, seed=1)
w, b = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
out = torch.empty(B * T, H, device=dev)
pa, pb, pc = (torch.empty(B, H, device=dev) for _ in range(3))
part = torch.empty(B, 12, device=dev)
lib = pkg.load_library()
p = _capi.ptr
pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, 2), dtype=torch.uint8, device=dev)
assert lib.ggcn_weight_pack(p(w), H, H, H, 2, 0, p(pack), None) == 0
st = _capi.stream_of(dev)
def run():
    assert lib.ggcn_block_fused(p(x), H, p(pack), p(pack), p(csr.graph_ops), p(b), p(b), p(b), B, T, H, H, p(g1), p(g2), None, H, p(out), H,
                                p(pa), p(pb), p(pc), p(part), 2, st) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for _ in range(300): run()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
for e in ev: e.record()
torch.cuda.synchronize()
ev[0].record()
for i in range(N):
    run(); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
med = statistics.median(t)
slow = [i for i, v in enumerate(t) if v > 1.05 * med]
print("median %.1f us  mean %.1f  p10 %.1f  p90 %.1f  max %.1f" % (med, statistics.mean(t), sorted(t)[N // 10], sorted(t)[9 * N // 10], max(t)))
print("launches > 1.05 x median: %d of %d; indices: %s" % (len(slow), N, slow[:80]))
print("series (us, every launch of the first 120):", " ".join("%.0f" % v for v in t[:120]))
