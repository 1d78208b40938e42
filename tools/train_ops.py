import os, sys, torch
sys.path.insert(0, "/root/repo")
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
B, T, H = 4096, 32, 768
adj = synth.dependency_batch(B, T, 4.0); rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
xx = torch.randn(B, T, H, device=dev, requires_grad=True)
g1 = torch.rand(B, H, device=dev, requires_grad=True); g2 = torch.rand(B, H, device=dev, requires_grad=True)
ls = []
for s in (1, 2):
    ww, bb = synth.layer_params(H, H, seed=s)
    m = pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad(): m.weight.copy_(torch.from_numpy(ww)); m.bias.copy_(torch.from_numpy(bb))
    ls.append(m.train())
def train_step():
    for m in ls:
        m.weight.grad = None; m.bias.grad = None
    xx.grad = None; g1.grad = None; g2.grad = None
    r = pkg.gated_gcn_block(xx, csr, g1, g2, *ls); (r["out"].sum() + 0.01 * r["xy"]).backward()
for _ in range(5): train_step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): train_step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=50, max_shapes_column_width=60))
