import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
    return statistics.median(ts)
B, T, H = 4096, 32, 768
N = B * T
x = torch.randn(N, H, device=dev); dh = torch.randn(N, H, device=dev); w = torch.randn(H, H, device=dev)
print("dW = x.t() @ dh (torch fp32):      %.1f us" % timeit(lambda: x.t().matmul(dh)))
print("dX = dh @ w.t() (torch fp32):      %.1f us" % timeit(lambda: dh.matmul(w.t())))
y = torch.randn(B, T, H, device=dev, requires_grad=True); g = torch.rand(B, H, device=dev, requires_grad=True)
def gate_pool():
    o = (y * g[:, None, :]).max(dim=1)[0].sum(); o.backward()
print("gate*y -> max -> backward (torch):  %.1f us" % timeit(gate_pool))
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
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
    for _ in range(3):
        r = pkg.gated_gcn_block(xx, csr, g1, g2, *ls); (r["out"].sum() + 0.01 * r["xy"]).backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
# steady state: the step alone (setup, packs and operand blocks exist), HIP events around forward + backward, median of 10 behind 3 untimed
def train_step():
    for m in ls:
        m.weight.grad = None; m.bias.grad = None
    xx.grad = None; g1.grad = None; g2.grad = None
    r = pkg.gated_gcn_block(xx, csr, g1, g2, *ls); (r["out"].sum() + 0.01 * r["xy"]).backward()
print("steady-state training step (forward + backward of the block, loss = out.sum() + 0.01 xy): %.1f us" % timeit(train_step, n=10))
def fwd_only():
    with torch.no_grad():
        pkg.gated_gcn_block(xx, csr, g1, g2, *ls, one_launch=False)
print("the two one-launch layers alone (what the training forward runs): %.1f us" % timeit(fwd_only, n=10))
