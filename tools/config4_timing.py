#!/usr/bin/env python3
"""BASELINE configs[3] (256 graphs x 512 tokens, degree 6, hidden 1024, fp16 features) and the
training step of config 2: timings of the non-headline paths.  Development tool."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth

dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
    return statistics.median(ts)

def layer(H):
    w, b = synth.layer_params(H, H, seed=1)
    m = pkg.GraphConvolution(H, H, None).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    return m

# ---- config 4
B, T, H = 256, 512, 1024
adj = synth.dependency_batch(B, T, 6.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
nnz = int(rp[-1])
m = layer(H)
for dt in (torch.float16, torch.float32):
    x = torch.randn(B, T, H, device=dev).to(dt)
    s = 2 if dt == torch.float16 else 4
    with torch.no_grad():
        t_layer = timeit(lambda: m(x, csr))
        t_lin = timeit(lambda: m.linear(x.view(B * T, H)))
    bytes_layer = 2 * s * B * T * H + 4 * (B * T + 1) + 4 * nnz + 4 * H * H + 4 * H
    print("config 4 %s: layer %.1f us (linear %.1f, aggregate ~%.1f) | nnz %d | %.0f M edges/s | algorithmic %.2f TB/s"
          % (str(dt).split(".")[-1], t_layer, t_lin, t_layer - t_lin, nnz, nnz / t_layer, bytes_layer / t_layer / 1e6))

# ---- config 2 training step (forward + backward of the gated block)
B, T, H = 4096, 32, 768
adj = synth.dependency_batch(B, T, 4.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
x = torch.randn(B, T, H, device=dev, requires_grad=True)
g1 = torch.sigmoid(torch.randn(B, H, device=dev)).requires_grad_(); g2 = torch.sigmoid(torch.randn(B, H, device=dev)).requires_grad_()
gc1, gc2 = layer(H).train(), layer(H).train()
def train_step():
    r = pkg.gated_gcn_block(x, csr, g1, g2, gc1, gc2)
    (r["out"].sum() + 0.01 * r["xy"]).backward()
print("config 2 forward+backward of the gated block: %.1f us" % timeit(train_step, 10))
with torch.no_grad():
    print("config 2 forward only (inference path):       %.1f us" % timeit(lambda: pkg.gated_gcn_block(x.detach(), csr, g1.detach(), g2.detach(), gc1, gc2)))
