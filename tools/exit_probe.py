"""One f16mx8 forward outside the accuracy window, then exit without asking: range_guard reports on stderr (tests/test_gpu_parity.py)."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
m = pkg.GraphConvolution(64, 64, None).to(dev)
w, b = synth.layer_params(64, 64, seed=1)
with torch.no_grad():
    m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    adj = torch.from_numpy(synth.dependency_batch(4, 20, 3.0, seed=1)).to(dev).float()
    x = torch.randn(4, 20, 64, device=dev) * 1000.0
    y = m(x, adj)
print("forward done, exiting without asking", float(y.abs().max()) > 0)
