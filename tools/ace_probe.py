import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
dev = torch.device("cuda:0")
print(json.dumps(bench.ace_block(pkg, synth, torch, dev, "f16mx8"))[:400])
print(json.dumps(bench.ace_block(pkg, synth, torch, dev, "f16mx8"))[:400])
