#!/usr/bin/env python3
"""dW = X^T . dH at config 2's shape (N = 131 072, K = F = 768): the native TN kernel (dweight_tn.hip) against the
transpose + pack form (GGCN_DWEIGHT_TRANSPOSE=1), same process, and both against float64 on a slice.  Development tool."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi
lib = pkg.load_library()
dev = torch.device("cuda:0")
N, K, F = int(os.environ.get("LAB_N", 131072)), 768, 768
x = torch.randn(N, K, device=dev)
g = torch.randn(N, F, device=dev) * 1e-3
ws = torch.empty(4 * lib.ggcn_dweight_workspace_bytes(N, K, F, 0), dtype=torch.uint8, device=dev)   # (room for more chunks: LAB_SPLITS)
dw = {}
def run(name):
    out = dw.setdefault(name, torch.empty(K, F, device=dev))
    _capi.check(lib.ggcn_dweight(_capi.ptr(x), K, _capi.ptr(g), F, N, K, F, _capi.ptr(out), F, 0, _capi.ptr(ws), None), "ggcn_dweight")
def mode(name):
    if name == "transpose": os.environ["GGCN_DWEIGHT_TRANSPOSE"] = "1"
    else: os.environ.pop("GGCN_DWEIGHT_TRANSPOSE", None)
    if "@" in name: os.environ["GGCN_LAB_DW_SPLITS"] = name.split("@")[1]
    else: os.environ.pop("GGCN_LAB_DW_SPLITS", None)
for name in ("tn", "transpose"):
    mode(name); run(name)
torch.cuda.synchronize()
ref = (x[:, :64].double().t() @ g.double()).float()
for name in ("tn", "transpose"):
    print("%-10s max|err| vs float64 (64 rows of dW): %.3g  (|dW| ~ %.3g)" % (name, float((dw[name][:64] - ref).abs().max()), float(ref.abs().max())))
print("tn vs transpose max|diff| %.3g" % float((dw["tn"] - dw["transpose"]).abs().max()))
times = {"tn": [], "transpose": []}
for sp in os.environ.get("LAB_SPLITS", "").split(","):
    if sp: times["tn@" + sp] = []
for rnd in range(8):
    for name in times:
        mode(name)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(name)
        e.record(); torch.cuda.synchronize()
        if rnd >= 2: times[name].append(a.elapsed_time(e) / 5 * 1e3)
for name, v in times.items():
    print("%-10s median %8.1f us" % (name, statistics.median(v)))
