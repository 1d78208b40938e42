#!/usr/bin/env python3
"""ggcn_layer_fused_h at config 4's shape (256 x 512 tokens, degree 6, hidden 1024, fp16), lab variants interleaved in one
process, next to ggcn_linear_h + ggcn_aggregate_h of the main build.  usage: long_timing.py [lab names...]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
dev = torch.device("cuda:0")
B, T, H = int(os.environ.get("LAB_GRAPHS", "256")), 512, 1024
adj = synth.dependency_batch(B, T, 6.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
PAD = int(os.environ.get("LAB_LDX_PAD", "0"))
x = torch.randn(B * T, H + PAD, device=dev).half()[:, :H]
LDX = H + PAD
w, b = synth.layer_params(H, H, seed=1)
w, b = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
out = torch.empty(B * T, H, device=dev, dtype=torch.float16)
hid = torch.empty(B * T, H, device=dev, dtype=torch.float16)
pa, pb = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
names = sys.argv[1:] or ["main"]
libs = {}
p = _capi.ptr
for n in names + ["two"]:
    path = pkg.lib_path() if n in ("main", "two") else os.path.join(os.path.dirname(__file__), "_lab", "libggcn_%s.so" % n)
    lib = ctypes.CDLL(path)
    for fn, (res, args) in _capi.PROTOTYPES.items():
        if hasattr(lib, fn):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, 3), dtype=torch.uint8, device=dev)
    assert lib.ggcn_weight_pack(p(w), H, H, H, 3, 0, p(pack), None) == 0
    libs[n] = (lib, pack)
def run(n):
    lib, pack = libs[n]
    if n == "two":
        rc = lib.ggcn_linear_h(p(x), LDX, p(pack), p(hid), H, B * T, H, H, 3, None)
        rc = rc or lib.ggcn_aggregate_h(p(hid), H, p(csr.rowptr), p(csr.colidx), None, p(b), B, T, H, p(g2), p(g1), p(g2), p(out), H, p(pa), p(pb), None)
    else:
        rc = lib.ggcn_layer_fused_h(p(x), LDX, p(pack), p(csr.rowptr), p(csr.colidx), None, p(b), B, T, H, H, p(g2), p(g1), p(g2), p(out), H, p(pa), p(pb), None)
    assert rc == 0, lib.ggcn_last_error()
ref = None
for n in ["two"] + names:
    run(n); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print("%-10s max|out - two| %.3g" % (n, float((out.float() - ref.float()).abs().max())))
for _ in range(100): run(names[0])
alln = names + ["two"]
times = {n: [] for n in alln}
for r in range(10):
    for n in alln:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(n)
        e.record(); torch.cuda.synchronize()
        if r >= 2: times[n].append(a.elapsed_time(e) / 5 * 1e3)
for n in alln: print("%-12s median %.1f us  min %.1f" % (n, statistics.median(times[n]), min(times[n])))
