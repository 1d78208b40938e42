#!/usr/bin/env python3
"""fp16 linear (precision f16) at config 4's shape, variants interleaved in one process.  usage: f16_timing.py [lab names...]"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi
dev = torch.device("cuda:0")
M, K, F = 131072, 1024, 1024
x = torch.randn(M, K, device=dev).half(); w = (torch.randn(K, F, device=dev) * 0.03); y = torch.empty(M, F, device=dev, dtype=torch.float16)
names = sys.argv[1:] or ["main"]
libs = {}
for n in names:
    path = pkg.lib_path() if n == "main" else os.path.join(os.path.dirname(__file__), "_lab", "libggcn_%s.so" % n)
    lib = ctypes.CDLL(path)
    for fn, (res, args) in _capi.PROTOTYPES.items():
        if hasattr(lib, fn):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    pack = torch.empty(lib.ggcn_weight_pack_bytes(K, F, 3), dtype=torch.uint8, device=dev)
    assert lib.ggcn_weight_pack(_capi.ptr(w), F, K, F, 3, 0, _capi.ptr(pack), None) == 0
    libs[n] = (lib, pack)
def run(n):
    lib, pack = libs[n]
    assert lib.ggcn_linear_h(_capi.ptr(x), K, _capi.ptr(pack), _capi.ptr(y), F, M, K, F, 3, None) == 0
ref = None
for n in names:
    run(n); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    print(n, "max|diff|", float((y.float() - ref.float()).abs().max()))
for _ in range(100): run(names[0])
times = {n: [] for n in names}
for r in range(10):
    for n in names:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run(n)
        e.record(); torch.cuda.synchronize()
        if r >= 2: times[n].append(a.elapsed_time(e) / 5 * 1e3)
for n in names: print("%-12s median %.1f us  min %.1f" % (n, statistics.median(times[n]), min(times[n])))
