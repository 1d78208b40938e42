#!/usr/bin/env python3
"""ggcn_dweight (bf16x3, TN form): the 256 x 256 tile against the 128 x 256 one -- same bits; time, board power and energy per
launch of each alone (the kernel sits at the power cap: short interleaved timings hide the difference).  Needs a lab library
whose plan reads GGCN_DW_TILE:  tools/labbuild.sh dwlab "-DGGCN_LAB_DW" dweight_tn.hip dweight_bx3.hip"""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi
import bench
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_lab", "libggcn_dwlab.so"))
for fn in ("ggcn_dweight_workspace_bytes", "ggcn_dweight"):
    getattr(lib, fn).restype, getattr(lib, fn).argtypes = _capi.PROTOTYPES[fn]
p = _capi.ptr
dev = torch.device("cuda:0")
for N, K, F in ((131072, 768, 768), (65536, 768, 768), (20000, 768, 768), (131072, 1024, 1024), (5000, 256, 512)):
    x = torch.randn(N, K, device=dev); dh = torch.randn(N, F, device=dev) * 1e-3
    res = {}
    for tile in ("128", "256"):
        os.environ["GGCN_DW_TILE"] = tile
        ws = torch.empty(lib.ggcn_dweight_workspace_bytes(N, K, F, _capi.PREC["bf16x3"]), dtype=torch.uint8, device=dev)
        dw = torch.empty(K, F, device=dev)
        f = lambda: lib.ggcn_dweight(p(x), K, p(dh), F, N, K, F, p(dw), F, _capi.PREC["bf16x3"], p(ws), None)
        for _ in range(200): f()
        torch.cuda.synchronize()
        with bench.PowerSampler() as ps:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_end, cnt = time.perf_counter() + 1.2, 0
            a.record()
            while time.perf_counter() < t_end:
                for _ in range(100): f()
                cnt += 100; torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
        us, pw = a.elapsed_time(e) / cnt * 1e3, ps.summary()
        res[tile] = (dw.clone(), us, pw.get("power_w") or float("nan"))
    print("N=%d K=%d F=%d: 128 x 256: %.1f us %.0f W %.4f J   256 x 256: %.1f us %.0f W %.4f J   bitwise equal %s" %
          (N, K, F, res["128"][1], res["128"][2], res["128"][1] * 1e-6 * res["128"][2], res["256"][1], res["256"][2],
           res["256"][1] * 1e-6 * res["256"][2], torch.equal(res["128"][0], res["256"][0])), flush=True)
