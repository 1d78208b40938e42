#!/usr/bin/env python3
"""Config 4 through ggcn_layer_fused_h, variants interleaved in ONE process on one box (boxes differ by several per cent):
   python tools/long_time.py [lab names ...]   -- always times the product library with MFMA neighbour sums (default) and
   with lane sums (GGCN_LONG_LANE_SUMS=1, read per call); lab names add tools/_lab/libggcn_<name>.so (e.g. `old`)."""
import ctypes, os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth

def load(path):
    lib = ctypes.CDLL(path)
    for fn, (res, args) in _capi.PROTOTYPES.items():
        if hasattr(lib, fn) and fn in ("ggcn_layer_fused_h", "ggcn_weight_pack_bytes", "ggcn_weight_pack"):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
    return lib

dev = torch.device("cuda:0")
B, T, H = 256, 512, 1024
adj = synth.dependency_batch(B, T, 6.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
x = torch.randn(B * T, H, device=dev).half()
w, b = synth.layer_params(H, H, seed=1)
w, b = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
out = torch.empty(B * T, H, device=dev, dtype=torch.float16)
pa, pb = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
p = _capi.ptr
variants = {}
main = load(pkg.lib_path())
for name in sys.argv[1:]:
    variants[name] = (load(os.path.join(ROOT, "tools", "_lab", "libggcn_%s.so" % name)), None)
variants["main, MFMA sums"] = (main, None)
variants["main, lane sums"] = (main, "1")
packs = {}
for name, (lib, _) in variants.items():
    pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, 3), dtype=torch.uint8, device=dev)
    assert lib.ggcn_weight_pack(p(w), H, H, H, 3, 0, p(pack), None) == 0
    packs[name] = pack

# LONG_LDX="1032,1088": the product library again with X rows that far apart (elements) -- does the 2 KiB row pitch cost L2 channels?
xpad = {}
for ld in [int(v) for v in os.environ.get("LONG_LDX", "").split(",") if v]:
    buf = torch.zeros(B * T, ld, device=dev, dtype=torch.float16)
    buf[:, :H] = x
    n = "main, ldx %d" % ld
    variants[n] = (main, None); packs[n] = packs["main, MFMA sums"]; xpad[n] = (buf, ld)

def run(name):
    lib, env = variants[name]
    xx, ldx = xpad.get(name, (x, H))
    if env: os.environ["GGCN_LONG_LANE_SUMS"] = env
    else: os.environ.pop("GGCN_LONG_LANE_SUMS", None)
    assert lib.ggcn_layer_fused_h(p(xx), ldx, p(packs[name]), p(csr.rowptr), p(csr.colidx), None, p(b), B, T, H, H, p(g2), p(g1), p(g2),
                                  p(out), H, p(pa), p(pb), None) == 0

for _ in range(100): run("main, MFMA sums")
torch.cuda.synchronize()
if os.environ.get("LAB_ENERGY"):   # each variant alone for ~1.5 s with the board power from sysfs beside it -> energy per launch
    import bench, time
    for n in variants:
        for _ in range(300): run(n)
        torch.cuda.synchronize()
        with bench.PowerSampler() as ps:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_end, cnt = time.perf_counter() + 1.5, 0
            a.record()
            while time.perf_counter() < t_end:
                for _ in range(100): run(n)
                cnt += 100
                torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
        us, pw = a.elapsed_time(e) / cnt * 1e3, ps.summary()
        print("%-24s %8.1f us per launch   %7.1f W (max %7.1f, sclk %s MHz)   %.4f J per launch" %
              (n, us, pw.get("power_w") or float("nan"), pw.get("power_w_max") or float("nan"), pw.get("sclk_dpm_mhz"),
               us * 1e-6 * (pw.get("power_w") or float("nan"))), flush=True)
    sys.exit(0)
times = {n: [] for n in variants}
for rnd in range(12):
    for n in variants:
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run(n)
        a.record()
        for _ in range(10): run(n)
        e.record(); torch.cuda.synchronize()
        times[n].append(a.elapsed_time(e) * 100.0)
for n, t in times.items():
    print("%-24s median %7.1f us   min %7.1f us" % (n, statistics.median(t), min(t)))
