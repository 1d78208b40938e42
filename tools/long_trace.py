#!/usr/bin/env python3
"""Phase timeline of layer_fused_long_kernel (lab build: tools/lab.py build ltrace:-DGGCN_LAB_TRACE_LONG) at config 4."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_lab", "libggcn_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "ltrace")))
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # extra halfs per row of X (leading dimension H + PAD)
for fn, (res, args) in _capi.PROTOTYPES.items():
    getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
dev = torch.device("cuda:0")
B, T, H = 256, 512, 1024
adj = synth.dependency_batch(B, T, 6.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
xfull = torch.randn(B * T, H + PAD, device=dev).half()
x = xfull
w, b = synth.layer_params(H, H, seed=1)
w, b = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
g1, g2 = torch.rand(B, H, device=dev), torch.rand(B, H, device=dev)
out = torch.empty(B * T, H, device=dev, dtype=torch.float16)
pa, pb = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
p = _capi.ptr
pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, 3), dtype=torch.uint8, device=dev)
assert lib.ggcn_weight_pack(p(w), H, H, H, 3, 0, p(pack), None) == 0
def run():
    assert lib.ggcn_layer_fused_h(p(x), H + PAD, p(pack), p(csr.rowptr), p(csr.colidx), None, p(b), B, T, H, H, p(g2), p(g1), p(g2),
                                  p(out), H, p(pa), p(pb), None) == 0
for _ in range(300): run()
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.ggcn_lab_trace_read_long.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.ggcn_lab_trace_read_long(buf.ctypes.data, buf.nbytes) == 0
full = buf.reshape(4096, 8)[:2048].astype(np.float64) * 0.01
t = full[:, :5]
t0 = t[:, 0].min()
names = ["main loop (+ CSR staging, prologue)", "tile write + barrier", "neighbour sums + stores", "pool reduction"]
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-40s median %6.2f us  p10 %6.2f  p90 %6.2f" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
d = t[:, 4] - t[:, 0]
print("%-40s median %6.2f us  p10 %6.2f  p90 %6.2f" % ("workgroup", np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
for nm, a_, b_ in (("wave 7 leaves the row loop after wave 0 by", 3, 7), ("shuffles + LDS partials", 3, 5), ("barrier wait", 5, 6), ("final max + pool stores", 6, 4)):
    d = full[:, b_] - full[:, a_]
    print("%-40s median %6.2f us  p10 %6.2f  p90 %6.2f" % (nm, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
print("kernel span %.1f us; workgroup starts: first wave until %.1f us" % (t[:, 4].max() - t0, np.sort(t[:, 0] - t0)[255]))
