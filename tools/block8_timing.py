#!/usr/bin/env python3
"""VERDICT r4 item 2 (i), measured: the two-layer block at config 2 through ggcn_block_fused (two four-wavefront workgroups per CU,
W1 / W12 tiles on separate XCD groups) and through ggcn_lab_block_fused8 (one eight-wavefront workgroup per CU that shares a row
block's X planes between its W1 and W12 tiles): bitwise comparison, then interleaved timing in one process.  Development tool."""
import os, statistics, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth
from ed_gated_gcn_amd.gated_block import _block_operands
lib = pkg.load_library(); dev = torch.device("cuda:0")
B, T, H = int(os.environ.get("LAB_GRAPHS", 4096)), int(os.environ.get("LAB_T", 32)), 768
adj = synth.dependency_batch(B, T, min(4.0, T), lengths=None if T == 32 else np.random.default_rng(1).integers(1, T + 1, size=B))
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
gen = torch.Generator().manual_seed(1)
x = torch.randn(B * T, H, generator=gen).to(dev)
g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev); g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
ls = []
for s in (1, 2):
    w, b = synth.layer_params(H, H, seed=s)
    m = pkg.GraphConvolution(H, H, None).to(dev).eval(); m.precision = "f16mx8"
    with torch.no_grad(): m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    ls.append(m)
st, P = _capi.stream_of(dev), _capi.ptr
pack1, pack12, mid = _block_operands(ls[0], ls[1], lib, st, precision="f16mx8")
b1, b2 = ls[0].bias.detach(), ls[1].bias.detach()
ops, ops2 = csr.graph_ops, csr.graph_ops2(1)
STAMPS = [None]
def bufs():
    e = lambda *s: torch.full(s, float("nan"), device=dev)
    return dict(xo=e(B * T, H), x1=e(B, H), y1=e(B, H), out=e(B, H), part=e(B, 12))
r4, r8 = bufs(), bufs()
def run4(r=r4):
    os.environ["GGCN_BLOCK_FORM"] = "4"   # (since late round 5 ggcn_block_fused itself takes the eight-wavefront kernel from 2048 graphs up)
    try:
        return _run4(r)
    finally:
        os.environ.pop("GGCN_BLOCK_FORM", None)
def _run4(r):
    return lib.ggcn_block_fused(P(x), H, P(pack1), P(pack12), P(ops), P(ops2), P(b1), P(mid), P(b2), B, T, H, H, P(g1), P(g2), None, H,
                                P(r["xo"]), H, P(r["x1"]), P(r["y1"]), P(r["out"]), P(r["part"]), _capi.PREC["f16mx8"], st)
def run8(r=r8):
    return lib.ggcn_lab_block_fused8(P(x), H, P(pack1), P(pack12), P(ops), P(ops2), P(b1), P(mid), P(b2), B, T, H, H, P(g1), P(g2),
                                     P(r["xo"]), H, P(r["x1"]), P(r["y1"]), P(r["out"]), P(r["part"]), P(STAMPS[0]), st)
_capi.check(run4(), "block_fused"); _capi.check(run8(), "lab_block_fused8"); torch.cuda.synchronize()
for k in r4:
    rows = slice(None)
    a, b = r4[k], r8[k]
    if k == "xo" and T < 32: pass
    same = torch.equal(a, b) or bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
    print("%-4s bitwise equal: %s" % (k, same))
if os.environ.get("LAB_ONLY") == "8":
    for _ in range(int(os.environ.get("LAB_REPS", 60))): run8()
    torch.cuda.synchronize(); sys.exit(0)
if os.environ.get("LAB_ONLY") == "4":
    for _ in range(int(os.environ.get("LAB_REPS", 60))): run4()
    torch.cuda.synchronize(); sys.exit(0)
if os.environ.get("LAB_ENERGY"):   # each form ALONE for 1.5 s with the board power beside it: the kernels sit at the power cap, and
    import time                    # five-launch interleaved timings (below) mix the two forms' power states
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for rep in range(2):
        for name, f in (("block_fused (4 wavefronts, 2 per CU)", run4), ("lab_block_fused8 (8 wavefronts, shared X)", run8)):
            for _ in range(300): f()
            torch.cuda.synchronize()
            with bench.PowerSampler() as ps:
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_end, cnt = time.perf_counter() + 1.5, 0
                a.record()
                while time.perf_counter() < t_end:
                    for _ in range(100): f()
                    cnt += 100; torch.cuda.synchronize()
                e.record(); torch.cuda.synchronize()
            us, pw = a.elapsed_time(e) / cnt * 1e3, ps.summary()
            print("%-44s %.1f us  %.0f W (sclk %s MHz)  %.4f J" % (name, us, pw.get("power_w") or float("nan"), pw.get("sclk_dpm_mhz"),
                                                                 us * 1e-6 * (pw.get("power_w") or float("nan"))), flush=True)
    sys.exit(0)
for _ in range(150): run4()
times = {"block_fused (4 wavefronts, 2 per CU)": [], "lab_block_fused8 (8 wavefronts, shared X)": []}
fns = [run4, run8]
for rnd in range(12):
    for name, f in zip(times, fns):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        if rnd >= 2: times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
for name, v in times.items(): print("%-44s median %.1f us  min %.1f" % (name, statistics.median(v), min(v)))
# phase stamps of the eight-wavefront form (10 ns ticks): per workgroup and group: start, main loop start, main loop end, end
grid = 8 * ((B // 4 * 3 + 7) // 8) if not os.environ.get("GGCN_LAB_BLOCK8_ROWMAJOR") else 4096
st_buf = torch.zeros(grid * 2 * 4, dtype=torch.int64, device=dev)
STAMPS[0] = st_buf
for _ in range(20): run8()
torch.cuda.synchronize()
s = st_buf.view(-1, 2, 4).cpu().double()
s = s[s[:, 1, 3] > 0]
for gname, gi in (("W1 group ", 0), ("W12 group", 1)):
    pro = (s[:, gi, 1] - s[:, gi, 0]) * 0.01; loop = (s[:, gi, 2] - s[:, gi, 1]) * 0.01; epi = (s[:, gi, 3] - s[:, gi, 2]) * 0.01
    print("%s per workgroup (us, median): prologue %.2f  main loop %.2f  epilogue %.2f  total %.2f   (n = %d)" %
          (gname, float(pro.median()), float(loop.median()), float(epi.median()), float(((s[:, gi, 3] - s[:, gi, 0]) * 0.01).median()), s.shape[0]))
span = (s[:, :, 3].max() - s[:, :, 0].min()) * 0.01
print("kernel span by stamps: %.1f us" % float(span))
