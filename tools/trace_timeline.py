#!/usr/bin/env python3
"""Timeline of the fused layer kernel from a lab build with -DGGCN_LAB_TRACE (tools/lab.py build
trace:-DGGCN_LAB_TRACE): per workgroup start / main-loop begin / main-loop end / end in 10 ns ticks
plus the CU it ran on.  Prints phase durations, the turnaround between consecutive workgroups of a
CU and how the two co-resident workgroups of a CU overlap.   usage: trace_timeline.py [variant] [mx8|bf16x3]"""
import ctypes, os, sys, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import _capi, synth

variant = sys.argv[1] if len(sys.argv) > 1 else "trace"
NOOUT = "noout" in sys.argv
BLOCK = "block" in sys.argv     # the whole gated block as one launch (ggcn_block_fused): 6 column tiles per row block
prec = 2 if (len(sys.argv) > 2 and sys.argv[2] == "mx8") or len(sys.argv) <= 2 else 0
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_lab", "libggcn_%s.so" % variant))
for fn, (res, args) in _capi.PROTOTYPES.items():
    getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
dev = torch.device("cuda:0")
B, T, H = 4096, 32, 768
adj = synth.dependency_batch(B, T, 4.0)
rowptr, colidx, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rowptr, colidx, B, T, dev)
gen = torch.Generator().manual_seed(1)
x = torch.randn(B * T, H, generator=gen).to(dev)
w = (torch.randn(H, H, generator=gen) * 0.05).to(dev)
b = torch.randn(H, generator=gen).to(dev)
g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev); g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
out = torch.empty(B * T, H, device=dev); pa = torch.empty(B, H, device=dev); pb = torch.empty(B, H, device=dev); pc = torch.empty(B, H, device=dev)
p = _capi.ptr
pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, prec), dtype=torch.uint8, device=dev)
assert lib.ggcn_weight_pack(p(w), H, H, H, prec, 0, p(pack), None) == 0
pack12 = torch.empty_like(pack)
assert lib.ggcn_weight_pack(p(w.t().contiguous()), H, H, H, prec, 0, p(pack12), None) == 0   # timing only: any weights do
part = torch.empty(B, 12, device=dev)
def run():
    if BLOCK:
        rc = lib.ggcn_block_fused(p(x), H, p(pack), p(pack12), p(csr.graph_ops), p(csr.graph_ops2(0 if prec == 0 else 1)), p(b), p(b), p(b), B, T, H, H,
                                  p(g1), p(g2), None, H, p(out), H, p(pa), p(pb), p(pc), p(part), prec, None)
        assert rc == 0
        return
    rc = lib.ggcn_layer_fused(p(x), H, p(pack), p(csr.rowmask), p(csr.graph_ops), p(b), B, T, H, H, None, p(g1), p(g2), None if NOOUT else p(out), H,
                              p(pa), p(pb), None, None, None, prec, None)
    assert rc == 0
for _ in range(int(os.environ.get('TRACE_WARMUP', '300'))):   # the chip needs ~0.1 s of load to settle
    run()
torch.cuda.synchronize()
run(); torch.cuda.synchronize()
n = 6144 if BLOCK else 3072
buf = np.zeros(n * 8, dtype=np.uint64)
lib.ggcn_lab_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.ggcn_lab_trace_read(buf.ctypes.data, buf.nbytes) == 0
tr = buf.reshape(n, 8)
t0 = tr[:, 3].min()
hw, xcc = tr[:, 1].astype(np.int64), tr[:, 2].astype(np.int64) & 0xF
start, lb, le, end = [(tr[:, k] - t0).astype(np.float64) / 100.0 for k in (3, 4, 5, 6)]   # microseconds
print("kernel span %.1f us; blocks %d" % (end.max(), n))
print("prologue  (start->loop)   median %.2f us  p10 %.2f p90 %.2f" % tuple(np.percentile(lb - start, [50, 10, 90])))
print("main loop                 median %.2f us  p10 %.2f p90 %.2f" % tuple(np.percentile(le - lb, [50, 10, 90])))
print("epilogue  (loop->end)     median %.2f us  p10 %.2f p90 %.2f" % tuple(np.percentile(end - le, [50, 10, 90])))
g0e = (tr[:, 7] - t0).astype(np.float64) / 100.0
print("epilogue: gate loads + first graph median %.2f us; each later graph %.2f us" % (np.median(g0e - le), np.median(end - g0e) / 3))
if BLOCK:
    second = (tr[:, 0].astype(np.int64) & 7) >= 4
    for name, sel in (("layer-1 tiles", ~second), ("layer-2 tiles", second)):
        print("  %s: main loop median %.2f us, epilogue median %.2f us (p90 %.2f), whole %.2f" % (
            name, np.median((le - lb)[sel]), np.median((end - le)[sel]), np.percentile((end - le)[sel], 90), np.median((end - start)[sel])))
print("HW_ID bits that vary: %s" % bin(int(np.bitwise_or.reduce(hw ^ hw[0]))))
cu_key = (xcc << 32) | (hw & 0xFF00)       # cu_id[11:8], sh_id[12], se_id[15:13]; bits 16-19 are the workgroup slot
cus = collections.defaultdict(list)
for i in range(n):
    cus[int(cu_key[i])].append(i)
print("distinct CU keys: %d; blocks per CU min/max %d/%d" % (len(cus), min(map(len, cus.values())), max(map(len, cus.values()))))
gaps, overlap_frac, conc = [], [], []
for key, ids in cus.items():
    ids.sort(key=lambda i: start[i])
    # lanes: greedy assignment of blocks to resident slots
    lanes = []
    for i in ids:
        for L in lanes:
            if end[L[-1]] <= start[i] + 0.005:
                gaps.append(start[i] - end[L[-1]]); L.append(i); break
        else:
            lanes.append([i])
    conc.append(len(lanes))
    # epilogue of i overlapped by a main loop of another block on the same CU?
    for i in ids:
        e0, e1 = le[i], end[i]
        cov = 0.0
        for j in ids:
            if j != i:
                cov += max(0.0, min(e1, le[j]) - max(e0, lb[j]))
        overlap_frac.append(cov / max(e1 - e0, 1e-9))
print("resident slots per CU: %s" % collections.Counter(conc))
print("turnaround gap (end -> next start on the CU slot): median %.2f us p90 %.2f" % tuple(np.percentile(gaps, [50, 90])))
print("fraction of an epilogue covered by the co-resident's main loop: median %.2f  mean %.2f" % (np.median(overlap_frac), np.mean(overlap_frac)))
# one CU, ASCII
key = sorted(cus)[0]
print("timeline of one CU (us): block start loop_begin loop_end end")
for i in sorted(cus[key], key=lambda i: start[i]):
    print("  blk %4d  %7.1f %7.1f %7.1f %7.1f" % (tr[i, 0], start[i], lb[i], le[i], end[i]))
# when do epilogues happen chip-wide: histogram of loop_end times
hist, edges = np.histogram(le, bins=40, range=(0, end.max()))
print("histogram of main-loop-end times (40 bins over the kernel):")
print(" ".join("%d" % h for h in hist))
# siblings = the 3 column tiles of one row block (tile_of_block: xcd = id & 7, slot = id >> 3)
n_wg = 3
ids = tr[:, 0].astype(np.int64)
m_tile = ((ids >> 3) // n_wg * 4 + (ids & 3)) * 2 + ((ids & 7) >> 2) if BLOCK else (ids >> 3) // n_wg * 8 + (ids & 7)
same_xcc = 0; spread = []; xcc_of_id_ok = 0
by_tile = collections.defaultdict(list)
for i in range(n):
    by_tile[int(m_tile[i])].append(i)
    xcc_of_id_ok += int(xcc[i] == (ids[i] & 7))
for t, members in by_tile.items():
    if len(members) == n_wg:
        same_xcc += int(len({int(xcc[i]) for i in members}) == 1)
        spread.append(max(start[i] for i in members) - min(start[i] for i in members))
print("XCC_ID == id & 7 for %d of %d blocks" % (xcc_of_id_ok, n))
print("row blocks whose column tiles all ran on one XCD: %d of %d" % (same_xcc, len(by_tile)))
print("start-time spread inside a sibling group: median %.2f us  p90 %.2f  max %.2f" % (np.median(spread), np.percentile(spread, 90), max(spread)))
spread_le = [max(le[i] for i in m) - min(le[i] for i in m) for m in by_tile.values() if len(m) == n_wg]
print("loop-end spread inside a sibling group:   median %.2f us  p90 %.2f  max %.2f" % (np.median(spread_le), np.percentile(spread_le, 90), max(spread_le)))
