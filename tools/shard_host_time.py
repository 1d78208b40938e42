#!/usr/bin/env python3
"""Host time per call of the N > 1 step loop's pieces at a 512-graph shard (world 1 over RCCL): hipGraph replay, all-gather
start (async), finish.  Development tool."""
import os, sys, time, statistics
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth, shard
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
B, T, H = 512, 32, 768
adj = synth.dependency_batch(B, T, 4.0)
rp, ci, _ = synth.csr_from_dense_host(adj)
csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
x = torch.randn(B, T, H, device=dev); g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
ls = []
for s in (1, 2):
    w, b = synth.layer_params(H, H, seed=s)
    m = pkg.GraphConvolution(H, H, None).to(dev).eval()
    with torch.no_grad(): m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
    ls.append(m)
head = torch.randn(H, 34, device=dev) / H ** 0.5
def forward():
    r = pkg.gated_gcn_block(x, csr, g1, g2, ls[0], ls[1]); r["payload"] = torch.mm(r["out"], head); return r
with torch.no_grad():
    for _ in range(3): forward()
    gs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): r = forward()
        gs.append((g, r))
gather = shard.PooledGather([B], 34, dev)
T = {"finish": [], "replay": [], "start": [], "step": []}
pending = []
for i in range(600):
    t0 = time.perf_counter()
    while len(pending) > 1: gather.finish(pending.pop(0))
    t1 = time.perf_counter()
    g, r = gs[i & 1]; g.replay()
    t2 = time.perf_counter()
    pending.append(gather.start(r["payload"]))
    t3 = time.perf_counter()
    if i >= 300:
        T["finish"].append(t1 - t0); T["replay"].append(t2 - t1); T["start"].append(t3 - t2); T["step"].append(t3 - t0)
torch.cuda.synchronize()
for k, v in T.items(): print("%-7s median %.1f us  p90 %.1f" % (k, statistics.median(v) * 1e6, sorted(v)[int(0.9 * len(v))] * 1e6))
# the same loop with a synchronous-on-stream gather (async_op=False) and with no gather at all
for mode in ("sync_op", "none"):
    torch.cuda.synchronize(); buf = torch.empty(B, 34, device=dev)
    t0 = time.perf_counter()
    for i in range(600):
        g, r = gs[i & 1]; g.replay()
        if mode == "sync_op": dist.all_gather_into_tensor(buf, r["payload"])
    torch.cuda.synchronize()
    print(mode, "loop: %.1f us per step" % ((time.perf_counter() - t0) / 600 * 1e6))
# what costs the ~13 us per step: an event record behind the replay? a second stream waiting for it? the copy?
side = torch.cuda.Stream(device=dev)
small = torch.zeros(512, 34, device=dev); small2 = torch.zeros(512, 34, device=dev)
evs = [torch.cuda.Event() for _ in range(4)]
def loop(mode, n=600):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        g, r = gs[i & 1]; g.replay()
        if mode == "record":
            evs[i & 3].record()
        elif mode == "record+sidewait":
            e = evs[i & 3]; e.record(); side.wait_event(e)
        elif mode == "record+sidewait+sidekernel":
            e = evs[i & 3]; e.record(); side.wait_event(e)
            with torch.cuda.stream(side): small2.add_(1.0)
        elif mode == "record+sidewait+sidecopy":
            e = evs[i & 3]; e.record(); side.wait_event(e)
            with torch.cuda.stream(side): small2.copy_(r["payload"], non_blocking=True)
        elif mode == "samestream_copy":
            small2.copy_(r["payload"], non_blocking=True)
    torch.cuda.synchronize()
    print("%-28s %.1f us per step" % (mode, (time.perf_counter() - t0) / n * 1e6), flush=True)
for m in ("none", "record+sidewait+sidekernel", "record+sidewait+sidecopy", "none"):
    loop(m)
# (a) a side-stream kernel that is the library's own; (b) the side stream joined back every step; (c) fork / join INSIDE a captured graph
from ed_gated_gcn_amd import _capi
lib = pkg.load_library()
part = torch.zeros(B, 12, device=dev); xyb = torch.zeros((), device=dev)
def loop2(mode, n=600):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    main = torch.cuda.current_stream()
    for i in range(n):
        g, r = gs[i & 1]; g.replay()
        e = evs[i & 3]; e.record(); side.wait_event(e)
        with torch.cuda.stream(side):
            if mode == "side_libkernel": lib.ggcn_overlap_reduce(_capi.ptr(part), B, H, _capi.ptr(xyb), ctypes.c_void_p(side.cuda_stream))
            elif mode == "side_add_joined": small2.add_(1.0)
            elif mode == "side_mm": torch.mm(r["out"], head, out=small2)
        if mode == "side_add_joined": main.wait_stream(side)
    torch.cuda.synchronize()
    print("%-28s %.1f us per step" % (mode, (time.perf_counter() - t0) / n * 1e6), flush=True)
import ctypes
for m in ("side_libkernel", "side_add_joined", "side_mm"):
    loop2(m)
try:
    g3s = []
    with torch.no_grad():
        for k in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    small2.add_(1.0)
                r = forward()
                cur.wait_stream(side)
            g3s.append(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(600): g3s[i & 1].replay()
    torch.cuda.synchronize()
    print("graph with a forked side kernel: %.1f us per step" % ((time.perf_counter() - t0) / 600 * 1e6))
except Exception as e:
    print("fork/join capture refused:", type(e).__name__, str(e)[:300])
# the all-gather captured INSIDE the graph (world 1: a copy node)
try:
    gbuf = [torch.empty(B, 34, device=dev) for _ in range(2)]
    g2s = []
    with torch.no_grad():
        for k in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                r = forward()
                dist.all_gather_into_tensor(gbuf[k], r["payload"])
            g2s.append(g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(600): g2s[i & 1].replay()
    torch.cuda.synchronize()
    print("gather captured in the graph: %.1f us per step" % ((time.perf_counter() - t0) / 600 * 1e6))
except Exception as e:
    print("capture of the all-gather refused:", type(e).__name__, str(e)[:300])
# block-only graph + EAGER dense head + async gather: does the event behind an eager kernel cost less than behind a graph launch?
try:
    with torch.no_grad():
        gb = []
        for k in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                r = pkg.gated_gcn_block(x, csr, g1, g2, ls[0], ls[1])
            gb.append((g, r))
        gather2 = shard.PooledGather([B], 34, dev)
        for variant in ("graph(block+xy) + eager mm + async gather", "all in graph + async gather"):
            pending = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(600):
                while len(pending) > 1: gather2.finish(pending.pop(0))
                if variant.startswith("graph(block"):
                    g, r = gb[i & 1]; g.replay(); pay = torch.mm(r["out"], head)
                else:
                    g, r = gs[i & 1]; g.replay(); pay = r["payload"]
                pending.append(gather2.start(pay))
            while pending: gather2.finish(pending.pop(0))
            torch.cuda.synchronize()
            print("%-45s %.1f us per step" % (variant, (time.perf_counter() - t0) / 600 * 1e6), flush=True)
except Exception as e:
    print("variant failed:", type(e).__name__, str(e)[:200])
dist.destroy_process_group()
