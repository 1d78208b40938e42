import os, sys, time, statistics
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
import torch, torch.distributed as dist
sys.path.insert(0, "/root/repo")
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
signal = torch.zeros(2, dtype=torch.int32, device=dev)
def forward():
    r = pkg.gated_gcn_block(x, csr, g1, g2, ls[0], ls[1], dense_head=(head, None, signal)); r["payload"] = r["logits"]; return r
n_l = 0
with torch.no_grad():
    for _ in range(3): forward(); n_l += 1
    gs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): r = forward()
        gs.append((g, r))
torch.cuda.synchronize(); n_l = int(signal[1].item())
gather = shard.PooledGather([B], 34, dev)
T = {"finish": [], "replay": [], "start": [], "step": []}
pending = []
for i in range(600):
    t0 = time.perf_counter()
    while len(pending) > 1: gather.finish(pending.pop(0))
    t1 = time.perf_counter()
    g, r = gs[i & 1]; g.replay(); n_l += 1
    t2 = time.perf_counter()
    pending.append(gather.start(r["payload"], gate=(signal, n_l)))
    t3 = time.perf_counter()
    if i >= 300:
        T["finish"].append(t1 - t0); T["replay"].append(t2 - t1); T["start"].append(t3 - t2); T["step"].append(t3 - t0)
while pending: gather.finish(pending.pop(0))
torch.cuda.synchronize()
for k, v in T.items(): print("%-7s median %.1f us  p90 %.1f" % (k, statistics.median(v) * 1e6, sorted(v)[int(0.9 * len(v))] * 1e6))
# pieces of start()
side = gather._side_stream(dev)
import ctypes
hip = shard._hip()
def t(f, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return dt
print("hipStreamWaitValue32 (already satisfied): %.1f us" % t(lambda: hip.hipStreamWaitValue32(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(signal.data_ptr() + 4), 1, 0, 0xFFFFFFFF)))
def ctx():
    with torch.cuda.stream(side): pass
print("stream context: %.1f us" % t(ctx))
print("event create + record: %.1f us" % t(lambda: torch.cuda.Event().record(side)))
buf = torch.empty(B, 34, device=dev)
def ag():
    with torch.cuda.stream(side):
        w = dist.all_gather_into_tensor(buf, r["payload"], async_op=True); w.wait()
print("all_gather on side + wait: %.1f us" % t(ag))
os._exit(0)
