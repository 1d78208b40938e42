#!/usr/bin/env python3
"""Does hipStreamWaitValue32 gate a stream on a value a KERNEL writes (plain device memory / signal memory)?  Probe for a
flag-based hand-off between the replay stream and the collective's stream (no event record on the replay stream)."""
import ctypes, os, sys, time
import torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
attr = ctypes.c_int(0)
# hipDeviceAttributeCanUseStreamWaitValue: look the enum up by probing the known neighbourhood is fragile; just try the call
A, B = torch.cuda.Stream(), torch.cuda.Stream()
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
GTE = 0

def trial(name, ptr, write):
    y = torch.zeros(4, device=dev)
    rc = hip.hipStreamWaitValue32(ctypes.c_void_p(B.cuda_stream), ctypes.c_void_p(ptr), 1, GTE, 0xFFFFFFFF)
    print(name, "hipStreamWaitValue32 rc", rc, flush=True)
    if rc != 0:
        return
    with torch.cuda.stream(B):
        y.fill_(7.0)
        evb = torch.cuda.Event(); evb.record(B)
    time.sleep(0.05)
    early = evb.query()
    print(name, "stream B ran before the flag was written:", early, flush=True)
    write()
    t0 = time.perf_counter()
    while not evb.query() and time.perf_counter() - t0 < 3.0:
        time.sleep(0.001)
    print(name, "stream B released after the write:", evb.query(), "in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    if not evb.query():
        print("STUCK: leaving without synchronising", flush=True)
        os._exit(3)

flag = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
def w_kernel():
    with torch.cuda.stream(A):
        flag.fill_(1)
trial("plain device memory, kernel write:", flag.data_ptr(), w_kernel)
sig = ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2)   # hipMallocSignalMemory
print("hipExtMallocWithFlags(signal) rc", rc, hex(sig.value or 0), flush=True)
if rc == 0:
    def w_stream():
        print("  hipStreamWriteValue32 rc", hip.hipStreamWriteValue32(ctypes.c_void_p(A.cuda_stream), sig, 1, 0), flush=True)
    trial("signal memory, hipStreamWriteValue32:", sig.value, w_stream)
# cost on the producer stream: 200 x (small kernel + write value) vs 200 x (small kernel + event record)
x = torch.zeros(1 << 20, device=dev)
def loop(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(A):
        for i in range(200):
            x.add_(1.0)
            if kind == "event":
                torch.cuda.Event().record(A)
            elif kind == "write" and rc == 0:
                hip.hipStreamWriteValue32(ctypes.c_void_p(A.cuda_stream), sig, i + 2, 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e6
for kind in ("none", "event", "write", "none"):
    print("producer loop %-6s %.1f us per step" % (kind, loop(kind)), flush=True)
os._exit(0)
