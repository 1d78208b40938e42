#!/usr/bin/env python3
"""Sub-word pooling (bert_amir5.py:600): ggcn_subword_pool vs torch.bmm on the GPU, reference-like
shapes (ORI_ML 31 words, ~1.6 sub-words per word, 12 x 768 features).  Development tool."""
import os, sys, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ed_gated_gcn_amd as pkg

dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e) * 1e3)
    return statistics.median(ts)
rng = np.random.default_rng(0)
for B, T, L, D in ((32, 31, 64, 9216), (256, 31, 64, 9216), (32, 31, 64, 768)):
    tr = np.zeros((B, T, L), np.float32)
    for b in range(B):
        off = 1
        for i in range(T):
            l = int(rng.integers(1, 4))
            if off + l >= L: break
            tr[b, i, off:off + l] = 1.0 / l; off += l
    a = torch.from_numpy(tr).to(dev); x = torch.randn(B, L, D, device=dev)
    us_h = t(lambda: pkg.subword_pool(a, x)); us_t = t(lambda: torch.bmm(a, x))
    byt = (B * L * D + B * T * D) * 4
    print("B=%d T=%d L=%d D=%d: subword_pool %.1f us (%.2f TB/s of x+y)   torch.bmm %.1f us" % (B, T, L, D, us_h, byt / us_h / 1e6, us_t))
