#!/usr/bin/env python3
"""Board power beside the one-launch layer kernels of the wider graphs (1.5 s each): is the kernel at the power cap (time =
energy / 1400 W) or below it (waiting for something)?  Development tool."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ed_gated_gcn_amd as pkg
from ed_gated_gcn_amd import synth
import bench
dev = torch.device("cuda:0")
for B, T, H in ((512, 231, 768), (512, 160, 768), (512, 256, 768), (1024, 100, 768), (2048, 64, 768), (4096, 32, 768)):
    adj = synth.dependency_batch(B, T, 4.0)
    rp, ci, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rp, ci, B, T, dev)
    x = torch.randn(B, T, H, device=dev); g1 = torch.rand(B, H, device=dev); g2 = torch.rand(B, H, device=dev)
    w, b = synth.layer_params(H, H, seed=1)
    m = pkg.GraphConvolution(H, H, None).to(dev); m.precision = "f16mx8"; m.fused_max_t = 256
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(w)); m.bias.copy_(torch.from_numpy(b))
        f = lambda: m.forward_gated(x, csr, store_gate=g2, pool_gate_a=g1, pool_gate_b=g2, want_pool_a=True, want_pool_b=True)
        for _ in range(300): f()
        torch.cuda.synchronize()
        with bench.PowerSampler() as ps:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_end, cnt = time.perf_counter() + 1.5, 0
            a.record()
            while time.perf_counter() < t_end:
                for _ in range(100): f()
                cnt += 100
                torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
    us, pw = a.elapsed_time(e) / cnt * 1e3, ps.summary()
    print("B=%d T=%d H=%d: %7.1f us per layer  %7.1f W (max %7.1f, sclk %s MHz)  %.4f J" %
          (B, T, H, us, pw.get("power_w") or float("nan"), pw.get("power_w_max") or float("nan"), pw.get("sclk_dpm_mhz"),
           us * 1e-6 * (pw.get("power_w") or float("nan"))), flush=True)
