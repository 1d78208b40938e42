#!/usr/bin/env python3
"""Kernel lab: build variants of libggcn_hip.so with extra -D flags and time one entry point of
each, interleaved in ONE process (cdna guide §5.4 rule 24).  Development tool, not product code.

  python tools/lab.py build  name1:-DFLAG_A name2:"-DFLAG_B -DFLAG_C" ...   (CPU box: cross-compiles)
  python tools/lab.py build  old@<git-rev>[:flags]                            (the csrc/ of that revision)
  python tools/lab.py time linear|aggregate|block [names...]                 (GPU box)

LAB_ENERGY=1 times every variant ALONE for 1.5 s with the board power beside it.  For kernels at the board's power cap (the block
kernel, config 4, dW) that is the yardstick for variants with a DIFFERENT power profile: the power controller averages over
milliseconds, so in a five-launch interleave the cheaper variant runs at the clock the dearer one leaves it -- the eight-wavefront
block kernel read 1-6 % slower interleaved and is 2-4 % faster in steady state.  Small variants of one loop
compare fine either way.
"""
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "ed-gated-gcn_amd", "csrc")
LAB = os.path.join(ROOT, "tools", "_lab")


def build(specs):
    os.makedirs(LAB, exist_ok=True)
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        name, _, rev = name.partition("@")
        src_dir, these = CSRC, srcs
        if rev:   # sources of another revision, extracted beside the lab libraries
            src_dir = os.path.join(LAB, "src_" + name, "ed-gated-gcn_amd", "csrc")
            os.makedirs(os.path.join(LAB, "src_" + name), exist_ok=True)
            tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "ed-gated-gcn_amd/csrc", "include"], check=True,
                                 capture_output=True).stdout
            subprocess.run(["tar", "-x", "-C", os.path.join(LAB, "src_" + name)], input=tar, check=True)
            these = [f for f in sorted(os.listdir(src_dir)) if f.endswith(".hip")]
        out = os.path.join(LAB, "libggcn_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-o", out] + flags.split() + [os.path.join(src_dir, s) for s in these]
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait():
            raise SystemExit("build of %s failed" % name)
        print("built", name)


def time_variants(what, names):
    import torch
    import ed_gated_gcn_amd as pkg
    from ed_gated_gcn_amd import _capi, synth
    dev = torch.device("cuda:0")
    B, T, H = int(os.environ.get("LAB_GRAPHS", "4096")), int(os.environ.get("LAB_T", "32")), 768
    N = B * T
    adj = synth.dependency_batch(B, T, 4.0)
    rowptr, colidx, _ = synth.csr_from_dense_host(adj)
    csr = pkg.BatchedCSR.from_arrays(rowptr, colidx, B, T, dev)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=gen).to(dev)
    if os.environ.get("LAB_ZERO_X"):
        x.zero_()
    g1 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    g2 = torch.sigmoid(torch.randn(B, H, generator=gen)).to(dev)
    w, b = synth.layer_params(H, H, seed=1)
    w, b = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    y = torch.empty(N, H, device=dev)
    out = torch.empty(N, H, device=dev)
    pa, pb = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
    pc, part = torch.empty(B, H, device=dev), torch.empty(B, (H + 63) // 64, device=dev)
    if not names:
        names = sorted(f[len("libggcn_"):-3] for f in os.listdir(LAB) if f.endswith(".so"))
    libs = {}
    for n in names:
        base = n.split("@")[0]  # "name@mx8" times the f16mx8 precision of that build
        path = os.path.join(LAB, "libggcn_%s.so" % base) if base != "main" else pkg.lib_path()
        lib = ctypes.CDLL(path)
        lib.ggcn_abi_version.restype = ctypes.c_int
        new_abi = lib.ggcn_abi_version() >= 7      # 7: graph_ops argument (ggcn_graph_operands blocks)
        lib._abi = lib.ggcn_abi_version()
        for fn, (res, args) in _capi.PROTOTYPES.items():
            if hasattr(lib, fn):
                if fn == "ggcn_layer_fused" and not new_abi:
                    args = args[:4] + args[5:]
                if fn == "ggcn_block_fused" and lib._abi < 10:   # before ABI 10: no graph_ops2 argument
                    args = args[:5] + args[6:]
                getattr(lib, fn).restype, getattr(lib, fn).argtypes = res, args
        lib._new_abi = new_abi
        prec = 4 if n.endswith("@mx6") else 2 if n.endswith("@mx8") else 0
        pack = torch.empty(lib.ggcn_weight_pack_bytes(H, H, prec), dtype=torch.uint8, device=dev)
        assert lib.ggcn_weight_pack(_capi.ptr(w), H, H, H, prec, 0, _capi.ptr(pack), None) == 0
        libs[n] = (lib, pack, prec)
    st = _capi.stream_of(dev)
    p = _capi.ptr

    def run(n):
        lib, pack, prec = libs[n]
        masks = (p(csr.rowmask), p(csr.graph_ops)) if lib._new_abi else (p(csr.rowmask),)
        gops = p(csr.graph_ops) if lib._new_abi else p(csr.rowmask)
        if what == "linear":
            rc = lib.ggcn_linear(p(x), H, p(w), H, p(pack), p(y), H, N, H, H, prec, st)
        elif what == "linear_pp":     # lab_pp.hip: ping-pong form of the f16mx8 linear (falls back to the plain linear)
            if hasattr(lib, "ggcn_lab_linear_pp"):
                lib.ggcn_lab_linear_pp.restype = ctypes.c_int
                lib.ggcn_lab_linear_pp.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
                rc = lib.ggcn_lab_linear_pp(p(x), H, p(pack), p(y), H, N, H, H, st)
            else:
                rc = lib.ggcn_linear(p(x), H, p(w), H, p(pack), p(y), H, N, H, H, prec, st)
        elif what == "linear_fp32":
            rc = lib.ggcn_linear(p(x), H, p(w), H, None, p(y), H, N, H, H, 1, st)
        elif what == "aggregate":
            rc = lib.ggcn_aggregate(p(x), H, p(csr.rowptr), p(csr.colidx), None, p(b), B, T, H, None, p(g1), p(g2),
                                    p(out), H, p(pa), p(pb), st)
        elif what == "fused_noout":   # pooled outputs only: no [N,F] store at all
            rc = lib.ggcn_layer_fused(p(x), H, p(pack), *masks, p(b), B, T, H, H, None, p(g1), p(g2),
                                      None, H, p(pa), p(pb), None, None, None, prec, st)
        elif what == "fused2":        # the block's two layers back to back: layer 2 reads what layer 1 wrote
            rc = lib.ggcn_layer_fused(p(x), H, p(pack), *masks, p(b), B, T, H, H, None, p(g1), p(g2),
                                      p(out), H, p(pa), p(pb), None, None, None, prec, st)
            rc = rc or lib.ggcn_layer_fused(p(out), H, p(pack), *masks, p(b), B, T, H, H, p(g2), p(g2), None,
                                            p(y), H, p(pa), None, None, None, None, prec, st)
        elif what == "blockfused":    # the whole gated block as one launch (timing: W12 := the same image)
            if lib._abi >= 10:   # ABI 10: the block's second layer reads ggcn_graph_operands2 blocks
                rc = lib.ggcn_block_fused(p(x), H, p(pack), p(pack), gops, p(csr.graph_ops2(0 if prec == 0 else 1)), p(b), p(b), p(b), B, T, H, H,
                                          p(g1), p(g2), None, H, p(out), H, p(pa), p(pb), p(pc), p(part), prec, st)
            else:
                rc = lib.ggcn_block_fused(p(x), H, p(pack), p(pack), gops, p(b), p(b), p(b), B, T, H, H, p(g1), p(g2),
                                          None, H, p(out), H, p(pa), p(pb), p(pc), p(part), prec, st)
        elif what == "fused":
            rc = lib.ggcn_layer_fused(p(x), H, p(pack), *masks, p(b), B, T, H, H, None, p(g1), p(g2),
                                      p(out), H, p(pa), p(pb), None, None, None, prec, st)
        else:
            raise SystemExit("unknown target " + what)
        assert rc == 0, lib.ggcn_last_error()

    ref = None
    for n in names:
        run(n)
        torch.cuda.synchronize()
        res = (y if what.startswith("linear") else out).clone()
        if ref is None:
            ref = res
        print("%-24s max|diff vs %s| = %.3g" % (n, names[0], float((res - ref).abs().max())))
    if os.environ.get("LAB_ENERGY"):   # each variant alone for ~1.5 s: board power from sysfs beside it (bench.py's sampler) -> energy per launch
        sys.path.insert(0, ROOT)
        import bench
        import time as _time
        for n in names:
            for _ in range(300):
                run(n)
            torch.cuda.synchronize()
            with bench.PowerSampler() as ps:
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t_end = _time.perf_counter() + 1.5
                cnt = 0
                a.record()
                while _time.perf_counter() < t_end:
                    for _ in range(100):
                        run(n)
                    cnt += 100
                    torch.cuda.synchronize()
                e.record()
                torch.cuda.synchronize()
            us = a.elapsed_time(e) / cnt * 1e3
            pw = ps.summary()
            print("%-10s %8.1f us per launch   %7.1f W (max %7.1f, sclk %s MHz)   %.4f J per launch" %
                  (n, us, pw.get("power_w") or float("nan"), pw.get("power_w_max") or float("nan"), pw.get("sclk_dpm_mhz"), us * 1e-6 * (pw.get("power_w") or float("nan"))), flush=True)
        return
    for _ in range(150):          # precondition: the chip needs ~50 ms of load to settle
        run(names[0])
    torch.cuda.synchronize()
    times = {n: [] for n in names}
    for rnd in range(12):
        for n in names:
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                run(n)
            e.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                times[n].append(a.elapsed_time(e) / 5 * 1e3)
    for n in names:
        print("%-24s median %8.1f us   min %8.1f us" % (n, statistics.median(times[n]), min(times[n])))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        time_variants(sys.argv[2], sys.argv[3:])
