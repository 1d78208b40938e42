#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy report of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/resources.py fused_layer.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ed-gated-gcn_amd", "csrc")


def main():
    src = sys.argv[1]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/dev/null"] + sys.argv[2:]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            m = re.search(r"\d+:\d+:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            if cur:
                rows.append(cur)
            name = t.split(":", 1)[1].strip()
            try:
                name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
            except OSError:
                pass
            cur = {"name": re.sub(r"ggcn::\(anonymous namespace\)::", "", name)[:90]}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    if cur:
        rows.append(cur)
    for r in rows:
        print("%-92s VGPR %-4s AGPR %-4s SGPR %-4s spill %-3s scratch %-5s LDS %-6s occ %s" % (
            r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")),
            r.get("VGPR Spill", r.get("VGPRs Spill", "?")), r.get("ScratchSize [bytes/lane]", "?"),
            r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))


if __name__ == "__main__":
    main()
