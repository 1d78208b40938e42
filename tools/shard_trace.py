#!/usr/bin/env python3
"""Per-kernel durations and gaps of the 512-graph shard step (bench.py --force-dist --graphs 512) from a rocprofv3 kernel trace.
usage: tools/shard_trace.py <dir with *kernel_trace.csv>.  Development tool."""
import csv, glob, sys, statistics, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]           # steady state: the second half of the run
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-72s n=%5d  median %.1f us" % (k, len(v), statistics.median(v)))
# period between consecutive block launches
starts = [int(r["Start_Timestamp"]) for r in rows if "layer_fused_kernel" in r["Kernel_Name"]]
per = [(b - a) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
print("step period (block start to block start): median %.1f us over %d steps" % (statistics.median(per), len(per)))
# a typical step: the kernels between two block starts, with their gaps
i0 = [i for i, r in enumerate(rows) if "layer_fused_kernel" in r["Kernel_Name"]][len(starts) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 12]:
    print("  +%7.1f us  %6.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80]))
