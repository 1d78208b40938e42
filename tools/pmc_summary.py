#!/usr/bin/env python3
"""Sum rocprofv3 --pmc CSVs per kernel (mean per dispatch).  Usage: pmc_summary.py <dir>"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in acc.items():
    lines.append("== %s" % k)
    for c, v in sorted(cs.items()):
        # a counter may report one row per dimension (XCD/SE): sum rows of one dispatch
        lines.append("   %-28s mean/dispatch-row %.4g  rows %d  total %.6g" % (c, sum(v) / len(v), len(v), sum(v)))
out = "\n".join(lines)
print(out)
open(os.path.join(d, "summary.txt"), "w").write(out + "\n")
