#!/bin/bash
# effective shader clock of each kernel = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md DVFS)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/clk_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -o clk -- python3 $R/"$@" > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
            acc[k].append((float(r["Counter_Value"]) / 8 / d / 1e9, d * 1e6))
for k, v in acc.items():
    if len(v) > 5:
        v = v[len(v)//4:]
        print("%-52s clock %.2f GHz  dur %.1f us  (n=%d)" % (k, sum(a for a, _ in v) / len(v), sum(b for _, b in v) / len(v), len(v)))
PY
