#!/bin/bash
# config 4 through the one-launch long-graph layer: MFMA neighbour sums (default) against the lane sums of round 3
# (GGCN_LONG_LANE_SUMS=1), same box, back to back; phase timeline of the lab trace build if present.
set -u
O=gpurun_out/r4; mkdir -p $O
TAG=${1:-ab}
if [ -f tools/_lab/libggcn_ltrace.so ]; then python tools/long_trace.py > $O/long_trace_$TAG.txt 2>&1; fi
python bench.py --config 4 --steps 200 --warmup 20 > $O/c4_mma_$TAG.json 2>$O/c4_mma_$TAG.err || exit 1
GGCN_LONG_LANE_SUMS=1 python bench.py --config 4 --steps 200 --warmup 20 > $O/c4_lane_$TAG.json 2>$O/c4_lane_$TAG.err || exit 1
python bench.py --config 4 --steps 200 --warmup 20 > $O/c4_mma2_$TAG.json 2>$O/c4_mma2_$TAG.err || exit 1
[ -f $O/long_trace_$TAG.txt ] && cat $O/long_trace_$TAG.txt
python - <<PY
import json
for n in ("mma","lane","mma2"):
    d=json.loads(open("$O/c4_%s_$TAG.json"%n).read().strip().splitlines()[-1]); print(n, "ms_per_step %.4f  kernel %.1f us" % (d["ms_per_step"], d["roofline"]["avg_launch_us"]))
PY
