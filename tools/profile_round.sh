#!/bin/bash
# One-call profile of the headline bench: rocprofv3 kernel stats (un-mixed with PMC), then the PMC passes of
# tools/pmc.sh, all on the same box.   usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R && python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
# (the profiled runs carry no appendix: no child processes, no box probes -- only the timed loop's kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --no-cpu-baseline --no-alt --no-config4 --no-box "$@" > $OUT/stats.log 2>&1
cd $R && bash tools/pmc.sh $TAG bench.py --steps 20 --warmup 5 --precondition 100 --no-cpu-baseline --no-alt --no-config4 --no-box "$@" > $OUT/pmc.txt 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cp $R/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt 2>/dev/null
echo done
