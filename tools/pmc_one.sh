#!/bin/bash
# One PMC pass with the given counters.  usage: tools/pmc_one.sh <tag> "<counters>" <python-script> [args...]
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc1_$TAG; mkdir -p $OUT/p1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT/p1 -o pmc -- python3 $R/"$@" > $OUT/p1.log 2>&1
python3 $R/tools/pmc_summary.py $OUT
