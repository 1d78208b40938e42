#!/bin/bash
# Collect PMC counters for a command, one rocprofv3 pass per counter group (PMC only: no
# --kernel-trace/--stats mixing beyond what rocprofv3 needs; MI355X_MICROARCH.md slot limits:
# SQ 8, TCC 4 with FETCH_SIZE=3 / WRITE_SIZE=2).  Usage: tools/pmc.sh <tag> <python-script> [args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o pmc -- python3 $R/"$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see $OUT/p$i.log)"
done
python3 $R/tools/pmc_summary.py $OUT
