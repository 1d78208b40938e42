#!/bin/bash
# Board power, power cap and shader clock while the headline forward runs in a loop (rocm-smi sampled from the side, once a
# second): evidence for the clock the block kernel runs at.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4; mkdir -p $O
python3 $R/bench.py --steps 40000 --warmup 50 --no-cpu-baseline --no-alt > $O/power_bench.json 2> $O/power_bench.err &
BP=$!
: > $O/power_probe.txt
for i in $(seq 1 45); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|Sensor junction" | tr -s ' \t' ' ' | tr '\n' ' ' >> $O/power_probe.txt
  echo >> $O/power_probe.txt
  kill -0 $BP 2>/dev/null || break
  sleep 1
done
wait $BP
rocm-smi --showmaxpower 2>/dev/null | grep -E "Max" | tr -s ' \t' ' ' >> $O/power_probe.txt
cat $O/power_probe.txt
