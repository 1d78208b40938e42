set -u
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh r05c4 --config 4 > gpurun_out/prof_r05c4.log 2>&1 && echo c4-done
cd /tmp && export TMPDIR=/tmp
LAB_PRECS=f16mx8 LAB_SHAPES="512,231,768" rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r05ace/stats -o ace -- python3 $R/tools/wide_timing.py > $R/gpurun_out/prof_r05ace.log 2>&1 && echo ace-stats-done
cd $R && LAB_PRECS=f16mx8 LAB_SHAPES="512,231,768" bash tools/pmc.sh r05ace tools/wide_timing.py > gpurun_out/prof_r05ace_pmc.log 2>&1 && echo ace-pmc-done
