#!/bin/bash
# rocprofv3 kernel statistics of the step kernels of the other tasks (one short bench run each), collected into one CSV
set -e
out=gpurun_out/${1:-tasks}
mkdir -p $out
export TMPDIR=/tmp
echo '"Task","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > $out/tasks_kernel_stats.csv
for t in singlecombat_shoot singlecombat_dodge_missile scenario1 scenario_nvn scenario3_nvn multiplecombat wvr_lowlevel heading approach hierarchical_singlecombat; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$t -o stats -- python3 bench.py --task $t --steps 400 --warmup 50 --device-only --no-configs --no-cpu-baseline --no-saturating > $out/$t.json 2> $out/$t.err
  grep -E "step_kernel|controller" $out/$t/stats_kernel_stats.csv | sed "s/^/\"$t\",/" >> $out/tasks_kernel_stats.csv
  find $out/$t -name "*trace.csv" -delete
done
cat $out/tasks_kernel_stats.csv | cut -c1-170
