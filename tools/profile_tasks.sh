#!/bin/bash
# rocprofv3 kernel statistics of the step kernels of the other tasks (one short bench run each), collected into one CSV
set -e
out=gpurun_out/${1:-tasks}
mkdir -p $out
export TMPDIR=/tmp
echo '"Task","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > $out/tasks_kernel_stats.csv
for spec in "singlecombat_shoot 1 0" "singlecombat_dodge_missile 1 0" "scenario1 1 0" "scenario_nvn 2 0" "scenario_nvn 4 0" "multiplecombat 2 0" "wvr_lowlevel 1 0" \
            "heading 1 0" "approach 1 0" "multiplecombat_dodge_missile 2 0" "multiplecombat_dodge_missile 4 0" "hierarchical_singlecombat 1 1" "scenario1 1 1" "scenario_nvn 2 1" "scenario_nvn 4 1"; do
  set -- $spec; t=$1; ps=$2; hier=$3
  extra=""; tag=$t
  if [ $ps != 1 ]; then extra="--per-side $ps"; tag="${t}_${ps}v${ps}"; fi
  if [ $hier = 1 ]; then extra="$extra --hierarchical"; tag="${tag}_as_shipped"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o stats -- python3 bench.py --task $t $extra --steps 400 --warmup 50 --device-only --no-configs --no-cpu-baseline --no-saturating --no-steady-state > $out/$tag.json 2> $out/$tag.err
  grep -E "step_kernel|controller" $out/$tag/stats_kernel_stats.csv | sed "s/^/\"$tag\",/" >> $out/tasks_kernel_stats.csv
  find $out/$tag -name "*trace.csv" -delete
done
cat $out/tasks_kernel_stats.csv | cut -c1-170
