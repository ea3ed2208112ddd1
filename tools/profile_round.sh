#!/bin/bash
# One GPU-box pass: the plain bench line and the rocprofv3 kernel statistics of the same command.
# usage (inside gpurun): bash tools/profile_round.sh <tag>
set -e
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --no-cpu-baseline --no-saturating > $out/stats_bench.json 2> $out/stats.err
# the two kinds of launch of that command on their own: device-resident steps only, host-boundary steps only
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dev -o stats -- python3 bench.py --device-only --no-configs --no-saturating --no-cpu-baseline --no-steady-state > $out/dev.json 2> $out/dev.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/host -o stats -- python3 bench.py --host-only > $out/host.json 2> $out/host.err
# (the PMC passes -- FETCH_SIZE / WRITE_SIZE, one counter per run, kernel-trace only -- are tools/pmc_traffic_tasks.sh)
find $out -name "*trace.csv" -delete
ls -la $out $out/stats
