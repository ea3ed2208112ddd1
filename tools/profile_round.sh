#!/bin/bash
# One GPU-box pass: plain bench, rocprofv3 kernel stats, and the two PMC passes (own runs, no tracing domains besides kernel-trace).
# usage (inside gpurun): bash tools/profile_round.sh <tag>
set -e
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --no-cpu-baseline --no-saturating > $out/stats_bench.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-saturating --checksum-calls 20 > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-saturating --checksum-calls 20 > $out/pmc_write.json 2> $out/pmc_write.err
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write --out $out/pmc_traffic.json
find $out -name "*.csv" -size +20M -delete
ls -la $out $out/stats
