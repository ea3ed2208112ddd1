#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the step kernels of further tasks -> $out/pmc_traffic.json
set -e
out=gpurun_out/${1:-pmc_tasks}
mkdir -p $out
export TMPDIR=/tmp
cp profiles/pmc_traffic.json $out/pmc_traffic.json
for spec in "singlecombat_shoot 2" "scenario1 2" "multiplecombat 4" "scenario_nvn 4"; do
  set -- $spec; t=$1; A=$2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${t}_fetch -o pmc -- python3 bench.py --task $t --steps 300 --warmup 100 --no-cpu-baseline --no-saturating --checksum-calls 20 > $out/${t}_fetch.json 2> $out/${t}_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${t}_write -o pmc -- python3 bench.py --task $t --steps 300 --warmup 100 --no-cpu-baseline --no-saturating --checksum-calls 20 > $out/${t}_write.json 2> $out/${t}_write.err
  python3 tools/pmc_traffic.py $out/${t}_fetch $out/${t}_write --task $t --agents $A --out $out/pmc_traffic.json | grep -E "bytes_per_aircraft_step|traffic_bytes"
  find $out -name "*.csv" -size +5M -delete
done
