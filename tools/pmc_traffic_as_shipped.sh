#!/bin/bash
# the PMC traffic passes (FETCH_SIZE, WRITE_SIZE: one counter per run) of the AS-SHIPPED configurations: hierarchical actions, the controller
# kernel's traffic added to the step kernel's (tools/pmc_traffic.py --hierarchical)
set -e
out=gpurun_out/${1:-pmc_as}
round=${2:-4}
mkdir -p $out
export TMPDIR=/tmp
cp profiles/pmc_traffic.json $out/pmc_traffic.json
for spec in "scenario1 1 2" "scenario_nvn 2 4" "scenario_nvn 4 8"; do
  set -- $spec; t=$1; ps=$2; A=$3
  extra="--hierarchical"; if [ $ps != 1 ]; then extra="$extra --per-side $ps"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${t}${A}_$c -o pmc -- python3 bench.py --task $t $extra --steps 300 --warmup 100 --device-only --no-configs --no-cpu-baseline --no-saturating --no-steady-state --checksum-calls 20 > $out/${t}${A}_$c.json 2> $out/${t}${A}_$c.err
  done
  python3 tools/pmc_traffic.py $out/${t}${A}_FETCH_SIZE $out/${t}${A}_WRITE_SIZE --task $t --agents $A --round $round --hierarchical --out $out/pmc_traffic.json | grep -E "bytes_per_aircraft_step|traffic_bytes|KB"
  find $out -name "*.csv" -size +5M -delete
done
