#!/bin/bash
# rollout-buffer measurement pass: bench line, rocprofv3 kernel stats, and the two PMC passes (separate runs, kernel-trace only)
set -e
out=gpurun_out/${1:-buf}
mkdir -p $out
export TMPDIR=/tmp
python3 bench_buffer.py > $out/bench.json 2> $out/bench.err
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench_buffer.py --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o pmc -- python3 bench_buffer.py --no-cpu-baseline --repeats 4 > $out/pmc_fetch.json 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o pmc -- python3 bench_buffer.py --no-cpu-baseline --repeats 4 > $out/pmc_write.json 2> $out/pmc_write.err
python3 - <<PY
import csv, glob, collections
for tag in ("pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(list)
    for p in glob.glob("$out/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(p)):
            if "returns_kernel" in r["Kernel_Name"] or "gather_rows" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(tag, k, "avg per dispatch %.1f KB-units (n=%d)" % (sum(v) / len(v), len(v)))
PY
find $out -name "*.csv" -size +20M -delete
grep -E "rbuf::|Name" $out/stats/stats_kernel_stats.csv | cut -c1-200
