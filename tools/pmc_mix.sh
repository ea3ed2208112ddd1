#!/bin/bash
# instruction mix / stall counters of the step kernel (one SQ pass, 8 counters)
set -e
tag=${1:-mix}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc -o pmc -- python3 bench.py --steps 300 --warmup 50 --device-only --no-configs --no-cpu-baseline --no-saturating --no-steady-state ${@:2} > $out/bench.json 2> $out/err.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "step_kernel" in r["Kernel_Name"] or "controller8_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        print("   %-22s avg per dispatch %14.1f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
