#!/bin/bash
# where the controller kernel's wave cycles go (one SQ pass): parked on s_waitcnt / barriers, issue-stalled, issuing; matrix-pipe busy cycles
# usage (inside gpurun): bash tools/pmc_controller.sh <tag> [per-side]
set -e
tag=${1:-ctl}
ps=${2:-4}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $out/pmc -o pmc -- python3 bench.py --task scenario_nvn --per-side $ps --hierarchical --steps 200 --warmup 50 --device-only --no-configs --no-cpu-baseline --no-saturating --no-steady-state > $out/bench.json 2> $out/err.txt
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "controller" in r["Kernel_Name"] or "step_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()):
        print("   %-26s avg per dispatch %16.1f  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
