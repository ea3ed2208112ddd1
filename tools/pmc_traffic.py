#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of bench.py (FETCH_SIZE in one, WRITE_SIZE in the other; they do not fit one pass on gfx950)
into HBM bytes per launch of the step kernel, following MI355X_MICROARCH.md's HBM section: both counters are memory-side
request tallies in KB; WRITE_SIZE is exact for dword-per-lane stores; FETCH_SIZE is only calibrated by the guide for 16-B
streaming reads (where it reports half), so it is calibrated here on the state-digest kernel, which reads a known byte count
(328 B per aircraft since round 4: 19 groups of 16 B + the 24-B fp64 position) with the step kernel's own 16-B-per-lane group pattern
(rounds 1-3: 324 B, 4 B per lane).

usage: pmc_traffic.py <fetch_dir> <write_dir> --task singlecombat --envs 4096 --agents 2 [--out profiles/pmc_traffic.json]"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

STATE_READ_BYTES = 16 * 19 + 8 * 3


def per_kernel(directory, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def pick(acc, needle):
    vals = [v for k, vs in acc.items() if needle in k for v in vs]
    if not vals:
        raise SystemExit(f"no dispatches of a kernel containing {needle!r}")
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--task", default="singlecombat"); ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--kernel", default="step_kernel")
    ap.add_argument("--hierarchical", action="store_true", help="an as-shipped run: the controller kernel's traffic is added to the step kernel's")
    ap.add_argument("--out", default=None)
    ap.add_argument("--round", type=int, default=2, help="build round the measurement belongs to (bench.py reports entries of round >= 2)")
    a = ap.parse_args()
    fetch, write = per_kernel(a.fetch_dir, "FETCH_SIZE"), per_kernel(a.write_dir, "WRITE_SIZE")
    lanes = a.envs * a.agents
    cal = pick(fetch, "state_checksum_kernel")
    cal_kb = sum(cal) / len(cal)
    factor = (STATE_READ_BYTES * lanes / 1024.0) / cal_kb          # true KB per counted KB for this access pattern
    f = pick(fetch, a.kernel); w = pick(write, a.kernel)
    f = f[len(f) // 4:]; w = w[len(w) // 4:]                       # drop the warm-up quarter
    fetch_kb, write_kb = sum(f) / len(f), sum(w) / len(w)
    ctl = None
    if a.hierarchical:                                             # controller8_kernel of the same steps (one launch per step, like the step kernel)
        cf = pick(fetch, "controller8_kernel"); cw = pick(write, "controller8_kernel")
        cf = cf[len(cf) // 4:]; cw = cw[len(cw) // 4:]
        ctl = {"FETCH_SIZE_KB_raw": sum(cf) / len(cf), "WRITE_SIZE_KB": sum(cw) / len(cw)}
        fetch_kb += ctl["FETCH_SIZE_KB_raw"]; write_kb += ctl["WRITE_SIZE_KB"]
    rec = {"task": a.task, "round": a.round, "envs_per_gpu": a.envs, "aircraft": lanes, "kernel": a.kernel + (" + controller8_kernel" if a.hierarchical else ""),
           "hierarchical": bool(a.hierarchical), "controller": ctl, "dispatches": len(f),
           "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
           "fetch_calibration": {"kernel": "state_checksum_kernel", "known_KB": STATE_READ_BYTES * lanes / 1024.0,
                                 "FETCH_SIZE_KB_raw": cal_kb, "factor": factor, "dispatches": len(cal)},
           "traffic_bytes_per_launch": (fetch_kb * factor + write_kb) * 1024.0,
           "bytes_per_aircraft_step": (fetch_kb * factor + write_kb) * 1024.0 / lanes}
    print(json.dumps(rec, indent=1))
    if a.out:
        try:
            allrec = json.load(open(a.out))
        except (OSError, ValueError):
            allrec = {"runs": []}
        allrec["runs"] = [r for r in allrec["runs"] if not (r["task"] == a.task and r["envs_per_gpu"] == a.envs and r["aircraft"] == lanes
                                                            and bool(r.get("hierarchical", False)) == bool(a.hierarchical))] + [rec]
        json.dump(allrec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
