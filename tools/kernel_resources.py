#!/usr/bin/env python3
"""Register / spill / scratch / LDS use of every kernel in the emitted gfx950 assembly (aircombat-selfplay_amd/build/aircombat_gfx950.s,
written by __graft_entry__.build_hip): the code-object metadata the VERDICT's spill targets are read from."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "aircombat-selfplay_amd", "build", "aircombat_gfx950.s")
txt = open(path).read()
rows = []
for b in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: int(re.search(r"\." + k + r":\s+(\S+)", b).group(1))
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    rows.append((name, int(b.split()[0]), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
try:
    dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
except OSError:
    dem = [r[0] for r in rows]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"{'kernel':78s} agpr vgpr vspill sspill scratch   lds")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d.replace("void ", ""))
    if filt in d:
        print(f"{d[:78]:78s} {r[1]:4d} {r[2]:4d} {r[3]:6d} {r[4]:6d} {r[5]:7d} {r[6]:5d}")
