#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 5's tables from the committed round profiles (bench line + rocprofv3 per-task kernel statistics):
python tools/design_tables.py 4"""
import csv
import json
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "4"
b = json.load(open(f"profiles/round{rnd}_bench.json"))
stats = {}
for r in csv.DictReader(open(f"profiles/round{rnd}_kernel_stats_tasks.csv")):
    stats.setdefault(r["Task"], []).append((r["Name"], float(r["AverageNs"]) / 1e3))


def k(task, needle):
    for name, us in stats.get(task, []):
        if needle in name:
            return us
    return float("nan")


print(f"headline: value {b['value'] / 1e6:.1f} M ({b['ms_per_step'] * 1e3:.2f} us/step, {b.get('value_mode', '')[:30]}), zero-copy views "
      f"{b.get('zero_copy_views_mode', {}).get('value', 0) / 1e6:.1f} M, steady {b['steady_state']['value'] / 1e6:.1f} M, benign {b['benign_actions']['value'] / 1e6:.1f} M, "
      f"hoarding {b.get('hoarding_caller_copy_fallback', {}).get('value', 0) / 1e6:.1f} M")
print(f"device-resident {b['device_resident']['value'] / 1e6:.1f} M ({b['device_resident']['ms_per_step'] * 1e3:.2f} us), kernel_ms {b['roofline']['kernel_ms'] * 1e3:.2f} us, "
      f"host-boundary kernel {b['roofline']['kernel_ms_host_boundary'] * 1e3:.2f} us, frac {b['roofline']['frac']:.4f}, traffic/algorithmic {b['roofline'].get('traffic_over_algorithmic')}")
print(f"saturating: {b['saturating']['kernel_ms'] * 1e3:.1f} us, {b['saturating']['value'] / 1e9:.2f} G, hbm frac {b['saturating']['hbm']['frac']:.3f}, valu frac {b['saturating']['valu']['frac']:.3f}")
print(f"cpu: {b['cpu_baseline']['value'] / 1e6:.2f} M on {b['cpu_baseline']['cores']} cores, single thread {b['cpu_baseline']['single_thread']['value'] / 1e6:.3f} M")
print()
print("| config | kernels (rocprofv3) | host boundary (`VecEnv.step`, default mode) | device-resident | PCIe share of the boundary step |")
print("|---|---|---|---|---|")
names = {("singlecombat_shoot", False): ("C3 `singlecombat_shoot` (quad form)", "singlecombat_shoot"),
         ("scenario1", False): ("C3 `scenario1`, control-index form (quad form)", "scenario1"),
         ("scenario1", True): ("**C3 `scenario1` AS SHIPPED** (hierarchical)", "scenario1_as_shipped"),
         ("scenario_nvn2", False): ("C4 `scenario_nvn` 2v2, control-index form (pair form)", "scenario_nvn_2v2"),
         ("scenario_nvn2", True): ("**C4 `scenario_nvn` 2v2 AS SHIPPED**", "scenario_nvn_2v2_as_shipped"),
         ("multiplecombat2", False): ("C4 legacy `multiplecombat` 2v2", "multiplecombat_2v2"),
         ("scenario_nvn4", False): ("C5 `scenario_nvn` 4v4, control-index form (512 workgroups, pair form)", "scenario_nvn_4v4"),
         ("scenario_nvn4", True): ("**C5 `scenario_nvn` 4v4 AS SHIPPED**", "scenario_nvn_4v4_as_shipped")}
for c in b["configs"]:
    per = c["aircraft_per_env"] // 2
    key = (c["task"] + (str(per) if per > 1 else ""), c["controller_ms"] is not None)
    label, tag = names[key]
    ctl, stp = k(tag, "controller"), k(tag, "step_kernel")
    kern = f"controller {ctl:.1f} + step **{stp:.1f} µs**" if key[1] else f"**{stp:.1f} µs**"
    print(f"| {label} | {kern} | {c['ms_per_step'] * 1e3:.1f} µs → **{c['value'] / 1e6:.0f} M** | {c['device_resident']['ms_per_step'] * 1e3:.1f} µs → {c['device_resident']['value'] / 1e6:.0f} M | "
          f"{100 * c['pcie_bound_frac']:.0f} % |")
for t, label in (("singlecombat_dodge_missile", "C3 `singlecombat_dodge_missile` (quad form)"), ("wvr_lowlevel", "`wvr` (gun only, three-wave form)"), ("heading", "C1 `heading` (one aircraft per env)"),
                 ("approach", "`approach`")):
    print(f"| {label} | {k(t, 'step_kernel'):.1f} µs | | | |")
print(f"| hierarchical `singlecombat` | controller {k('hierarchical_singlecombat_as_shipped', 'controller'):.1f} + step {k('hierarchical_singlecombat_as_shipped', 'step_kernel'):.1f} µs | | | |")
