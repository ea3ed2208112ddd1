#!/usr/bin/env python3
"""Diagnostic: free-flight (open-loop) difference between the fp32 HIP step and the float64 oracle — no state injection at all.

Both start from the same initial conditions and take the same action sequence. Per env step it records, over all aircraft, the
position difference (NEU, m), the attitude difference (roll / pitch / yaw, rad), the velocity difference (m/s), the observation and
reward differences, and whether the discrete decisions of the flight control system agree (leading-edge-flap switch on alpha / Mach,
trailing-edge-flap switch on calibrated airspeed / Mach, turbine phase word). Writes gpurun_out/open_loop_<mode>.json: the curves the
frozen envelopes of tests/test_gpu_open_loop.py were taken from.

usage: open_loop.py straight|random [steps] [envs]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aircombat_selfplay_amd as pkg  # noqa: E402
from oracle import oracle  # noqa: E402
from open_loop_util import OpenLoopPair  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "straight"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    E = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    pair = OpenLoopPair(pkg, oracle, E, spread=True)
    rng = np.random.default_rng(20250321)
    rows = []
    age = np.zeros(E, dtype=np.int64)
    act = np.tile(np.array([20, 19, 20, 0], dtype=np.float32), (E, 2, 1))
    for step in range(steps):
        if mode == "random" and step % 5 == 0:
            act = np.stack([rng.integers(0, n, size=(E, 2)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32)
        m = pair.step(act)
        live = m["live"]
        age += 1
        row = {"step": step + 1, "live_envs": int(live.sum()), "max_age": int(age.max())}
        for k in ("pos_m", "att_rad", "vel_ms", "obs", "rew"):
            v = m[k][live]
            row[k + "_max"] = float(v.max()) if v.size else None
            row[k + "_p50"] = float(np.median(v)) if v.size else None
            if v.size:
                row[k + "_age_of_max"] = int(age[live][np.unravel_index(v.argmax(), v.shape)[0]])
        age[pair.last_reset] = 0
        rows.append(row)
        if (step + 1) % 50 == 0:
            print(row, flush=True)
    out = {"mode": mode, "envs": E, "steps": steps, "rows": rows, "horizon_steps": pair.horizon.tolist(),
           "horizon_reason": pair.reason, "done_mismatch_envs": int(pair.done_mismatch.sum())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"open_loop_{mode}.json"), "w") as f:
        json.dump(out, f)
    h = pair.horizon
    print(f"{mode}: horizon (first step a discrete FCS decision differs) min {h.min()} p10 {np.percentile(h, 10):.0f} median {np.median(h):.0f} "
          f"never {int((h >= steps).sum())}/{E}; reasons {pair.reason_counts()}")


if __name__ == "__main__":
    main()
