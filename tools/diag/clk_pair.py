#!/usr/bin/env python3
"""Cycle stamps of workgroup 0 of a pair-form scenario kernel (variants/libclk.so from tools/build_clk_variant.sh): where the
environment wave and the flight wave spend a step."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AIRCOMBAT_HIP_LIB"] = os.path.join(ROOT, "variants", "libclk.so")
import aircombat_selfplay_amd as pkg

per_side = int(sys.argv[1]) if len(sys.argv) > 1 else 2
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = pkg.default_nvn_config(per_side, task="scenario_nvn") if per_side > 1 else pkg.default_config("scenario1")
cls = pkg.HipShareVecEnv if per_side > 1 else pkg.HipVecEnv
env = cls(cfg, E, seed=1)
env.reset()
rng = np.random.default_rng(0)
fn = env.lib.dll.ac_debug_clocks
fn.argtypes = [ctypes.c_void_p]
A = env.num_agents
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 120):
    a = np.concatenate([np.stack([rng.integers(0, n, size=(E, A)) for n in (41, 41, 41, 30)], axis=-1), rng.random((E, A, 4)) < 0.05], axis=-1).astype(np.float32)
    env.step(a)
    if it % 40 == 39:
        clk = np.zeros(256, dtype=np.uint64)
        fn(clk.ctypes.data)
        c = clk.astype(np.int64)
        print(f"---- step {it + 1}: env wave: prologue {c[1] - c[0]}")
        for sub in range(6):
            b = 2 + 8 * sub
            f = 129 + 4 * sub
            print(f"  sub {sub}: env wait-pose {c[b + 1] - c[b]:6d} pose {c[b + 2] - c[b + 1]:6d} missiles {c[b + 3] - c[b + 2]:6d} scatter {c[b + 4] - c[b + 3]:6d} "
                  f"rest {(c[b + 8] if sub < 5 else c[60]) - c[b + 4]:6d} | flight wait-run {c[f] - (c[f - 1] if sub else c[128]):6d} tick {c[f + 2] - c[f]:6d} "
                  f"next pose {c[f + 3] - c[f + 2]:6d}")
        print(f"  rewards: to rewards (obs + 1v1 terminations) {c[70] - c[65]}  gun-track distances + first-eval {c[71] - c[70]}  missile posture walk {c[72] - c[71]}  "
              f"terms {c[73] - c[72]}  team mean {c[74] - c[73]}  terminations {c[66] - c[74]}")
        print(f"  weapons: target {c[80] - c[62]} launch {c[81] - c[80]} gun {c[82] - c[81]} chaff {c[63] - c[82]} | geometry {c[83] - c[63]} incoming {c[64] - c[83]} | "
              f"terminations: walk {c[84] - c[74]} codes {c[85] - c[84]} last {c[66] - c[85]}")
        print(f"  tail: wait-final {c[61] - c[60]} props {c[62] - c[61]} weapons {c[63] - c[62]} geometry+incoming {c[64] - c[63]} obs {c[65] - c[64]} "
              f"term+rewards {c[66] - c[65]} reset+stores {c[67] - c[66]} outputs {c[68] - c[67]}  | total {c[68] - c[0]}  flight store done {c[160] - c[128]}")
env.close()
