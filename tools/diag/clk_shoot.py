#!/usr/bin/env python3
"""Cycle stamps of workgroup 0 of the pair-form 1v1 missile kernels (variants/libclk.so from tools/build_clk_variant.sh):
where the environment wave and the flight wave spend a step of singlecombat_shoot / singlecombat_dodge_missile."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AIRCOMBAT_HIP_LIB"] = os.path.join(ROOT, "variants", "libclk.so")
import aircombat_selfplay_amd as pkg

task = sys.argv[1] if len(sys.argv) > 1 else "singlecombat_shoot"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = pkg.HipVecEnv(pkg.default_config(task), E, seed=1)
env.reset()
rng = np.random.default_rng(0)
fn = env.lib.dll.ac_debug_clocks
fn.argtypes = [ctypes.c_void_p]
shoot = env.act_dim == 5
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 120):
    cols = [rng.integers(0, n, size=(E, 2)) for n in (41, 41, 41, 30)] + ([rng.integers(0, 2, size=(E, 2))] if shoot else [])
    env.step(np.stack(cols, axis=-1).astype(np.float32))
    if it % 40 == 39:
        clk = np.zeros(256, dtype=np.uint64)
        fn(clk.ctypes.data)
        c = clk.astype(np.int64)
        print(f"---- step {it + 1}: env wave: prologue {c[1] - c[0]}  to first substep {c[2] - c[1]}")
        for sub in range(12):
            b = 2 + 8 * sub
            f = 129 + 4 * sub
            if b + 8 >= 128:
                break
            nxt = c[b + 8] if sub < 11 and c[b + 8] > c[b] else c[52]
            print(f"  sub {sub:2d}: env wait+pose {c[b + 2] - c[b]:6d} missiles {c[b + 3] - c[b + 2]:6d} rest {nxt - max(c[b + 3], c[b + 2]):6d} | "
                  f"flight wait-run {c[f] - (c[f - 1] if sub else c[128]):6d} tick {c[f + 2] - c[f]:6d} post-pose {c[f + 3] - c[f + 2]:5d}")
        print(f"  tail: task.step {c[53] - c[52]} stores+outputs {c[54] - c[53]} | total {c[54] - c[0]}  flight store done {c[160] - c[128]}")
env.close()
