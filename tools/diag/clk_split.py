#!/usr/bin/env python3
"""Cycle stamps of the dynamics wave of workgroup 0 of the three-wave singlecombat kernel (variants/libclk.so from
tools/build_clk_variant.sh): the four pieces of every tick and the waits at the three barriers between them."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AIRCOMBAT_HIP_LIB"] = os.path.join(ROOT, "variants", "libclk.so")
import aircombat_selfplay_amd as pkg

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = pkg.HipVecEnv(pkg.default_config("singlecombat"), E, seed=1)
env.reset()
rng = np.random.default_rng(0)
fn = env.lib.dll.ac_debug_clocks
fn.argtypes = [ctypes.c_void_p]
for it in range(80):
    env.step(np.stack([rng.integers(0, n, size=(E, 2)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32))
    if it % 40 == 39:
        clk = np.zeros(256, dtype=np.uint64)
        fn(clk.ctypes.data)
        c = clk.astype(np.int64)
        print(f"---- step {it + 1}: prologue {c[1] - c[0]}, to first tick {c[2] - c[1]}")
        for sub in range(6):
            b = 2 + 8 * sub
            nxt = c[b + 8] if sub < 5 else c[50]
            print(f"  tick {sub}: p1 {c[b + 1] - c[b]:5d} wait {c[b + 2] - c[b + 1]:5d} | p2 {c[b + 3] - c[b + 2]:5d} wait {c[b + 4] - c[b + 3]:5d} | "
                  f"p3 {c[b + 5] - c[b + 4]:5d} wait {c[b + 6] - c[b + 5]:5d} | p4 {nxt - c[b + 6]:5d}   tick {nxt - c[b]:6d}")
            h = 64 + 8 * sub   # the helper waves' arrival at B1 / B2 / B3, as slack before the dynamics wave's own arrival
            print(f"          systems wave early by {c[b + 1] - c[h]:5d} / {c[b + 3] - c[h + 1]:5d} / {c[b + 5] - c[h + 2]:5d}   "
                  f"kinematics wave early by {c[b + 1] - c[h + 4]:5d} / {c[b + 3] - c[h + 5]:5d} / {c[b + 5] - c[h + 6]:5d}")
        print(f"  finish wait {c[51] - c[50]}  tail (finish -> task.step) {c[52] - c[51]}  task.step {c[53] - c[52]}  stores+outputs {c[54] - c[53]}  total {c[54] - c[0]}")
        # (the observation rows leave from the kinematics wave in this form: the dynamics wave writes the reward / done / info scalars only)
        print(f"  task.step: to obs {c[58] - c[52]}  terminations {c[59] - c[58]}  rewards+reset {c[53] - c[59]} | state stores {c[55] - c[53]}  reward / done / info scalars {c[54] - c[55]}")
env.close()
