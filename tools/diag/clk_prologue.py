#!/usr/bin/env python3
"""Where the prologue of the three-wave singlecombat kernel goes (dynamics wave of workgroup 0; variants/libclk.so from
tools/build_clk_variant.sh): kernel entry -> table loads issued (kernel arguments have arrived) -> state loads issued -> action row asked
for -> tables in LDS (the table loads have returned) -> workgroup barrier -> first tick; once with the steps taken through
VecEnv.step (actions read from mapped host memory) and once device-resident."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIRCOMBAT_HIP_LIB", os.path.join(ROOT, "variants", "libclk.so"))
import torch
import aircombat_selfplay_amd as pkg

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = pkg.HipVecEnv(pkg.default_config("singlecombat"), E, seed=1, copy=False)
env.reset()
rng = np.random.default_rng(0)
fn = env.lib.dll.ac_debug_clocks
fn.argtypes = [ctypes.c_void_p]
acts = [np.stack([rng.integers(0, n, size=(E, 2)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32) for _ in range(8)]
dev = [torch.from_numpy(a).cuda() for a in acts]


def show(label):
    clk = np.zeros(256, dtype=np.uint64)
    fn(clk.ctypes.data)
    c = clk.astype(np.int64)
    print(f"{label}: dynamics wave: tables asked {c[123] - c[0]:5d} flight state asked {c[124] - c[0]:5d} task record asked {c[125] - c[0]:5d} | barrier reached by dynamics {c[120] - c[0]:5d} systems {c[121] - c[0]:5d} kinematics {c[122] - c[0]:5d} | prologue {c[1] - c[0]:5d} | to first tick {c[2] - c[1]:5d} | first tick p1 {c[3] - c[2]:5d} | six ticks {c[50] - c[2]:6d} | "
          f"env layer {c[53] - c[50]:5d} | state stores {c[55] - c[53]:5d} | rest of the stores {c[54] - c[55]:5d} | whole step {c[54] - c[0]}")


for it in range(60):
    env.step(acts[it % 8])
    if it % 20 == 19:
        show("host boundary  ")
for it in range(60):
    env.step_device(dev[it % 8].data_ptr())
    if it % 20 == 19:
        env.sync()
        show("device-resident")
import time
for it in range(60):            # device-resident inputs and outputs, but one step at a time with a host round trip in between, like the boundary
    env.step_device(dev[it % 8].data_ptr())
    env.sync()
    time.sleep(20e-6)
    if it % 20 == 19:
        show("device, synced ")
for it in range(60):            # the boundary's launch (actions from mapped host memory, second copy of the outputs to it) back to back
    cur = env._cur = env._cur ^ 1
    env.lib.dll.ac_step_host_async(env._h, cur)
    if it % 20 == 19:
        env.lib.dll.ac_step_host_wait(env._h)
        show("host, unsynced ")
env.lib.dll.ac_step_host_wait(env._h)
env.close()
