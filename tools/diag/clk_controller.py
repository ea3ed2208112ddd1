#!/usr/bin/env python3
"""Cycle stamps of wave 0 of workgroup 0 of the low-level controller kernel (variants/libclk.so from tools/build_clk_variant.sh)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["AIRCOMBAT_HIP_LIB"] = os.path.join(ROOT, "variants", "libclk.so")
import torch
torch.cuda.init()          # (before the first handle: torch ships its own copy of the HIP runtime)
import aircombat_selfplay_amd as pkg

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = pkg.HipVecEnv(pkg.default_config("hierarchical_singlecombat", hierarchical=True), E, seed=1)
env.reset()
rng = np.random.default_rng(0)
fn = env.lib.dll.ac_debug_clocks
fn.argtypes = [ctypes.c_void_p]
names = ["stage", "layer 1", "LayerNorm 1", "layer 2", "LayerNorm 2", "GRU products", "GRU gates", "state store", "LayerNorm 3", "heads", "fifth-tile sums", "argmax + out"]
dev = [torch.from_numpy(np.stack([rng.integers(0, n, size=(E, 2)) for n in (3, 5, 3)], axis=-1).astype(np.float32)).cuda() for _ in range(4)]
DEVICE = os.environ.get("AC_DEVICE_RESIDENT", "1") == "1"     # 0: through VecEnv.step (the [3,5,3] rows are read from mapped host memory)
print("device-resident steps" if DEVICE else "host-boundary steps")
for it in range(80):
    if DEVICE:
        env.step_device(dev[it % 4].data_ptr()); env.sync()
    else:
        env.step(np.stack([rng.integers(0, n, size=(E, 2)) for n in (3, 5, 3)], axis=-1).astype(np.float32))
    if it % 40 == 39:
        clk = np.zeros(256, dtype=np.uint64)
        fn(clk.ctypes.data)
        c = clk.astype(np.int64)
        print(f"---- step {it + 1}: total {c[212] - c[200]}")
        print("  " + "  ".join(f"{nm} {c[201 + i] - c[200 + i]}" for i, nm in enumerate(names)))
env.close()
