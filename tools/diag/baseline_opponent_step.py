#!/usr/bin/env python3
"""Device-resident steps of a hierarchical task against the scripted opponent (use_baseline = 1 PursueAgent / 2 ManeuverAgent): run
under rocprofv3 --kernel-trace --stats to see scripted_inputs_kernel + controller + step kernel per step."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import aircombat_selfplay_amd as pkg

task = sys.argv[1] if len(sys.argv) > 1 else "scenario1"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = pkg.default_config(task, hierarchical=True)
cfg.use_baseline = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = pkg.HipVecEnv(cfg, E, seed=1)
env.reset()
rng = np.random.default_rng(0)
A = env.num_agents
acts = [torch.from_numpy(np.concatenate([np.stack([rng.integers(0, n, size=(E, A)) for n in (3, 5, 3)], axis=-1),
                                         (rng.random((E, A, env.act_dim - 3)) < 0.05)], axis=-1).astype(np.float32)).cuda() for _ in range(16)]
torch.cuda.synchronize()
for i in range(50):
    env.step_device(acts[i & 15].data_ptr())
env.sync()
t0 = time.perf_counter()
K = 400
for i in range(K):
    env.step_device(acts[i & 15].data_ptr())
env.sync()
print(f"{task} hierarchical, use_baseline={cfg.use_baseline}, {E} envs: {(time.perf_counter() - t0) / K * 1e6:.2f} us per step (device-resident)")
env.close()
