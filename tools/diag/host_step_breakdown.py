#!/usr/bin/env python3
"""Where a VecEnv.step(numpy) of the headline config (or: <envs> <task> <per_side>) spends its time on the host: the action copy into the mapped buffer, the launch
call, the wait. (time.perf_counter_ns around the three pieces; ~0.1 us of timer overhead each.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aircombat_selfplay_amd as pkg

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
task = sys.argv[2] if len(sys.argv) > 2 else "singlecombat"
per_side = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = pkg.default_nvn_config(per_side, task=task) if per_side > 1 else pkg.default_config(task)
env = (pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv)(cfg, E, seed=1)
env.reset()
rng = np.random.default_rng(0)
A = env.num_agents
acts = [np.concatenate([np.stack([rng.integers(0, n, size=(E, A)) for n in (41, 41, 41, 30)], axis=-1),
                        rng.random((E, A, env.act_dim - 4)) < 0.05], axis=-1).astype(np.float32) for _ in range(16)]
dll = env.lib.dll
h = env._h
now = time.perf_counter_ns
for label, do_copy in (("copy + launch + wait", True), ("launch + wait (actions left as they are)", False)):
    tc = tl = tw = 0
    K = 3000
    for it in range(K + 300):
        if it == 300:
            tc = tl = tw = 0
            T0 = now()
        cur = env._cur = env._cur ^ 1
        t0 = now()
        if do_copy:
            np.copyto(env._sets[cur]["actions"], acts[it & 15])
        t1 = now()
        dll.ac_step_host_async(h, cur)
        t2 = now()
        dll.ac_step_host_wait(h)
        t3 = now()
        tc += t1 - t0; tl += t2 - t1; tw += t3 - t2
    T1 = now()
    print(f"{label}: copy {tc / K / 1e3:.2f} us  launch call {tl / K / 1e3:.2f} us  wait {tw / K / 1e3:.2f} us   loop total {(T1 - T0) / K / 1e3:.2f} us/step")
# the same through VecEnv.step
t0 = now()
for it in range(3000):
    env.step(acts[it & 15])
print(f"VecEnv.step: {(now() - t0) / 3000 / 1e3:.2f} us/step")
env.close()
