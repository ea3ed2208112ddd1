#!/usr/bin/env python3
"""The controller kernel's two forms side by side in ONE process on one device (interleaved rounds: cdna_hip_programming.md rule 24):
controller8_kernel with 32 and with 64 aircraft per workgroup (AIRCOMBAT_CTL_ROWS; the four-wave kernel of rounds 2-3 it was first measured
against here -- 22.7 / 39.2 / 74.5 us -- is gone), at the three
batches BASELINE's as-shipped configs call it with: 8192 aircraft (C3 scenario1), 16 384 (C4 2v2), 32 768 (C5 4v4). HIP events around the
controller kernel and the step kernel of every device-resident step (ac_step_timed_device)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aircombat_selfplay_amd as pkg

E = 4096
ROUNDS, STEPS = 5, 200
cases = [("scenario1", 1), ("scenario_nvn", 2), ("scenario_nvn", 4)]
FORMS = (("1", "32 rows per workgroup", "32"), ("1", "64 rows per workgroup", "64"))
rng = np.random.default_rng(0)
for task, per_side in cases:
    envs = {}
    for form, name, rows in FORMS:
        os.environ["AIRCOMBAT_CTL_ROWS"] = rows
        form = name
        cfg = pkg.default_config(task, hierarchical=True) if per_side == 1 else pkg.default_nvn_config(per_side, task=task, hierarchical=True)
        cls = pkg.HipShareVecEnv if cfg.n_agents > 2 else pkg.HipVecEnv
        envs[form] = cls(cfg, E, seed=1, copy=False)
        envs[form].reset()
    first = FORMS[0][1]
    A = envs[first].num_agents
    pool = []
    for _ in range(8):
        a = np.stack([rng.integers(0, n, size=(E, A)) for n in (3, 5, 3)], axis=-1).astype(np.float32)
        a = np.concatenate([a, (rng.random((E, A, envs[first].act_dim - 3)) < 0.05).astype(np.float32)], axis=-1)
        pool.append(torch.from_numpy(a).cuda())
    ptrs = [t.data_ptr() for t in pool]
    res = {nm: [] for _, nm, _ in FORMS}
    ctl, stp = C.c_float(), C.c_float()
    for r in range(ROUNDS + 1):
        for _, form, _ in FORMS:
            env = envs[form]
            tc = ts = 0.0
            for i in range(STEPS):
                env.lib.check(env.lib.ac_step_timed_device(env._h, ptrs[i % 8], C.byref(ctl), C.byref(stp)), "ac_step_timed_device")
                tc += ctl.value; ts += stp.value
            if r:
                res[form].append((tc / STEPS * 1e3, ts / STEPS * 1e3))
    for _, form, _ in FORMS:
        name = form
        c = sorted(x[0] for x in res[form]); s = sorted(x[1] for x in res[form])
        print(f"{task} x{per_side} ({E * A} aircraft) {name:18s}: controller median {c[len(c) // 2]:6.2f} us (min {c[0]:6.2f})   step kernel median {s[len(s) // 2]:6.2f} us")
    # the two forms must drive the same episode: same observations after the same actions (argmax near-ties aside)
    o0, o1 = envs[FORMS[0][1]].device_tensors()[1].cpu().numpy(), envs[FORMS[-1][1]].device_tensors()[1].cpu().numpy()
    print(f"    observations of the two handles after {(ROUNDS + 1) * STEPS} steps: {100.0 * np.mean(np.all(o0 == o1, axis=-1)):.2f} % of the aircraft rows identical")
    for env in envs.values():
        env.close()
