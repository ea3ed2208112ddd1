#!/usr/bin/env python3
"""What handing the caller arrays of its own costs per VecEnv.step at the headline batch (4096 envs x 2): the step's outputs sit in
page-locked host memory the GPU has just written (cache-cold for the CPU); a fresh numpy array above 128 KB is an mmap + a page fault
per 4 KB. Variants: views (copy=False), three fresh .copy()s, np.copyto into arrays allocated once (no allocation, same memcpy),
the copies alone without a step in between (cache-warm source)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aircombat_selfplay_amd as pkg

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = pkg.default_config("singlecombat")
env = pkg.HipVecEnv(cfg, E, seed=1, copy=False)
env.reset()
rng = np.random.default_rng(0)
acts = [np.stack([rng.integers(0, n, size=(E, 2)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32) for _ in range(16)]
now = time.perf_counter_ns
K = 3000
pool = [[np.empty_like(env._sets[0][k]) for k in ("obs", "rew", "done")] for _ in range(4)]


def run(label, after):
    for it in range(K + 300):
        if it == 300:
            t0 = now()
        res = env.step(acts[it & 15])
        after(it, res)
    dt = (now() - t0) / K / 1e3
    print(f"{label:60s} {dt:7.2f} us/step  {E * 2 / dt:7.1f} M agent-steps/s")


run("views of the step's buffer set (copy=False)", lambda it, r: None)
run("three fresh .copy()s", lambda it, r: (r[0].copy(), r[1].copy(), r[2].copy()))
run("np.copyto into arrays allocated once (ring of 4)", lambda it, r: [np.copyto(d, s) for d, s in zip(pool[it & 3], r[:3])])
keep = []
run("fresh copies, the previous step's kept alive (runner pattern)", lambda it, r: keep.__setitem__(slice(None), [(r[0].copy(), r[1].copy(), r[2].copy())]))
r = env.step(acts[0])
t0 = now()
for it in range(K):
    r[0].copy(), r[1].copy(), r[2].copy()
print(f"{'three fresh .copy()s of a warm source, no step':60s} {(now() - t0) / K / 1e3:7.2f} us")
t0 = now()
for it in range(K):
    [np.copyto(d, s) for d, s in zip(pool[it & 3], r[:3])]
print(f"{'np.copyto of a warm source into the ring, no step':60s} {(now() - t0) / K / 1e3:7.2f} us")
env.close()
