#!/usr/bin/env python3
"""Diagnostic: replay tests/test_gpu_parity.py::test_scenario_weapon_tasks_match_oracle[scenario_nvn-4-closing-0] and print, per step,
the device's munition slots next to the oracle's missile list for one env until they disagree (status / incoming-missile block)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import aircombat_selfplay_amd as pkg
from oracle import oracle
from parity_util import TASK_FIELDS, obs_bounds

per_side = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = pkg.default_nvn_config(per_side, task="scenario_nvn")
for i in range(2 * per_side):
    cfg.init[i].lon_deg += 0.013 * (i % 3) + (0.02 if i >= per_side else 0.0)
    cfg.init[i].psi_deg = (7.0 + 3.0 * i) if i < per_side else (171.0 + 2.0 * i)
    cfg.init[i].h_sl_ft += 300.0 * i
    if i >= per_side:
        cfg.init[i].lat_geod_deg = 60.06
A, E, seed = cfg.n_agents, 4, 1234
env = pkg.HipShareVecEnv(cfg, E, seed=seed)
ocfg = oracle.config_from_ac(cfg)
ocfg.task = oracle.TASK_SCENARIO_NVN
ref = oracle.OracleVecEnv(ocfg, E, chaff_seed=seed)
env.reset(); ref.reset()
names = env.lib.state_field_names()
fdm_fields = np.array([k for k, nm in enumerate(names) if nm and not nm.startswith("x_") and nm not in TASK_FIELDS])
rng = np.random.default_rng(11)
ST = {-1: "----", 0: "LNCH", 1: "HIT ", 2: "MISS"}
for step in range(steps):
    for e in range(E):
        for a in range(A):
            v = env.get_state(e, a)
            v[fdm_fields] = ref.envs[e].export_state(a)[fdm_fields]
            env.set_state(e, a, v)
    act = np.stack([rng.integers(0, n, size=(E, A)) for n in (41, 41, 41, 30)], axis=-1).astype(np.float32)
    act[:, :, :4] = np.array([20, 18.6, 20, 15], dtype=np.float32) + rng.integers(-2, 3, size=(E, A, 4))
    bits = (rng.random((E, A, 4)) < 0.6).astype(np.float32)
    act = np.concatenate([act, bits], axis=-1)
    res = env.step(act)
    obs, rew, done = res[0], res[2], res[3]
    robs, rrew, rdone, rinfo = ref.step(act)
    tol, free = obs_bounds(robs, 10.0)
    bad = (np.abs(obs - robs) > tol) & ~free
    for e in range(E):
        dev = {}
        for a in range(A):
            for k in range(2):
                m = env.get_missile(e, a, k)
                if m[0] >= 0:
                    dev[(a, 2 - k)] = m
        orc = {}
        for m in ref.envs[e].missiles():
            orc.setdefault((int(m[11]), None), []).append(m)
        line = f"step {step:3d} env {e}: device " + " ".join(f"{a}.{u}:{ST[int(m[0])]}t{m[9]:.2f}" for (a, u), m in sorted(dev.items()))
        line += " | oracle " + " ".join(f"{int(m[11])}->{int(m[12])}:{ST[int(m[0])]}t{m[9]:.2f}" for m in ref.envs[e].missiles())
        if bad[e].any() or (done[e] != rdone[e]).any():
            print(line)
            print("   MISMATCH obs idx", np.argwhere(bad[e])[:8].tolist(), "done", done[e, :, 0].astype(int), rdone[e, :, 0].astype(int))
            for (a, u), m in sorted(dev.items()):
                print(f"     dev {a}.{u} st {int(m[0])} pos {m[1]:.2f} {m[2]:.2f} {m[3]:.2f} t {m[9]:.4f}")
            for m in ref.envs[e].missiles():
                print(f"     orc {int(m[11])}->{int(m[12])} st {int(m[0])} pos {m[1]:.2f} {m[2]:.2f} {m[3]:.2f} t {m[9]:.4f}")
            g = [env.get_state(e, a) for a in range(A)]
            ix = {nm: k for k, nm in enumerate(names) if nm}
            print("     dev status", [int(x[ix['status']]) for x in g], "bloods", [round(float(x[ix['bloods']]), 1) for x in g])
            print("     orc status", [ref.envs[e].status(a) for a in range(A)])
            sys.exit(0)
        elif e == 0:
            print(line)
print("no mismatch")
