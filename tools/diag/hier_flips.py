#!/usr/bin/env python3
"""Diagnostic: where the low-level controller's argmax indices differ from the oracle's in the hierarchical 1v1 missile tasks."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aircombat_selfplay_amd as pkg
from oracle import oracle
from parity_util import TASK_FIELDS
task = sys.argv[1] if len(sys.argv) > 1 else "hierarchical_singlecombat_dodge_missile"
cfg = pkg.default_config(task, hierarchical=True)
cfg.init[1].lon_deg, cfg.init[1].lat_geod_deg, cfg.init[1].psi_deg = 120.02, 60.06, 171.0
cfg.init[0].psi_deg = 9.0
A, E = 2, 6
env = pkg.HipVecEnv(cfg, E, seed=5)
ref = oracle.OracleVecEnv(oracle.config_from_ac(cfg), E, chaff_seed=5)
obs = env.reset(); robs = ref.reset()
names = env.lib.state_field_names()
ix = {nm: k for k, nm in enumerate(names) if nm}
fdm_fields = np.array([k for k, nm in enumerate(names) if nm and not nm.startswith("x_") and nm not in TASK_FIELDS])
rng = np.random.default_rng(23)
hi = np.stack([rng.integers(0, n, size=(E, A)) for n in (3, 5, 3)], axis=-1).astype(np.float32)
prev_obs, prev_robs = obs.copy(), robs.copy()
for step in range(120):
    if step % 7 == 0:
        hi = np.stack([rng.integers(0, n, size=(E, A)) for n in (3, 5, 3)], axis=-1).astype(np.float32)
    act = hi if env.act_dim == 3 else np.concatenate([hi, (rng.random((E, A, env.act_dim - 3)) < 0.3).astype(np.float32)], axis=-1)
    for e in range(E):
        for a in range(A):
            v = env.get_state(e, a); v[fdm_fields] = ref.envs[e].export_state(a)[fdm_fields]; env.set_state(e, a, v)
            env.set_controller_state(e, a, ref.envs[e].get_rnn(a)[0])
    obs, rew, done, info = env.step(act)
    robs, rrew, rdone, rinfo = ref.step(act)
    for e in range(E):
        for a in range(A):
            hid, low = env.get_controller_state(e, a)
            rh, rlow = ref.envs[e].get_rnn(a)
            if (low[:4].astype(int) != rlow).any():
                st = int(env.get_state(e, a)[ix["status"]])
                print(f"step {step} env {e} agent {a} status {st} low {low[:4].astype(int).tolist()} orc {rlow.tolist()} | input obs diff {np.abs(prev_obs[e, a, :9] - prev_robs[e, a, :9]).max():.2e} "
                      f"dev obs9 {np.round(prev_obs[e, a, :9], 4).tolist()} orc {np.round(prev_robs[e, a, :9], 4).tolist()} reset {int(rinfo[e][3])}")
    prev_obs, prev_robs = obs.copy(), robs.copy()
