#!/usr/bin/env python3
"""Diagnostic: tests/test_gpu_baseline_builds.py::test_small_batch_both_kernel_forms_match_oracle[*-singlecombat_shoot-1] replayed
with the device's munition slots printed next to the oracle's missiles at the first observation mismatch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aircombat_selfplay_amd as pkg
from oracle import oracle
from parity_util import TASK_FIELDS, obs_bounds
from test_gpu_baseline_builds import make_cfg, actions_for
task = sys.argv[1] if len(sys.argv) > 1 else "singlecombat_shoot"
cfg = make_cfg(pkg, task, 1)
E, seed = 6, 77
env = pkg.HipVecEnv(cfg, E, seed=seed)
ref = oracle.OracleVecEnv(oracle.config_from_ac(cfg), E, chaff_seed=seed, env_ids=list(range(E)))
env.reset(); ref.reset()
names = env.lib.state_field_names()
ix = {nm: k for k, nm in enumerate(names) if nm}
fdm_fields = np.array([k for k, nm in enumerate(names) if nm and not nm.startswith("x_") and nm not in TASK_FIELDS])
rng = np.random.default_rng(seed)
for step in range(12):
    for e in range(E):
        for a in range(2):
            v = env.get_state(e, a); v[fdm_fields] = ref.envs[e].export_state(a)[fdm_fields]; env.set_state(e, a, v)
    act = actions_for(rng, E, 2, env.act_dim, gentle=True)
    obs, rew, done, info = env.step(act)
    robs, rrew, rdone, rinfo = ref.step(act)
    tol, free = obs_bounds(robs, 10.0)
    bad = (np.abs(obs - robs) > tol) & ~free
    print("step", step, "bits", act[:, :, 4].astype(int).tolist(), "bad", np.argwhere(bad).tolist())
    for e in sorted(set(np.argwhere(bad)[:, 0].tolist()))[:2]:
        print("  env", e, "obs dev", np.round(obs[e, :, 15:], 5).tolist(), "orc", np.round(robs[e, :, 15:], 5).tolist())
        for a in range(2):
            g = env.get_state(e, a)
            print("   agent", a, "remaining", g[ix["remaining"]], "last_missile", g[ix["last_missile"]], "shoot_action", g[ix["shoot_action"]], "oracle remaining", ref.envs[e].export_state(a)[ix["remaining"]])
            for k in range(4):
                m = env.get_missile(e, a, k)
                if m[0] >= 0:
                    print(f"   dev {a}.{k} st {int(m[0])} pos {m[1]:.3f} {m[2]:.3f} {m[3]:.3f} vel {m[4]:.3f} {m[5]:.3f} {m[6]:.3f} th {m[7]:.5f} psi {m[8]:.5f} t {m[9]:.3f}")
        for m in ref.envs[e].missiles():
            print(f"   orc {int(m[11])}->{int(m[12])} st {int(m[0])} pos {m[1]:.3f} {m[2]:.3f} {m[3]:.3f} vel {m[4]:.3f} {m[5]:.3f} {m[6]:.3f} th {m[7]:.5f} psi {m[8]:.5f} t {m[9]:.3f}")
    if bad.any():
        break
