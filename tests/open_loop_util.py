"""Free-flight comparison harness shared by tests/test_gpu_open_loop.py and tools/diag/open_loop.py: the HIP step and the float64
oracle from the same initial conditions under the same actions, with NO state injection, and the per-step differences north_star
names (position, attitude, velocity, reward).

Every kernel form a BASELINE config launches flies through here: the task picks the kernel family, the environment variables
AIRCOMBAT_SPLIT / AIRCOMBAT_QUAD (set by the caller BEFORE the pair is built: ac_create reads them per handle) pin the form.
`substeps` = agent_interaction_steps: 6 is every shipped YAML's; 1 makes an env step ONE FDM tick, so that the discrete decisions of
the flight control system are compared after every tick instead of after every sixth."""
import ctypes as C

import numpy as np

KTSTOFPS = 1.68781

# (psi A, psi B, h_sl ft A, h_sl ft B, u fps A, u fps B): eight starts around the shipped one (WVR_selfplay.yaml:15-40: 20 000 ft, 800 fps, head-on)
STARTS = ((0.0, 180.0, 20000.0, 20000.0, 800.0, 800.0), (35.0, 200.0, 24000.0, 18000.0, 700.0, 900.0), (90.0, 270.0, 16000.0, 26000.0, 950.0, 650.0),
          (310.0, 140.0, 28000.0, 22000.0, 600.0, 1000.0), (180.0, 0.0, 19000.0, 21000.0, 850.0, 750.0), (225.0, 45.0, 30000.0, 15000.0, 1000.0, 600.0),
          (10.0, 170.0, 22000.0, 22500.0, 780.0, 820.0), (270.0, 100.0, 17500.0, 27000.0, 900.0, 700.0))

# task -> (config factory arguments, straight-and-level action row): the reference's straight-fly control indices [20, 19, 20, 0]
# (model/baseline.py:168), every weapon / shoot bit 0 (nothing is ever launched: what is compared is the FLIGHT of these kernel forms)
STRAIGHT = {"singlecombat": [20, 19, 20, 0], "singlecombat_shoot": [20, 19, 20, 0, 0], "multiplecombat": [20, 19, 20, 0],
            "scenario_nvn": [20, 19, 20, 0, 0, 0, 0, 0], "scenario1": [20, 19, 20, 0, 0, 0, 0, 0]}


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def make_config(pkg, task, per_side, start, substeps):
    if task in ("multiplecombat", "scenario_nvn"):
        cfg = pkg.default_nvn_config(per_side, task=task)
    else:
        cfg = pkg.default_config(task)
    A = cfg.n_agents
    for a in range(A):
        side, k = (0, a) if a < A // 2 else (1, a - A // 2)
        cfg.init[a].psi_deg, cfg.init[a].h_sl_ft, cfg.init[a].u_fps = start[side] + 3.0 * k, start[2 + side] + 400.0 * k, start[4 + side] + 15.0 * k
        if A > 2:   # the shipped NvN YAMLs put both teams on one meridian, exactly head-on: stagger them like the parity suite does
            cfg.init[a].lon_deg += 0.013 * (a % 3) + (0.03 if side else 0.0)
    cfg.agent_interaction_steps = substeps
    cfg.max_steps = 9000 * 6 // substeps
    return cfg


class OpenLoopPair:
    """len(STARTS) handles x (E / len(STARTS)) envs of `task`, each next to its oracle twin."""

    def __init__(self, pkg, oracle, E, spread=True, task="singlecombat", per_side=1, substeps=6, n_starts=None):
        self.L = oracle.lib()
        self.L.f16_vcas_from_qc.restype = C.c_double
        self.L.f16_vcas_from_qc.argtypes = [C.c_double]
        starts = STARTS if spread else STARTS[:1]
        if n_starts is not None:
            starts = starts[:n_starts]
        per = max(1, E // len(starts))
        self.parts = []
        self.task, self.substeps = task, substeps
        for s in starts:
            cfg = make_config(pkg, task, per_side, s, substeps)
            env = pkg.HipVecEnv(cfg, per, seed=77)
            ocfg = oracle.config_from_ac(cfg)
            ref = oracle.OracleVecEnv(ocfg, per, chaff_seed=77)
            obs, robs = env.reset(), ref.reset()
            assert obs.shape == robs.shape, (obs.shape, robs.shape)
            assert np.abs(obs - robs).max() < 2e-3, np.abs(obs - robs).max()
            self.parts.append((env, ref, per))
        self.A = self.parts[0][0].num_agents
        self.altitude_limit = float(cfg.altitude_limit)
        self.E = per * len(starts)
        names = self.parts[0][0].lib.state_field_names()
        self.ix = {nm: k for k, nm in enumerate(names) if nm}
        self.k = 0
        self.horizon = np.full(self.E, 1 << 30, dtype=np.int64)     # first step at which a discrete decision differed (never: huge)
        self.reason = [""] * self.E
        self.done_mismatch = np.zeros(self.E, dtype=bool)
        self.unexplained = []                               # done mismatches that neither a differing decision nor a threshold explains
        self.last_reset = np.zeros(self.E, dtype=bool)      # envs whose episode ended (and restarted) in the last step()

    def straight_action(self):
        return np.tile(np.array(STRAIGHT[self.task], dtype=np.float32), (self.E, self.A, 1))

    def discrete(self, st):
        """The decisions the next tick's flight control system takes from this state (f16.xml:325-335,814-832: gear stays down, so the
        leading-edge flap is 0.262 rad above alpha 0.0873, else -0.0349 above Mach 0.9; trailing-edge flap 0.349 rad below 250 kt,
        -0.0349 above Mach 0.9), the turbine's phase word and the aircraft status."""
        ix = self.ix
        alpha, mach, qc = st[ix["alpha"]], st[ix["mach"]], st[ix["qc"]]
        vc_kts = self.L.f16_vcas_from_qc(float(qc)) / KTSTOFPS
        lef = 2 if alpha > 0.0873 else (1 if mach > 0.9 else 0)
        tef = 2 if vc_kts < 250.0 else (1 if mach > 0.9 else 0)
        return ("lef", lef), ("tef", tef), ("engine", int(st[ix["eng"]])), ("status", int(st[ix["status"]]))

    def near_a_threshold(self, st, pose, k):
        """A done flag that differs with every EARLIER decision equal is legitimate only where a continuous quantity sits on its
        threshold: the side that did NOT terminate must be within a hair of LowAltitude's limit (the other side's altitude is
        within the position envelope of it and just across) or of Overload / ExtremeState's load factor 10. Anything else (a
        timeout, SafeReturn, an altitude a hundred metres off the limit) is a termination bug. `st`, `pose`: the state and pose of
        the aircraft on the side whose done flag is False."""
        ix = self.ix
        margin = 1.0 + 8.0 * (0.02 + 15.0 * (k * self.substeps / 3600.0) ** 3)
        if 0.0 <= pose[2] - self.altitude_limit <= margin:
            return "altitude"
        for nm, off in (("npx", 0.0), ("npy", 0.0), ("npz", 1.0)):
            if 0.0 <= 10.0 - abs(st[ix[nm]] + off) <= 0.1:
                return "load factor"
        return None

    def others_of(self, a):
        """The aircraft behind the relative-geometry blocks of agent a's observation, in block order: partners, then enemies."""
        A, h = self.A, self.A // 2
        team = range(0, h) if a < h else range(h, A)
        foes = range(h, A) if a < h else range(0, h)
        return [j for j in team if j != a] + list(foes)

    def side_flags_free(self, poses):
        """[A, A-1] bool: the side flag of a block is the sign of the HORIZONTAL cross product v_ego x (p_other - p_ego)
        (utils.py:58-83): undefined where the other aircraft is dead ahead or astern in plan view -- here to within the angle the
        free-flight position envelope subtends at that range (oracle poses: lon lat alt | rpy | v NED | NEU)."""
        pos_env = 0.02 + 15.0 * (self.k * self.substeps / 3600.0) ** 3
        free = np.zeros((self.A, self.A - 1), dtype=bool)
        for a in range(self.A):
            pa, va = poses[a][1][9:11], poses[a][1][6:8]
            for b, j in enumerate(self.others_of(a)):
                d = poses[j][1][9:11] - pa
                nd, nv = np.linalg.norm(d), np.linalg.norm(va)
                s2 = abs(va[0] * d[1] - va[1] * d[0]) / max(nd * nv, 1e-9)
                free[a, b] = s2 < 2e-3 + 4.0 * pos_env / max(nd, 100.0)
        return free

    def obs_difference(self, obs, robs, free):
        """max |d observation| per aircraft, with the two conditioning rules of the relative-geometry blocks [du, dh, AO, TA, R / 1e4,
        side] (tests/parity_util.py): the side flag where it is undefined (side_flags_free), and acos turning an fp32 rounding of its
        argument into 3e-7 / sin(angle)."""
        d = np.abs(obs - robs)
        for b in range(self.A - 1):
            o = 9 + 6 * b
            if o + 5 >= obs.shape[-1]:
                break
            d[..., o + 5] = np.where(free[..., b], 0.0, d[..., o + 5])
            for col in (o + 2, o + 3):
                d[..., col] = np.maximum(0.0, d[..., col] - 3e-7 / np.maximum(np.sin(robs[..., col]), 1e-4))
        return d.max(axis=-1)

    def step(self, act):
        """One env step of every handle and twin. Returns per-aircraft differences [E, A] and `live` [E]: envs still inside their
        horizon (no discrete decision has differed yet, dones agree)."""
        self.k += 1
        A = self.A
        out = {k: [] for k in ("pos_m", "att_rad", "vel_ms", "obs", "rew")}
        e0 = 0
        for env, ref, per in self.parts:
            a = act[e0:e0 + per]
            obs, rew, done, _ = env.step(a)
            robs, rrew, rdone, rinfo = ref.step(a)
            self.last_reset[e0:e0 + per] = rinfo[:, 3] != 0
            pos, att, vel = np.zeros((per, A)), np.zeros((per, A)), np.zeros((per, A))
            free = np.zeros((per, A, A - 1), dtype=bool)
            for e in range(per):
                g = e0 + e
                states = None
                if self.horizon[g] > self.k:
                    states = [(env.get_state(e, ag), ref.envs[e].export_state(ag)) for ag in range(A)]
                poses = [(env.get_entity(e, ag), ref.envs[e].pose(ag)) for ag in range(A)]
                free[e] = self.side_flags_free(poses)
                for ag in range(A):
                    ge, oe = poses[ag]
                    pos[e, ag] = np.linalg.norm(ge[9:12] - oe[9:12])
                    att[e, ag] = np.abs(wrap(ge[3:6] - oe[3:6])).max()
                    vel[e, ag] = np.linalg.norm(ge[6:9] - oe[6:9])
                if states is not None:                      # decisions first: a status that differs explains a done flag that differs
                    for ag in range(A):
                        dg, do = self.discrete(states[ag][0]), self.discrete(states[ag][1])
                        for (nm, x), (_, y) in zip(dg, do):
                            if x != y and self.horizon[g] > self.k:
                                self.horizon[g], self.reason[g] = self.k, nm
                if (done[e] != rdone[e]).any() and not self.done_mismatch[g]:
                    self.done_mismatch[g] = True
                    if self.horizon[g] >= self.k:           # no decision differed BEFORE this step: a threshold has to explain it
                        why = []
                        for ag in np.argwhere(done[e, :, 0] != rdone[e, :, 0])[:, 0]:
                            if done[e, ag, 0]:              # the device terminated this aircraft, the oracle flies on: look at the oracle's
                                why.append(self.near_a_threshold(ref.envs[e].export_state(ag), poses[ag][1], self.k))
                            else:
                                why.append(self.near_a_threshold(env.get_state(e, ag), poses[ag][0], self.k))
                        if any(w is None for w in why):
                            self.unexplained.append((g, self.k, done[e, :, 0].tolist(), rdone[e, :, 0].tolist()))
                        self.horizon[g], self.reason[g] = self.k, "done (" + ", ".join(w or "unexplained" for w in why) + ")"
            out["pos_m"].append(pos); out["att_rad"].append(att); out["vel_ms"].append(vel)
            out["obs"].append(self.obs_difference(obs, robs, free)); out["rew"].append(np.abs(rew - rrew)[..., 0])
            e0 += per
        res = {k: np.concatenate(v, axis=0) for k, v in out.items()}
        res["live"] = self.horizon > self.k
        return res

    def reason_counts(self):
        c = {}
        for r in self.reason:
            if r:
                c[r] = c.get(r, 0) + 1
        return c

    def close(self):
        for env, _, _ in self.parts:
            env.close()
