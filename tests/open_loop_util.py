"""Free-flight comparison harness shared by tests/test_gpu_open_loop.py and tools/diag/open_loop.py: the HIP step and the float64
oracle from the same initial conditions under the same actions, with NO state injection, and the per-step differences north_star
names (position, attitude, velocity, reward)."""
import ctypes as C

import numpy as np

KTSTOFPS = 1.68781

# (psi A, psi B, h_sl ft A, h_sl ft B, u fps A, u fps B): eight starts around the shipped one (WVR_selfplay.yaml:15-40: 20 000 ft, 800 fps, head-on)
STARTS = ((0.0, 180.0, 20000.0, 20000.0, 800.0, 800.0), (35.0, 200.0, 24000.0, 18000.0, 700.0, 900.0), (90.0, 270.0, 16000.0, 26000.0, 950.0, 650.0),
          (310.0, 140.0, 28000.0, 22000.0, 600.0, 1000.0), (180.0, 0.0, 19000.0, 21000.0, 850.0, 750.0), (225.0, 45.0, 30000.0, 15000.0, 1000.0, 600.0),
          (10.0, 170.0, 22000.0, 22500.0, 780.0, 820.0), (270.0, 100.0, 17500.0, 27000.0, 900.0, 700.0))


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


class OpenLoopPair:
    """len(STARTS) handles x (E / len(STARTS)) envs of SingleCombat (BASELINE C2's task), each next to its oracle twin."""

    def __init__(self, pkg, oracle, E, spread=True):
        self.L = oracle.lib()
        self.L.f16_vcas_from_qc.restype = C.c_double
        self.L.f16_vcas_from_qc.argtypes = [C.c_double]
        starts = STARTS if spread else STARTS[:1]
        per = E // len(starts)
        self.parts = []
        for s in starts:
            cfg = pkg.default_config("singlecombat")
            for a in range(2):
                cfg.init[a].psi_deg, cfg.init[a].h_sl_ft, cfg.init[a].u_fps = s[a], s[2 + a], s[4 + a]
            env = pkg.HipVecEnv(cfg, per)
            ref = oracle.OracleVecEnv(oracle.config_from_ac(cfg), per)
            obs, robs = env.reset(), ref.reset()
            assert np.abs(obs - robs).max() < 2e-3
            self.parts.append((env, ref, per))
        self.E = per * len(starts)
        names = self.parts[0][0].lib.state_field_names()
        self.ix = {nm: k for k, nm in enumerate(names) if nm}
        self.k = 0
        self.horizon = np.full(self.E, 1 << 30, dtype=np.int64)     # first step at which a discrete decision differed (never: huge)
        self.reason = [""] * self.E
        self.done_mismatch = np.zeros(self.E, dtype=bool)
        self.last_reset = np.zeros(self.E, dtype=bool)      # envs whose episode ended (and restarted) in the last step()

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

    def step(self, act):
        """One env step of every handle and twin. Returns per-aircraft differences [E, 2] and `live` [E]: envs still inside their
        horizon (no discrete decision has differed yet, dones agree)."""
        self.k += 1
        out = {k: [] for k in ("pos_m", "att_rad", "vel_ms", "obs", "rew")}
        e0 = 0
        for env, ref, per in self.parts:
            a = act[e0:e0 + per]
            obs, rew, done, _ = env.step(a)
            robs, rrew, rdone, rinfo = ref.step(a)
            self.last_reset[e0:e0 + per] = rinfo[:, 3] != 0
            pos, att, vel = np.zeros((per, 2)), np.zeros((per, 2)), np.zeros((per, 2))
            for e in range(per):
                g = e0 + e
                if (done[e] != rdone[e]).any() and not self.done_mismatch[g]:
                    self.done_mismatch[g] = True
                    if self.horizon[g] > self.k:
                        self.horizon[g], self.reason[g] = self.k, "done"
                for ag in range(2):
                    ge, oe = env.get_entity(e, ag), ref.envs[e].pose(ag)
                    pos[e, ag] = np.linalg.norm(ge[9:12] - oe[9:12])
                    att[e, ag] = np.abs(wrap(ge[3:6] - oe[3:6])).max()
                    vel[e, ag] = np.linalg.norm(ge[6:9] - oe[6:9])
                    if self.horizon[g] > self.k:
                        dg, do = self.discrete(env.get_state(e, ag)), self.discrete(ref.envs[e].export_state(ag))
                        for (nm, x), (_, y) in zip(dg, do):
                            if x != y:
                                self.horizon[g], self.reason[g] = self.k, nm
                                break
            out["pos_m"].append(pos); out["att_rad"].append(att); out["vel_ms"].append(vel)
            out["obs"].append(np.abs(obs - robs).max(axis=-1)); out["rew"].append(np.abs(rew - rrew)[..., 0])
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
