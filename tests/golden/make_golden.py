#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own Python (runs only in the build container, where /root/reference exists).

The reference's pure-Python part of the step() path (geometry, rewards, terminations, missile engine, observation builders)
is imported from /root/reference and driven with duck-typed fake aircraft; inputs and the reference's outputs are written to
tests/golden/*.npz. The FDM (third-party ``jsbsim`` wheel) is not importable here, so nothing FDM-related is pinned.

Absent third-party modules are replaced by empty in-process stand-ins so that the imports resolve:
  jsbsim, colorama, wandb  – never called by the code exercised here
  gymnasium                – only ``spaces.Box/MultiDiscrete/Discrete/Tuple`` containers and ``seeding.np_random``
  pymap3d                  – ``geodetic2ned`` / ``ned2geodetic`` are supplied by the WGS84 closed forms below (our own
                             restatement of pymap3d's published algorithm; the missile fixtures therefore pin everything
                             except that geodesy call, which stays "parity unpinned")
Only data (inputs / expected outputs) is stored; no reference source text.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------- stand-ins
def _geodetic2ecef(lat, lon, alt):
    a, f = 6378137.0, 1 / 298.257223563
    b = a * (1 - f)
    lat, lon = np.radians(lat), np.radians(lon)
    N = a ** 2 / np.hypot(a * np.cos(lat), b * np.sin(lat))
    return (N + alt) * np.cos(lat) * np.cos(lon), (N + alt) * np.cos(lat) * np.sin(lon), (N * (b / a) ** 2 + alt) * np.sin(lat)


def _ecef2geodetic(x, y, z):
    a, f = 6378137.0, 1 / 298.257223563
    b = a * (1 - f)
    r = np.sqrt(x * x + y * y + z * z)
    E = np.sqrt(a * a - b * b)
    u = np.sqrt(0.5 * (r * r - E * E) + 0.5 * np.hypot(r * r - E * E, 2 * E * z))
    hxy = np.hypot(x, y)
    huE = np.hypot(u, E)
    Beta = np.arctan(huE / u * z / hxy)
    dBeta = ((b * u - a * huE + E * E) * np.sin(Beta)) / (a * huE / np.cos(Beta) - E * E * np.cos(Beta))
    Beta += dBeta
    lat = np.arctan(a / b * np.tan(Beta))
    lon = np.arctan2(y, x)
    alt = np.hypot(z - b * np.sin(Beta), hxy - a * np.cos(Beta))
    if x * x / a ** 2 + y * y / a ** 2 + z * z / b ** 2 < 1:
        alt = -alt
    return np.degrees(lat), np.degrees(lon), alt


def geodetic2ned(lat, lon, h, lat0, lon0, h0):
    x, y, z = _geodetic2ecef(lat, lon, h)
    x0, y0, z0 = _geodetic2ecef(lat0, lon0, h0)
    dx, dy, dz = x - x0, y - y0, z - z0
    la, lo = np.radians(lat0), np.radians(lon0)
    t = np.cos(lo) * dx + np.sin(lo) * dy
    e = -np.sin(lo) * dx + np.cos(lo) * dy
    u = np.cos(la) * t + np.sin(la) * dz
    n = -np.sin(la) * t + np.cos(la) * dz
    return n, e, -u


def ned2geodetic(n, e, d, lat0, lon0, h0):
    x0, y0, z0 = _geodetic2ecef(lat0, lon0, h0)
    la, lo = np.radians(lat0), np.radians(lon0)
    u = -d
    t = np.cos(la) * u - np.sin(la) * n
    dz = np.sin(la) * u + np.cos(la) * n
    dx = np.cos(lo) * t - np.sin(lo) * e
    dy = np.sin(lo) * t + np.cos(lo) * e
    return _ecef2geodetic(x0 + dx, y0 + dy, z0 + dz)


def install_standins():
    for name in ("jsbsim", "colorama", "wandb"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["colorama"].Fore = type("Fore", (), {"LIGHTRED_EX": "", "LIGHTGREEN_EX": "", "LIGHTYELLOW_EX": ""})
    sys.modules["wandb"].agent = None
    pm = types.ModuleType("pymap3d")
    pm.geodetic2ned, pm.ned2geodetic = geodetic2ned, ned2geodetic
    sys.modules["pymap3d"] = pm
    g = types.ModuleType("gymnasium")
    sp = types.ModuleType("gymnasium.spaces")
    ut = types.ModuleType("gymnasium.utils")
    sd = types.ModuleType("gymnasium.utils.seeding")

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape = np.full(shape, low), np.full(shape, high), shape

    class Discrete:
        def __init__(self, n):
            self.n = n

    class MultiDiscrete:
        def __init__(self, nvec):
            self.nvec = np.array(nvec)
            self.shape = self.nvec.shape

    class MultiBinary:
        def __init__(self, n):
            self.n, self.shape = n, (n,)

    class Tuple(tuple):
        def __new__(cls, xs):
            return tuple.__new__(cls, xs)

    sp.Box, sp.Discrete, sp.MultiDiscrete, sp.MultiBinary, sp.Tuple, sp.Space = Box, Discrete, MultiDiscrete, MultiBinary, Tuple, object
    g.spaces, g.Space, g.Env = sp, object, object
    sd.np_random = lambda seed=None: (np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))), seed)
    ut.seeding, g.utils = sd, ut
    sys.modules.update({"gymnasium": g, "gymnasium.spaces": sp, "gymnasium.utils": ut, "gymnasium.utils.seeding": sd})
    sys.path.insert(0, REF)


# ----------------------------------------------------------------------------------------------- fakes
class FakeAircraft:
    """Duck-typed AircraftSimulator: the getters and flags the reference's task / reward / termination code reads."""
    ALIVE, CRASH, SHOTDOWN = 0, 1, 2

    def __init__(self, uid, color="Red", dt=1 / 60, origin=(120.0, 60.0, 0.0), num_missiles=2):
        self.uid, self.color, self.dt = uid, color, dt
        self.lon0, self.lat0, self.alt0 = origin
        self.status = 0
        self.bloods = 100
        self.num_missiles = num_missiles
        self.partners, self.enemies, self.launch_missiles, self.under_missiles = [], [], [], []
        self.props = {}
        self._geodetic, self._position, self._posture, self._velocity = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)

    is_alive = property(lambda s: s.status == 0)
    is_crash = property(lambda s: s.status == 1)
    is_shotdown = property(lambda s: s.status == 2)

    def crash(self):
        self.status = 1

    def shotdown(self):
        self.status = 2

    def get_geodetic(self):
        return self._geodetic

    def get_position(self):
        return self._position

    def get_rpy(self):
        return self._posture

    def get_velocity(self):
        return self._velocity

    def check_missile_warning(self):
        for m in self.under_missiles:
            if m.is_alive:
                return m
        return None

    def get_property_value(self, prop):
        return self.props[prop.name_jsbsim]

    def get_property_values(self, props):
        return [self.get_property_value(p) for p in props]

    def set_property_value(self, prop, value):
        self.props[prop.name_jsbsim] = min(max(value, prop.min), prop.max)

    def set_pose(self, lon, lat, alt_m, rpy, v_ned_mps, uvw_mps=(250.0, 0.0, 0.0), vc=250.0, npilot=(0.0, 0.0, -1.0), sim_time=20.0):
        """Everything _update_properties caches plus the catalogue values get_obs / terminations read."""
        from envs.JSBSim.utils.utils import LLA2NEU
        self._geodetic[:] = (lon, lat, alt_m)
        self._position[:] = LLA2NEU(lon, lat, alt_m, self.lon0, self.lat0, self.alt0)
        self._posture[:] = rpy
        self._velocity[:] = v_ned_mps
        p = self.props
        p["position/long-gc-deg"], p["position/lat-geod-deg"], p["position/h-sl-m"] = lon, lat, alt_m
        p["attitude/roll-rad"], p["attitude/pitch-rad"], p["attitude/heading-true-rad"] = rpy
        p["velocities/v-north-mps"], p["velocities/v-east-mps"], p["velocities/v-down-mps"] = v_ned_mps
        p["velocities/u-mps"], p["velocities/v-mps"], p["velocities/w-mps"] = uvw_mps
        p["velocities/vc-mps"] = vc
        p["accelerations/n-pilot-x-norm"], p["accelerations/n-pilot-y-norm"], p["accelerations/n-pilot-z-norm"] = npilot
        p["simulation/sim-time-sec"] = sim_time
        p["detect/extreme-state"] = 0


class FakeEnv:
    def __init__(self, agents, center=(120.0, 60.0, 0.0), max_steps=9000):
        self.agents = {a.uid: a for a in agents}
        self.center_lon, self.center_lat, self.center_alt = center
        self.current_step = 0
        self.time_interval = 0.1
        self._tempsims = {}
        self.ego_ids = [a.uid for a in agents if a.uid[0] == agents[0].uid[0]]
        self.enm_ids = [a.uid for a in agents if a.uid[0] != agents[0].uid[0]]

    def add_temp_simulator(self, sim):
        self._tempsims[sim.uid] = sim
        return sim


def link(agents):
    for a in agents:
        for b in agents:
            if a is b:
                continue
            (a.partners if a.uid[0] == b.uid[0] else a.enemies).append(b)


def make_config(**kw):
    base = dict(max_steps=9000, altitude_limit=2500, acceleration_limit_x=10.0, acceleration_limit_y=10.0, acceleration_limit_z=10.0,
                PostureReward_scale=15.0, PostureReward_potential=True, PostureReward_orientation_version="v2",
                PostureReward_range_version="v3", AltitudeReward_safe_altitude=4.0, AltitudeReward_danger_altitude=3.5,
                AltitudeReward_Kv=0.2, EventDrivenReward_scale=1, EventDrivenReward_potential=True, MissilePostureReward_scale=30,
                max_attack_angle=45, max_attack_distance=14000, min_attack_interval=25,
                aircraft_configs={"A0100": {"color": "Blue", "missile": 2}, "B0100": {"color": "Red", "missile": 2}})
    base.update(kw)
    return type("EnvConfig", (object,), base)


def random_pose(rng, ac, spread_km=30.0, alt=(2000.0, 9000.0)):
    lon = 120.0 + rng.uniform(-1, 1) * spread_km / 55.0
    lat = 60.0 + rng.uniform(-1, 1) * spread_km / 111.0
    alt_m = rng.uniform(*alt)
    rpy = (rng.uniform(-1.2, 1.2), rng.uniform(-0.6, 0.6), rng.uniform(0, 2 * np.pi))
    sp = rng.uniform(120, 420)
    hdg = rng.uniform(0, 2 * np.pi)
    vd = rng.uniform(-80, 80)
    vned = (sp * np.cos(hdg), sp * np.sin(hdg), vd)
    uvw = (sp * rng.uniform(0.9, 1.0), rng.uniform(-15, 15), rng.uniform(-30, 40))
    npil = (rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-9, 3))
    ac.set_pose(lon, lat, alt_m, rpy, vned, uvw, vc=rng.uniform(100, 400), npilot=npil, sim_time=rng.uniform(0, 40))


def pose_vector(ac):
    p = ac.props
    return np.array([ac._geodetic[0], ac._geodetic[1], ac._geodetic[2], *ac._posture, *ac._velocity,
                     p["velocities/u-mps"], p["velocities/v-mps"], p["velocities/w-mps"], p["velocities/vc-mps"],
                     p["accelerations/n-pilot-x-norm"], p["accelerations/n-pilot-y-norm"], p["accelerations/n-pilot-z-norm"],
                     p["simulation/sim-time-sec"], ac.status, ac.bloods, p["detect/extreme-state"]])


# ----------------------------------------------------------------------------------------------- generators
def gen_geometry(rng):
    from envs.JSBSim.utils.utils import get_AO_TA_R, get2d_AO_TA_R, in_range_deg, in_range_rad
    n = 1200
    ego = rng.normal(size=(n, 6)) * np.array([20000, 20000, 3000, 250, 250, 60])
    enm = rng.normal(size=(n, 6)) * np.array([20000, 20000, 3000, 250, 250, 60])
    enm[:20] = ego[:20] + rng.normal(size=(20, 6)) * 1e-9      # R -> 0
    ego[20:40, 3:] = 0.0                                           # |v_ego| = 0
    enm[40:60, 3:] = 0.0
    enm[60:80, :3] = ego[60:80, :3] + ego[60:80, 3:] * 40.0      # dead ahead: AO = 0
    enm[80:100, :3] = ego[80:100, :3] - ego[80:100, 3:] * 40.0   # dead astern: AO = pi
    with np.errstate(all="ignore"):
        out3 = np.array([get_AO_TA_R(a, b, True) for a, b in zip(ego, enm)], dtype=float)
        out2 = np.array([get2d_AO_TA_R(a, b, True) for a, b in zip(ego, enm)], dtype=float)
    ang = rng.uniform(-2000, 2000, size=400)
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), ego=ego, enm=enm, out3d=out3, out2d=out2, ang=ang,
             in_range_deg=np.array([in_range_deg(a) for a in ang]), in_range_rad=np.array([in_range_rad(a) for a in ang]))


def gen_reward_functions(rng):
    """PostureReward orientation/range functions and AltitudeReward on grids (pure functions of their inputs)."""
    from envs.JSBSim.reward_functions import PostureReward, AltitudeReward
    cfg = make_config()
    pr = PostureReward(cfg)
    AO = rng.uniform(0, np.pi, 600)
    TA = rng.uniform(0, np.pi, 600)
    R = np.concatenate([rng.uniform(0, 12, 300), rng.uniform(0, 80, 300)])
    TA[:5] = 0.0
    TA[5:10] = np.pi / 2
    with np.errstate(all="ignore"):
        orn = np.array([pr.orientation_fn(a, t) for a, t in zip(AO, TA)])
        rngv = np.array([pr.range_fn(r) for r in R])
    ar = AltitudeReward(cfg)
    z = rng.uniform(0.5, 8, 600)
    vz = rng.uniform(-0.6, 0.6, 600)
    alt = []
    for zz, vv in zip(z, vz):
        a = FakeAircraft("A0100")
        a._position[:] = (0, 0, zz * 1000)
        a._velocity[:] = (0, 0, vv * 340)
        alt.append(ar.get_reward(None, FakeEnv([a]), "A0100"))
    np.savez_compressed(os.path.join(OUT, "reward_functions.npz"), AO=AO, TA=TA, R=R, orientation=orn, range=rngv, z=z, vz=vz, altitude=np.array(alt))


def gen_singlecombat_sequences(rng):
    """SingleCombatTask over scripted two-aircraft pose sequences: obs (15), terminations, rewards incl. reset seeding,
    potential differencing, the die-flag latch and the sequential termination order."""
    from envs.JSBSim.tasks.singlecombat_task import SingleCombatTask
    cfg = make_config()
    episodes = []
    for ep in range(24):
        task = SingleCombatTask(cfg)
        a, b = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
        link([a, b])
        env = FakeEnv([a, b])
        random_pose(rng, a); random_pose(rng, b)
        if ep % 4 == 1:
            random_pose(rng, b, spread_km=3.0)   # close-in geometry
        task.reset(env)
        frames = [dict(step=0, pose=np.stack([pose_vector(a), pose_vector(b)]), obs=np.stack([task.get_obs(env, u) for u in env.agents]),
                       rew=np.zeros(2), done=np.zeros(2))]
        T = 14
        for t in range(1, T + 1):
            env.current_step = t
            for ac in (a, b):
                if ac.is_alive:
                    random_pose(rng, ac, alt=(2000.0, 9000.0) if ep % 3 else (2300.0, 5000.0))
            # scripted events
            if ep % 6 == 2 and t == 6:
                b.shotdown()
            if ep % 6 == 3 and t == 5:
                a.props["accelerations/n-pilot-z-norm"] = -12.5
                a.props["simulation/sim-time-sec"] = 30.0
            if ep % 6 == 4 and t == 7:
                b.props["detect/extreme-state"] = 1
            if ep % 6 == 5 and t == 4:
                env.current_step = 9000
            pose = np.stack([pose_vector(a), pose_vector(b)])
            obs = np.stack([task.get_obs(env, u) for u in env.agents])
            info = {"current_step": env.current_step}
            done = []
            for u in env.agents:
                d, info = task.get_termination(env, u, info)
                done.append(d)
            rew = []
            for u in env.agents:
                r, info = task.get_reward(env, u, info)
                rew.append(r)
            frames.append(dict(step=env.current_step, pose=pose, obs=obs, rew=np.array(rew, dtype=float), done=np.array(done, dtype=float),
                               status_after=np.array([a.status, b.status], dtype=float)))
            if all(done):
                break
        episodes.append(frames)
    flat = {}
    for i, frames in enumerate(episodes):
        flat[f"ep{i}_pose"] = np.stack([f["pose"] for f in frames])
        flat[f"ep{i}_obs"] = np.stack([f["obs"] for f in frames])
        flat[f"ep{i}_rew"] = np.stack([f["rew"] for f in frames])
        flat[f"ep{i}_done"] = np.stack([f["done"] for f in frames])
        flat[f"ep{i}_step"] = np.array([f["step"] for f in frames], dtype=float)
        flat[f"ep{i}_extreme"] = np.array([0.0])
    flat["n_episodes"] = np.array([len(episodes)], dtype=float)
    np.savez_compressed(os.path.join(OUT, "singlecombat_sequences.npz"), **flat)


def gen_artillery(rng):
    """SingleCombatTask.step with use_artillery (singlecombat_task.py:162-188): every aircraft drains the blood of each ALIVE enemy by
    orientation_fn(AO) * distance_fn(R / 1000) per env step. Scripted close-in geometries (inside and outside the 30 deg cone and the
    1 km / 3 km range bands, dead targets, dead shooters -- the rule does not ask whether the SHOOTER is alive), bloods after every step."""
    from envs.JSBSim.tasks.singlecombat_task import SingleCombatTask
    cfg = make_config(use_artillery=True)
    task = SingleCombatTask(cfg)
    assert task.use_artillery
    poses, bloods = [], []
    for ep in range(40):
        a, b = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
        link([a, b])
        env = FakeEnv([a, b])
        task.reset(env)
        for t in range(12):
            # shooter a somewhere, target b placed at a chosen range / off-boresight angle from a's velocity vector, and vice versa at random
            random_pose(rng, a, spread_km=5.0)
            dist = rng.choice([300.0, 900.0, 1000.0, 1500.0, 2500.0, 3000.0, 3400.0]) * rng.uniform(0.97, 1.03)
            off = np.deg2rad(rng.choice([0.0, 5.0, 15.0, 29.0, 30.0, 31.0, 60.0, 170.0]) + rng.uniform(-0.4, 0.4))
            v = np.array(a._velocity); vhat = v / np.linalg.norm(v)
            perp = np.cross(vhat, rng.normal(size=3)); perp /= np.linalg.norm(perp)
            # get_AO_TA_R is fed get_position() (N, E, U) + get_velocity() (vN, vE, vDOWN): the cone is about that mixed vector
            d = np.cos(off) * vhat + np.sin(off) * perp
            pa = np.array(a._position)
            pb = pa + dist * d
            random_pose(rng, b, spread_km=5.0)
            lon, lat, alt = utils_neu2lla(pb)
            b.set_pose(lon, lat, alt, tuple(b._posture), tuple(b._velocity))
            if ep % 5 == 1 and t >= 6:
                b.shotdown()
            if ep % 5 == 2 and t >= 4:
                a.crash()
            before = np.stack([pose_vector(a), pose_vector(b)])
            task.step(env)
            poses.append(before)
            bloods.append([a.bloods, b.bloods])
    np.savez_compressed(os.path.join(OUT, "artillery.npz"), pose=np.array(poses), bloods_after=np.array(bloods))


def utils_neu2lla(neu):
    from envs.JSBSim.utils.utils import NEU2LLA
    return NEU2LLA(*neu, 120.0, 60.0, 0.0)


def gen_missile(rng):
    """MissileSimulator fly-outs (AIM-9L defaults and the AIM_9M/AIM_120B parameter set) against scripted targets."""
    from envs.JSBSim.core.simulatior import MissileSimulator, AIM_120B
    from envs.JSBSim.utils.utils import LLA2NEU
    cases = []
    specs = [
        (MissileSimulator, 0, dict(dist=6000.0, tgt_speed=250.0, tgt_turn=0.0, aspect=np.pi, steps=1500)),     # head-on, hit
        (MissileSimulator, 0, dict(dist=9000.0, tgt_speed=300.0, tgt_turn=0.0, aspect=0.0, steps=4200)),       # tail chase
        (MissileSimulator, 0, dict(dist=12000.0, tgt_speed=280.0, tgt_turn=0.12, aspect=1.3, steps=4200)),     # turning target
        (MissileSimulator, 0, dict(dist=25000.0, tgt_speed=330.0, tgt_turn=0.0, aspect=0.1, steps=4200)),      # out of range: slows / recedes
        (AIM_120B, 1, dict(dist=15000.0, tgt_speed=260.0, tgt_turn=0.0, aspect=np.pi, steps=1800)),
        (AIM_120B, 1, dict(dist=8000.0, tgt_speed=260.0, tgt_turn=0.25, aspect=2.0, steps=1800)),
        (MissileSimulator, 0, dict(dist=5000.0, tgt_speed=250.0, tgt_turn=0.0, aspect=np.pi, steps=900, kill_target_at=120)),  # target dies first
    ]
    for cls, model, sp in specs:
        parent, target = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
        parent.set_pose(120.0, 60.0, 6000.0, (0.1, 0.05, 0.0), (260.0, 0.0, -5.0))
        # target placed `dist` north of the parent
        tgt_lat = 60.0 + sp["dist"] / 111320.0
        hdg = sp["aspect"]
        target.set_pose(120.0, tgt_lat, 6200.0, (0.0, 0.0, hdg), (sp["tgt_speed"] * np.cos(hdg), sp["tgt_speed"] * np.sin(hdg), 0.0))
        m = cls.create(parent, target, "A01001") if cls is MissileSimulator else cls.create(parent, target, "A01001", "AIM-120B")
        rows = []
        tpos = target._position.copy()
        for k in range(sp["steps"]):
            # target kinematics (a plain scripted point mass in NEU; velocity z stored as "down" like the aircraft cache)
            hdg += sp["tgt_turn"] * parent.dt
            target._velocity[:] = (sp["tgt_speed"] * np.cos(hdg), sp["tgt_speed"] * np.sin(hdg), 2.0 * np.sin(0.01 * k))
            tpos += parent.dt * np.array([target._velocity[0], target._velocity[1], -target._velocity[2]])
            target._position[:] = tpos
            if sp.get("kill_target_at") == k:
                target.crash()
            alive_before = target.is_alive
            m.run()
            rows.append(np.concatenate([[k, alive_before, m._MissileSimulator__status], target._position, target._velocity,
                                        m.get_position(), m.get_velocity(), m.get_rpy()[1:], [m._t, m._m, m._geodetic[2]]]))
            if m.is_done and k > 5 and rows[-2][2] != 0 and rows[-3][2] != 0:
                break
        cases.append((model, np.array(rows), np.concatenate([parent._geodetic, parent._position, parent._velocity, parent._posture])))
    flat = {"n": np.array([len(cases)], dtype=float)}
    for i, (model, rows, par) in enumerate(cases):
        flat[f"c{i}_model"] = np.array([model], dtype=float)
        flat[f"c{i}_rows"] = rows
        flat[f"c{i}_parent"] = par
    np.savez_compressed(os.path.join(OUT, "missile.npz"), **flat)


def gen_missile_task_obs(rng):
    """21-value observation of SingleCombatDodgeMissileTask (3-D AO/TA, unclipped, missile-warning block)."""
    from envs.JSBSim.tasks.singlecombat_with_missile_task import SingleCombatShootMissileTask
    from envs.JSBSim.core.simulatior import MissileSimulator
    cfg = make_config()
    task = SingleCombatShootMissileTask(cfg)
    poses, obs, msl = [], [], []
    for i in range(200):
        a, b = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
        link([a, b])
        env = FakeEnv([a, b])
        random_pose(rng, a); random_pose(rng, b)
        mrow = np.zeros(7)
        if i % 2:
            m = MissileSimulator.create(b, a, "B01001")
            m._position[:] = a._position + rng.normal(size=3) * np.array([4000, 4000, 800])
            m._velocity[:] = rng.normal(size=3) * np.array([400, 400, 80])
            mrow = np.concatenate([[1.0], m._position, m._velocity])
        poses.append(np.stack([pose_vector(a), pose_vector(b)]))
        obs.append(task.get_obs(env, "A0100"))
        msl.append(mrow)
    np.savez_compressed(os.path.join(OUT, "missile_task_obs.npz"), pose=np.array(poses), obs=np.array(obs), missile=np.array(msl))


def gen_heading(rng):
    """HeadingTask: obs(12), HeadingReward, UnreachHeading with the env.np_random draws (PCG64 seeded like gymnasium)."""
    from envs.JSBSim.tasks.heading_task import HeadingTask
    from envs.JSBSim.core.catalog import Catalog as c
    cfg = make_config(max_steps=10000, aircraft_configs={"A0100": {"color": "Blue", "max_heading_increment": 180, "max_altitude_increment": 7000,
                                                                   "max_velocities_u_increment": 100, "check_interval": 30}},
                      PostureReward_potential=False, EventDrivenReward_potential=False)
    task = HeadingTask(cfg)

    class HeadingAircraft(FakeAircraft):
        # the three delta properties are recomputed on every read by their catalogue update() (catalog.py:340-356)
        def get_property_value(self, prop):
            p = self.props
            if prop.name_jsbsim == "position/delta-altitude-to-target-m":
                return float(np.clip((p["tc/h-sl-ft"] - p["position/h-sl-ft"]) * 0.3048, -40000, 40000))
            if prop.name_jsbsim == "position/delta-heading-to-target-deg":
                x = (p["tc/target-heading-deg"] - p["attitude/psi-deg"]) % 360
                return float(np.clip(x - 360 if x > 180 else x, -180, 180))
            if prop.name_jsbsim == "position/delta-velocities_u-to-target-mps":
                return float(np.clip(p["tc/target-velocity-u-mps"] - p["velocities/u-mps"], -1400, 1400))
            return p[prop.name_jsbsim]

    a = HeadingAircraft("A0100", "Blue")
    env = FakeEnv([a])
    seed = 12345
    env.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    bitgen_state = env.np_random.bit_generator.state
    env.heading_turn_counts = 0
    # property plumbing of the extra catalogue: derived values are recomputed by their update() on read in the real wrapper;
    # here they are supplied directly
    hdg0, alt0_ft, u0 = 35.0, 20000.0, 800.0
    a.props.update({"tc/target-heading-deg": hdg0, "tc/h-sl-ft": alt0_ft, "tc/target-velocity-u-mps": u0 * 0.3048, "heading_check_time": 0.0})
    rows = []
    psi, h_ft, u_mps, roll, p, q = hdg0, alt0_ft, u0 * 0.3048, 0.0, 0.0, 0.0
    task.reset(env)
    for t in range(1, 1000):
        env.current_step = t
        # the scripted aircraft tracks its current target heading (so several checks pass) until it drifts away after t = 620
        psi = (a.props["tc/target-heading-deg"] + rng.normal() * 1.5 + (0.25 * (t - 620) if t > 620 else 0.0)) % 360
        h_ft += rng.normal() * 8
        u_mps += rng.normal() * 0.3
        roll = float(np.clip(roll + rng.normal() * 0.01, -0.5, 0.5))
        p, q = rng.normal() * 0.02, rng.normal() * 0.02
        sim_time = t * 0.1

        a.props.update({
            "position/h-sl-ft": h_ft, "attitude/psi-deg": psi,
            "position/h-sl-m": h_ft * 0.3048, "attitude/roll-rad": roll, "attitude/pitch-rad": 0.02,
            "velocities/u-mps": u_mps, "velocities/v-mps": 0.5, "velocities/w-mps": 3.0, "velocities/vc-mps": 200.0,
            "velocities/p-rad_sec": p, "velocities/q-rad_sec": q, "simulation/sim-time-sec": sim_time,
            "accelerations/n-pilot-x-norm": 0.0, "accelerations/n-pilot-y-norm": 0.0, "accelerations/n-pilot-z-norm": -1.0,
            "detect/extreme-state": 0,
        })
        a._position[:] = (0, 0, h_ft * 0.3048)
        a._velocity[:] = (0, 0, 0.0)
        obs = task.get_obs(env, "A0100")
        info = {"current_step": t}
        done, info = task.get_termination(env, "A0100", info)
        rew, info = task.get_reward(env, "A0100", info)
        rows.append(np.concatenate([[t, psi, h_ft, u_mps, roll, p, q, sim_time, done, rew, env.heading_turn_counts,
                                     a.props["tc/target-heading-deg"], a.props["tc/h-sl-ft"], a.props["tc/target-velocity-u-mps"],
                                     a.props["heading_check_time"]], obs]))
        if done:
            break
    m = (1 << 64) - 1
    st, inc = bitgen_state["state"]["state"], bitgen_state["state"]["inc"]
    np.savez_compressed(os.path.join(OUT, "heading.npz"), rows=np.array(rows), init=np.array([hdg0, alt0_ft, u0 * 0.3048]),
             pcg64=np.array([st >> 64, st & m, inc >> 64, inc & m], dtype=np.uint64))



def gen_approach(rng):
    """ApproachTask (tasks/approach_task.py): the heading observation, AltitudeReward alone, LowAltitude / ExtremeState / Overload /
    Timeout in that order. A scripted descent through the safe, danger and limit altitudes ends the episode by LowAltitude."""
    from envs.JSBSim.tasks.approach_task import ApproachTask
    cfg = make_config(max_steps=10000, aircraft_configs={"A0100": {"color": "Blue"}}, PostureReward_potential=False, EventDrivenReward_potential=False)
    task = ApproachTask(cfg)

    class ApproachAircraft(FakeAircraft):
        def get_property_value(self, prop):
            p = self.props
            if prop.name_jsbsim == "position/delta-altitude-to-target-m":
                return float(np.clip((p["tc/h-sl-ft"] - p["position/h-sl-ft"]) * 0.3048, -40000, 40000))
            if prop.name_jsbsim == "position/delta-heading-to-target-deg":
                x = (p["tc/target-heading-deg"] - p["attitude/psi-deg"]) % 360
                return float(np.clip(x - 360 if x > 180 else x, -180, 180))
            if prop.name_jsbsim == "position/delta-velocities_u-to-target-mps":
                return float(np.clip(p["tc/target-velocity-u-mps"] - p["velocities/u-mps"], -1400, 1400))
            return p[prop.name_jsbsim]

    a = ApproachAircraft("A0100", "Blue")
    env = FakeEnv([a])
    hdg0, alt0_ft, u0 = 35.0, 16000.0, 800.0
    a.props.update({"tc/target-heading-deg": hdg0, "tc/h-sl-ft": alt0_ft, "tc/target-velocity-u-mps": u0 * 0.3048, "heading_check_time": 0.0})
    rows = []
    psi, h_ft, u_mps, roll = hdg0, alt0_ft, u0 * 0.3048, 0.0
    a.props.update({"position/h-sl-ft": h_ft, "position/h-sl-m": h_ft * 0.3048})
    a._position[:] = (0, 0, h_ft * 0.3048)
    task.reset(env)
    for t in range(1, 400):
        env.current_step = t
        psi = (psi + rng.normal() * 3.0) % 360
        sink = 30.0 + 0.25 * t                       # m/s down, growing: crosses 4 km, 3.5 km and the 2.5 km limit
        h_ft -= sink * 0.1 / 0.3048
        u_mps += rng.normal() * 0.3
        roll = float(np.clip(roll + rng.normal() * 0.01, -0.5, 0.5))
        a.props.update({
            "position/h-sl-ft": h_ft, "attitude/psi-deg": psi, "position/h-sl-m": h_ft * 0.3048, "attitude/roll-rad": roll, "attitude/pitch-rad": -0.1,
            "velocities/u-mps": u_mps, "velocities/v-mps": 0.5, "velocities/w-mps": 3.0, "velocities/vc-mps": 200.0,
            "velocities/p-rad_sec": 0.0, "velocities/q-rad_sec": 0.0, "simulation/sim-time-sec": t * 0.1,
            "accelerations/n-pilot-x-norm": 0.0, "accelerations/n-pilot-y-norm": 0.0, "accelerations/n-pilot-z-norm": -1.0, "detect/extreme-state": 0,
        })
        a._position[:] = (0, 0, h_ft * 0.3048)
        a._velocity[:] = (0, 0, sink)
        obs = task.get_obs(env, "A0100")
        done, info = task.get_termination(env, "A0100", {"current_step": t})
        rew, info = task.get_reward(env, "A0100", info)
        rows.append(np.concatenate([[t, psi, h_ft, u_mps, roll, sink, t * 0.1, done, rew], obs]))
        if done:
            break
    np.savez_compressed(os.path.join(OUT, "approach.npz"), rows=np.array(rows), init=np.array([hdg0, alt0_ft, u0 * 0.3048]))


def gen_multicombat_sequences(rng):
    """MultipleCombatTask (2v2) over scripted four-aircraft pose sequences. The tail of MultipleCombatEnv.step
    (multiplecombat_env.py:160-182: obs, rewards for every agent, team mean, then terminations) is replayed here around the
    reference's task object, because the env class itself cannot be built without the jsbsim wheel."""
    from envs.JSBSim.tasks.multiplecombat_task import MultipleCombatTask
    acs = {u: {"color": "Blue" if u[0] == "A" else "Red", "missile": 0} for u in ("A0100", "A0200", "B0100", "B0200")}
    cfg = make_config(aircraft_configs=acs, EventDrivenReward_potential=False)
    flat = {}
    n_ep = 16
    for ep in range(n_ep):
        task = MultipleCombatTask(cfg)
        agents = [FakeAircraft(u, acs[u]["color"]) for u in acs]
        link(agents)
        env = FakeEnv(agents)
        for a in agents:
            random_pose(rng, a, spread_km=12.0 if ep % 2 else 40.0)
        task.reset(env)
        poses, obss, rews, dones, steps = [np.stack([pose_vector(a) for a in agents])], [np.stack([task.get_obs(env, u) for u in env.agents])], [np.zeros(4)], [np.zeros(4)], [0]
        for t in range(1, 13):
            env.current_step = t
            for a in agents:
                if a.is_alive:
                    random_pose(rng, a, spread_km=12.0 if ep % 2 else 40.0, alt=(2000.0, 9000.0) if ep % 3 else (2300.0, 5000.0))
            if ep % 5 == 1 and t == 4:
                agents[2].shotdown()
            if ep % 5 == 1 and t == 7:
                agents[3].shotdown()
            if ep % 5 == 2 and t == 5:
                agents[0].props["detect/extreme-state"] = 1
            if ep % 5 == 3 and t == 6:
                agents[1].props["accelerations/n-pilot-y-norm"] = 10.5
                agents[1].props["simulation/sim-time-sec"] = 25.0
            if ep % 5 == 4 and t == 8:
                env.current_step = 9000
            pose = np.stack([pose_vector(a) for a in agents])
            obs = np.stack([task.get_obs(env, u) for u in env.agents])
            info = {"current_step": env.current_step}
            rewards = {}
            for u in env.agents:
                r, info = task.get_reward(env, u, info)
                rewards[u] = [r]
            ego = np.mean([rewards[u] for u in env.ego_ids])
            enm = np.mean([rewards[u] for u in env.enm_ids])
            rew = np.array([ego if u in env.ego_ids else enm for u in env.agents])
            done = []
            for u in env.agents:
                d, info = task.get_termination(env, u, info)
                done.append(d)
            poses.append(pose); obss.append(obs); rews.append(rew); dones.append(np.array(done, dtype=float)); steps.append(env.current_step)
            if all(done):
                break
        flat[f"ep{ep}_pose"] = np.stack(poses); flat[f"ep{ep}_obs"] = np.stack(obss); flat[f"ep{ep}_rew"] = np.stack(rews)
        flat[f"ep{ep}_done"] = np.stack(dones); flat[f"ep{ep}_step"] = np.array(steps, dtype=float)
    flat["n_episodes"] = np.array([n_ep], dtype=float)
    np.savez_compressed(os.path.join(OUT, "multicombat_sequences.npz"), **flat)



class KeyedRNG:
    """Stand-in for the global, unseeded np.random the reference's decoy test draws from (env_base.py:153): splitmix64 of a
    seed and a key naming the (substep, missile, chaff) pair under test — the generator the oracle and the kernel implement."""

    def __init__(self, seed=1):
        self.seed, self.n = seed, 0

    def rand(self, tick, msl_parent, msl_num, chaff_parent, chaff_local):
        M = (1 << 64) - 1
        k = ((tick & 0xffffffff) << 32) | ((msl_parent & 0xff) << 24) | ((msl_num & 0xff) << 16) | ((chaff_parent & 0xff) << 8) | (chaff_local & 0xff)
        z = (self.seed * 0x9E3779B97F4A7C15 + k * 0xD1B54A32D192ED03) & M
        self.n += 1
        z ^= z >> 30; z = (z * 0xBF58476D1CE4E5B9) & M
        z ^= z >> 27; z = (z * 0x94D049BB133111EB) & M
        z ^= z >> 31
        return (z >> 40) / 16777216.0


class WeaponEnv(FakeEnv):
    """FakeEnv plus the projectile part of BaseEnv.step's substep loop (env_base.py:139-154), replayed around the reference's
    own MissileSimulator / ChaffSimulator objects (the aircraft are scripted poses, held during the six substeps)."""

    def __init__(self, agents):
        super().__init__(agents)
        self._chaffsims = {}
        self.rng = KeyedRNG(1)
        self.tick = 0
        self.uids = [a.uid for a in agents]

    def add_chaff_simulator(self, sim):
        self._chaffsims[sim.uid] = sim
        return sim

    def run_projectiles(self, substeps=6):
        for a in self.agents.values():            # AircraftSimulator.run: bloods <= 0 -> shotdown (simulatior.py:220-222)
            if a.is_alive and a.bloods <= 0:
                a.shotdown()
        for _ in range(substeps):
            self.tick += 1
            for sim in self._tempsims.values():
                sim.run()
            for sim in self._chaffsims.values():
                sim.run()
            for missile in self._tempsims.values():
                if missile.is_done:
                    continue
                for chaff in self._chaffsims.values():
                    if chaff.is_done:
                        continue
                    if np.linalg.norm(chaff.get_position() - missile.get_position()) <= chaff.effective_radius:
                        # key: missile = (launcher index, uid number), chaff = (releaser index, its release index)
                        cp = self.uids.index(chaff.uid[:5])
                        local = [c.uid for c in self._chaffsims.values() if c.uid[:5] == chaff.uid[:5]].index(chaff.uid)
                        if self.rng.rand(self.tick, self.uids.index(missile.uid[:5]), int(missile.uid[5:]), cp, local) < 0.85:
                            missile.missed()


def engagement_pose(rng, shooter, target, dist_m, off_deg, alt_m=6000.0):
    """Put `target` dist_m ahead of `shooter` (who flies north), off_deg off the nose; both roughly co-altitude."""
    shooter.set_pose(120.0, 60.0, alt_m, (0.05, 0.02, 0.0), (250.0, 0.0, -2.0), (250.0, 1.0, 5.0), vc=240.0,
                     npilot=(0.1, 0.0, -1.1), sim_time=30.0)
    b = np.radians(off_deg)
    dn, de = dist_m * np.cos(b), dist_m * np.sin(b)
    lat = 60.0 + dn / 111412.0
    lon = 120.0 + de / (111320.0 * np.cos(np.radians(60.0)))
    hdg = rng.uniform(0, 2 * np.pi)
    target.set_pose(lon, lat, alt_m + rng.uniform(-300, 300), (0.0, 0.0, hdg), (230.0 * np.cos(hdg), 230.0 * np.sin(hdg), 3.0),
                    (230.0, -1.0, 4.0), vc=225.0, npilot=(0.0, 0.0, -1.0), sim_time=30.0)


def missile_rows(env, agents):
    """Per frame: every missile ever launched in launch order: parent, target, status, pos, vel; plus per-agent counters."""
    rows = []
    seen = []
    for a in agents:
        for m in a.launch_missiles:
            seen.append(m)
    # global launch order = order of appearance in the under_missiles lists merged by creation: keep creation order by id list
    return seen


def _scenario_task(cls, cfg):
    import torch
    orig = torch.load
    torch.load = lambda f, map_location=None, **kw: orig(f, map_location="cpu", **kw)   # the reference asks for 'cuda'
    try:
        return cls(cfg)
    finally:
        torch.load = orig


def gen_scenario_sequences(rng):
    """Scenario1 (1v1) and Scenario2_NvN (2v2) weapon rules, the 11 reward terms with their shared-list behaviour, the NvN
    observation layout, chaff release and the decoy draw — over scripted engagements."""
    from envs.JSBSim.tasks.scenario1_task import Scenario1
    from envs.JSBSim.tasks.scenario2_task import Scenario2_NvN
    import envs.JSBSim.tasks.scenario1_task as s1mod
    flat = {}
    ep_id = 0
    for family in ("s1", "nvn"):
        if family == "s1":
            uids = ("A0100", "B0100")
        else:
            uids = ("A0100", "A0200", "B0100", "B0200")
        acs = {u: {"color": "Blue" if u[0] == "A" else "Red", "missile": 2} for u in uids}
        cfg = make_config(aircraft_configs=acs, EventDrivenReward_potential=(family == "s1"))
        for ep in range(8):
            task = _scenario_task(Scenario1 if family == "s1" else Scenario2_NvN, cfg)
            agents = [FakeAircraft(u, acs[u]["color"]) for u in uids]
            link(agents)
            env = WeaponEnv(agents)
            order = []      # missiles in creation order
            orig_add = env.add_temp_simulator

            def add(sim, _o=order, _f=orig_add):
                _o.append(sim)
                return _f(sim)
            env.add_temp_simulator = add
            A = len(agents)
            shooter, target = agents[0], agents[A // 2]
            # episode kinds: 0 gun duel, 1/2 tail chase with chaff (long: missiles time out / get decoyed, second launches reuse
            # uids), 3 random poses, 4 head-on at long range, 5 random with a scripted blood loss
            kind = ep % 6
            T = {0: 40, 1: 330, 2: 330, 3: 30, 4: 120, 5: 30}[kind]
            for a in agents:
                random_pose(rng, a, spread_km=8.0, alt=(2600.0, 9000.0))

            def chase_pose(t):
                # both fly north along the 120E meridian; positions advance with time so that missiles and chaff see a moving target
                sep0 = {0: 2000.0, 1: 6000.0, 2: 4500.0, 4: 30000.0}.get(kind, 5000.0)
                vs, vt = 255.0, (-240.0 if kind == 4 else 235.0)
                ys = vs * 0.1 * t
                yt = sep0 + vt * 0.1 * t
                off = 25.0 * np.sin(0.05 * t) if kind != 0 else 3.0
                shooter.set_pose(120.0, 60.0 + ys / 111412.0, 6000.0, (0.02, 0.01, 0.0), (vs, 0.0, -1.0), (vs, 0.5, 4.0), vc=240.0,
                                 npilot=(0.1, 0.0, -1.05), sim_time=12.0 + 0.1 * t)
                hdg = np.pi if kind == 4 else 0.0
                target.set_pose(120.0 + off / 55660.0, 60.0 + yt / 111412.0, 6050.0, (0.0, 0.0, hdg), (vt, 0.0, 1.0), (abs(vt), -0.5, 3.0),
                                vc=225.0, npilot=(0.0, 0.0, -1.0), sim_time=12.0 + 0.1 * t)
            if kind in (0, 1, 2, 4):
                chase_pose(0)
            task.reset(env)
            frames = []
            for t in range(1, T + 1):
                env.current_step = t
                for k, a in enumerate(agents):
                    if not a.is_alive:
                        continue
                    if kind in (0, 1, 2, 4) and a in (shooter, target):
                        continue
                    if t % 3 == 0 or kind in (3, 5):
                        random_pose(rng, a, spread_km=8.0, alt=(2600.0, 9000.0))
                if kind in (0, 1, 2, 4) and shooter.is_alive and target.is_alive:
                    chase_pose(t)
                if kind == 5 and t == 12:
                    agents[A - 1].bloods = 0     # dies at the next run()
                pose = np.stack([pose_vector(a) for a in agents])
                env.run_projectiles(6)
                if kind == 0:
                    bits = np.tile(np.array([1, 0, 0, 0]), (A, 1))
                elif kind in (1, 2, 4):
                    bits = np.tile(np.array([0, 1, 1, 1]), (A, 1))
                    if kind == 2:
                        bits[:, 0] = (t % 2)
                else:
                    bits = (rng.random((A, 4)) < 0.4).astype(int)
                for k, u in enumerate(env.agents):
                    if family == "nvn" or u in env.ego_ids:       # what normalize_action does with action[-4:] (the controller net is bypassed)
                        task._shoot_action[u] = list(bits[k])
                    else:                                         # Scenario1's other team flies the scripted baseline: weapon bits [0,0,0,0]
                        task._shoot_action[u] = [0, 0, 0, 0]      # (scenario1_task.py:38-39)
                task.step(env)
                obs = np.stack([task.get_obs(env, u) for u in env.agents])
                info = {"current_step": env.current_step}
                if family == "s1":
                    done = []
                    for u in env.agents:
                        d, info = task.get_termination(env, u, info)
                        done.append(d)
                    rew = []
                    for u in env.agents:
                        r, info = task.get_reward(env, u, info)
                        rew.append(r)
                    rew = np.array(rew, dtype=float)
                else:
                    rewards = {}
                    for u in env.agents:
                        r, info = task.get_reward(env, u, info)
                        rewards[u] = [r]
                    ego = np.mean([rewards[u] for u in env.ego_ids]); enm = np.mean([rewards[u] for u in env.enm_ids])
                    rew = np.array([ego if u in env.ego_ids else enm for u in env.agents])
                    done = []
                    for u in env.agents:
                        d, info = task.get_termination(env, u, info)
                        done.append(d)
                counters = np.array([[task.remaining_gun[u], task.remaining_missiles_AIM_9M[u], task.remaining_missiles_AIM_120B[u],
                                      task.remaining_chaff_flare[u], agents[k].bloods, agents[k].status] for k, u in enumerate(env.agents)], dtype=float)
                mrows = np.zeros((12, 9))
                for k, m in enumerate(order[:12]):
                    mrows[k] = [1 + uids.index(m.parent_aircraft.uid), uids.index(m.target_aircraft.uid), m._MissileSimulator__status,
                                *m.get_position(), *m.get_velocity()]
                frames.append(dict(pose=pose, bits=bits, obs=obs, rew=rew, done=np.array(done, dtype=float), counters=counters, msl=mrows,
                                   nchaff=len(env._chaffsims), draws=env.rng.n, step=env.current_step, nmsl=len(order)))
                if all(done):
                    break
            for key in ("pose", "bits", "obs", "rew", "done", "counters", "msl"):
                flat[f"ep{ep_id}_{key}"] = np.stack([f[key] for f in frames])
            flat[f"ep{ep_id}_misc"] = np.array([[f["nchaff"], f["draws"], f["step"], f["nmsl"]] for f in frames], dtype=float)
            flat[f"ep{ep_id}_family"] = np.array([0.0 if family == "s1" else 1.0])
            ep_id += 1
    flat["n_episodes"] = np.array([ep_id], dtype=float)
    np.savez_compressed(os.path.join(OUT, "scenario_sequences.npz"), **flat)


def gen_multicombat_dodge(rng):
    """MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-145; 2v2 and 4v4): the 21-value observation against the
    enemy with the agent's own team index with the missile-warning block, the rule-based launch at enemies[0] from the one-second lock
    window (angle, distance, interval, remaining rounds, alive), the base-class missile flown by MultipleCombatEnv.step's substep
    loop, the four reward terms (MissilePostureReward's shared remembered missile included) and MultipleCombatTask's terminations.
    No env of the reference constructs this class; the task object itself is driven here like the others."""
    from envs.JSBSim.tasks.multiplecombat_with_missile_task import MultipleCombatDodgeMissileTask
    flat = {}
    ep_id = 0
    for per_side in (2, 4):
        uids = tuple(f"{team}0{k + 1}00" for team in "AB" for k in range(per_side))
        acs = {u: {"color": "Blue" if u[0] == "A" else "Red", "missile": 2} for u in uids}
        cfg = make_config(aircraft_configs=acs, EventDrivenReward_potential=False, min_attack_interval=25, max_attack_distance=14000, max_attack_angle=45)
        for ep in range(6 if per_side == 2 else 3):
            task = MultipleCombatDodgeMissileTask(cfg)
            assert task.num_agents == 4 or True
            agents = [FakeAircraft(u, acs[u]["color"]) for u in uids]
            link(agents)
            env = WeaponEnv(agents)
            order = []
            orig_add = env.add_temp_simulator

            def add(sim, _o=order, _f=orig_add):
                _o.append(sim)
                return _f(sim)
            env.add_temp_simulator = add
            A = len(agents)
            # kinds: 0 tail chase inside every gate (two launches 25 steps apart, the second ego aircraft locks the same enemies[0]);
            # 1 random poses; 2 head-on from beyond max_attack_distance, closing; 3 tail chase in which the shooter dies of blood loss
            # after its first launch; 4 chase just outside the lock cone for a while, then inside; 5 random, close
            kind = ep % 6
            T = {0: 220, 1: 30, 2: 270, 3: 150, 4: 180, 5: 30}[kind]
            shooter, wing, target = agents[0], agents[1], agents[per_side]
            for a in agents:
                random_pose(rng, a, spread_km=8.0 if kind == 5 else 25.0, alt=(2600.0, 9000.0))

            def chase_pose(t):
                sep0 = {0: 5000.0, 2: 17000.0, 3: 4000.0, 4: 6000.0}[kind]
                vs, vt = 255.0, (-240.0 if kind == 2 else 235.0)
                ys, yt = vs * 0.1 * t, sep0 + vt * 0.1 * t
                off_m = 0.0
                if kind == 4:
                    off_m = 9000.0 if t < 20 else 600.0          # 56 deg off the nose, then 6 deg
                shooter.set_pose(120.0, 60.0 + ys / 111412.0, 6000.0, (0.02, 0.01, 0.0), (vs, 0.0, -1.0), (vs, 0.5, 4.0), vc=240.0,
                                 npilot=(0.1, 0.0, -1.05), sim_time=12.0 + 0.1 * t)
                wing.set_pose(120.0 - 300.0 / 55660.0, 60.0 + (ys - 800.0) / 111412.0, 6100.0, (0.0, 0.0, 0.0), (vs, 0.0, 0.0), (vs, 0.0, 3.0), vc=240.0,
                              npilot=(0.0, 0.0, -1.0), sim_time=12.0 + 0.1 * t)
                hdg = np.pi if kind == 2 else 0.0
                target.set_pose(120.0 + (off_m + 25.0 * np.sin(0.05 * t)) / 55660.0, 60.0 + yt / 111412.0, 6050.0, (0.0, 0.0, hdg), (vt, 0.0, 1.0),
                                (abs(vt), -0.5, 3.0), vc=225.0, npilot=(0.0, 0.0, -1.0), sim_time=12.0 + 0.1 * t)
            scripted = kind in (0, 2, 3, 4)
            if scripted:
                chase_pose(0)
            task.reset(env)
            frames = []
            for t in range(1, T + 1):
                env.current_step = t
                for a in agents:
                    if not a.is_alive or (scripted and a in (shooter, wing, target)):
                        continue
                    if t % 3 == 0 or not scripted:
                        random_pose(rng, a, spread_km=8.0 if kind == 5 else 25.0, alt=(2600.0, 9000.0))
                if scripted:
                    chase_pose(t)           # (a dead aircraft keeps being posed: nothing reads its pose but the missile aimed at it)
                if kind == 3 and t == 14:
                    shooter.bloods = 0      # dies at the next run()
                pose = np.stack([pose_vector(a) for a in agents])
                env.run_projectiles(6)
                task.step(env)
                obs = np.stack([task.get_obs(env, u) for u in env.agents])
                info = {"current_step": env.current_step}
                rewards = {}
                for u in env.agents:
                    r, info = task.get_reward(env, u, info)
                    rewards[u] = [r]
                ego = np.mean([rewards[u] for u in env.ego_ids]); enm = np.mean([rewards[u] for u in env.enm_ids])
                rew = np.array([ego if u in env.ego_ids else enm for u in env.agents])
                done = []
                for u in env.agents:
                    d, info = task.get_termination(env, u, info)
                    done.append(d)
                counters = np.array([[task.remaining_missiles[u], task._last_shoot_time[u], sum(task.lock_duration[u]), agents[k].bloods, agents[k].status]
                                     for k, u in enumerate(env.agents)], dtype=float)
                mrows = np.zeros((16, 9))
                for k, m in enumerate(order[:16]):
                    mrows[k] = [1 + uids.index(m.parent_aircraft.uid), uids.index(m.target_aircraft.uid), m._MissileSimulator__status,
                                *m.get_position(), *m.get_velocity()]
                frames.append(dict(pose=pose, obs=obs, rew=rew, done=np.array(done, dtype=float), counters=counters, msl=mrows,
                                   step=env.current_step, nmsl=len(order)))
                if all(done):
                    break
            for key in ("pose", "obs", "rew", "done", "counters", "msl"):
                flat[f"ep{ep_id}_{key}"] = np.stack([f[key] for f in frames])
            flat[f"ep{ep_id}_misc"] = np.array([[f["step"], f["nmsl"]] for f in frames], dtype=float)
            flat[f"ep{ep_id}_per_side"] = np.array([float(per_side)])
            ep_id += 1
    flat["n_episodes"] = np.array([ep_id], dtype=float)
    flat["min_attack_interval"] = np.array([25.0])
    np.savez_compressed(os.path.join(OUT, "multicombat_dodge_sequences.npz"), **flat)


def gen_curriculum_table():
    from envs.JSBSim.utils.utils import calculate_coordinates_heading_by_curriculum
    res = calculate_coordinates_heading_by_curriculum(60.1, 120.0, 11.119, list(range(0, 181)))
    np.savez_compressed(os.path.join(OUT, "curriculum_spawn.npz"), table=np.array(res, dtype=float))


def gen_baseline_actor(rng):
    """The low-level controller (envs/JSBSim/model/baseline_actor.py BaselineActor + model/baseline_model.pt) driven like
    HierarchicalSingleCombatTask.normalize_action does (singlecombat_task.py:223-256): 12 inputs -> 4 argmax indices, GRU
    state carried over a sequence. Logits are read from the reference module's own head layers."""
    import torch
    from envs.JSBSim.model.baseline_actor import BaselineActor
    actor = BaselineActor()
    actor.load_state_dict(torch.load(os.path.join(REF, "envs/JSBSim/model/baseline_model.pt"), map_location=torch.device("cpu")))
    actor.eval()
    S, T = 16, 24
    d_alt, d_hdg, d_vel = np.array([0.1, 0, -0.1]), np.array([-np.pi / 6, -np.pi / 12, 0, np.pi / 12, np.pi / 6]), np.array([0.05, 0, -0.05])
    X = np.zeros((S, T, 12)); ACT = np.zeros((S, T, 4), dtype=np.int64); H = np.zeros((S, T, 128)); LOG = np.zeros((S, T, 153))
    with torch.no_grad():
        for s_ in range(S):
            h = np.zeros((1, 1, 128))
            roll, pitch = rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.2)
            alt, u = rng.uniform(3000, 9000), rng.uniform(180, 330)
            for t_ in range(T):
                roll += rng.normal(0, 0.15); pitch = np.clip(pitch + rng.normal(0, 0.05), -1.2, 1.2)
                alt += rng.normal(0, 40); u = np.clip(u + rng.normal(0, 4), 120, 400)
                x = np.array([d_alt[rng.integers(3)], d_hdg[rng.integers(5)], d_vel[rng.integers(3)], alt / 5000,
                              np.sin(roll), np.cos(roll), np.sin(pitch), np.cos(pitch), u / 340, rng.normal(0, 0.02),
                              rng.normal(0, 0.05), (u * rng.uniform(0.75, 1.0)) / 340])
                if s_ >= S - 4:                       # a few wild inputs: large magnitudes exercise the LayerNorms
                    x = x * rng.uniform(0.2, 4.0, size=12)
                X[s_, t_] = x
                a, h2 = actor(x[None, :], h)
                xt = torch.from_numpy(x[None, :]).float()
                feat = actor.base(xt)
                feat, _ = actor.rnn(feat, torch.from_numpy(h).float())
                LOG[s_, t_] = np.concatenate([m.logits_net(feat).numpy().ravel() for m in actor.act.action_outs])
                h = h2.numpy()
                ACT[s_, t_] = a.numpy().ravel(); H[s_, t_] = h.ravel()
    np.savez_compressed(os.path.join(OUT, "baseline_actor.npz"), x=X, action=ACT, hidden=H.astype(np.float32), logits=LOG.astype(np.float32))


def gen_wvr_sequences(rng, which="wvr"):
    """WVRTask (tasks/WVR_task.py:10-90): the 15-value SingleCombatTask observation, the unlimited gun (-5 blood on the farthest
    enemy inside 3 km and 5 deg, dead shooters included), eight reward terms, terminations WITHOUT SafeReturn.
    which="maneuver": Maneuver_curriculum (tasks/singlecombat_task.py:264-359): the same gun, nine reward terms (adds
    RelativeAltitude), the ordinary 1v1 terminations (SafeReturn included)."""
    from envs.JSBSim.tasks.WVR_task import WVRTask
    from envs.JSBSim.tasks.singlecombat_task import Maneuver_curriculum
    if which == "maneuver":
        WVRTask = Maneuver_curriculum
    flat = {}
    uids = ("A0100", "B0100")
    acs = {u: {"color": "Blue" if u[0] == "A" else "Red", "missile": 2} for u in uids}
    cfg = make_config(aircraft_configs=acs, max_steps=60)
    for ep in range(4):
        task = _scenario_task(WVRTask, cfg)
        agents = [FakeAircraft(u, acs[u]["color"]) for u in uids]
        link(agents)
        env = WeaponEnv(agents)
        env.reset_simulators_curriculum = lambda angle: None     # the spawn is the env's business; poses are scripted here
        shooter, target = agents
        kind = ep % 4      # 0 tail chase inside the gun envelope, 1 random poses, 2 slow closure into the envelope, 3 runs to max_steps

        def chase_pose(t):
            sep0 = {0: 1500.0, 2: 3600.0, 3: 9000.0}[kind]
            vs, vt = 255.0, (235.0 if kind != 2 else 215.0)
            ys, yt = vs * 0.1 * t, sep0 + vt * 0.1 * t
            off = 20.0 * np.sin(0.07 * t)
            shooter.set_pose(120.0, 60.0 + ys / 111412.0, 6000.0, (0.02, 0.01, 0.0), (vs, 0.0, -1.0), (vs, 0.5, 4.0), vc=240.0,
                             npilot=(0.1, 0.0, -1.05), sim_time=12.0 + 0.1 * t)
            target.set_pose(120.0 + off / 55660.0, 60.0 + yt / 111412.0, 6050.0, (0.0, 0.0, 0.0), (vt, 0.0, 1.0), (vt, -0.5, 3.0),
                            vc=225.0, npilot=(0.0, 0.0, -1.0), sim_time=12.0 + 0.1 * t)
        if kind == 1:
            for a in agents:
                random_pose(rng, a, spread_km=3.0, alt=(2600.0, 9000.0))
        else:
            chase_pose(0)
        task.reset(env)
        frames = []
        for t in range(1, 70):
            env.current_step = t
            if kind == 1:
                for a in agents:
                    if a.is_alive:
                        random_pose(rng, a, spread_km=3.0, alt=(2300.0 if t > 25 else 2600.0, 9000.0))
            else:
                chase_pose(t)
            pose = np.stack([pose_vector(a) for a in agents])
            env.run_projectiles(0)          # AircraftSimulator.run: bloods <= 0 -> shot down
            task.step(env)
            obs = np.stack([task.get_obs(env, u) for u in env.agents])
            info = {"current_step": env.current_step}
            done = []
            for u in env.agents:
                d, info = task.get_termination(env, u, info)
                done.append(d)
            rew = []
            for u in env.agents:
                r, info = task.get_reward(env, u, info)
                rew.append(r)
            state = np.array([[a.bloods, a.status] for a in agents], dtype=float)
            frames.append(dict(pose=pose, obs=obs, rew=np.array(rew, dtype=float), done=np.array(done, dtype=float), state=state, step=t))
            if all(done):
                break
        for key in ("pose", "obs", "rew", "done", "state"):
            flat[f"ep{ep}_{key}"] = np.stack([f[key] for f in frames])
        flat[f"ep{ep}_step"] = np.array([f["step"] for f in frames], dtype=float)
    flat["n_episodes"] = np.array([4.0])
    flat["max_steps"] = np.array([60.0])
    np.savez_compressed(os.path.join(OUT, f"{which}_sequences.npz"), **flat)


def gen_rwr_obs(rng):
    """Observation builders of the RWR task variants: Scenario1_RWR (23 values: the 21-value layout with the missile block forced
    to zero and two reserved slots, scenario1_task.py:213-314) and Scenario2_RWR (11 + 6 per other aircraft + 6, i.e. the NvN layout
    plus two trailing reserved slots, scenario2_task.py:403-476)."""
    from envs.JSBSim.tasks.scenario1_task import Scenario1_RWR
    from envs.JSBSim.tasks.scenario2_task import Scenario2_RWR, Scenario2
    from envs.JSBSim.core.simulatior import MissileSimulator
    out = {}
    # "legacy": Scenario2 itself keeps MultipleCombatShootMissileTask's 21-value observation against the enemy with the same index
    for fam, cls, uids in (("s1", Scenario1_RWR, ("A0100", "B0100")), ("nvn", Scenario2_RWR, ("A0100", "A0200", "B0100", "B0200")),
                           ("legacy", Scenario2, ("A0100", "A0200", "B0100", "B0200"))):
        acs = {u: {"color": "Blue" if u[0] == "A" else "Red", "missile": 2} for u in uids}
        task = _scenario_task(cls, make_config(aircraft_configs=acs))
        poses, obs, msl = [], [], []
        for i in range(40):
            agents = [FakeAircraft(u, acs[u]["color"]) for u in uids]
            link(agents)
            env = FakeEnv(agents)
            for a in agents:
                random_pose(rng, a, spread_km=10.0)
            mrow = np.zeros(7)
            if i % 2:
                m = MissileSimulator.create(agents[-1], agents[0], "B01001")
                m._position[:] = agents[0]._position + rng.normal(size=3) * np.array([4000, 4000, 800])
                m._velocity[:] = rng.normal(size=3) * np.array([400, 400, 80])
                mrow = np.concatenate([[1.0], m._position, m._velocity])
            poses.append(np.stack([pose_vector(a) for a in agents]))
            obs.append(np.stack([task.get_obs(env, u) for u in uids]))
            msl.append(mrow)
        out[f"{fam}_pose"], out[f"{fam}_obs"], out[f"{fam}_missile"] = np.array(poses), np.array(obs), np.array(msl)
    np.savez_compressed(os.path.join(OUT, "rwr_obs.npz"), **out)


def gen_baseline_agents(rng):
    """Scripted opponents of the `use_baseline` YAMLs (envs/JSBSim/model/baseline.py): PursueAgent.set_delta_value /
    BaselineAgent.get_observation on random two-aircraft poses, and a ManeuverAgent('triangle') sequence (turn schedule + latched
    initial heading). Outputs are the 12 controller inputs; the controller itself is pinned by baseline_actor.npz."""
    import torch
    from envs.JSBSim.model.baseline import PursueAgent, ManeuverAgent
    orig = torch.load
    torch.load = lambda f, map_location=None, **kw: orig(f, map_location="cpu", **kw)   # the reference asks for 'cuda'
    try:
        pursue, man = PursueAgent(agent_id=1), ManeuverAgent(agent_id=1, maneuver="triangle")
    finally:
        torch.load = orig
    a, b = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
    link([a, b])
    env = FakeEnv([a, b])
    # the three delta properties are read (and ignored) by get_observation's state_var
    for k, prop in enumerate(pursue.state_var[:3]):
        for ac in (a, b):
            ac.props[prop.name_jsbsim] = 0.0
    P = 160
    poses = np.zeros((P, 2, 20)); delta = np.zeros((P, 3)); obs = np.zeros((P, 12))
    for k in range(P):
        random_pose(rng, a); random_pose(rng, b, spread_km=12.0 if k % 3 else 2.0)
        poses[k, 0], poses[k, 1] = pose_vector(a), pose_vector(b)
        dv = pursue.set_delta_value(env, None, 0)
        delta[k] = dv
        obs[k] = pursue.get_observation(env, None, dv)[0]
    # maneuver: 70 steps with a 1.5 s turn interval so that several schedule entries are crossed
    man.turn_interval = 1.5
    man.reset()
    T = 70
    mposes = np.zeros((T, 20)); mdelta = np.zeros((T, 3)); mobs = np.zeros((T, 12))
    for t_ in range(T):
        random_pose(rng, b)
        mposes[t_] = pose_vector(b)
        dv = man.set_delta_value(env, None)
        mdelta[t_] = dv
        mobs[t_] = man.get_observation(env, None, dv)[0]
    np.savez_compressed(os.path.join(OUT, "baseline_agents.npz"), poses=poses, pursue_delta=delta, pursue_obs=obs,
                        man_poses=mposes, man_delta=mdelta, man_obs=mobs, man_turn_interval=1.5, time_interval=env.time_interval)


def gen_rollout_buffer(rng):
    """ReplayBuffer / SharedReplayBuffer (algorithms/utils/buffer.py): inserts with episode ends and time-limit ends, the four
    compute_returns modes, the normalised advantages, after_update, and the mini-batches of recurrent_generator for a recorded
    chunk permutation (torch.randperm is replaced by that recorded permutation while the generator runs)."""
    import torch
    from algorithms.utils.buffer import ReplayBuffer, SharedReplayBuffer
    sp = sys.modules["gymnasium.spaces"]
    T, E, A, OBS, SH, H = 12, 3, 2, 5, 10, 4
    obs_space, share_space = sp.Box(low=-10, high=10, shape=(OBS,)), sp.Box(low=-10, high=10, shape=(SH,))
    act_space = sp.MultiDiscrete([41, 41, 41, 30])
    out = {"dims": np.array([T, E, A, OBS, SH, 4, 1, H])}
    f32 = lambda x: x.astype(np.float32)
    stream = {
        "obs": f32(rng.normal(size=(T + 1, E, A, OBS))), "share_obs": f32(rng.normal(size=(T + 1, E, A, SH))),
        "actions": f32(rng.integers(0, 30, size=(T, E, A, 4))), "rewards": f32(rng.normal(size=(T, E, A, 1)) * 3),
        "masks": f32(rng.random((T, E, A, 1)) > 0.15), "bad_masks": f32(rng.random((T, E, A, 1)) > 0.1),
        "active_masks": f32(rng.random((T, E, A, 1)) > 0.2),
        "logp": f32(rng.normal(size=(T, E, A, 1))), "logp_shared": f32(rng.normal(size=(T, E, A, 4))),
        "values": f32(rng.normal(size=(T, E, A, 1))), "next_value": f32(rng.normal(size=(E, A, 1))),
        "rnn_a": f32(rng.normal(size=(T + 1, E, A, 1, H))), "rnn_c": f32(rng.normal(size=(T + 1, E, A, 1, H))),
    }
    out.update({"in_" + k: v for k, v in stream.items()})
    L, MB = 4, 3                                   # data_chunk_length, num_mini_batch: 36 rows per column set -> 9 chunks, 3 per batch
    perm = rng.permutation(T * E * A // L)
    out["perm"], out["chunk"] = perm, np.array([L, MB])
    real_randperm = torch.randperm
    for shared in (False, True):
        for proper in (False, True):
            for gae in (False, True):
                args = types.SimpleNamespace(buffer_size=T, n_rollout_threads=E, gamma=0.99, use_proper_time_limits=proper, use_gae=gae,
                                             gae_lambda=0.95, recurrent_hidden_size=H, recurrent_hidden_layers=1)
                buf = SharedReplayBuffer(args, A, obs_space, share_space, act_space) if shared else ReplayBuffer(args, A, obs_space, act_space)
                buf.obs[0] = stream["obs"][0]
                buf.rnn_states_actor[0], buf.rnn_states_critic[0] = stream["rnn_a"][0], stream["rnn_c"][0]
                if shared:
                    buf.share_obs[0] = stream["share_obs"][0]
                for t in range(T):
                    kw = dict(obs=stream["obs"][t + 1], actions=stream["actions"][t], rewards=stream["rewards"][t], masks=stream["masks"][t],
                              action_log_probs=stream["logp_shared" if shared else "logp"][t], value_preds=stream["values"][t],
                              rnn_states_actor=stream["rnn_a"][t + 1], rnn_states_critic=stream["rnn_c"][t + 1], bad_masks=stream["bad_masks"][t])
                    if shared:
                        kw.update(share_obs=stream["share_obs"][t + 1], active_masks=stream["active_masks"][t])
                    buf.insert(**kw)
                assert buf.step == 0
                buf.compute_returns(stream["next_value"])
                key = f"{'shared' if shared else 'single'}_{'proper' if proper else 'plain'}_{'gae' if gae else 'mc'}"
                out[key + "_returns"] = buf.returns.copy()
                out[key + "_advantages"] = buf.advantages.copy()
                if proper and gae:      # one mode per buffer class is enough for the data movement
                    out[key + "_masks"], out[key + "_bad_masks"] = buf.masks.copy(), buf.bad_masks.copy()
                    torch.randperm = lambda n: torch.from_numpy(perm.copy())
                    try:
                        if shared:
                            batches = list(buf.recurrent_generator(buf.advantages, MB, L))
                        else:
                            batches = list(ReplayBuffer.recurrent_generator(buf, MB, L))
                    finally:
                        torch.randperm = real_randperm
                    for b, batch in enumerate(batches):
                        for j, arr in enumerate(batch):
                            out[f"{key}_batch{b}_{j}"] = np.asarray(arr)
                    buf.after_update()
                    out[key + "_after_obs0"], out[key + "_after_masks0"] = buf.obs[0].copy(), buf.masks[0].copy()
    np.savez_compressed(os.path.join(OUT, "rollout_buffer.npz"), **out)


def gen_acmi_records():
    """The reference's own Tacview records for a scripted sequence: BaseSimulator.log for two aircraft,
    MissileSimulator.log for an AIM-120B flown by the reference's missile engine until it is done (position records, then removal +
    explosion once, then removal), ChaffSimulator.log for a cloud released on frame 1 (alive, then removed after 20 s), the frame
    layout of BaseEnv.render (env_base.py:207-250). Stored: the inputs of every frame and the text the reference produced."""
    from envs.JSBSim.core.simulatior import AIM_120B, BaseSimulator, ChaffSimulator
    a, b = FakeAircraft("A0100", "Blue"), FakeAircraft("B0100", "Red")
    a.model = b.model = "f16"
    a.set_pose(120.0, 60.0, 6000.0, (0.1, 0.05, 0.3), (240.0, 60.0, -5.0))
    b.set_pose(120.01, 60.02, 6100.0, (-0.2, 0.02, 3.3), (-250.0, -20.0, 1.0))
    m = AIM_120B.create(a, b, "A01002", "AIM-120B")
    chaff = None
    frames, inputs = [], []
    tpos = b._position.copy()
    done_frames = 0
    for f in range(200):
        for _ in range(6):                                  # agent_interaction_steps substeps of 1/60 s
            tpos += a.dt * np.array([b._velocity[0], b._velocity[1], -b._velocity[2]])
            b._position[:] = tpos
            m.run()
            if chaff is not None:
                chaff.run()
        from envs.JSBSim.utils.utils import NEU2LLA
        b._geodetic[:] = NEU2LLA(*b._position, b.lon0, b.lat0, b.alt0)   # (AircraftSimulator.run refreshes the cache every substep)
        if f == 1:                                           # task.step releases chaff after the substeps (scenario1_task.py:97-103)
            chaff = ChaffSimulator.create(parent=b, uid="B010012", chaff_model="CHF")
        if f == 30 and chaff is not None:
            chaff._t = 25.0                                  # past its 20 s life: the next run() ends it
        lines = [f"#{(f + 1) * 0.1:.2f}"]
        for sim in (a, b):                                   # env._jsbsims, then _tempsims, then _chaffsims
            lines.append(BaseSimulator.log(sim))
        msg = m.log()
        if msg is not None:
            lines.append(msg)
        if chaff is not None:
            lines.append(chaff.log())
        frames.append("\n".join(lines) + "\n")             # render() writes log_msg + "\n" per simulator
        inputs.append(np.concatenate([a._geodetic, a._posture, b._geodetic, b._posture, [m._MissileSimulator__status], m.get_position(),
                                      m.get_rpy(), [0 if chaff is None else (1 if chaff.is_alive else 2)]]))
        done_frames += 1 if m.is_done else 0
        if done_frames >= 3 and f > 34:
            break
    np.savez_compressed(os.path.join(OUT, "acmi_records.npz"), frames=np.array(frames), inputs=np.array(inputs),
                        center=np.array([a.lon0, a.lat0, a.alt0]), missile_radius=np.array([m._Rc]))


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not present: golden vectors are generated in the build container only")
    install_standins()
    if len(sys.argv) > 1 and sys.argv[1] == "acmi":         # this fixture alone
        gen_acmi_records()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "artillery":
        gen_artillery(np.random.default_rng(86))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "multicombat_dodge":
        gen_multicombat_dodge(np.random.default_rng(87))
        return
    rng = np.random.default_rng(20250321)
    gen_geometry(rng)
    gen_reward_functions(rng)
    gen_singlecombat_sequences(rng)
    gen_missile(rng)
    gen_missile_task_obs(rng)
    gen_heading(rng)
    gen_curriculum_table()
    gen_multicombat_sequences(np.random.default_rng(77))
    gen_scenario_sequences(np.random.default_rng(78))
    gen_baseline_actor(np.random.default_rng(79))
    gen_baseline_agents(np.random.default_rng(80))
    gen_rwr_obs(np.random.default_rng(81))
    gen_wvr_sequences(np.random.default_rng(82))
    gen_wvr_sequences(np.random.default_rng(83), which="maneuver")
    gen_rollout_buffer(np.random.default_rng(84))
    gen_approach(np.random.default_rng(85))
    gen_acmi_records()
    gen_artillery(np.random.default_rng(86))
    gen_multicombat_dodge(np.random.default_rng(87))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
