"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes access to oracle/_build/liboracle.so, the CPU (plain C, float64) restatement of the reference's step() path
(oracle/f16_fdm.c, geodesy.c, combat_env.c). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product (aircombat-selfplay_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
OR_MAX_AC = 8
TASK_HEADING, TASK_SINGLECOMBAT, TASK_DODGE_MISSILE, TASK_SHOOT_MISSILE, TASK_MULTICOMBAT = 0, 1, 2, 3, 4
TASK_SCENARIO1, TASK_SCENARIO_NVN, TASK_WVR, TASK_MANEUVER = 5, 6, 7, 8
STATE_LEN = 80


class F16Init(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lon_deg", "lat_geod_deg", "h_sl_ft", "psi_deg", "u_fps", "v_fps", "w_fps", "p", "q", "r")]


class OrEnvConfig(C.Structure):
    _fields_ = [
        ("task", C.c_int), ("n_aircraft", C.c_int), ("n_ego", C.c_int), ("sim_freq", C.c_int),
        ("agent_interaction_steps", C.c_int), ("max_steps", C.c_int),
        ("center_lon", C.c_double), ("center_lat", C.c_double), ("center_alt", C.c_double),
        ("altitude_limit", C.c_double), ("acc_limit_x", C.c_double), ("acc_limit_y", C.c_double), ("acc_limit_z", C.c_double),
        ("init", F16Init * OR_MAX_AC), ("num_missiles", C.c_int * OR_MAX_AC),
        ("posture_scale", C.c_double), ("posture_potential", C.c_int),
        ("altitude_scale", C.c_double), ("altitude_potential", C.c_int),
        ("event_scale", C.c_double), ("event_potential", C.c_int),
        ("heading_scale", C.c_double), ("heading_potential", C.c_int),
        ("missile_posture_scale", C.c_double),
        ("shoot_penalty_scale", C.c_double), ("shoot_penalty_potential", C.c_int),
        ("alt_safe", C.c_double), ("alt_danger", C.c_double), ("alt_kv", C.c_double),
        ("max_attack_angle", C.c_double), ("max_attack_distance", C.c_double), ("min_attack_interval", C.c_int),
        ("max_heading_increment", C.c_double), ("max_altitude_increment", C.c_double),
        ("max_velocities_u_increment", C.c_double), ("check_interval", C.c_double),
        ("use_artillery", C.c_int),
        ("relative_altitude_scale", C.c_double), ("relative_altitude_KH", C.c_double), ("gun_scale", C.c_double),
        ("chaff_seed", C.c_uint64),
        ("legacy_obs", C.c_int),
        ("rwr", C.c_int),
        ("use_baseline", C.c_int),
        ("hierarchical", C.c_int),
        ("approach", C.c_int),
    ]


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.or_env_sizeof.restype = C.c_size_t
        L.or_env_config_sizeof.restype = C.c_size_t
        L.or_env_default_config.argtypes = [C.POINTER(OrEnvConfig), C.c_int]
        L.or_env_init.argtypes = [C.c_void_p, C.POINTER(OrEnvConfig)]
        L.or_env_seed_pcg64.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.or_env_uniform.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.or_env_uniform.restype = C.c_double
        L.or_env_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.or_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.or_env_obs_dim.argtypes = [C.c_int]
        L.or_env_act_dim.argtypes = [C.c_int]
        L.or_env_act_dim_h.argtypes = [C.c_int, C.c_int]
        L.or_env_obs_dim_n.argtypes = [C.c_int, C.c_int]
        L.or_state_export.argtypes = [C.c_void_p, C.c_int, dp, C.c_int]
        L.or_state_import.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_get_pose.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_num_missiles.argtypes = [C.c_void_p]
        L.or_env_get_missile.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_status.argtypes = [C.c_void_p, C.c_int]
        L.or_env_set_status.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.or_env_bloods.argtypes = [C.c_void_p, C.c_int]
        L.or_env_bloods.restype = C.c_double
        L.or_env_set_bloods.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.or_get_AO_TA_R.argtypes = [dp, dp, C.c_int, dp]
        L.or_posture_reward.argtypes = [C.c_double] * 3
        L.or_posture_reward.restype = C.c_double
        L.or_altitude_reward.argtypes = [C.c_double] * 5
        L.or_altitude_reward.restype = C.c_double
        L.or_lla2neu.argtypes = [C.c_double] * 6 + [dp]
        L.or_neu2lla.argtypes = [C.c_double] * 6 + [dp]
        L.f16_atmosphere.argtypes = [C.c_double] + [dp] * 5
        L.f16_atmosphere_bias.argtypes = [C.c_double, C.c_double] + [dp] * 6
        L.f16_test_aero_sums.argtypes = [dp, dp]
        L.f16_test_pid.argtypes = [dp] + [C.c_double] * 6
        L.f16_test_pid.restype = C.c_double
        L.f16_test_turbine_run.argtypes = [dp, C.c_double, C.c_double, C.c_double]
        L.f16_tab1.argtypes = [C.c_int, C.c_int, C.c_double]
        L.f16_tab1.restype = C.c_double
        L.f16_tab2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.f16_tab2.restype = C.c_double
        L.f16_vcas_from_mach.argtypes = [C.c_double, C.c_double]
        L.f16_vcas_from_mach.restype = C.c_double
        L.f16_kinemat.argtypes = [C.c_double, C.c_double, dp, dp, C.c_int, C.c_double]
        L.f16_kinemat.restype = C.c_double
        L.f16_test_fcs.argtypes = [dp, dp, dp]
        L.f16_test_fcs.restype = None
        L.f16_test_massbalance.argtypes = [dp, dp, dp, dp]
        L.f16_test_massbalance.restype = None
        L.f16_test_pilot_accel.argtypes = [dp] * 6
        L.f16_test_pilot_accel.restype = None
        L.f16_test_aero_frame.argtypes = [C.c_double, C.c_double] + [dp] * 5
        L.f16_test_aero_frame.restype = None
        L.f16_test_thruster_moment.argtypes = [dp] * 3
        L.f16_test_thruster_moment.restype = None
        L.or_posture_orientation.argtypes = [C.c_double] * 2
        L.or_posture_orientation.restype = C.c_double
        L.or_posture_range.argtypes = [C.c_double]
        L.or_posture_range.restype = C.c_double
        L.or_env_task_reset.argtypes = [C.c_void_p]
        L.or_env_evaluate.argtypes = [C.c_void_p] * 5
        L.or_env_set_pose.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_set_step.argtypes = [C.c_void_p, C.c_int]
        L.or_env_add_missile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, dp, dp]
        L.or_env_clear_missiles.argtypes = [C.c_void_p]
        L.or_env_heading_targets.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.or_env_heading_pose.argtypes = [C.c_void_p] + [C.c_double] * 11
        L.or_env_heading_get.argtypes = [C.c_void_p, dp]
        L.or_missile_raw_run.argtypes = [dp, C.c_int, dp, dp, C.c_int]
        L.or_env_set_shoot4.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.or_env_task_step.argtypes = [C.c_void_p]
        L.or_env_run_projectiles.argtypes = [C.c_void_p, C.c_int]
        L.or_env_get_counters.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_get_misc.argtypes = [C.c_void_p, dp]
        L.or_bench_run.argtypes = [C.POINTER(OrEnvConfig), C.c_int, C.c_int, C.c_uint64, dp, C.POINTER(C.c_long)]
        L.or_bench_run.restype = C.c_long
        L.or_actor_load.argtypes = [C.c_char_p]
        L.or_actor_forward.argtypes = [dp, dp, C.POINTER(C.c_int), dp]
        L.or_actor_forward.restype = None
        L.or_env_get_rnn.argtypes = [C.c_void_p, C.c_int, dp, C.POINTER(C.c_int)]
        L.or_env_set_rnn.argtypes = [C.c_void_p, C.c_int, dp]
        L.or_env_get_ctl_gaps.argtypes = [C.c_void_p, C.c_int, dp]
        assert L.or_env_config_sizeof() == C.sizeof(OrEnvConfig), "OrEnvConfig layout mismatch"
        _lib = L
    return _lib


def default_config(task):
    c = OrEnvConfig()
    lib().or_env_default_config(C.byref(c), task)
    return c


def copy_config(cfg):
    c = OrEnvConfig()
    C.memmove(C.byref(c), C.byref(cfg), C.sizeof(OrEnvConfig))
    return c


def config_from_ac(ac_cfg):
    """Oracle config with the same scalars as a product AcConfig (duck-typed: same field names where they overlap)."""
    c = default_config(int(ac_cfg.task))
    c.n_aircraft = ac_cfg.n_agents
    for name in ("n_ego", "sim_freq", "agent_interaction_steps", "max_steps", "center_lon", "center_lat", "center_alt",
                 "altitude_limit", "acc_limit_x", "acc_limit_y", "acc_limit_z", "posture_scale", "posture_potential",
                 "altitude_scale", "altitude_potential", "event_scale", "event_potential", "missile_posture_scale",
                 "shoot_penalty_scale", "shoot_penalty_potential", "alt_safe", "alt_danger", "alt_kv", "max_attack_angle",
                 "max_attack_distance", "min_attack_interval", "use_artillery", "hierarchical", "heading_scale", "heading_potential",
                 "max_heading_increment", "max_altitude_increment", "max_velocities_u_increment", "check_interval", "use_baseline", "rwr", "legacy_obs",
                 "approach"):
        setattr(c, name, getattr(ac_cfg, name))
    for i in range(OR_MAX_AC):
        src, dst = ac_cfg.init[i], c.init[i]
        dst.lon_deg, dst.lat_geod_deg, dst.h_sl_ft, dst.psi_deg = src.lon_deg, src.lat_geod_deg, src.h_sl_ft, src.psi_deg
        dst.u_fps, dst.v_fps, dst.w_fps = src.u_fps, src.v_fps, src.w_fps
        dst.p, dst.q, dst.r = src.p_rad_sec, src.q_rad_sec, src.r_rad_sec
        c.num_missiles[i] = ac_cfg.num_missiles[i]
    return c


class OracleEnv:
    """One env instance of the CPU restatement (BaseEnv semantics: reset() / step() without auto-reset)."""

    def __init__(self, cfg, pcg64_state=None):
        L = lib()
        self.L = L
        self.cfg = cfg
        self._buf = C.create_string_buffer(L.or_env_sizeof())
        self.p = C.cast(self._buf, C.c_void_p)
        L.or_env_init(self.p, C.byref(cfg))
        self.A = cfg.n_aircraft
        self.obs_dim = L.or_env_obs_dim_n(cfg.task, cfg.n_aircraft) + (2 if cfg.rwr else 0)
        if cfg.task == TASK_SCENARIO_NVN and cfg.legacy_obs:
            self.obs_dim = 21
        self.act_dim = L.or_env_act_dim_h(cfg.task, cfg.hierarchical)
        if cfg.task == TASK_MULTICOMBAT and cfg.legacy_obs:   # hierarchical_multiplecombat_shoot (or_env_init)
            self.obs_dim = 21
            self.act_dim = 4 if cfg.hierarchical else 5
        if cfg.hierarchical:
            actor_load()
        if pcg64_state is not None:
            self.seed_from_numpy(pcg64_state)

    def seed_from_numpy(self, bitgen_state):
        st, inc = bitgen_state["state"]["state"], bitgen_state["state"]["inc"]
        m = (1 << 64) - 1
        self.L.or_env_seed_pcg64(self.p, st >> 64, st & m, inc >> 64, inc & m)

    def reset(self):
        obs = np.zeros((self.A, self.obs_dim))
        self.L.or_env_reset(self.p, obs.ctypes.data)
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.A, self.act_dim)
        obs = np.zeros((self.A, self.obs_dim))
        rew = np.zeros(self.A)
        done = np.zeros(self.A, dtype=np.uint8)
        info = np.zeros(4, dtype=np.int32)
        self.L.or_env_step(self.p, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, info.ctypes.data)
        return obs, rew, done.astype(bool), info

    def task_reset(self):
        self.L.or_env_task_reset(self.p)

    def evaluate(self):
        obs = np.zeros((self.A, self.obs_dim))
        rew = np.zeros(self.A)
        done = np.zeros(self.A, dtype=np.uint8)
        info = np.zeros(4, dtype=np.int32)
        self.L.or_env_evaluate(self.p, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, info.ctypes.data)
        return obs, rew, done.astype(bool), info

    def set_pose(self, i, pose):
        buf = (C.c_double * 20)(*[float(v) for v in pose])
        self.L.or_env_set_pose(self.p, i, buf)

    def set_step(self, k):
        self.L.or_env_set_step(self.p, int(k))

    def add_missile(self, parent, target, model, pos, vel):
        return self.L.or_env_add_missile(self.p, parent, target, model, (C.c_double * 3)(*pos), (C.c_double * 3)(*vel))

    def export_state(self, i):
        out = (C.c_double * STATE_LEN)()
        self.L.or_state_export(self.p, i, out, STATE_LEN)
        return np.array(out[:])

    def task_record(self, i):
        """The task bookkeeping of aircraft i by name (the tail of or_state_export's vector)."""
        out = (C.c_double * STATE_LEN)()
        k = self.L.or_state_export(self.p, i, out, STATE_LEN)
        v = out[:k]
        names = ("status", "die_flag", "remaining", "pre_remaining", "shoot_action", "last_missile", "last_shoot_time", "lock_bits", "lock_pos", "cur_step")
        rec = {nm: int(round(v[k - 10 + j])) for j, nm in enumerate(names)}
        rec["bloods"] = v[k - 17]
        return rec

    def import_state(self, i, vec):
        buf = (C.c_double * STATE_LEN)(*[float(v) for v in vec])
        self.L.or_state_import(self.p, i, buf)

    def pose(self, i):
        out = (C.c_double * 12)()
        self.L.or_env_get_pose(self.p, i, out)
        return np.array(out[:])

    def missiles(self):
        n = self.L.or_env_num_missiles(self.p)
        res = []
        for k in range(n):
            out = (C.c_double * 14)()
            self.L.or_env_get_missile(self.p, k, out)
            res.append(np.array(out[:]))
        return res

    def get_rnn(self, i):
        """(hidden[128], low_action[4]) of the low-level controller for aircraft i (hierarchical tasks)."""
        h = np.zeros(128)
        low = (C.c_int * 4)()
        self.L.or_env_get_rnn(self.p, i, h.ctypes.data_as(C.POINTER(C.c_double)), low)
        return h, np.array(low[:])

    def ctl_gaps(self, i):
        """Top-two logit gap of each of the controller's four heads at its last call for aircraft i."""
        g = np.zeros(4)
        self.L.or_env_get_ctl_gaps(self.p, i, g.ctypes.data_as(C.POINTER(C.c_double)))
        return g

    def set_rnn(self, i, h):
        h = np.ascontiguousarray(h, dtype=np.float64)
        self.L.or_env_set_rnn(self.p, i, h.ctypes.data_as(C.POINTER(C.c_double)))

    def status(self, i):
        return self.L.or_env_status(self.p, i)

    def set_status(self, i, s):
        self.L.or_env_set_status(self.p, i, s)

    def set_bloods(self, i, b):
        self.L.or_env_set_bloods(self.p, i, b)


class OracleVecEnv:
    """E oracle envs behind the reference's VecEnv semantics (auto-reset when every agent is done)."""

    def __init__(self, cfg, num_envs, chaff_seed=None, env_ids=None):
        """``env_ids``: which envs of a larger batch these are (a sample of a full-size product batch replayed here)."""
        self.envs = []
        for e in range(num_envs):
            if chaff_seed is not None:   # the product keys its decoy draws with seed + env index
                cfg = copy_config(cfg)
                cfg.chaff_seed = chaff_seed + (env_ids[e] if env_ids is not None else e)
            self.envs.append(OracleEnv(cfg))
        self.num_envs = num_envs
        self.num_agents = cfg.n_aircraft
        self.obs_dim, self.act_dim = self.envs[0].obs_dim, self.envs[0].act_dim

    def reset(self):
        return np.stack([e.reset() for e in self.envs])

    def step(self, actions):
        obs, rew, done, info = [], [], [], []
        for e, a in zip(self.envs, actions):
            o, r, d, i = e.step(a)
            if i[3]:
                o = e.reset()
            obs.append(o); rew.append(r); done.append(d); info.append(i)
        return np.stack(obs), np.stack(rew)[..., None], np.stack(done)[..., None], np.stack(info)


ACTOR_WEIGHTS = os.path.join(os.path.dirname(HERE), "aircombat-selfplay_amd", "data", "baseline_actor.f32")


def actor_load(path=ACTOR_WEIGHTS):
    rc = lib().or_actor_load(path.encode())
    if rc != 0:
        raise RuntimeError(f"or_actor_load({path}) -> {rc}")


def actor_forward(x, h):
    """One call of the low-level controller: x[12], h[128] -> (action[4], new h[128], logits[153])."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.array(h, dtype=np.float64).ravel().copy()
    act = (C.c_int * 4)()
    logits = np.zeros(153)
    dp = C.POINTER(C.c_double)
    lib().or_actor_forward(x.ctypes.data_as(dp), h.ctypes.data_as(dp), act, logits.ctypes.data_as(dp))
    return np.array(act[:]), h, logits


def bench_run(cfg, n_envs, steps, seed=20250321):
    """Time the CPU restatement: returns (agent_steps, seconds, episodes). Single thread."""
    sec = C.c_double()
    eps = C.c_long()
    n = lib().or_bench_run(C.byref(cfg), n_envs, steps, seed, C.byref(sec), C.byref(eps))
    return n, sec.value, eps.value
