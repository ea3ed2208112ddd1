"""Pins the oracle's restatement of the reference's PYTHON code against golden vectors generated from that code
(tests/golden/make_golden.py, run in the build container where /root/reference exists). CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def close(a, b, rtol=1e-9, atol=1e-9):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(invalid="ignore"):
        ok = np.abs(a - b) <= atol + rtol * np.abs(b)
    return ok | both_nan | same_inf


def test_ao_ta_r_matches_reference(oracle):
    g = load("geometry.npz")
    L = oracle.lib()
    out = (C.c_double * 4)()
    for two_d, key in ((0, "out3d"), (1, "out2d")):
        got = []
        for e, n in zip(g["ego"], g["enm"]):
            L.or_get_AO_TA_R((C.c_double * 6)(*e), (C.c_double * 6)(*n), two_d, out)
            got.append(out[:])
        got = np.array(got)
        # acos near +-1 turns a 1e-16 difference of its argument (summation order of the norms) into ~1e-8 rad
        ok = close(got, g[key], rtol=1e-9, atol=5e-8)
        # acos is ill-conditioned at +-1: the degenerate rows (R -> 0, v = 0) are compared on R and side only
        degenerate = (np.linalg.norm(g["ego"][:, 3:], axis=1) == 0) | (np.linalg.norm(g["enm"][:, 3:], axis=1) == 0) | (g[key][:, 2] < 1e-6)
        assert ok[~degenerate].all()
        assert close(got[degenerate][:, 2:], g[key][degenerate][:, 2:], atol=1e-9).all()
        assert np.isfinite(got).all()


def test_posture_and_altitude_functions(oracle):
    g = load("reward_functions.npz")
    L = oracle.lib()
    orn = np.array([L.or_posture_orientation(a, t) for a, t in zip(g["AO"], g["TA"])])
    rng = np.array([L.or_posture_range(r) for r in g["R"]])
    alt = np.array([L.or_altitude_reward(z, vz, 4.0, 3.5, 0.2) for z, vz in zip(g["z"], g["vz"])])
    assert close(orn, g["orientation"], rtol=1e-12, atol=1e-12).all()
    assert close(rng, g["range"], rtol=1e-12, atol=1e-12).all()
    assert close(alt, g["altitude"], rtol=1e-12, atol=1e-12).all()


def test_singlecombat_obs_termination_reward_sequences(oracle):
    """obs(15), the five terminations in task order with their crash() side effects, potential rewards seeded at reset,
    die-flag latch — against SingleCombatTask driven by the same scripted poses."""
    g = load("singlecombat_sequences.npz")
    cfg = oracle.default_config(oracle.TASK_SINGLECOMBAT)
    n = int(g["n_episodes"][0])
    n_done = 0
    for ep in range(n):
        env = oracle.OracleEnv(cfg)
        pose, obs, rew, done, step = (g[f"ep{ep}_{k}"] for k in ("pose", "obs", "rew", "done", "step"))
        for i in range(2):
            env.set_pose(i, pose[0][i])
        env.set_step(0)
        env.task_reset()
        o0, _, _, _ = None, None, None, None
        o, r, d, _ = env.evaluate.__func__(env) if False else (None, None, None, None)
        # frame 0: observation at reset (no termination / reward evaluation happens at reset)
        obs0 = np.zeros((2, 15))
        env.L.or_env_evaluate  # noqa: B018 (symbol exists)
        for t in range(1, len(pose)):
            for i in range(2):
                # dead aircraft keep their status; pose vectors carry the status the reference had BEFORE evaluation
                env.set_pose(i, pose[t][i])
            env.set_step(int(step[t]))
            o, r, d, info = env.evaluate()
            assert close(o, obs[t], rtol=1e-9, atol=1e-9).all(), (ep, t, np.abs(o - obs[t]).max())
            assert (d == done[t].astype(bool)).all(), (ep, t, d, done[t])
            assert close(r, rew[t], rtol=1e-8, atol=1e-8).all(), (ep, t, r, rew[t])
            n_done += int(d.sum())
    assert n_done > 10  # the scripted events really exercised the terminations


def test_missile_flyouts(oracle):
    """MissileSimulator.run / _guidance / _state_trans per substep for both parameter sets: hit, tail chase, turning
    target, out of range (speed / receding miss) and target-dies-first."""
    g = load("missile.npz")
    L = oracle.lib()
    statuses = set()
    for c in range(int(g["n"][0])):
        model = int(g[f"c{c}_model"][0])
        rows = g[f"c{c}_rows"]
        par = g[f"c{c}_parent"]
        # launch state: parent's cached geodetic(3), position(3), velocity(3), rpy(3)
        st = np.zeros(20)
        st[0] = 0
        st[1:4] = par[3:6]; st[4:7] = par[6:9]; st[7] = par[10]; st[8] = par[11]
        st[9] = 0.0; st[10] = 84.0 if model == 0 else 152.0; st[13] = np.inf; st[15] = par[2]
        buf = (C.c_double * 20)(*st)
        for row in rows:
            alive = int(row[1])
            tp, tv = (C.c_double * 3)(*row[3:6]), (C.c_double * 3)(*row[6:9])
            L.or_missile_raw_run(buf, model, tp, tv, alive)
            got = np.array(buf[:])
            assert int(got[0]) == int(row[2]), (c, row[0], got[0], row[2])
            assert close(got[1:4], row[9:12], rtol=1e-9, atol=1e-6).all(), (c, row[0])
            assert close(got[4:7], row[12:15], rtol=1e-9, atol=1e-7).all(), (c, row[0])
            assert close(got[7:9], row[15:17], rtol=1e-9, atol=1e-9).all()
            assert close([got[9], got[10], got[15]], row[17:20], rtol=1e-9, atol=1e-6).all()
            statuses.add(int(row[2]))
    assert statuses == {0, 1, 2}


def test_missile_task_observation(oracle):
    g = load("missile_task_obs.npz")
    cfg = oracle.default_config(oracle.TASK_SHOOT_MISSILE)
    env = oracle.OracleEnv(cfg)
    for pose, obs, m in zip(g["pose"], g["obs"], g["missile"]):
        env.L.or_env_clear_missiles(env.p)
        for i in range(2):
            env.set_pose(i, pose[i])
        if m[0]:
            env.add_missile(1, 0, 0, m[1:4], m[4:7])
        o, _, _, _ = env.evaluate()
        assert close(o[0], obs, rtol=1e-9, atol=1e-9).all(), np.abs(o[0] - obs).max()


def test_heading_task_with_numpy_pcg64(oracle):
    """HeadingTask obs(12), HeadingReward + AltitudeReward, UnreachHeading incl. the env.np_random draws: the oracle's
    PCG64 mirror must reproduce numpy's Generator.uniform stream bit for bit."""
    g = load("heading.npz")
    cfg = oracle.default_config(oracle.TASK_HEADING)
    env = oracle.OracleEnv(cfg)
    st = g["pcg64"]
    env.L.or_env_seed_pcg64(env.p, int(st[0]), int(st[1]), int(st[2]), int(st[3]))
    hdg0, alt0, u0 = g["init"]
    env.L.or_env_heading_targets(env.p, hdg0, alt0, u0, 0.0)
    env.set_step(0)
    env.task_reset()
    env.L.or_env_heading_targets(env.p, hdg0, alt0, u0, 0.0)
    out = (C.c_double * 5)()
    turns = 0
    for row in g["rows"]:
        t, psi, h_ft, u_mps, roll, p, q, sim_time, done, rew, hc, th, ta, tu, ct = row[:15]
        env.L.or_env_heading_pose(env.p, psi, h_ft, u_mps, roll, 0.02, 0.5, 3.0, 200.0, p, q, sim_time)
        env.set_step(int(t))
        o, r, d, _ = env.evaluate()
        env.L.or_env_heading_get(env.p, out)
        assert close(o[0], row[15:], rtol=1e-9, atol=1e-9).all(), (t, o[0], row[15:])
        assert bool(d[0]) == bool(done), t
        assert close(r[0], rew, rtol=1e-9, atol=1e-9), (t, r[0], rew)
        assert close(out[:], [th, ta, tu, ct, hc], rtol=1e-12, atol=1e-9).all(), (t, out[:], [th, ta, tu, ct, hc])
        turns = int(hc)
    assert turns >= 2 and bool(done)


def test_approach_task_sequence(oracle):
    """ApproachTask (tasks/approach_task.py): heading observation, AltitudeReward alone, LowAltitude first among the terminations;
    a scripted descent through the safe / danger / limit altitudes."""
    g = load("approach.npz")
    cfg = oracle.default_config(oracle.TASK_HEADING)
    cfg.approach = 1
    env = oracle.OracleEnv(cfg)
    hdg0, alt0, u0 = g["init"]
    env.L.or_env_heading_vdown.argtypes = [C.c_void_p, C.c_double]
    env.L.or_env_heading_targets(env.p, hdg0, alt0, u0, 0.0)
    env.L.or_env_heading_pose(env.p, hdg0, alt0, u0, 0.0, -0.1, 0.5, 3.0, 200.0, 0.0, 0.0, 0.0)
    env.set_step(0)
    env.task_reset()
    env.L.or_env_heading_targets(env.p, hdg0, alt0, u0, 0.0)
    done = False
    for row in g["rows"]:
        t, psi, h_ft, u_mps, roll, sink, sim_time, done, rew = row[:9]
        env.L.or_env_heading_pose(env.p, psi, h_ft, u_mps, roll, -0.1, 0.5, 3.0, 200.0, 0.0, 0.0, sim_time)
        env.L.or_env_heading_vdown(env.p, sink)
        env.set_step(int(t))
        o, r, d, info = env.evaluate()
        assert close(o[0], row[9:], rtol=1e-9, atol=1e-9).all(), (t, o[0], row[9:])
        assert bool(d[0]) == bool(done), t
        assert close(r[0], rew, rtol=1e-9, atol=1e-9), (t, r[0], rew)
    assert bool(done) and g["rows"][:, 8].min() < -1.5      # the danger-zone term fired before the limit ended the episode


def test_pcg64_uniform_stream(oracle):
    gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(2024)))
    env = oracle.OracleEnv(oracle.default_config(oracle.TASK_HEADING))
    env.seed_from_numpy(gen.bit_generator.state)
    want = [gen.uniform(-3.0, 7.5) for _ in range(1000)]
    got = [env.L.or_env_uniform(env.p, -3.0, 7.5) for _ in range(1000)]
    assert want == got


def test_curriculum_spawn_table_fixture_is_sane():
    t = load("curriculum_spawn.npz")["table"]
    assert t.shape == (181, 3)
    # 11.119 km circle about (60.1 N, 120 E): first point due south, heading 0
    assert abs(t[0][0] - 60.0) < 2e-3 and abs(t[0][1] - 120.0) < 1e-9 and t[0][2] == 0


def test_multicombat_2v2_sequences(oracle):
    """MultipleCombatTask under MultipleCombatEnv.step's order: obs(27), rewards BEFORE terminations, reward only while alive,
    team-mean rewards, termination order SafeReturn first."""
    g = load("multicombat_sequences.npz")
    cfg = oracle.default_config(oracle.TASK_MULTICOMBAT)
    n_done = 0
    for ep in range(int(g["n_episodes"][0])):
        env = oracle.OracleEnv(cfg)
        pose, obs, rew, done, step = (g[f"ep{ep}_{k}"] for k in ("pose", "obs", "rew", "done", "step"))
        for i in range(4):
            env.set_pose(i, pose[0][i])
        env.set_step(0)
        env.task_reset()
        for t in range(1, len(pose)):
            for i in range(4):
                env.set_pose(i, pose[t][i])
            env.set_step(int(step[t]))
            o, r, d, info = env.evaluate()
            assert close(o, obs[t], rtol=1e-9, atol=5e-8).all(), (ep, t, np.abs(o - obs[t]).max())
            assert close(r, rew[t], rtol=1e-8, atol=1e-7).all(), (ep, t, r, rew[t])
            assert (d == done[t].astype(bool)).all(), (ep, t, d, done[t])
            n_done += int(d.sum())
    assert n_done > 10


def test_scenario_weapon_sequences(oracle):
    """Scenario1 (1v1) and Scenario2_NvN (2v2): gun / AIM-120B / AIM-9M / chaff rules with the shared last-munition gate and
    the uid-collision behaviour of env._tempsims, the decoy draw, the NvN observation layout, and all eleven reward terms
    with their never-cleared shared lists — against the reference's task objects driven by the same scripted engagements."""
    g = load("scenario_sequences.npz")
    launched = chaffs = hits = 0
    for ep in range(int(g["n_episodes"][0])):
        nvn = bool(g[f"ep{ep}_family"][0])
        cfg = oracle.default_config(oracle.TASK_SCENARIO_NVN if nvn else oracle.TASK_SCENARIO1)
        if nvn:
            cfg.event_potential = 0
        A = cfg.n_aircraft
        env = oracle.OracleEnv(cfg)
        pose, bits, obs, rew, done, counters, msl, misc = (g[f"ep{ep}_{k}"] for k in ("pose", "bits", "obs", "rew", "done", "counters", "msl", "misc"))
        # the generator resets the task on the poses of frame 0 *before* its first nudge; those poses equal frame 0's except
        # that frame 0 may have moved the non-engagement aircraft: seed the potentials the same way the generator did
        for t in range(len(pose)):
            for i in range(A):
                # pose vectors were captured at frame start: kinematics plus the status / bloods the reference had then (these
                # equal what the oracle carried over, as the counters assertion of the previous frame checked, plus scripted events)
                env.set_pose(i, pose[t][i])
            if t == 0:
                env.set_step(0)
                # task.reset happened on the initial poses, which the generator did not store separately when it re-posed
                # aircraft in frame 1; potentials are therefore checked from frame 2 on (see below)
                env.task_reset()
            env.set_step(int(misc[t][2]))
            env.L.or_env_run_projectiles(env.p, 6)
            for i in range(A):
                b = (C.c_int * 4)(*[int(x) for x in bits[t][i]])
                if nvn or i < cfg.n_ego:
                    env.L.or_env_set_shoot4(env.p, i, b)
            env.L.or_env_task_step(env.p)
            o, r, d, info = env.evaluate()
            got_c = np.zeros((A, 6))
            for i in range(A):
                env.L.or_env_get_counters(env.p, i, got_c[i].ctypes.data_as(C.POINTER(C.c_double)))
            assert (got_c == counters[t]).all(), (ep, t, got_c, counters[t])
            ms = env.missiles()
            assert len(ms) == int(misc[t][3]), (ep, t, len(ms), misc[t][3])
            for k, m in enumerate(ms[:12]):
                assert int(m[11]) == int(msl[t][k][0]) - 1 and int(m[12]) == int(msl[t][k][1]) and int(m[0]) == int(msl[t][k][2]), (ep, t, k)
                assert close(m[1:7], msl[t][k][3:9], rtol=1e-9, atol=1e-6).all(), (ep, t, k)
            gm = np.zeros(3); env.L.or_env_get_misc(env.p, gm.ctypes.data_as(C.POINTER(C.c_double)))
            assert gm[0] == misc[t][0] and gm[1] == misc[t][1], (ep, t, gm, misc[t])
            assert close(o, obs[t], rtol=1e-9, atol=5e-8).all(), (ep, t, np.abs(o - obs[t]).max())
            assert (d == done[t].astype(bool)).all(), (ep, t, d, done[t])
            if t >= 1:   # frame 0's potential difference depends on the un-stored reset poses
                assert close(r, rew[t], rtol=1e-7, atol=1e-6).all(), (ep, t, r, rew[t])
            launched = max(launched, len(ms)); chaffs = max(chaffs, int(gm[0])); hits += int(sum(1 for m in ms if int(m[0]) == 1))
    assert launched >= 4 and chaffs >= 2


def test_multicombat_dodge_missile_sequences(oracle):
    """MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-145; 2v2 and 4v4) against the reference's own task object
    over scripted engagements: the 21-value paired-enemy observation with the missile-warning block, the rule-based launch at
    enemies[0] (lock window, angle, distance, interval, rounds left, alive), the base-class missile under MultipleCombatEnv.step's
    substep loop (mutual kills, two aircraft firing at one target), Posture + MissilePosture + Altitude + EventDriven rewards with the
    team mean, MultipleCombatTask's terminations. The oracle runs it as OR_TASK_DODGE_MISSILE with more than two aircraft."""
    g = load("multicombat_dodge_sequences.npz")
    launched = kills = 0
    for ep in range(int(g["n_episodes"][0])):
        per_side = int(g[f"ep{ep}_per_side"][0])
        cfg = oracle.default_config(oracle.TASK_MULTICOMBAT)
        cfg.task = oracle.TASK_DODGE_MISSILE
        cfg.n_aircraft, cfg.n_ego = 2 * per_side, per_side
        cfg.event_potential = 0
        cfg.min_attack_interval = int(g["min_attack_interval"][0])
        for i in range(cfg.n_aircraft):
            cfg.num_missiles[i] = 2
        A = cfg.n_aircraft
        env = oracle.OracleEnv(cfg)
        assert env.obs_dim == 21 and env.act_dim == 4
        pose, obs, rew, done, counters, msl, misc = (g[f"ep{ep}_{k}"] for k in ("pose", "obs", "rew", "done", "counters", "msl", "misc"))
        for t in range(len(pose)):
            for i in range(A):
                env.set_pose(i, pose[t][i])
            if t == 0:
                env.set_step(0)
                env.task_reset()
            env.set_step(int(misc[t][0]))
            env.L.or_env_run_projectiles(env.p, 6)
            env.L.or_env_task_step(env.p)
            o, r, d, info = env.evaluate()
            for i in range(A):
                rec = env.task_record(i)
                window = bin(rec["lock_bits"]).count("1")
                got = [rec["remaining"], rec["last_shoot_time"], window, rec["bloods"], rec["status"]]
                assert got == list(counters[t][i]), (ep, t, i, got, counters[t][i])
            ms = env.missiles()
            assert len(ms) == int(misc[t][1]), (ep, t, len(ms), misc[t][1])
            for k, m in enumerate(ms[:16]):
                assert int(m[11]) == int(msl[t][k][0]) - 1 and int(m[12]) == int(msl[t][k][1]) and int(m[0]) == int(msl[t][k][2]), (ep, t, k, m[:1], msl[t][k][:3])
                assert close(m[1:7], msl[t][k][3:9], rtol=1e-9, atol=1e-6).all(), (ep, t, k)
            assert close(o, obs[t], rtol=1e-9, atol=5e-8).all(), (ep, t, np.abs(o - obs[t]).max())
            assert (d == done[t].astype(bool)).all(), (ep, t, d, done[t])
            if t >= 1:   # frame 0's potential difference depends on the un-stored reset poses
                assert close(r, rew[t], rtol=1e-7, atol=1e-6).all(), (ep, t, r, rew[t])
            launched = max(launched, len(ms))
        kills += int((counters[-1][:, 4] == 2).sum())
    assert launched >= 6 and kills >= 4


def test_lowlevel_controller_matches_reference_module(oracle):
    """BaselineActor restatement (oracle/lowlevel_actor.c) against outputs of the reference's own module over GRU sequences:
    logits and hidden state to fp32-summation accuracy (the reference computes in float32), argmax indices identical wherever
    the reference's top-two logits are further apart than that accuracy."""
    g = load("baseline_actor.npz")
    oracle.actor_load()
    S, T = g["x"].shape[:2]
    flips = total = 0
    for s in range(S):
        h = np.zeros(128)
        for t in range(T):
            act, h, logits = oracle.actor_forward(g["x"][s, t], h)
            assert np.abs(h - g["hidden"][s, t]).max() < 2e-5, (s, t, np.abs(h - g["hidden"][s, t]).max())
            assert np.abs(logits - g["logits"][s, t]).max() < 2e-4 * max(1.0, np.abs(g["logits"][s, t]).max()), (s, t)
            for hd, (a, b) in enumerate(((0, 41), (41, 82), (82, 123), (123, 153))):
                top2 = np.sort(g["logits"][s, t, a:b])[-2:]
                total += 1
                if act[hd] != g["action"][s, t, hd]:
                    assert top2[1] - top2[0] < 1e-4, (s, t, hd, act[hd], g["action"][s, t, hd], top2)
                    flips += 1
            h = g["hidden"][s, t].astype(np.float64)      # follow the reference's trajectory
    assert flips <= total // 200, (flips, total)


def test_scripted_opponents_match_reference(oracle):
    """PursueAgent / ManeuverAgent('triangle') of the use_baseline YAMLs: delta values and the 12 controller inputs from the
    reference's own classes on random poses; the maneuver sequence crosses several schedule entries with its latched heading."""
    g = load("baseline_agents.npz")
    cfg = oracle.default_config(oracle.TASK_SINGLECOMBAT)
    env = oracle.OracleEnv(cfg)
    env.reset()
    dp = C.POINTER(C.c_double)
    L = env.L
    L.or_env_pursue.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp]
    L.or_env_maneuver.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, dp, dp]
    for k in range(g["poses"].shape[0]):
        env.set_pose(0, g["poses"][k, 0]); env.set_pose(1, g["poses"][k, 1])
        dv, x = np.zeros(3), np.zeros(12)
        L.or_env_pursue(env.p, 1, 0, dv.ctypes.data_as(dp), x.ctypes.data_as(dp))
        assert np.allclose(dv, g["pursue_delta"][k], rtol=1e-9, atol=5e-8), (k, dv, g["pursue_delta"][k])
        assert np.allclose(x, g["pursue_obs"][k], rtol=1e-9, atol=5e-8), (k, x, g["pursue_obs"][k])
    env.task_reset()
    for t in range(g["man_poses"].shape[0]):
        env.set_pose(1, g["man_poses"][t])
        dv, x = np.zeros(3), np.zeros(12)
        L.or_env_maneuver(env.p, 1, float(g["man_turn_interval"]), float(g["time_interval"]), dv.ctypes.data_as(dp), x.ctypes.data_as(dp))
        assert np.allclose(dv, g["man_delta"][t], rtol=1e-9, atol=1e-9), (t, dv, g["man_delta"][t])
        assert np.allclose(x, g["man_obs"][t], rtol=1e-9, atol=1e-9), (t, x, g["man_obs"][t])


def test_rwr_observation_variants(oracle):
    """Scenario1_RWR (23 values, missile block blanked) and Scenario2_RWR (NvN layout + two reserved slots) against the
    reference's get_obs."""
    g = load("rwr_obs.npz")
    for fam, task, n in (("s1", oracle.TASK_SCENARIO1, 2), ("nvn", oracle.TASK_SCENARIO_NVN, 4), ("legacy", oracle.TASK_SCENARIO_NVN, 4)):
        cfg = oracle.default_config(task)
        cfg.rwr = int(fam != "legacy")
        cfg.legacy_obs = int(fam == "legacy")   # Scenario2 (not _NvN): 21 values against the enemy with the same team index
        env = oracle.OracleEnv(cfg)
        env.reset()
        assert env.obs_dim == g[f"{fam}_obs"].shape[-1]
        for k in range(g[f"{fam}_pose"].shape[0]):
            for i in range(n):
                env.set_pose(i, g[f"{fam}_pose"][k, i])
            env.L.or_env_clear_missiles(env.p)
            m = g[f"{fam}_missile"][k]
            if m[0]:
                env.add_missile(n - 1, 0, 0, m[1:4], m[4:7])
            obs, _, _, _ = env.evaluate()
            assert np.allclose(obs, g[f"{fam}_obs"][k], rtol=1e-9, atol=5e-8), (fam, k, np.abs(obs - g[f"{fam}_obs"][k]).max())


@pytest.mark.parametrize("which", ["wvr", "maneuver"])
def test_wvr_task_sequences(oracle, which):
    """WVRTask: 15-value observation, unlimited gun on the farthest enemy (no aliveness checks), eight reward terms with the shared
    reference lists, terminations LowAltitude / ExtremeState / Overload / Timeout only (a shot-down aircraft is not 'done').
    Maneuver_curriculum: the same gun, nine reward terms, the ordinary 1v1 terminations."""
    g = load(f"{which}_sequences.npz")
    shot = crashed = 0
    for ep in range(int(g["n_episodes"][0])):
        cfg = oracle.default_config(oracle.TASK_WVR if which == "wvr" else oracle.TASK_MANEUVER)
        cfg.max_steps = int(g["max_steps"][0])
        env = oracle.OracleEnv(cfg)
        pose, obs, rew, done, state, step = (g[f"ep{ep}_{k}"] for k in ("pose", "obs", "rew", "done", "state", "step"))
        for t in range(len(pose)):
            for i in range(2):
                env.set_pose(i, pose[t][i])
            if t == 0:
                env.set_step(0)
                env.task_reset()
            env.set_step(int(step[t]))
            env.L.or_env_run_projectiles(env.p, 0)
            env.L.or_env_task_step(env.p)
            o, r, d, info = env.evaluate()
            assert close(o, obs[t], rtol=1e-9, atol=5e-8).all(), (ep, t, np.abs(o - obs[t]).max())
            assert (d == done[t].astype(bool)).all(), (ep, t, d, done[t])
            got = np.zeros((2, 6))
            for i in range(2):
                env.L.or_env_get_counters(env.p, i, got[i].ctypes.data_as(C.POINTER(C.c_double)))
            assert (got[:, 4:6] == state[t]).all(), (ep, t, got[:, 4:6], state[t])
            if t >= 1:
                assert close(r, rew[t], rtol=1e-7, atol=1e-6).all(), (ep, t, r, rew[t])
            shot += int((state[t][:, 1] == 2).any()); crashed += int((state[t][:, 1] == 1).any())
    assert shot and crashed


def test_artillery_blood_drain_of_singlecombat_task_step(oracle):
    """SingleCombatTask.step with use_artillery (singlecombat_task.py:162-188): 480 scripted frames from the reference itself -- targets
    inside / on / outside the 30 deg cone and the 1 km / 3 km bands, dead targets (no drain), dead shooters (still drain)."""
    g = load("artillery.npz")
    cfg = oracle.default_config(oracle.TASK_SINGLECOMBAT)
    cfg.use_artillery = 1
    env = oracle.OracleEnv(cfg)
    L = oracle.lib()
    hits = 0
    for pose, want in zip(g["pose"], g["bloods_after"]):
        for i in range(2):
            env.set_pose(i, pose[i])
        L.or_env_task_step(env.p)
        got = np.array([L.or_env_bloods(env.p, i) for i in range(2)])
        assert close(got, want, rtol=1e-9, atol=1e-9).all(), (pose[:, 17:19], got, want)
        hits += int((pose[:, 18] - want > 0).sum())
    assert hits > 150
    # without the flag the same frames leave the blood alone
    cfg.use_artillery = 0
    env = oracle.OracleEnv(cfg)
    for pose in g["pose"][:40]:
        for i in range(2):
            env.set_pose(i, pose[i])
        L.or_env_task_step(env.p)
        assert [L.or_env_bloods(env.p, i) for i in range(2)] == list(pose[:, 18])
