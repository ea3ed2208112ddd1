"""The F-16's own wiring in the FDM oracle — fcs_run() (f16.xml:317-992 read by hand), massbalance_run() (f16.xml:62-92,272-316) and the
pilot-station acceleration — against an independent GENERIC reading of the same XML (tests/golden/f16_fcs_check.npz, written by
tests/golden/make_f16_fcs_check.py: the <flight_control> section interpreted component by component in document order with the
semantics of the reference's FGSwitch / FGGain / FGSummer / FGPID / FGKinemat / FGFCSFunction sources; <metrics>, <mass_balance> and the
tanks evaluated per FGMassBalance.cpp:181-262), plus the expectations the reference's JSBSim unit tests make about these blocks
(tests/golden/jsbsim_relations.npz, make_jsbsim_relations.py)."""
import ctypes as C
import importlib.util
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dbl(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def fcs():
    return np.load(os.path.join(GOLD, "f16_fcs_check.npz"))


@pytest.fixture(scope="module")
def gen():
    """The generator module, for expand_inputs() only (knots -> per-tick inputs); importing it does not touch the reference tree."""
    spec = importlib.util.spec_from_file_location("make_f16_fcs_check", os.path.join(GOLD, "make_f16_fcs_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_flight_control_component_inventory(fcs):
    """What the generic reading found in <flight_control>: the section the oracle restates has exactly these components."""
    counts = {k: int(v) for k, v in fcs["component_counts"]}
    assert counts == {"summer": 13, "switch": 11, "pure_gain": 9, "kinematic": 9, "aerosurface_scale": 7, "scheduled_gain": 5, "pid": 3, "fcs_function": 1}


def test_fcs_run_against_the_generic_reading_of_flight_control(oracle, fcs, gen):
    """512 input sequences x 200 ticks (every switch branch, every clip, PID triggers 0 and 1, every kinematic in motion; 64 of them
    checked at every tick, the rest every 10th tick — the section's memory carries any earlier difference forward): every named surface,
    the throttle position, the three PID outputs and the actuator positions, 1e-9."""
    L = oracle.lib()
    knots, linear, gear0 = fcs["knots"], fcs["linear"], fcs["gear_pos0"]
    full, sampled = fcs["out_full"], fcs["out_sampled"]
    assert [str(p) for p in fcs["in_props"]] == gen.IN_PROPS and [str(p) for p in fcs["out_props"]] == gen.OUT_PROPS
    assert float(fcs["fcs_dt"]) == 1.0 / 120.0
    nfull, every = full.shape[0], gen.KNOT_EVERY
    worst = 0.0
    out, outp = dbl(np.zeros(16))
    for s in range(knots.shape[0]):
        X = gen.expand_inputs(knots[s], linear[s])
        st, stp = dbl(np.zeros(18))
        st[17] = gear0[s]                        # gear/gear-pos-norm starts DOWN (FGFCS.cpp:81) unless the sequence set it
        for t in range(X.shape[0]):
            x, xp = dbl(X[t])
            L.f16_test_fcs(stp, xp, outp)
            if s < nfull:
                want = full[s, t]
            elif t % every == every - 1:
                want = sampled[s - nfull, t // every]
            else:
                continue
            err = np.abs(out - want) / np.maximum(1.0, np.abs(want))
            worst = max(worst, float(err.max()))
            assert err.max() <= 1e-9, (s, t, [str(p) for p in fcs["out_props"][err > 1e-9]], out[err > 1e-9], want[err > 1e-9])
    assert worst <= 1e-9


def header_defines(path):
    txt = open(path).read()
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"#define\s+(F16_\w+)\s+\(([-+0-9.eE]+)\)", txt)}


@pytest.mark.parametrize("header", ["oracle/f16_tables.h", "aircombat-selfplay_amd/csrc/f16_tables.h"])
def test_metrics_mass_and_tank_constants_of_both_headers(fcs, header):
    """<metrics>, <mass_balance>, the point masses and the four tanks as the second tokenizer reads them, against the constants BOTH
    generated headers carry (the oracle's and the device's)."""
    d = header_defines(os.path.join(ROOT, header))
    m = {k[5:]: fcs[k] for k in fcs.files if k.startswith("mass|")}
    for key, name in (("wingarea", "WINGAREA"), ("wingspan", "WINGSPAN"), ("chord", "CHORD"), ("ixx", "IXX"), ("iyy", "IYY"), ("izz", "IZZ"),
                      ("ixy", "IXY"), ("ixz", "IXZ"), ("iyz", "IYZ"), ("emptywt", "EMPTYWT")):
        assert d["F16_" + name] == float(m[key]), key
    for key, name in (("loc_AERORP", "AERORP"), ("loc_EYEPOINT", "EYEPOINT"), ("loc_VRP", "VRP"), ("cg", "CG")):
        assert [d[f"F16_{name}_{a}"] for a in "XYZ"] == list(m[key]), key
    assert float(m["negated"]) == 1.0            # negated_crossproduct_inertia="true" (f16.xml:62), the branch massbalance_run hard-codes
    assert len(m["pm_weight"]) == 2 and len(m["tank_contents"]) == 4 and float(m["n_engines"]) == 1.0
    for i in range(2):
        assert d[f"F16_PM{i}_WEIGHT"] == m["pm_weight"][i] and [d[f"F16_PM{i}_{a}"] for a in "XYZ"] == list(m["pm_xyz"][i])
    for i in range(4):
        assert d[f"F16_TANK{i}_CONTENTS"] == m["tank_contents"][i] and d[f"F16_TANK{i}_CAPACITY"] == m["tank_capacity"][i]
        assert [d[f"F16_TANK{i}_{a}"] for a in "XYZ"] == list(m["tank_xyz"][i])
    for tag in ("milthrust", "maxthrust", "bypassratio", "tsfc", "atsfc", "idlen1", "idlen2", "maxn1", "maxn2", "augmented", "augmethod", "injected"):
        assert d["F16_ENG_" + tag.upper()] == float(m["eng_" + tag]), tag      # F100-PW-229.xml scalars
    assert list(m["thruster_xyz"]) == [0.0, 0.0, 0.0]     # thrust acts at the structural origin: propulsion_run's moment arm is the CG itself


def massbalance(L, tanks, pm, cg_tanks):
    o, op = dbl(np.zeros(31))
    L.f16_test_massbalance(dbl(tanks)[1], dbl(pm)[1], dbl(cg_tanks)[1], op)
    return o


def test_massbalance_run_against_the_generic_reading(oracle, fcs):
    """200 tank loadings (incl. the as-shipped one and the empty, pilot-less aircraft): weight, CG, J, J^-1 and the tanks' share of J."""
    L = oracle.lib()
    for tanks, pm, want in zip(fcs["mb_tanks"], fcs["mb_pm_weight"], fcs["mb_result"]):
        cg = massbalance(L, tanks, pm, np.zeros(3))[1:4]       # first pass: the CG does not depend on the tank inertia's reference point
        got = massbalance(L, tanks, pm, cg)                    # second pass: tank inertia about the CG of the previous pass = this CG
        assert np.all(np.abs(got[:13] - want[:13]) <= 1e-9 * np.maximum(1.0, np.abs(want[:13]))), (tanks, got[:13], want[:13])
        assert np.all(np.abs(got[13:22] - want[13:22]) <= 1e-9 * np.abs(want[13:22]).max())
        assert np.all(np.abs(got[22:] - want[22:]) <= 1e-9 * np.maximum(1.0, np.abs(want[22:])))


@pytest.fixture(scope="module")
def rel():
    return np.load(os.path.join(GOLD, "jsbsim_relations.npz"))


def test_inertia_matrix_expectations_of_TestPointMassInertia(oracle, fcs, rel):
    """TestPointMassInertia.testInertiaMatrix runs on the F-16 itself (script f16_test -> aircraft f16): J * Jinv is the identity to 7
    places; with the point-mass weight and both internal tanks at 0 the weight is the empty weight and inertia/ixz-slugs_ft2 is the number
    in the file's <ixz> element (the test reads it from the XML: the recorded value), i.e. no parallel-axis term is left and the
    negated_crossproduct_inertia branch puts the file's value, sign included, at J(1,3)."""
    L = oracle.lib()
    m = {k[5:]: fcs[k] for k in fcs.files if k.startswith("mass|")}
    places = int(rel["inertia_identity_places"])
    for tanks in (m["tank_contents"], [0, 0, 0, 0], [1234.5, 2900.0, 0, 0]):
        cg = massbalance(L, tanks, m["pm_weight"], np.zeros(3))[1:4]
        got = massbalance(L, tanks, m["pm_weight"], cg)
        ident = got[4:13].reshape(3, 3) @ got[13:22].reshape(3, 3)
        assert np.all(np.abs(ident - rel["inertia_identity"]) < 0.5 * 10.0 ** -places)
    got = massbalance(L, [0, 0, 0, 0], [0, 0], m["cg"])
    assert round(got[0] - float(m["emptywt"]), places) == 0                 # inertia/weight-lbs == inertia/empty-weight-lbs
    assert round(got[4 + 2] - float(rel["f16_ixz_expected"]), places) == 0   # J(1,3) holds the file's ixz
    assert float(rel["f16_ixz_expected"]) == float(m["ixz"])


def test_tank_inertia_scales_with_contents_as_in_TestFuelTanksInertia(oracle, fcs, rel):
    """TestFuelTanksInertia.test_fuel_tanks_inertia: a tank's inertia varies as its contents (ratio 0.5 -> 0.5, delta 1e-7). The F-16's
    tanks carry no <radius> (generic reading asserts it), so their local inertia is 0 and what scales is the parallel-axis share of J."""
    L = oracle.lib()
    m = {k[5:]: fcs[k] for k in fcs.files if k.startswith("mass|")}
    ratio, delta = float(rel["tank_ratio"]), float(rel["tank_delta"])
    cg = np.array([-190.0, 1.5, -3.0])
    full = massbalance(L, m["tank_contents"], m["pm_weight"], cg)[22:]
    part = massbalance(L, ratio * m["tank_contents"], m["pm_weight"], cg)[22:]
    assert np.all(np.abs(part - ratio * full) <= delta)
    assert np.abs(full).max() > 100.0


def test_pilot_station_acceleration_relations_of_TestAccelerometer(oracle, rel):
    """TestAccelerometer.testOrbit: a body in free fall (no body-frame force, no rotation relative to the inertial frame) reads 0 at the
    pilot station (1e-8); testSpinningBodyOnOrbit: spinning at r_inertial = 1 rad/s about body z with the CG offset along structural
    y by d, the station at the structural origin reads (0, d r^2, 0) (a-pilot-y / cg-y = 1 to 1e-8): the w x (w x r) term on the
    INERTIAL rates with r = StructuralToBody(station)."""
    L = oracle.lib()
    o, op = dbl(np.zeros(3))
    z = np.zeros(3)
    L.f16_test_pilot_accel(dbl([-193.0, 0.0, -5.1])[1], dbl([-336.2, 0.0, 29.5])[1], dbl(z)[1], dbl(z)[1], dbl(z)[1], op)
    assert np.all(np.abs(o - rel["orbit_a_pilot"]) <= float(rel["orbit_delta"]))
    cgy_in = 17.0
    L.f16_test_pilot_accel(dbl([0.0, cgy_in, 0.0])[1], dbl(z)[1], dbl(z)[1], dbl(z)[1], dbl([0.0, 0.0, float(rel["spin_r_inertial"])])[1], op)
    assert abs(o[0] - rel["spin_a_pilot_x"]) <= float(rel["spin_delta"]) and abs(o[2] - rel["spin_a_pilot_z"]) <= float(rel["spin_delta"])
    assert abs(o[1] / (cgy_in / 12.0) - float(rel["spin_ay_over_cgy_ft"])) <= float(rel["spin_delta"])


def test_initial_geodetic_latitude_of_TestInitialConditions(oracle, rel):
    """TestInitialConditions.test_set_initial_geodetic_latitude: after run_ic, position/lat-geod-deg is the ic/lat-geod-deg that was set and
    the altitude above sea level is unchanged (assertAlmostEqual, 7 places) — f16_reset places the aircraft by geodetic latitude."""
    O = oracle
    places = int(rel["ic_places"])
    for dlat in rel["ic_lat_shift_deg"]:
        for lat0, h in ((60.0, 20000.0), (35.2, 9000.0), (-12.0, 31000.0)):
            cfg = O.default_config(O.TASK_SINGLECOMBAT)
            cfg.init[0].lat_geod_deg = lat0 + float(dlat)
            cfg.init[0].h_sl_ft = h
            env = O.OracleEnv(cfg)
            env.reset()
            lon, lat, alt_m = env.pose(0)[:3]        # position/lat-geod-deg and position/h-sl-ft * 0.3048, as AircraftSimulator caches them
            assert round(lat - (lat0 + float(dlat)), places) == 0
            assert abs(alt_m / 0.3048 - h) < 1e-5    # h_sl goes through radius - sea-level radius at 2e7 ft: 1e-5 ft is fp64's floor there


def test_aerodynamic_force_frame_and_moment_transfer_of_TestAeroFuncFrame(oracle, rel):
    """TestAeroFuncFrame.testAeroFrame (the DRAG / SIDE / LIFT axis form of f16.xml:994-1925), 256 random frames: the body-axis force is
    Tw2b * diag(-1, 1, -1) * (DRAG, SIDE, LIFT), the moment is the (ROLL, PITCH, YAW) sums taken at the reference point plus
    cross((cg - rp) / 12 with y negated, Fb) — aerodynamics_run's wind_to_body() / aero_frame() at the test's own 7 places. A sign flip
    of either arm of the cross product, of the lift / drag negation or of one Tw2b element fails it (checked below on the numbers)."""
    L = oracle.lib()
    X, fb, mb = rel["aeroframe_inputs"], rel["aeroframe_fb"], rel["aeroframe_mb"]
    places = int(rel["aeroframe_places"])
    assert len(X) >= 200 and np.array_equal(rel["aeroframe_fw"], X[:, 8:11])            # forces/fw*-aero-lbs ARE the axis sums
    f, fp = dbl(np.zeros(3))
    m, mp = dbl(np.zeros(3))
    for x, wf, wm in zip(X, fb, mb):
        L.f16_test_aero_frame(x[0], x[1], dbl(x[2:5])[1], dbl(x[5:8])[1], dbl(x[8:14])[1], fp, mp)
        assert all(round(a - b, places) == 0 for a, b in zip(f, wf)), (x, f, wf)          # assertAlmostEqual(places=7)
        assert all(round(a - b, places) == 0 for a, b in zip(m, wm)), (x, m, wm)
    # the fixture discriminates: each of these wrong readings misses the recorded numbers by far more than the criterion
    x, wf, wm = X[0], fb[0], mb[0]
    ca, sa, cb, sb = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1])
    T = np.array([[ca * cb, -ca * sb, -sa], [sb, cb, 0.0], [sa * cb, -sa * sb, ca]])
    arm = np.array([x[2] - x[5], x[6] - x[3], x[4] - x[7]]) / 12.0
    right = T @ (np.array([-1.0, 1.0, -1.0]) * x[8:11])
    assert np.abs(right - wf).max() < 1e-7 and np.abs(x[11:14] + np.cross(arm, right) - wm).max() < 1e-6
    assert np.abs(T @ x[8:11] - wf).max() > 1.0                                          # drag / lift not negated
    assert np.abs(T.T @ (np.array([-1.0, 1.0, -1.0]) * x[8:11]) - wf).max() > 1.0        # Tb2w instead of Tw2b
    assert np.abs(x[11:14] + np.cross(-arm, right) - wm).max() > 1.0                     # rp - cg instead of cg - rp
    assert np.abs(x[11:14] + np.cross(arm * [1, -1, 1], right) - wm).max() > 1.0         # structural y not negated


def test_moment_of_a_force_at_the_structural_origin_of_CheckMomentsUpdate(oracle, rel, fcs):
    """CheckMomentsUpdate.test_moments_update: a force applied at the structural origin with the CG at (CGx, CGz) gives
    My = Fx * CGz - Fz * CGx (CG in feet, delta 1e-7). The F-16's thruster sits at the structural origin (f16.xml:259-270; the generic
    reading's thruster_xyz == 0 is asserted in test_metrics_mass_and_tank_constants_of_both_headers), so this is propulsion_run's moment arm."""
    L = oracle.lib()
    assert list(fcs["mass|thruster_xyz"]) == [0.0, 0.0, 0.0]
    X, want, delta = rel["origin_force_inputs"], rel["origin_force_my"], float(rel["origin_force_delta"])
    assert len(X) >= 200
    m, mp = dbl(np.zeros(3))
    for (fx, fz, cgx, cgz), my in zip(X, want):
        L.f16_test_thruster_moment(dbl([cgx, 0.0, cgz])[1], dbl([fx, 0.0, fz])[1], mp)
        assert abs(m[1] - my) <= delta, (fx, fz, cgx, cgz, m, my)
        assert abs(-m[1] - my) > 1e-3 or abs(my) < 1e-3                                  # the opposite arm misses it
    # the thrust itself is along body x only: its pitching moment is thrust * CGz / 12 (a 5.1 in arm on the as-shipped loading)
    L.f16_test_thruster_moment(dbl([-193.0, 0.0, -5.1])[1], dbl([10000.0, 0.0, 0.0])[1], mp)
    assert abs(m[1] - 10000.0 * (-5.1 / 12.0)) < 1e-9 and m[0] == 0.0 and abs(m[2]) == 0.0


def _golden_module(name):
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if here not in sys.path:
        sys.path.insert(0, here)
    spec = importlib.util.spec_from_file_location(name, os.path.join(here, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("case,min_expectations", [("TestGain.test_conditions", 80), ("TestSwitch.test_conditions", 33), ("TestSwitch.test_nested", 6),
                                                   ("TestFunctions.test_functions", 28)])
def test_generic_component_reading_passes_the_references_component_tests(case, min_expectations):
    """The fixture fcs_run() is held to (f16_fcs_check.npz) is the generic interpreter's reading of <switch>, <pure_gain>, <summer>,
    <fcs_function>. The reference holds JSBSim's own unit tests of those component types with their system files (TestGain.py / gain.xml,
    TestSwitch.py / switch.xml, TestFunctions.py / function.xml): tests/golden/make_jsbsim_components.py ran them as they are against the
    interpreter and logged what they set, when they ran the FDM and what they asserted. Replayed here on the stored (tokenised) system
    file with the same `Component` class: every expectation holds at the test's own criterion (assertEqual exact, assertAlmostEqual
    7 places)."""
    import json
    comp = _golden_module("make_jsbsim_components")
    rec = json.loads(str(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsbsim_components.npz"))[case]))
    sysm = comp.System(rec["tree"])          # (the fixture carries the covered components of the test's system file only)
    assert sysm.uncovered == []
    n = 0
    for ev in rec["events"]:
        if ev[0] == "set":
            sysm.st.set(ev[1], ev[2])
        elif ev[0] == "run":
            sysm.run()
        else:
            _, name, want, places, delta = ev
            got = sysm.st.get(name)
            if delta is not None:
                assert abs(got - want) <= delta, (name, got, want)
            elif places is None:
                assert got == want, (name, got, want)
            else:
                assert round(abs(got - want), places) == 0, (name, got, want)
            n += 1
    assert n >= min_expectations


def test_fcs_fixture_is_what_the_generic_interpreter_produces_today(fcs):
    """The interpreter was extended after the fixture was written (nested <test> groups, for TestSwitch): its reading of the F-16's own
    section must not have moved. Needs the reference's f16.xml: runs in the build container, skipped on the GPU box."""
    gen = _golden_module("make_f16_fcs_check")
    if not os.path.exists(gen.F16):
        pytest.skip("the reference's f16.xml is not on this machine")
    fc = next(gen.find_all(gen.parse(gen.F16), "flight_control"))
    for seq in (0, 5, 13, 37, 63):                      # (5, 13, 37: sequences that start with the gear already up)
        ctl = gen.F16FlightControl(fc)
        if fcs["gear_pos0"][seq] == 0.0:
            ctl.st.set("gear/gear-pos-norm", 0.0)
        X = gen.expand_inputs(fcs["knots"][seq], fcs["linear"][seq])
        for t in range(gen.NT):
            y = ctl.tick(dict(zip(gen.IN_PROPS, X[t])))
            assert np.array_equal(np.asarray(y), fcs["out_full"][seq, t]), (seq, t)
