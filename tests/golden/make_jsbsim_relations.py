#!/usr/bin/env python3
"""The expectations four more of the reference's JSBSim unit tests make about blocks the FDM oracle restates, as plain numbers
(/root/reference/envs/JSBSim/data/tests; runs only in the build container).

Like make_jsbsim_blocks.py this loads the reference's test scripts with in-process stand-ins for the absent ``jsbsim`` wheel; here
the tests assert RELATIONS between outputs of the wheel rather than tables, so the stand-in ``fdm`` hands out symbolic tokens and
every ``assertAlmostEqual(expression, number)`` the test makes is logged as (what was compared, the number, places / delta):

  TestPointMassInertia.py:96-124 testInertiaMatrix (script f16_test = the F-16 itself)
        J * Jinv == identity element by element (7 places); with inertia/pointmass-weight-lbs and both internal tanks at 0:
        inertia/weight-lbs == inertia/empty-weight-lbs and inertia/ixz-slugs_ft2 == float(<ixz> of f16.xml)
  TestFuelTanksInertia.py:61-88 test_fuel_tanks_inertia
        halving propulsion/tank/contents-lbs halves the tank's inertias, delta 1e-7
  TestAccelerometer.py:55-62 testOrbit              accelerations/a-pilot-{x,y,z}-ft_sec2 == 0 in free fall, delta 1e-8
  TestAccelerometer.py:154-199 testSpinningBodyOnOrbit   r_inertial = 1 rad/s about z, CG offset along y: a-pilot-x == 0, a-pilot-z == 0,
        a-pilot-y / (inertia/cg-y-in / 12) == 1, delta 1e-8
  TestInitialConditions.py:322-345 test_set_initial_geodetic_latitude
        after run_ic: ic/h-agl-ft unchanged, position/lat-geod-deg == the ic/lat-geod-deg that was set (shift -30 deg), 7 places

  TestAeroFuncFrame.py:58-101,206-221 testAeroFrame (the DRAG / SIDE / LIFT axis form the F-16's <aerodynamics> uses)
        per frame, from the axis sums Fa = (DRAG, SIDE, LIFT) and (ROLL, PITCH, YAW): forces/fw{x,y,z}-aero-lbs == Fa,
        forces/fb{x,y,z}-aero-lbs == Tw2b * diag(-1, 1, -1) * Fa, moments/{l,m,n}-aero-lbsft == M_MRC + cross((cg - rp) / 12, Fb) with cg, rp
        in structural inches and their y negated (7 places). These tests do ARITHMETIC on the wheel's outputs, so the stand-in here is
        numeric (NumFDM): it hands out random property values per frame, the test's own code computes what it expects of the wheel, and
        every assertAlmostEqual(number, <output property>) is logged as (property, number). Tw2b is the wheel's (auxiliary.get_Tw2b());
        the stand-in builds it as the test's own getTs2b() (stability -> body, alpha) times the rotation by beta about stability z,
        which is FGAuxiliary.cpp:256-264 element by element.
  CheckMomentsUpdate.py:44-76 test_moments_update (weather-balloon: buoyancy acts at the structural origin, gas_cell location 0 0 0)
        moments/m-buoyancy-lbsft == Fbx * CGz - Fbz * CGx with the CG in feet (delta 1e-7): the moment of a force applied at the
        structural origin — where the F-16's thruster sits (f16.xml:259-270).

Output: tests/golden/jsbsim_relations.npz (numbers only). tests/test_oracle_f16_wiring.py applies them to the oracle's functions.
"""
import importlib.util
import math
import os
import sys
import types
import xml.etree.ElementTree as et

import numpy as np

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
DATA = os.path.join(REF, "envs", "JSBSim", "data")
TESTS = os.path.join(DATA, "tests")
OUT = os.path.dirname(os.path.abspath(__file__))


class Sym:
    """A symbolic output of the wheel; arithmetic builds a readable expression string."""

    def __init__(self, expr):
        self.expr = expr

    def _bin(self, other, op, swap=False):
        o = other.expr if isinstance(other, Sym) else repr(float(other))
        return Sym(f"({o} {op} {self.expr})" if swap else f"({self.expr} {op} {o})")

    def __mul__(self, o): return self._bin(o, "*")
    def __rmul__(self, o): return self._bin(o, "*", True)
    def __truediv__(self, o): return self._bin(o, "/")
    def __rtruediv__(self, o): return self._bin(o, "/", True)
    def __sub__(self, o): return self._bin(o, "-")
    def __rsub__(self, o): return self._bin(o, "-", True)
    def __add__(self, o): return self._bin(o, "+")
    __radd__ = __add__
    def __neg__(self): return Sym(f"(-{self.expr})")
    def __float__(self): return float("nan")       # the tests format values into their failure messages with %f


class SymMatrix:
    def __init__(self, name):
        self.name = name

    def __mul__(self, other):
        return SymMatrix(f"{self.name}*{other.name}")

    def __getitem__(self, ij):
        return Sym(f"{self.name}[{ij[0]},{ij[1]}]")


class SymFDM:
    def __init__(self, log):
        self.log, self.sets = log, {}

    def __getitem__(self, k):
        if k in self.sets and not isinstance(self.sets[k], Sym):
            return self.sets[k]
        return Sym(k)

    def __setitem__(self, k, v):
        self.sets[k] = v
        self.log.setdefault("sets", []).append((k, v.expr if isinstance(v, Sym) else float(v)))

    def load_model(self, *a, **k): return True
    def load_script(self, *a, **k): return True
    def load_ic(self, *a, **k): return True
    def run_ic(self): return True
    def set_aircraft_path(self, *a): pass
    def set_output_directive(self, *a): pass

    def run(self):
        self.log["runs"] = self.log.get("runs", 0) + 1
        return self.log["runs"] < 3           # `while fdm.run():` loops see two frames

    def get_mass_balance(self):
        return types.SimpleNamespace(get_J=lambda: SymMatrix("J"), get_Jinv=lambda: SymMatrix("Jinv"))


class Out(float):
    """An output property of the wheel the test compares a computed number with: carries its name, worthless as a number."""

    def __new__(cls, name):
        o = float.__new__(cls, float("nan"))
        o.name = name
        return o


class NumFDM:
    """Numeric stand-in: every frame (run()) draws fresh values for the properties the test READS as inputs; the properties it compares
    its own arithmetic with (`outputs`) come back as Out(name). `frame_inputs` keeps what each frame handed out."""

    def __init__(self, log, rng, outputs, draw):
        self.log, self.rng, self.outputs, self.draw = log, rng, set(outputs), draw
        self.frames, self.cur, self.max_frames = [], None, log.get("max_frames", 4)
        self.sets = {}
        self._new_frame()

    def _new_frame(self):
        self.cur = {}
        self.frames.append(self.cur)

    def __getitem__(self, k):
        if k in self.outputs:
            return Out(k)
        if k in self.sets:
            return self.sets[k]
        if k not in self.cur:
            self.cur[k] = float(self.draw(k, self.rng))
        return self.cur[k]

    def __setitem__(self, k, v):
        self.sets[k] = float(v)

    def load_model(self, *a, **k): return True
    def load_script(self, *a, **k): return True
    def run_ic(self): return True
    def set_aircraft_path(self, *a): pass
    def set_output_directive(self, *a): pass

    def run(self):
        if len(self.frames) > self.max_frames:
            return False
        self._new_frame()
        return True

    def get_auxiliary(self):
        fdm = self

        class Aux:
            def get_Tw2b(self_):
                a, b = fdm["aero/alpha-rad"], fdm["aero/beta-rad"]
                ca, sa, cb, sb = math.cos(a), math.sin(a), math.cos(b), math.sin(b)
                Ts2b = np.asmatrix([[ca, 0., -sa], [0., 1., 0.], [sa, 0., ca]])      # == the test's getTs2b()
                Tw2s = np.asmatrix([[cb, -sb, 0.], [sb, cb, 0.], [0., 0., 1.]])
                return Ts2b * Tw2s                                                  # FGAuxiliary.cpp:256-264

            def get_Tb2w(self_):
                return self_.get_Tw2b().T

        return Aux()


def load(name, log, fdm_factory=None):
    utils = types.ModuleType("JSBSim_utils")
    if fdm_factory is None:
        fdm_factory = lambda: SymFDM(log)                                            # noqa: E731

    class Sandbox:
        def path_to_jsbsim_file(self, *parts):
            return os.path.join(DATA, *parts)

        def __call__(self, *parts):
            return os.path.join("/tmp", "ac_relations_sandbox", *parts)

    class JSBSimTestCase:
        sandbox = Sandbox()

        def setUp(self): pass
        def tearDown(self): pass

        def create_fdm(self):
            self.fdm = fdm_factory()
            return self.fdm

        def load_script(self, name):
            log["script"] = name

        def get_aircraft_xml_tree(self, script):
            use = et.parse(os.path.join(DATA, "scripts", script + ".xml")).getroot().find("use")
            ac = use.attrib["aircraft"]
            log["aircraft"] = ac
            return et.parse(os.path.join(DATA, "aircraft", ac, ac + ".xml"))

        def assertAlmostEqual(self, a, b, places=7, delta=None, msg=None):
            if isinstance(b, Out) or isinstance(a, Out):
                name, val = (b.name, a) if isinstance(b, Out) else (a.name, b)
                log.setdefault("outputs", []).append((name, float(val), places if delta is None else None, delta))
                return
            log.setdefault("asserts", []).append((a.expr if isinstance(a, Sym) else float(a), b.expr if isinstance(b, Sym) else float(b),
                                                  places if delta is None else None, delta))

        def assertEqual(self, a, b, msg=None): pass
        def assertTrue(self, a, msg=None): pass

    utils.JSBSimTestCase = JSBSimTestCase
    utils.RunTest = lambda cls: None
    utils.ExecuteUntil = lambda fdm, t: None
    utils.CreateFDM = lambda sandbox: fdm_factory()
    utils.append_xml = lambda n: n if n.endswith(".xml") else n + ".xml"

    class _Tree:                                   # CopyAircraftDef(script, sandbox) -> (tree, aircraft name, path); the tests only edit / write it
        def __init__(self, path): self.t = et.parse(path)
        def getroot(self): return self.t.getroot()
        def findall(self, path): return self.t.findall(path)
        def write(self, *a, **k): pass

    def copy_def(script_path, sandbox):
        ac = et.parse(script_path).getroot().find("use").attrib["aircraft"]
        return _Tree(os.path.join(DATA, "aircraft", ac, ac + ".xml")), ac, os.path.join(DATA, "aircraft")

    utils.CopyAircraftDef = copy_def
    utils.isDataMatching = utils.FindDifferences = lambda *a, **k: None
    sys.modules["JSBSim_utils"] = utils
    sys.modules.setdefault("jsbsim", types.ModuleType("jsbsim"))
    if "pandas" not in sys.modules:
        try:
            import pandas  # noqa: F401
        except ImportError:
            sys.modules["pandas"] = types.ModuleType("pandas")
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(TESTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    out = {}

    # ---- TestPointMassInertia.testInertiaMatrix
    log = {}
    m = load("TestPointMassInertia", log)
    tc = m.TestPointMassInertia()
    tc.testInertiaMatrix()
    assert log["script"] == "f16_test" and log["aircraft"] == "f16", log
    ident = np.full((3, 3), np.nan)
    places = set()
    ixz = None
    weight_rel = False
    for a, b, pl, delta in log["asserts"]:
        if isinstance(a, str) and a.startswith("J*Jinv["):
            i, j = (int(x) for x in a[len("J*Jinv["):-1].split(","))
            ident[i, j] = b
            places.add(pl)
        elif a == "inertia/ixz-slugs_ft2":
            ixz = b
        elif a == "inertia/weight-lbs" and b == "inertia/empty-weight-lbs":
            weight_rel = True
    zeroed = {k: v for k, v in log["sets"]}
    assert zeroed == {"inertia/pointmass-weight-lbs": 0.0, "propulsion/tank[0]/contents-lbs": 0.0, "propulsion/tank[1]/contents-lbs": 0.0}, zeroed
    assert weight_rel and ixz is not None and not np.isnan(ident).any() and len(places) == 1
    out["inertia_identity"] = ident
    out["inertia_identity_places"] = np.array(places.pop())
    out["f16_ixz_expected"] = np.array(ixz)

    # ---- TestFuelTanksInertia.test_fuel_tanks_inertia
    log = {}
    m = load("TestFuelTanksInertia", log)
    tc = m.TestFuelTanksInertia()
    tc.test_fuel_tanks_inertia()
    ratios, deltas = set(), set()
    for a, b, pl, delta in log["asserts"]:
        if isinstance(a, str) and isinstance(b, str) and a.endswith("slug_ft2") and b.startswith("(0.5 * "):
            ratios.add(0.5)
            deltas.add(delta)
    assert ratios == {0.5} and len(deltas) == 1, log["asserts"]
    out["tank_ratio"], out["tank_delta"] = np.array(0.5), np.array(deltas.pop())

    # ---- TestAccelerometer.testOrbit / testSpinningBodyOnOrbit
    log = {}
    m = load("TestAccelerometer", log)
    tc = m.TestAccelerometer()
    tc.AddAccelerometersToAircraft = lambda path: None
    cwd = os.getcwd()
    os.makedirs("/tmp/ac_relations_sandbox", exist_ok=True)
    os.chdir("/tmp/ac_relations_sandbox")         # testOrbit writes its edited copy of the script into the working directory
    try:
        tc.testOrbit()
    finally:
        os.chdir(cwd)
    ap = {}
    for a, b, pl, delta in log["asserts"]:
        if isinstance(a, str) and a.startswith("accelerations/a-pilot-"):
            ap[a[len("accelerations/a-pilot-")]] = (b, delta)
    out["orbit_a_pilot"] = np.array([ap[c][0] for c in "xyz"])
    out["orbit_delta"] = np.array(ap["x"][1])
    log = {}
    m = load("TestAccelerometer", log)
    tc = m.TestAccelerometer()
    tc.AddAccelerometersToAircraft = lambda path: None
    tc.testSpinningBodyOnOrbit()
    sets = dict(log["sets"])
    omega = 0.00007292115
    out["spin_r_inertial"] = np.array(sets["ic/r-rad_sec"] + omega)            # the test sets ic/r = 1 - omega so that r_inertial = 1
    assert abs(sets["ic/phi-rad"] - 0.5 * math.pi) < 1e-15 and sets["ic/p-rad_sec"] == 0.0 and sets["ic/q-rad_sec"] == 0.0
    got = {}
    for a, b, pl, delta in log["asserts"]:
        if a == "fcs/accelerometer/X": got["x"] = (b, delta)
        if a == "fcs/accelerometer/Z": got["z"] = (b, delta)
        if a == "(fcs/accelerometer/Y / (inertia/cg-y-in / 12.0))": got["y"] = (b, delta)
        if isinstance(a, str) and a.startswith("accelerations/a-pilot-") and isinstance(b, str):
            assert b == "fcs/accelerometer/" + a[len("accelerations/a-pilot-")].upper()   # a-pilot IS what the accelerometer at that station reads
            got.setdefault("same", []).append(delta)
    assert len(got["same"]) == 3
    out["spin_a_pilot_x"], out["spin_a_pilot_z"], out["spin_ay_over_cgy_ft"] = np.array(got["x"][0]), np.array(got["z"][0]), np.array(got["y"][0])
    out["spin_delta"] = np.array(got["y"][1])

    # ---- TestInitialConditions.test_set_initial_geodetic_latitude
    log = {}
    m = load("TestInitialConditions", log)
    tc = m.TestInitialConditions()
    pd = sys.modules["pandas"]
    real_read_csv = getattr(pd, "read_csv", None)
    pd.read_csv = lambda *a, **k: {"Time": [0.0], "Latitude Geodetic (deg)": [Sym("csv/lat-geod-deg")]}
    try:
        tc.test_set_initial_geodetic_latitude()
    finally:
        if real_read_csv is not None:
            pd.read_csv = real_read_csv
    shift = None
    rels = set()
    for k, v in log["sets"]:
        if k == "ic/lat-geod-deg":
            assert v.startswith("(ic/lat-geod-deg - ")
            shift = -float(v[len("(ic/lat-geod-deg - "):-1])
    for a, b, pl, delta in log["asserts"]:
        rels.add((a, b, pl))
    assert ("position/lat-geod-deg", "(ic/lat-geod-deg - 30.0)", 7) in rels and ("ic/h-agl-ft", "ic/h-agl-ft", 7) in rels, rels
    out["ic_lat_shift_deg"] = np.array([shift, 0.0, 12.5])       # the test's own shift, and two more of ours
    out["ic_places"] = np.array(7)

    # ---- TestAeroFuncFrame.testAeroFrame (numeric stand-in; numpy 2 dropped np.mat, which the test script uses: same thing as asmatrix)
    if not hasattr(np, "mat"):
        np.mat = np.asmatrix
    NF = 256
    log = {"max_frames": NF}
    rng = np.random.default_rng(20251005)
    force_out = [f"forces/f{fr}{ax}-aero-lbs" for fr in "wbs" for ax in "xyz"]
    moment_out = [f"moments/{n}-aero-lbsft" for n in ("l", "m", "n", "roll-stab", "pitch-stab", "yaw-stab", "roll-wind", "pitch-wind", "yaw-wind")]

    def draw_aero(k, rng):
        if k == "aero/alpha-rad": return rng.uniform(-0.6, 1.4)          # the F-16's alpha table runs to 45 deg; beyond is fine for a frame relation
        if k == "aero/beta-rad": return rng.uniform(-0.6, 0.6)
        if k.startswith("metrics/aero-rp-") or k.startswith("inertia/cg-"):
            return rng.uniform(-400.0, 400.0)                            # structural inches, y included (the F-16's are 0: the relation is general)
        return rng.uniform(-3.0e4, 3.0e4)                                # an aero function's value (lbs / lbs ft)

    fdms = []

    def factory():
        fdms.append(NumFDM(log, rng, force_out + moment_out, draw_aero))
        return fdms[-1]

    m = load("TestAeroFuncFrame", log, factory)
    tc = m.TestAeroFuncFrame()
    tc.setUp()
    axes = {ax.attrib["name"]: [f.attrib["name"] for f in ax.findall("function")] for ax in tc.tree.findall("aerodynamics/axis")}
    assert list(axes) == ["DRAG", "SIDE", "LIFT", "ROLL", "PITCH", "YAW"], list(axes)      # X15.xml: the axis form f16.xml:994-1925 uses too
    tc.testAeroFrame()
    fdm = fdms[-1]
    rp_frame = fdm.frames[0]                                             # the test reads metrics/aero-rp-* once, before its loop
    frames = [f for f in fdm.frames[1:] if "aero/alpha-rad" in f]
    assert len(frames) == NF, len(frames)
    per = {}
    for name, val, pl, delta in log["outputs"]:
        per.setdefault(name, []).append(val)
        assert pl == 7 and delta is None
    assert all(len(per[n]) == NF for n in force_out + moment_out), {n: len(v) for n, v in per.items()}
    inp = np.zeros((NF, 14))
    for i, f in enumerate(frames):
        sums = []
        for ax in ("DRAG", "SIDE", "LIFT", "ROLL", "PITCH", "YAW"):
            acc = 0.0
            for fn in axes[ax]:
                acc += f[fn]                                             # the test's own summation order
            sums.append(acc)
        inp[i] = [f["aero/alpha-rad"], f["aero/beta-rad"], f["inertia/cg-x-in"], f["inertia/cg-y-in"], f["inertia/cg-z-in"],
                  rp_frame["metrics/aero-rp-x-in"], rp_frame["metrics/aero-rp-y-in"], rp_frame["metrics/aero-rp-z-in"]] + sums
    out["aeroframe_inputs"] = inp                                        # alpha, beta, cg xyz [in], rp xyz [in], DRAG SIDE LIFT ROLL PITCH YAW sums
    out["aeroframe_fw"] = np.array([per[f"forces/fw{a}-aero-lbs"] for a in "xyz"]).T
    out["aeroframe_fb"] = np.array([per[f"forces/fb{a}-aero-lbs"] for a in "xyz"]).T
    out["aeroframe_mb"] = np.array([per[f"moments/{a}-aero-lbsft"] for a in "lmn"]).T
    out["aeroframe_places"] = np.array(7)

    # ---- CheckMomentsUpdate.test_moments_update (numeric stand-in, one relation per run of the test: run it NM times)
    NM = 200
    rng = np.random.default_rng(20251006)
    mom_in, mom_out, mom_delta = [], [], set()

    def draw_mom(k, rng):
        if k.startswith("forces/"): return rng.uniform(-3.0e4, 3.0e4)
        if k.startswith("inertia/cg-"): return rng.uniform(-300.0, 300.0)
        if k == "simulation/dt": return 1.0 / 120.0
        return rng.uniform(1.0, 100.0)                                   # weights, contents, point-mass location: only CheckCGPosition reads them

    pd = sys.modules.get("pandas")
    if pd is None:
        import pandas as pd
    real_read_csv = pd.read_csv
    for _ in range(NM):
        log = {"max_frames": 8}
        fdms.clear()

        def factory_m():
            fdms.append(NumFDM(log, rng, ["moments/m-buoyancy-lbsft"], draw_mom))
            return fdms[-1]

        m = load("CheckMomentsUpdate", log, factory_m)       # RunTest is a no-op in the stand-in utils: importing does not run anything
        tc = m.CheckMomentsUpdate()

        class _Col:
            def __init__(self, v): self.iloc = [v]

        def fake_csv(*a, **k):
            f = fdms[-1]
            return {"M_{Buoyant} (ft-lbs)": _Col(Out("moments/m-buoyancy-lbsft")), "F_{Buoyant x} (lbs)": _Col(f["forces/fbx-buoyancy-lbs"]),
                    "F_{Buoyant z} (lbs)": _Col(f["forces/fbz-buoyancy-lbs"])}

        pd.read_csv = fake_csv
        try:
            tc.test_moments_update()
        finally:
            pd.read_csv = real_read_csv
        recs = [r for r in log["outputs"] if r[0] == "moments/m-buoyancy-lbsft"]
        assert len(recs) == 2, recs
        name, val, pl, delta = recs[0]                       # the in-memory check (:66-76); the second one re-reads the same frame from the CSV
        f = [fr for fr in fdms[-1].frames if "forces/fbx-buoyancy-lbs" in fr][0]
        mom_in.append([f["forces/fbx-buoyancy-lbs"], f["forces/fbz-buoyancy-lbs"], f["inertia/cg-x-in"], f["inertia/cg-z-in"]])
        mom_out.append(val)
        mom_delta.add(delta)
    assert mom_delta == {1e-7}
    out["origin_force_inputs"] = np.array(mom_in)            # Fbx, Fbz [lbs], CG x, z [in]
    out["origin_force_my"] = np.array(mom_out)               # lbs ft
    out["origin_force_delta"] = np.array(1e-7)
    np.savez_compressed(os.path.join(OUT, "jsbsim_relations.npz"), **out)
    for k, v in out.items():
        print(f"{k:28s} {v.tolist()}" if v.size <= 12 else f"{k:28s} shape {v.shape}")


if __name__ == "__main__":
    main()
