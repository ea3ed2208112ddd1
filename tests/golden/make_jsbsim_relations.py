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


def load(name, log):
    utils = types.ModuleType("JSBSim_utils")

    class Sandbox:
        def path_to_jsbsim_file(self, *parts):
            return os.path.join(DATA, *parts)

        def __call__(self, *parts):
            return os.path.join("/tmp", "ac_relations_sandbox", *parts)

    class JSBSimTestCase:
        sandbox = Sandbox()

        def create_fdm(self):
            self.fdm = SymFDM(log)
            return self.fdm

        def load_script(self, name):
            log["script"] = name

        def get_aircraft_xml_tree(self, script):
            use = et.parse(os.path.join(DATA, "scripts", script + ".xml")).getroot().find("use")
            ac = use.attrib["aircraft"]
            log["aircraft"] = ac
            return et.parse(os.path.join(DATA, "aircraft", ac, ac + ".xml"))

        def assertAlmostEqual(self, a, b, places=7, delta=None, msg=None):
            log.setdefault("asserts", []).append((a.expr if isinstance(a, Sym) else float(a), b.expr if isinstance(b, Sym) else float(b),
                                                  places if delta is None else None, delta))

        def assertEqual(self, a, b, msg=None): pass
        def assertTrue(self, a, msg=None): pass

    utils.JSBSimTestCase = JSBSimTestCase
    utils.RunTest = lambda cls: None
    utils.ExecuteUntil = lambda fdm, t: None
    utils.CreateFDM = lambda sandbox: SymFDM(log)
    utils.append_xml = lambda n: n if n.endswith(".xml") else n + ".xml"

    class _Tree:                                   # CopyAircraftDef(script, sandbox) -> (tree, aircraft name, path); the tests only edit / write it
        def __init__(self, path): self.t = et.parse(path)
        def getroot(self): return self.t.getroot()
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
    np.savez_compressed(os.path.join(OUT, "jsbsim_relations.npz"), **out)
    for k, v in out.items():
        print(f"{k:28s} {v.tolist()}")


if __name__ == "__main__":
    main()
