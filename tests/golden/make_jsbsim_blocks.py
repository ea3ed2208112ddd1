#!/usr/bin/env python3
"""Golden vectors for the generic JSBSim blocks the FDM oracle restates, taken from the REFERENCE's own JSBSim unit tests
(/root/reference/envs/JSBSim/data/tests; runs only in the build container, where /root/reference exists).

Those tests are scripts that drive the ``jsbsim`` wheel (absent here) and assert its outputs against numbers and closed forms
they hold themselves. This script loads each test module with in-process stand-ins for ``jsbsim`` / ``JSBSim_utils`` and a
recording ``fdm``: every ``fdm[...]`` read hands back a token, and every ``assertAlmostEqual`` the test makes on such a token is
logged as (inputs the test had set, expected value). What lands in tests/golden/jsbsim_blocks.npz is therefore the reference
tests' own expectations as plain numbers:

  TestStdAtmosphere.py:27-60,62-160   ISA lapse table walk (temperature and pressure at every breakpoint and half way, standard
                                      day and delta-T = 15 K): (h_sl_ft, delta_T_R, expected T [R]), (.., expected P [psf])
  TestDensityAltitude.py:29-72        (h_sl_ft, delta_T_R, expected density altitude [ft])
  TestPressureAltitude.py:29-72       (h_sl_ft, delta_T_R, expected pressure altitude [ft])
  TestKinematic.py:26-82              c172r flap <kinematic> (detents 0/10/20/30 deg, times 0/2/1/1 s, aircraft/c172r/c172r.xml:351-372):
                                      per 1/120 s frame the commanded value and the expected flap angle
  TestTurbine.py:26-34,37-40,43-62    the default spool-up law on a grid, and the spool-up / spool-down
                                      trajectory of the F100-PW-229 (engine/F100-PW-229.xml) that runScript predicts frame by frame with its ``seek``

  TestIntegrators.py:36-146           <pid>: ab2 / ab3 / rect integration of sin(8 pi t), a positive trigger suspends and a negative one
                                      resets the integral, and the kp / ki / kd terms alone (integrators.xml), dt = 0.005 s

Only numbers are stored; no reference source text.
"""
import importlib.util
import os
import sys
import types
import xml.etree.ElementTree as et

import numpy as np

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
DATA = os.path.join(REF, "envs", "JSBSim", "data")
TESTS = os.path.join(DATA, "tests")
OUT = os.path.dirname(os.path.abspath(__file__))


class Token:
    """What fdm['some/output'] returns: remembers which property it is and the inputs set when it was read."""

    def __init__(self, prop, inputs):
        self.prop, self.inputs, self.scale = prop, dict(inputs), 1.0

    def _scaled(self, k):
        t = Token(self.prop, self.inputs)
        t.scale = self.scale * k
        return t

    def __rtruediv__(self, num):      # expected / fdm[...]  (asserted equal to 1.0)  ->  expected value = num
        return ("ratio", float(num), self)

    def __truediv__(self, den):       # fdm[...] / expected  (asserted equal to 1.0)
        return ("ratio", float(den), self)

    def __lt__(self, other):          # `if density_alt < 1E-9` style guards never see tokens; keep comparisons harmless
        return False

    def __neg__(self):                # relations between two outputs of the wheel carry no number: they become inert tuples
        return ("neg", self)

    def __add__(self, other):
        return ("sum", self, other)

    __radd__ = __add__


class RecordingFDM(dict):
    OUTPUTS = ("atmosphere/T-R", "atmosphere/P-psf", "atmosphere/density-altitude", "atmosphere/pressure-altitude",
               "atmosphere/rho-slugs_ft3", "atmosphere/T-sl-R", "atmosphere/a-sl-fps", "atmosphere/P-sl-psf",
               "atmosphere/rho-sl-slugs_ft3", "fcs/flap-pos-deg",
               "test/output-pid-rect", "test/output-pid-trap", "test/output-pid-ab2", "test/output-pid-ab3", "test/output-integrator",
               "pid/negative-combined", "pid/kp-alone", "pid/ki-alone", "pid/kd-alone", "pid/kp-alone-inverted-input-sign",
               "pid/ki-alone-inverted-input-sign", "pid/kd-alone-inverted-input-sign")

    def __init__(self, log):
        super().__init__()
        self.log = log
        self.frame = 0
        self.time = 0.0
        self.dt = 1.0 / 120.0          # FGFDMExec's default dT (FGFDMExec.cpp:96); these tests never change it
        self["simulation/sim-time-sec"] = 0.0
        self["fcs/flap-cmd-norm"] = 0.0

    def __getitem__(self, k):
        if k in self.OUTPUTS:
            if k == "fcs/flap-pos-deg" and self.frame == 0 and not self.log.get("_started"):
                return 0.0            # assertEqual(fdm['fcs/flap-pos-deg'], 0.0) before the sequence starts
            return Token(k, {"h": self.get("ic/h-sl-ft", 0.0), "dT": self.get("atmosphere/delta-T", 0.0),
                             "frame": self.frame, "cmd": self.get("fcs/flap-cmd-norm", 0.0)})
        return super().__getitem__(k)

    def __setitem__(self, k, v):
        if k == "fcs/flap-cmd-norm" and v != 0.0:
            self.log["_started"] = True
        super().__setitem__(k, v)

    def load_model(self, *_a, **_k): return True
    def load_ic(self, *_a, **_k): return True
    def run_ic(self): return True

    def run(self):                     # sim time is the running fp64 sum of dT (FGFDMExec.cpp:196-203)
        self.frame += 1
        self.time += self.dt
        super().__setitem__("simulation/sim-time-sec", self.time)
        self.log.setdefault("cmd_by_frame", []).append(self.get("fcs/flap-cmd-norm", 0.0))
        self.log.setdefault("pid_in_by_frame", []).append((self.get("test/input", 0.0), self.get("test/trigger", 0.0)))
        return True

    def set_dt(self, dt):
        self.dt = dt


def load_test_module(name, log):
    """Import a reference test script with stand-ins for the absent jsbsim wheel and its helper module."""
    utils = types.ModuleType("JSBSim_utils")

    class JSBSimTestCase:
        def __init__(self):
            self.sandbox = None

        def setUp(self, *args):
            pass

        def create_fdm(self):
            return RecordingFDM(log)

        def assertEqual(self, a, b):
            pass

        def assertAlmostEqual(self, a, b, places=7, delta=None):
            log["_delta"] = delta
            if (isinstance(a, tuple) and a[0] in ("neg", "sum")) or (isinstance(b, tuple) and b[0] in ("neg", "sum")):
                return
            if isinstance(a, (Token, tuple)) and isinstance(b, (Token, tuple)):
                return      # a round trip between two outputs of the wheel (rho at the density altitude): no number to take
            for x, y in ((a, b), (b, a)):
                if isinstance(x, Token):                       # fdm[...] ~= expected
                    x.inputs["delta"] = delta
                    log.setdefault(x.prop, []).append((x.inputs, float(y)))
                    return
                if isinstance(x, tuple) and x and x[0] == "ratio":   # 1.0 ~= expected / fdm[...]
                    _, num, tok = x
                    log.setdefault(tok.prop, []).append((tok.inputs, float(num) * float(y)))
                    return

    class FlightModel:                # JSBSim_utils.FlightModel: loads an aircraft plus an extra <system> file, start() -> fdm
        def __init__(self, test_case, name):
            self.fdm = RecordingFDM(log)

        def include_system_test_file(self, fname):
            pass

        def before_loading(self):
            pass

        def start(self):
            self.before_loading()
            return self.fdm

    utils.JSBSimTestCase = JSBSimTestCase
    utils.FlightModel = FlightModel
    utils.RunTest = lambda cls: None
    utils.CreateFDM = lambda sandbox: RecordingFDM(log)
    utils.append_xml = lambda n: n if n.endswith(".xml") else n + ".xml"
    utils.CopyAircraftDef = utils.ExecuteUntil = lambda *a, **k: None
    sys.modules["JSBSim_utils"] = utils
    sys.modules.setdefault("jsbsim", types.ModuleType("jsbsim"))
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(TESTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rows(entries, key):
    return np.array([[e[0]["h"], e[0]["dT"], e[1]] for e in entries], dtype=np.float64) if key != "frame" else None


def main():
    out = {}

    # ---- ISA walk: the standard day and a 15 K bias (the other TestStdAtmosphere cases change sea-level pressure or add a
    # temperature gradient, which the reference's env never does)
    log = {}
    m = load_test_module("TestStdAtmosphere", log)
    tc = m.TestStdAtmosphere()
    tc.setUp()
    tc.test_std_atmosphere()
    tc.test_temperature_bias()
    out["isa_T"] = rows(log["atmosphere/T-R"], "h")
    out["isa_P"] = rows(log["atmosphere/P-psf"], "h")
    out["isa_sl"] = np.array([log["atmosphere/T-sl-R"][0][1], log["atmosphere/P-sl-psf"][0][1], log["atmosphere/rho-sl-slugs_ft3"][0][1],
                              log["atmosphere/a-sl-fps"][0][1]])

    # ---- density / pressure altitude tables
    for name, prop, key in (("TestDensityAltitude", "atmosphere/density-altitude", "density_altitude"),
                            ("TestPressureAltitude", "atmosphere/pressure-altitude", "pressure_altitude")):
        log = {}
        m = load_test_module(name, log)
        tc = getattr(m, name)()
        getattr(tc, "test_" + name[4:].lower())()
        out[key] = rows(log[prop], "h")

    # ---- kinematic timing
    log = {}
    m = load_test_module("TestKinematic", log)
    tc = m.TestKinematic()
    tc.testKinematicTiming()
    exp = log["fcs/flap-pos-deg"]
    out["kin_frame"] = np.array([e[0]["frame"] for e in exp], dtype=np.int64)
    out["kin_expected"] = np.array([e[1] for e in exp], dtype=np.float64)
    out["kin_cmd_by_frame"] = np.array(log["cmd_by_frame"], dtype=np.float64)     # command in force during run() number k+1
    flap = et.parse(os.path.join(DATA, "aircraft", "c172r", "c172r.xml")).getroot().find("flight_control/channel/kinematic")
    out["kin_detents"] = np.array([float(s.find("position").text) for s in flap.iter("setting")])
    out["kin_times"] = np.array([float(s.find("time").text) for s in flap.iter("setting")])
    out["kin_dt"] = np.array(1.0 / 120.0)

    # ---- turbine: seek, the default spool-up law, and runScript's frame-by-frame prediction for the F-16's engine
    log = {}
    m = load_test_module("TestTurbine", log)
    eng = et.parse(os.path.join(DATA, "engine", "F100-PW-229.xml")).getroot()
    idleN1, maxN1 = float(eng.find("idlen1").text), float(eng.find("maxn1").text)
    idleN2, maxN2 = float(eng.find("idlen2").text), float(eng.find("maxn2").text)
    BPR = float(eng.find("bypassratio").text)
    N1f, N2f = maxN1 - idleN1, maxN2 - idleN2
    tc = m.TestTurbine.__new__(m.TestTurbine)
    dt = 1.0 / 120.0
    tc.delay = 90.0 * dt / (BPR + 3.0)                      # TestTurbine.py:92
    sp = []
    for sigma in (1.0, 0.74, 0.31):
        tc.fdm = {"atmosphere/sigma": sigma}
        for n2norm in np.linspace(-0.05, 1.1, 24):
            sp.append((n2norm, sigma, tc.defaultSpoolUp(n2norm)))
    out["spool_grid"] = np.array(sp)
    out["spool_dt"] = np.array(dt)
    out["engine_consts"] = np.array([idleN1, maxN1, idleN2, maxN2, BPR])
    traj = []
    for sigma in (1.0, 0.53):
        tc.fdm = {"atmosphere/sigma": sigma}
        n1, n2, thr = idleN1, idleN2, 1.0
        for _ in range(20000):
            N2norm = (n2 - idleN2) / N2f
            if n2 >= 100.0:
                thr = 0.0                                   # :47-49 trigger the spool down
            up = tc.defaultSpoolUp(N2norm)
            n1n = m.seek(n1, idleN1 + thr * N1f, up, up * 2.4)     # :85-88 the legacy <bypassratio> law: N1 down x2.4, N2 down x3
            n2n = m.seek(n2, idleN2 + thr * N2f, up, up * 3.0)
            traj.append((sigma, thr, n1, n2, N2norm, n1n, n2n))
            n1, n2 = n1n, n2n
            if thr == 0.0 and (n2 - idleN2) / N2f == 0.0:
                break
    out["spool_traj"] = np.array(traj)
    # ---- <pid>: integration schemes, trigger semantics (positive suspends, negative resets), kp / ki / kd terms
    for method, tag in (("test_integrators", "integ"), ("test_pid", "pid")):
        log = {}
        m = load_test_module("TestIntegrators", log)
        tc = m.TestIntegrators()
        getattr(tc, method)()
        out[f"{tag}_in_by_frame"] = np.array(log["pid_in_by_frame"], dtype=np.float64)     # (input, trigger) in force during run() k+1
        for prop in ("test/output-pid-ab2", "test/output-pid-ab3", "test/output-pid-rect", "pid/kp-alone", "pid/ki-alone", "pid/kd-alone"):
            if prop in log:
                key = f"{tag}|{prop.split('/')[-1]}"
                out[key] = np.array([[e[0]["frame"], e[1], e[0]["delta"] if e[0]["delta"] is not None else 5e-8] for e in log[prop]])
    out["pid_gains"] = np.array([2.0, 0.5, -1.5])      # kp, ki, kd set by test_pid (TestIntegrators.py:115-117)
    out["pid_dt"] = np.array(0.005)                    # FDMIntegrators.before_loading (:26-28)
    np.savez_compressed(os.path.join(OUT, "jsbsim_blocks.npz"), **out)
    for k, v in out.items():
        print(f"{k:20s} {v.shape}")


if __name__ == "__main__":
    main()
