#!/usr/bin/env python3
"""The reference's own unit tests of the flight-control COMPONENT types, run against the generic interpreter that pins the F-16's wiring.

tests/golden/f16_fcs_check.npz (the fixture oracle/f16_fdm.c's fcs_run() is held to) comes from make_f16_fcs_check.py's generic
reading of <switch>, <pure_gain>, <summer>, <fcs_function>, ... . What says that THAT reading of a component is JSBSim's? The reference
holds JSBSim's unit tests of those very component types, with the system files they load:

  TestGain.py:27-57       tests/gain.xml      <pure_gain>: numeric gain, property gain, '-' on the gain property, '-' on the input
  TestSwitch.py:25-129    tests/switch.xml    <switch>: lt / eq / == / gt / ge / le, '-' on the tested property and on the value property,
                                               AND (default) and logic="OR", <default>, groups of nested <test>s (GitHub issue #176)
  TestFunctions.py:27-160 tests/function.xml  <fcs_function> with <sum> / <product> / <sin> / <p> / <v>, '-' on an <input>, <summer> with and
                                               without <bias>  (the test's other functions -- random, rotations, interpolate1d, quotient,
                                               not, pi -- are not part of the F-16's flight_control and are reported as not covered)

This script loads those test scripts as they are, with a stand-in for the absent ``jsbsim`` wheel whose FDM IS the generic interpreter
(``Component`` / ``Store`` of make_f16_fcs_check.py over the same tokenizer): ``fdm[name] = v`` sets a property, ``fdm.run()`` /
``run_ic()`` run the file's components once in document order, ``fdm[name]`` reads a property. Every assertion the tests make on a
covered property is checked on the spot (a failure stops the script) and logged. Output: tests/golden/jsbsim_components.npz -- per
test the tokenised system file (a data file of the reference's tests, as JSON) and the event list (set / run / expect), so that
tests/test_oracle_f16_wiring.py can replay them where /root/reference does not exist. Runs only in the build container.
"""
import importlib.util
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_f16_aero_check import parse  # noqa: E402
from make_f16_fcs_check import Component, Store, children  # noqa: E402

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
TESTS = os.path.join(REF, "envs", "JSBSim", "data", "tests")


class Named(float):
    """A property value that remembers which property it was read from."""
    def __new__(cls, v, name):
        o = float.__new__(cls, v)
        o.name = name
        return o


class System:
    """One <system> file under the generic interpreter (also used by the replaying test: build from the stored tree)."""

    def __init__(self, tree):
        self.st = Store()
        root = next(c for c in tree[2] if c[0] == "system")
        for pr in children(root, "property"):
            self.st.set(pr[3].strip(), float(pr[1].get("value", 0.0)))
        self.comps, self.uncovered = [], []
        for ch in children(root, "channel"):
            for c in ch[2]:
                if c[0] in ("description", "documentation"):
                    continue
                try:
                    comp = Component(c, self.st)
                    comp.run(self.st)               # construction-time dry run: an element the interpreter does not know raises here
                    comp.__init__(c, self.st)       # (state back to the initial one)
                    for n in comp.out_nodes:
                        self.st.set(n, 0.0)
                    self.comps.append(comp)
                except (ValueError, KeyError, StopIteration, AssertionError, IndexError, TypeError):
                    self.uncovered.append(c[1].get("name", c[0]))

    def run(self):
        for c in self.comps:
            c.run(self.st)

    def covered(self, name):
        return self.st.has(name) and name not in self.uncovered


def load(name, events, holder):
    """Import the reference's test script `name` with the stand-in JSBSim_utils; `events` receives the replay log."""
    utils = types.ModuleType("JSBSim_utils")

    class FDM:
        def __init__(self):
            self.sys = None
        def set_aircraft_path(self, *a): pass
        def set_systems_path(self, *a): pass
        def load_model(self, *a, **k): return True
        def run_ic(self):
            return self.run()
        def run(self):
            events.append(["run"])
            self.sys.run()
            return True
        def __setitem__(self, k, v):
            events.append(["set", k, float(v)])
            self.sys.st.set(k, float(v))
        def __getitem__(self, k):
            if self.sys.covered(k):
                return Named(self.sys.st.get(k), k)
            return Named(float("nan"), k)

    class FlightModel:
        def __init__(self, tc, model):
            self.fdm = FDM()
        def include_system_test_file(self, fname):
            tree = parse(os.path.join(TESTS, fname))
            holder["file"], holder["tree"] = fname, tree
            self.fdm.sys = System(tree)
            holder["uncovered"] = list(self.fdm.sys.uncovered)
        def start(self):
            self.fdm.run_ic()
            return self.fdm

    class JSBSimTestCase:
        def _log(self, a, b, places, delta):
            if not isinstance(a, Named) or math.isnan(a) or isinstance(b, Named) and math.isnan(b):
                holder["skipped"] = holder.get("skipped", 0) + 1
                return False
            events.append(["expect", a.name, float(b), places, delta])
            return True

        def assertAlmostEqual(self, a, b, places=None, msg=None, delta=None):
            pl = 7 if places is None and delta is None else places
            if self._log(a, b, pl, delta):
                ok = abs(float(a) - float(b)) <= delta if delta is not None else round(abs(float(a) - float(b)), pl) == 0
                assert ok, (name, a.name, float(a), float(b))

        def assertEqual(self, a, b, msg=None):
            if self._log(a, b, None, None):
                assert float(a) == float(b), (name, a.name, float(a), float(b))

        def assertTrue(self, a, msg=None): pass

    utils.JSBSimTestCase, utils.FlightModel = JSBSimTestCase, FlightModel
    utils.RunTest = lambda cls: None
    sys.modules["JSBSim_utils"] = utils
    sys.modules.setdefault("jsbsim", types.ModuleType("jsbsim"))
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(TESTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def prune(tree, uncovered):
    """The tokenised system file without the components the interpreter does not cover."""
    def keep(node):
        return not (isinstance(node, list) and len(node) == 4 and isinstance(node[1], dict) and node[1].get("name") in uncovered)
    def walk(node):
        return [node[0], node[1], [walk(c) for c in node[2] if keep(c)], node[3]]
    return walk(tree)


def main():
    out = {}
    for script, cls, methods in (("TestGain", "TestGain", ["test_conditions"]), ("TestSwitch", "TestSwitch", ["test_conditions", "test_nested"]),
                                 ("TestFunctions", "TestFunctions", ["test_functions"])):
        for meth in methods:
            events, holder = [], {}
            mod = load(script, events, holder)
            tc = getattr(mod, cls)()
            getattr(tc, meth)()
            n_exp = sum(1 for e in events if e[0] == "expect")
            names = sorted({e[1] for e in events if e[0] == "expect"})
            key = f"{script}.{meth}"
            # (only the components the interpreter covers travel with the fixture: the rest of the test's system file is not needed to replay it)
            tree = prune(holder["tree"], set(holder["uncovered"]))
            out[key] = np.array(json.dumps({"file": holder["file"], "tree": tree, "events": events, "uncovered": holder["uncovered"],
                                            "skipped": holder.get("skipped", 0)}))
            print(f"{key}: {n_exp} expectations hold on {len(names)} properties ({', '.join(names)}); "
                  f"{holder.get('skipped', 0)} assertions on properties outside the F-16's component set not covered: {holder['uncovered']}")
    np.savez_compressed(os.path.join(HERE, "jsbsim_components.npz"), **out)


if __name__ == "__main__":
    main()
