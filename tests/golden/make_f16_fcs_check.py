#!/usr/bin/env python3
"""Independent pin of the F-16's own wiring: the <flight_control> section of f16.xml (which component feeds which, in which order,
with which gains, limits and switch tests) and the <metrics> / <mass_balance> / <propulsion> constants.

oracle/f16_fdm.c's fcs_run() and massbalance_run() are f16.xml:317-992 and :37-92,255-316 read by hand. This script reads the same
XML text with the tokenizer of make_f16_aero_check.py (no code shared with tools/gen_f16_tables.py or the oracle) and interprets it
GENERICALLY, component semantics taken from the JSBSim sources the reference vendors (R/envs/JSBSim/data/src/models/flight_control/):

  channels and components in document order, every tick        FGFCS.cpp:153-178
  <input> with a leading '-' negates; <output> nodes + the component's own name node all receive Output   FGFCSComponent.cpp:122-150,300-323
  <clipto> min / max                                          FGFCSComponent.cpp:266-290
  <switch>: default first, then the first passing <test> wins  FGSwitch.cpp:125-155, conditions FGCondition.cpp:102-205
  <pure_gain> / <scheduled_gain> / <aerosurface_scale> (zero-centred domain -> range)   FGGain.cpp:138-170
  <summer>                                                     FGSummer.cpp:72-86
  <fcs_function>                                               FGFCSFunction.cpp:84-97
  <pid> (default integrator = Adams-Bashforth 2, trigger semantics, Kd * (in - in_prev) / dt)   FGPID.cpp:84-98,154-204
  <kinematic> (re-reads its FIRST output node as its state)    FGKinemat.cpp:99-170
  the tied FGFCS surface properties (-pos-rad / -pos-deg / -pos-norm are three views of one surface: writing deg sets rad)
                                                               FGFCS.cpp:200-330,697-742; gear defaults DOWN :81
  dt latched by every component at load time = 1/120 s         FGFCSComponent.cpp:58, FGFDMExec.cpp:96, R/envs/JSBSim/core/simulatior.py:165-169

It then drives that interpreter with 512 input sequences x 200 ticks (commands, alpha, Mach, calibrated airspeed, body rates, load
factors, attitude: piecewise linear / piecewise constant between stored knots, chosen so that every switch branch, every clip, the
three PID trigger regimes and every kinematic are exercised) and stores the knots and every named surface / throttle / PID output.
The mass model is read the same way: <metrics>, <mass_balance> (base inertia with its negated_crossproduct_inertia flag, empty weight,
CG, point masses) and the <propulsion> tanks, evaluated per FGMassBalance.cpp:181-262 + FGPropulsion.cpp:554-575 for random tank
contents. Runs only in the build container (/root/reference present); output: tests/golden/f16_fcs_check.npz (numbers and property
names only).
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_f16_aero_check import ENG, F16, find_all, lookup1, parse, read_table  # noqa: E402  (the second tokenizer, not the table generator)

FCS_DT = 1.0 / 120.0
RADTODEG = 57.295779513082320876798154814105          # FGJSBBase radtodeg
DEGTORAD = 0.017453292519943295769236907684886
NT = 200                                               # ticks per sequence
KNOT_EVERY = 10
NK = NT // KNOT_EVERY + 1

# inputs of the flight control system that come from outside it (the rest of the FDM, the pilot); order = column order of the fixture
IN_PROPS = ["fcs/aileron-cmd-norm", "fcs/elevator-cmd-norm", "fcs/rudder-cmd-norm", "fcs/throttle-cmd-norm", "gear/gear-cmd-norm",
            "velocities/vc-kts", "velocities/mach", "velocities/p-aero-rad_sec", "velocities/q-aero-rad_sec", "velocities/r-aero-rad_sec",
            "velocities/vg-fps", "velocities/v-fps", "attitude/pitch-rad", "attitude/roll-rad", "accelerations/n-pilot-z-norm",
            "accelerations/n-pilot-y-norm", "aero/alpha-rad"]
OUT_PROPS = ["fcs/aileron-pos-rad", "fcs/elevator-pos-rad", "fcs/rudder-pos-rad", "fcs/lef-pos-rad", "fcs/flaperon-mix-rad",
             "fcs/speedbrake-pos-rad", "fcs/throttle-pos-norm", "gear/gear-pos-norm", "fcs/tef-control", "fcs/left-aileron-pos-norm",
             "fcs/elevator-pos-norm", "fcs/rudder-pos-norm", "fcs/roll-rate-pid", "fcs/g-load-pid", "fcs/yaw-load-pid", "fcs/speedbrake-pos-deg"]
SURFACES = ("left-aileron", "right-aileron", "elevator", "rudder", "flap", "speedbrake", "spoiler")


def expand_inputs(knots, linear):
    """knots [NK][n_in] at ticks 0, 10, .. 200; linear[n_in] picks interpolation (1) or hold (0). -> [NT][n_in], the value in force
    during tick t. Shared with the test (tests/test_oracle_f16_wiring.py imports this module for it)."""
    t = np.arange(NT, dtype=np.float64)
    k = (t // KNOT_EVERY).astype(int)
    f = (t - k * KNOT_EVERY) / KNOT_EVERY
    lin = knots[k] + f[:, None] * (knots[k + 1] - knots[k])
    return np.where(np.asarray(linear, bool)[None, :], lin, knots[k])


class Store:
    """The property tree as far as the FCS sees it: plain nodes plus FGFCS's tied surface triples."""

    def __init__(self):
        self.p = {}
        self.surf = {s: {"rad": 0.0, "deg": 0.0, "norm": 0.0} for s in SURFACES}

    def _tied(self, name):
        if name.startswith("fcs/") and "-pos-" in name:
            base, form = name[4:].rsplit("-pos-", 1)
            if base.startswith("mag-"):
                return base[4:], "mag"
            if base in self.surf and form in ("rad", "deg", "norm"):
                return base, form
        return None

    def has(self, name):
        return self._tied(name) is not None or name in self.p

    def get(self, name):
        t = self._tied(name)
        if t:
            return abs(self.surf[t[0]]["rad"]) if t[1] == "mag" else self.surf[t[0]][t[1]]
        return self.p[name]

    def set(self, name, v):
        t = self._tied(name)
        if t:
            s = self.surf[t[0]]
            if t[1] == "rad":
                s["rad"], s["deg"] = v, v * RADTODEG
            elif t[1] == "deg":
                s["rad"], s["deg"] = v * DEGTORAD, v
            else:
                s["norm"] = v
        else:
            self.p[name] = v

    def value_of(self, token):
        """FGPropertyValue / FGParameterValue: a number, or a property name with an optional leading '-'."""
        token = token.strip()
        try:
            return float(token)
        except ValueError:
            pass
        sign = 1.0
        if token.startswith("-"):
            sign, token = -1.0, token[1:]
        return sign * self.get(token)


def child(node, tag):
    return next((c for c in node[2] if c[0] == tag), None)


def children(node, tag):
    return [c for c in node[2] if c[0] == tag]


def clamp(lo, v, hi):
    return lo if v < lo else (hi if v > hi else v)


OPS = {"EQ": "eq", "NE": "ne", "GT": "gt", "GE": "ge", "LT": "lt", "LE": "le", "==": "eq", "!=": "ne", ">": "gt", ">=": "ge", "<": "lt", "<=": "le"}


def eval_function(node, st):
    """FGFunction subset the F-16's fcs_function uses: product / sum / cos / sin / property / value / table."""
    tag = node[0]
    if tag == "function":
        return eval_function(next(c for c in node[2] if c[0] != "description"), st)
    if tag == "product":
        v = 1.0
        for c in node[2]:
            v *= eval_function(c, st)
        return v
    if tag == "sum":
        return sum(eval_function(c, st) for c in node[2])
    if tag in ("cos", "sin"):
        return getattr(math, tag)(eval_function(node[2][0], st))
    if tag in ("property", "p"):
        return st.value_of(node[3])
    if tag in ("value", "v"):
        return float(node[3])
    if tag == "table":
        ivs, rk, _, v = read_table(node)
        assert len(ivs) == 1
        return lookup1(rk, v, st.get(ivs[0][1]))
    raise ValueError(tag)


class Component:
    def __init__(self, node, st):
        self.node, self.kind, self.name = node, node[0], node[1]["name"]
        self.output = 0.0
        self.inputs = [c[3].strip() for c in children(node, "input")]
        self.out_nodes = [c[3].strip() for c in children(node, "output")]
        own = self.name if "/" in self.name else "fcs/" + self.name.lower().replace(" ", "-")
        self.out_nodes.append(own)                      # bind(): after the explicit <output> nodes
        for n in self.out_nodes:
            if not st.has(n):
                st.set(n, self.output)                  # a node that already exists keeps its value
        clip = child(node, "clipto")
        self.clip = (child(clip, "min")[3], child(clip, "max")[3]) if clip is not None else None
        if self.kind == "kinematic":
            sets = children(child(node, "traverse"), "setting")
            self.detents = [float(child(s, "position")[3]) for s in sets]
            self.times = [float(child(s, "time")[3]) for s in sets]
            self.scale = child(node, "noscale") is None
        elif self.kind == "pid":
            self.in_prev = self.in_prev2 = self.i_total = 0.0
            ki = child(node, "ki")
            self.int_type = "ab2" if ki is None or ki[1].get("type", "") not in ("rect", "trap", "ab3") else ki[1]["type"]
            self.has_ki = ki is not None
            self.gains = [child(node, k)[3] if child(node, k) is not None else "0.0" for k in ("kp", "ki", "kd")]
            trig = child(node, "trigger")
            self.trigger = trig[3].strip() if trig is not None else None
            self.standard = node[1].get("type", "") == "standard"
        elif self.kind == "switch":
            self.tests = []
            d = child(node, "default")
            if d is not None:
                self.tests.append((None, None, d[1]["value"]))
            def group(t):   # FGCondition.cpp:55-100: the element's data lines, then its nested <test> groups, under one AND / OR
                leaves = []
                for ln in t[3].strip().splitlines():
                    if ln.strip():
                        a, op, b = ln.split()
                        leaves.append((a, OPS[op] if op in OPS else OPS[op.upper()], b))
                return (t[1].get("logic", "AND"), leaves, [group(c) for c in children(t, "test")])
            for t in children(node, "test"):
                self.tests.append((group(t), None, t[1]["value"]))
        elif self.kind in ("pure_gain", "scheduled_gain", "aerosurface_scale"):
            g = child(node, "gain")
            self.gain = g[3] if g is not None else "1.0"
            if self.kind == "scheduled_gain":
                ivs, self.trow, _, self.tval = read_table(child(node, "table"))
                assert len(ivs) == 1
                self.tvar = ivs[0][1]
            if self.kind == "aerosurface_scale":
                dom, rng = child(node, "domain"), child(node, "range")
                self.in_min, self.in_max = (float(child(dom, "min")[3]), float(child(dom, "max")[3])) if dom is not None else (-1.0, 1.0)
                self.out_min, self.out_max = float(child(rng, "min")[3]), float(child(rng, "max")[3])
                zc = child(node, "zero_centered")
                self.zero_centered = not (zc is not None and zc[3].strip() in ("0", "false"))
        elif self.kind == "summer":
            b = child(node, "bias")
            self.bias = float(b[3]) if b is not None else 0.0
        elif self.kind != "fcs_function":
            raise ValueError("component type not in the F-16's flight_control: " + self.kind)

    def compare(self, st, a, op, b):
        x, y = st.value_of(a), st.value_of(b)
        return {"eq": x == y, "ne": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[op]

    def run(self, st):
        k = self.kind
        if k == "switch":
            passed, default_out = False, 0.0
            def evaluate(g):   # FGCondition::Evaluate, FGCondition.cpp:150-205
                logic, leaves, subs = g
                r = [self.compare(st, *c) for c in leaves] + [evaluate(x) for x in subs]
                return all(r) if logic == "AND" else any(r)
            for cond, _, value in self.tests:
                if cond is None:
                    default_out = st.value_of(value)
                else:
                    passed = evaluate(cond)
                if passed:
                    self.output = st.value_of(value)
                    break
            if not passed:
                self.output = default_out
        elif k == "pure_gain":
            self.output = st.value_of(self.gain) * st.value_of(self.inputs[0])
        elif k == "scheduled_gain":
            self.output = st.value_of(self.gain) * lookup1(self.trow, self.tval, st.get(self.tvar)) * st.value_of(self.inputs[0])
        elif k == "aerosurface_scale":
            x = st.value_of(self.inputs[0])
            if self.zero_centered:
                self.output = 0.0 if x == 0.0 else ((x / self.in_max) * self.out_max if x > 0 else (x / self.in_min) * self.out_min)
            else:
                self.output = self.out_min + ((x - self.in_min) / (self.in_max - self.in_min)) * (self.out_max - self.out_min)
            self.output *= st.value_of(self.gain)
        elif k == "summer":
            self.output = 0.0
            for i in self.inputs:
                self.output += st.value_of(i)
            self.output += self.bias
        elif k == "fcs_function":
            self.output = eval_function(child(self.node, "function"), st)
            if self.inputs:
                self.output *= st.value_of(self.inputs[0])
        elif k == "pid":
            x = st.value_of(self.inputs[0])
            dval = (x - self.in_prev) / FCS_DT
            test = st.value_of(self.trigger) if self.trigger else 0.0
            i_delta = 0.0
            if abs(test) < 0.000001 and self.has_ki:
                i_delta = {"rect": x, "trap": 0.5 * (x + self.in_prev), "ab2": 1.5 * x - 0.5 * self.in_prev,
                           "ab3": (23.0 * x - 16.0 * self.in_prev + 5.0 * self.in_prev2) / 12.0}[self.int_type]
            if test < 0.0:
                self.i_total = 0.0
            kp, ki, kd = (st.value_of(g) for g in self.gains)
            self.i_total += ki * FCS_DT * i_delta
            self.output = kp * (x + self.i_total + kd * dval) if self.standard else kp * x + self.i_total + kd * dval
            self.in_prev2 = 0.0 if test < 0.0 else self.in_prev
            self.in_prev = x
        elif k == "kinematic":
            dt0 = FCS_DT
            x = st.value_of(self.inputs[0])
            det, tt = self.detents, self.times
            if self.scale:
                x *= det[-1]
            out = st.get(self.out_nodes[0])
            x = clamp(det[0], x, det[-1])
            while dt0 > 0.0 and not (abs(x - out) <= 2.0 * sys.float_info.epsilon * max(abs(x), abs(out))):
                ind = 1
                while (det[ind] < out) if x < out else (det[ind] <= out):
                    ind += 1                            # an IndexError here = the state left the detent range: not a valid fixture
                if tt[ind] <= 0.0:
                    out = x
                    break
                rate = (det[ind] - det[ind - 1]) / tt[ind]
                this_in = clamp(det[ind - 1], x, det[ind])
                this_dt = abs((this_in - out) / rate)
                if dt0 < this_dt:
                    this_dt = dt0
                    out = out + this_dt * rate if out < x else out - this_dt * rate
                else:
                    out = this_in
                dt0 -= this_dt
            self.output = out
        if self.clip:
            lo, hi = st.value_of(self.clip[0]), st.value_of(self.clip[1])
            if hi - lo >= 0.0:
                self.output = clamp(lo, self.output, hi)
        for n in self.out_nodes:
            st.set(n, self.output)


class F16FlightControl:
    def __init__(self, fc_node):
        self.st = st = Store()
        # FGFCS's own nodes and their start values (FGFCS.cpp:69-89,697-782)
        for n in ("fcs/aileron-cmd-norm", "fcs/elevator-cmd-norm", "fcs/rudder-cmd-norm", "fcs/flap-cmd-norm", "fcs/speedbrake-cmd-norm",
                  "fcs/spoiler-cmd-norm", "fcs/pitch-trim-cmd-norm", "fcs/roll-trim-cmd-norm", "fcs/yaw-trim-cmd-norm", "fcs/steer-cmd-norm",
                  "fcs/throttle-cmd-norm", "fcs/throttle-pos-norm", "gear/unit[1]/WOW", "gear/unit[2]/WOW", "velocities/u-fps", "aero/alpha-deg"):
            st.set(n, 0.0)
        st.set("gear/gear-cmd-norm", 1.0)
        st.set("gear/gear-pos-norm", 1.0)
        for p in IN_PROPS:
            if not st.has(p):
                st.set(p, 0.0)
        for pr in children(fc_node, "property"):        # interface properties declared by the section
            st.set(pr[3].strip(), float(pr[1].get("value", 0.0)))
        self.channels = [[Component(c, st) for c in ch[2] if c[0] not in ("description", "documentation")] for ch in children(fc_node, "channel")]

    def tick(self, inputs):
        st = self.st
        for k, v in inputs.items():
            st.set(k, v)
        st.set("aero/alpha-deg", st.get("aero/alpha-rad") * RADTODEG)
        st.set("fcs/throttle-pos-norm", st.get("fcs/throttle-cmd-norm"))       # FGFCS::Run copies cmd -> pos before the channels (:162)
        for ch in self.channels:
            for c in ch:
                c.run(st)
        return [st.get(p) for p in OUT_PROPS]


def make_knots(rng, seq):
    """One sequence's knots. Regimes rotate with the sequence number so that every branch of every switch sees traffic."""
    regime = seq % 8
    k = np.zeros((NK, len(IN_PROPS)))
    lin = np.ones(len(IN_PROPS), dtype=np.int8)
    col = {p: i for i, p in enumerate(IN_PROPS)}

    def walk(lo, hi, step, start=None):
        x = rng.uniform(lo, hi) if start is None else start
        out = []
        for _ in range(NK):
            out.append(x)
            x = clamp(lo, x + rng.normal(0.0, step) + (rng.uniform(lo, hi) - x) * (rng.random() < 0.15), hi)
        return np.array(out)

    for c in ("fcs/aileron-cmd-norm", "fcs/elevator-cmd-norm", "fcs/rudder-cmd-norm"):      # the policy's commands: held, like the env does
        k[:, col[c]] = rng.uniform(-1.0, 1.0, NK) * (rng.random(NK) < 0.8)
        lin[col[c]] = seq % 3 == 0
    k[:, col["fcs/throttle-cmd-norm"]] = rng.uniform(0.0, 0.9, NK)
    lin[col["fcs/throttle-cmd-norm"]] = 0
    k[:, col["gear/gear-cmd-norm"]] = 1.0 if regime not in (5, 6) else (0.0 if regime == 5 else rng.integers(0, 2, NK))
    lin[col["gear/gear-cmd-norm"]] = 0
    vc_lo, vc_hi = {0: (230.0, 270.0), 1: (0.0, 30.0), 2: (2.0, 25.0)}.get(regime, (120.0, 750.0))
    k[:, col["velocities/vc-kts"]] = walk(vc_lo, vc_hi, 0.08 * (vc_hi - vc_lo))
    k[:, col["velocities/mach"]] = walk(0.75, 1.05, 0.05) if regime in (0, 3) else walk(0.1, 1.8, 0.15)
    k[:, col["velocities/p-aero-rad_sec"]] = walk(-3.0, 3.0, 0.6)
    k[:, col["velocities/q-aero-rad_sec"]] = walk(-0.8, 0.8, 0.15)
    k[:, col["velocities/r-aero-rad_sec"]] = walk(-0.6, 0.6, 0.1)
    k[:, col["velocities/vg-fps"]] = walk(60.0, 170.0, 15.0) if regime in (1, 2) else walk(200.0, 1500.0, 80.0)
    k[:, col["velocities/v-fps"]] = walk(-40.0, 60.0, 12.0)
    k[:, col["attitude/pitch-rad"]] = walk(-1.4, 1.4, 0.2)
    k[:, col["attitude/roll-rad"]] = walk(-3.1, 3.1, 0.5)
    k[:, col["accelerations/n-pilot-z-norm"]] = walk(-9.0, 4.0, 1.0)
    k[:, col["accelerations/n-pilot-y-norm"]] = walk(-1.5, 1.5, 0.3)
    if regime == 4:          # deep stall: alpha through 53 deg with little sideslip velocity -> speedbrake limiter
        k[:, col["aero/alpha-rad"]] = walk(0.8, 1.1, 0.06)
        k[:, col["velocities/v-fps"]] = walk(-5.0, 30.0, 8.0)
    elif regime in (5, 6, 7):  # around the leading-edge-flap and elevator-scheduler breakpoints
        k[:, col["aero/alpha-rad"]] = walk(-0.05, 0.35, 0.05)
    else:
        k[:, col["aero/alpha-rad"]] = walk(-0.6, 0.7, 0.12)
    return k, lin


def mass_model(root):
    """<metrics> / <mass_balance> / <propulsion> tanks + thruster as plain numbers (inches, lbs, slug ft2)."""
    met, mb, prop = (next(find_all(root, t)) for t in ("metrics", "mass_balance", "propulsion"))

    def xyz(loc):
        return [float(child(loc, a)[3]) for a in "xyz"]
    out = {"wingarea": float(child(met, "wingarea")[3]), "wingspan": float(child(met, "wingspan")[3]), "chord": float(child(met, "chord")[3])}
    for loc in children(met, "location"):
        out["loc_" + loc[1]["name"]] = xyz(loc)
    for a in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz", "emptywt"):
        out[a] = float(child(mb, a)[3])
    out["negated"] = 0.0 if mb[1].get("negated_crossproduct_inertia") == "false" else 1.0
    out["cg"] = xyz(next(loc for loc in children(mb, "location") if loc[1]["name"] == "CG"))
    pms = children(mb, "pointmass")
    out["pm_weight"] = [float(child(p, "weight")[3]) for p in pms]
    out["pm_xyz"] = [xyz(child(p, "location")) for p in pms]
    tanks = children(prop, "tank")
    out["tank_xyz"] = [xyz(child(t, "location")) for t in tanks]
    out["tank_contents"] = [float(child(t, "contents")[3]) for t in tanks]
    out["tank_capacity"] = [float(child(t, "capacity")[3]) for t in tanks]
    assert all(child(t, "radius") is None for t in tanks)     # no radius => FGTank's local inertia stays 0 (FGTank.cpp:61,411)
    eng = child(prop, "engine")
    out["thruster_xyz"] = xyz(child(child(eng, "thruster"), "location"))
    out["n_engines"] = float(len(children(prop, "engine")))
    eng_root = next(find_all(parse(ENG), "turbine_engine"))
    for tag in ("milthrust", "maxthrust", "bypassratio", "tsfc", "atsfc", "idlen1", "idlen2", "maxn1", "maxn2", "augmented", "augmethod", "injected"):
        out["eng_" + tag] = float(child(eng_root, tag)[3])
    return out


LBTOSLUG = 1.0 / 32.174049
INCHTOFT = 1.0 / 12.0


def mass_balance(m, tank_lbs, pm_weight, cg_for_tanks=None):
    """FGMassBalance::Run (FGMassBalance.cpp:181-262); tank inertia about `cg_for_tanks` (FGPropulsion::CalculateTankInertias ->
    GetPointmassInertia with MassBalance's CURRENT vXYZcg, i.e. last pass's: FGFDMExec.cpp:572 loads it before MassBalance runs)."""
    def pm_inertia(cg, mass, r):     # FGMassBalance::GetPointmassInertia: v = StructuralToBody(r) about cg, [in] -> [ft], x and z flipped
        v = np.array([cg[0] - r[0], r[1] - cg[1], cg[2] - r[2]]) * INCHTOFT
        sv = mass * v
        xx, yy, zz = sv * v
        xy, xz, yz = -sv[0] * v[1], -sv[0] * v[2], -sv[1] * v[2]
        return np.array([[yy + zz, xy, xz], [xy, xx + zz, yz], [xz, yz, xx + yy]])
    tank_xyz = np.array(m["tank_xyz"])
    weight = m["emptywt"] + float(np.sum(tank_lbs)) + float(np.sum(pm_weight))
    moment = m["emptywt"] * np.array(m["cg"]) + sum(w * np.array(r) for w, r in zip(pm_weight, m["pm_xyz"])) + sum(w * r for w, r in zip(tank_lbs, tank_xyz))
    cg = moment / weight
    s = 1.0 if m["negated"] else -1.0
    J = np.array([[m["ixx"], -s * m["ixy"], s * m["ixz"]], [-s * m["ixy"], m["iyy"], -s * m["iyz"]], [s * m["ixz"], -s * m["iyz"], m["izz"]]])
    J = J + pm_inertia(cg, LBTOSLUG * m["emptywt"], m["cg"])
    for w, r in zip(pm_weight, m["pm_xyz"]):
        J = J + pm_inertia(cg, LBTOSLUG * w, r)
    tcg = cg if cg_for_tanks is None else cg_for_tanks
    tankJ = sum(pm_inertia(tcg, LBTOSLUG * w, r) for w, r in zip(tank_lbs, tank_xyz))
    J = J + tankJ
    return weight, cg, J, np.linalg.inv(J), tankJ


def main():
    root = parse(F16)
    fc = next(find_all(root, "flight_control"))
    counts = {}
    for ch in children(fc, "channel"):
        for c in ch[2]:
            counts[c[0]] = counts.get(c[0], 0) + 1
    print("flight_control components:", counts)

    rng = np.random.default_rng(20251004)
    NS, NFULL = 512, 64
    knots = np.zeros((NS, NK, len(IN_PROPS)))
    linear = np.zeros((NS, len(IN_PROPS)), dtype=np.int8)
    gear0 = np.ones(NS)
    full = np.zeros((NFULL, NT, len(OUT_PROPS)))
    sampled = np.zeros((NS - NFULL, NT // KNOT_EVERY, len(OUT_PROPS)))
    branch = {}
    for s in range(NS):
        knots[s], linear[s] = make_knots(rng, s)
        fcs = F16FlightControl(fc)
        if s % 8 == 5:                                   # gear already up (a user-set start state): the 0.436 rad LEF branch
            gear0[s] = 0.0
            fcs.st.set("gear/gear-pos-norm", 0.0)
        X = expand_inputs(knots[s], linear[s])
        for t in range(NT):
            y = fcs.tick(dict(zip(IN_PROPS, X[t])))
            if s < NFULL:
                full[s, t] = y
            elif t % KNOT_EVERY == KNOT_EVERY - 1:
                sampled[s - NFULL, t // KNOT_EVERY] = y
            for nm in ("fcs/lef-pos-rad", "fcs/tef-pos-rad", "fcs/speedbrake-alpha-limiter", "fcs/aileron-pid-trigger", "fcs/elevator-pid-trigger",
                       "fcs/rudder-pid-trigger"):
                branch.setdefault(nm, set()).add(round(fcs.st.get(nm), 4))
    print("switch outputs seen:", {k: sorted(v) for k, v in branch.items()})
    out = {"in_props": np.array(IN_PROPS), "out_props": np.array(OUT_PROPS), "knots": knots, "linear": linear, "gear_pos0": gear0,
           "out_full": full, "out_sampled": sampled, "fcs_dt": np.array(FCS_DT), "component_counts": np.array(sorted(counts.items()), dtype=object).astype(str)}

    m = mass_model(root)
    for k, v in m.items():
        out["mass|" + k] = np.array(v, dtype=np.float64)
    n = 200
    tanks = rng.uniform(0.0, 1.0, (n, len(m["tank_contents"]))) * np.array(m["tank_capacity"])
    tanks[:20, 2:] = 0.0
    tanks[0] = m["tank_contents"]
    tanks[1] = 0.0
    pmw = np.tile(np.array(m["pm_weight"]), (n, 1))
    pmw[1] = 0.0                                          # TestPointMassInertia.testInertiaMatrix: no pilot, empty tanks
    rows = []
    for i in range(n):
        w, cg, J, Ji, tJ = mass_balance(m, tanks[i], pmw[i])
        rows.append(np.concatenate([[w], cg, J.ravel(), Ji.ravel(), tJ.ravel()]))
    out["mb_tanks"], out["mb_pm_weight"], out["mb_result"] = tanks, pmw, np.array(rows)
    np.savez_compressed(os.path.join(HERE, "f16_fcs_check.npz"), **out)
    print("wrote f16_fcs_check.npz:", {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("mass|")})


if __name__ == "__main__":
    main()
