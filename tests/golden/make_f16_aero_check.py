#!/usr/bin/env python3
"""Independent pin of the oracle's F-16 aerodynamic build-up (FGAerodynamics::Run summation over f16.xml:994-1925) and of the table data.

tools/gen_f16_tables.py turns f16.xml / F100-PW-229.xml into oracle/f16_tables.h and csrc/f16_tables.h, and oracle/f16_fdm.c's
aerodynamics_run() hard-codes which property multiplies which table: a mis-parsed table or a mis-read product would be the same on
the CPU and the GPU side. This script shares no code with either. It (1) reads every <tableData> block of the two XML files with
its own tokenizer and stores the raw tables under their XML names, and (2) interprets the <aerodynamics> section generically
(<axis> -> <function> -> <product> of <property> / <value> / <table>, JSBSim semantics: clamped linear look-up, FGTable.cpp:443-516)
for random property values, storing inputs and the six axis sums. Runs only in the build container (/root/reference present);
output: tests/golden/f16_aero_check.npz (numbers only).
"""
import os
import re

import numpy as np

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
F16 = os.path.join(REF, "envs/JSBSim/data/aircraft/f16/f16.xml")
ENG = os.path.join(REF, "envs/JSBSim/data/engine/F100-PW-229.xml")
OUT = os.path.dirname(os.path.abspath(__file__))

TAG = re.compile(r"<(/?)([A-Za-z_][\w.-]*)((?:\s+[\w:.-]+\s*=\s*\"[^\"]*\")*)\s*(/?)>|([^<]+)")
ATTR = re.compile(r"([\w:.-]+)\s*=\s*\"([^\"]*)\"")


def parse(path):
    """Tiny XML reader (elements, attributes, text; comments and the prolog dropped) -> nested [tag, attrs, children, text]."""
    text = re.sub(r"<!--.*?-->", "", open(path, encoding="utf-8", errors="replace").read(), flags=re.S)
    text = re.sub(r"<\?.*?\?>", "", text, flags=re.S)
    root = ["#root", {}, [], ""]
    stack = [root]
    for m in TAG.finditer(text):
        close, tag, attrs, selfclose, chars = m.groups()
        if chars is not None:
            stack[-1][3] += chars
        elif close:
            assert stack[-1][0] == tag, (stack[-1][0], tag)
            stack.pop()
        else:
            node = [tag, dict(ATTR.findall(attrs or "")), [], ""]
            stack[-1][2].append(node)
            if not selfclose:
                stack.append(node)
    return root


def find_all(node, tag):
    for ch in node[2]:
        if ch[0] == tag:
            yield ch
        yield from find_all(ch, tag)


def read_table(tnode):
    ivs = [(iv[1].get("lookup", "row"), iv[3].strip()) for iv in tnode[2] if iv[0] == "independentVar"]
    data = next(ch for ch in tnode[2] if ch[0] == "tableData")[3]
    lines = [[float(x) for x in ln.split()] for ln in data.strip().splitlines() if ln.strip()]
    if len(ivs) == 1:
        arr = np.array(lines)
        assert arr.shape[1] == 2
        return ivs, arr[:, 0], np.zeros(0), arr[:, 1]
    cols = np.array(lines[0])
    body = np.array(lines[1:])
    assert body.shape[1] == len(cols) + 1
    assert [k for k, _ in ivs] == ["row", "column"]
    return ivs, body[:, 0], cols, body[:, 1:]


def lookup1(x, y, key):      # FGTable::GetValue(double), FGTable.cpp:443-476: clamp, then linear
    if key <= x[0]:
        return y[0]
    if key >= x[-1]:
        return y[-1]
    r = int(np.searchsorted(x, key, side="left"))
    f = (key - x[r - 1]) / (x[r] - x[r - 1])
    return y[r - 1] + f * (y[r] - y[r - 1])


def lookup2(rk, ck, v, rkey, ckey):   # FGTable::GetValue(double, double), :480-516
    def frac(ax, key):
        r = 1
        while r < len(ax) - 1 and ax[r] < key:
            r += 1
        f = (key - ax[r - 1]) / (ax[r] - ax[r - 1])
        return r, min(max(f, 0.0), 1.0)
    r, rf = frac(rk, rkey)
    c, cf = frac(ck, ckey)
    c1 = rf * (v[r, c - 1] - v[r - 1, c - 1]) + v[r - 1, c - 1]
    c2 = rf * (v[r, c] - v[r - 1, c]) + v[r - 1, c]
    return c1 + cf * (c2 - c1)


PROPS = ["aero/alpha-rad", "aero/beta-rad", "velocities/mach", "aero/qbar-psf", "aero/bi2vel", "aero/ci2vel",
         "velocities/p-aero-rad_sec", "velocities/q-aero-rad_sec", "velocities/r-aero-rad_sec", "fcs/elevator-pos-rad",
         "fcs/aileron-pos-rad", "fcs/rudder-pos-rad", "fcs/lef-pos-rad", "fcs/flaperon-mix-rad", "fcs/speedbrake-pos-rad",
         "gear/gear-pos-norm", "aero/h_b-mac-ft"]


def main():
    out = {}
    root = parse(F16)
    eng = parse(ENG)
    names = []
    for src, tree in (("f16", root), ("eng", eng)):
        for fn in list(find_all(tree, "function")) + list(find_all(tree, "scheduled_gain")):
            tabs = list(find_all(fn, "table"))
            if not tabs:
                continue
            ivs, rk, ck, v = read_table(tabs[0])
            nm = fn[1]["name"]
            names.append(nm)
            out["tab|%s|rows" % nm], out["tab|%s|cols" % nm], out["tab|%s|vals" % nm] = rk, ck, v
    out["table_names"] = np.array(names)

    met = next(find_all(root, "metrics"))
    consts = {"metrics/Sw-sqft": float(next(find_all(met, "wingarea"))[3]), "metrics/bw-ft": float(next(find_all(met, "wingspan"))[3]),
              "metrics/cbarw-ft": float(next(find_all(met, "chord"))[3])}
    aero = next(find_all(root, "aerodynamics"))
    kclge = next(f for f in aero[2] if f[0] == "function")
    _, gx, _, gy = read_table(next(find_all(kclge, "table")))
    axes = [a for a in aero[2] if a[0] == "axis"]
    assert [a[1]["name"] for a in axes] == ["DRAG", "SIDE", "LIFT", "ROLL", "PITCH", "YAW"]

    def evaluate(pv):
        pv = dict(pv)
        pv.update(consts)
        pv["aero/function/kCLge"] = lookup1(gx, gy, pv["aero/h_b-mac-ft"])
        sums = []
        for ax in axes:
            tot = 0.0
            for fn in ax[2]:
                if fn[0] != "function":
                    continue
                prod = next(ch for ch in fn[2] if ch[0] == "product")
                val = 1.0
                for term in prod[2]:
                    if term[0] == "property":
                        val *= pv[term[3].strip()]
                    elif term[0] == "value":
                        val *= float(term[3])
                    elif term[0] == "table":
                        ivs, rk, ck, v = read_table(term)
                        val *= lookup1(rk, v, pv[ivs[0][1]]) if len(ivs) == 1 else lookup2(rk, ck, v, pv[ivs[0][1]], pv[ivs[1][1]])
                    else:
                        raise ValueError(term[0])
                tot += val
            sums.append(tot)
        return sums

    rng = np.random.default_rng(20250404)
    n = 400
    lo = np.array([-0.25, -0.6, 0.0, 20.0, 0.005, 0.002, -3.0, -1.5, -1.5, -0.5, -0.4, -0.55, 0.0, 0.0, 0.0, 0.0, 0.0])
    hi = np.array([0.85, 0.6, 2.2, 1500.0, 0.05, 0.02, 3.0, 1.5, 1.5, 0.5, 0.4, 0.55, 0.45, 0.36, 1.05, 1.0, 1.3])
    X = rng.uniform(lo, hi, size=(n, len(PROPS)))
    X[:20, 0] = rng.uniform(-0.4, -0.18, 20)       # below / above the alpha axis
    X[20:40, 0] = rng.uniform(0.78, 1.0, 20)
    X[40:60, 2] = rng.uniform(2.0, 3.0, 20)        # beyond the Mach axes
    X[:, 5] = X[:, 4] * consts["metrics/cbarw-ft"] / consts["metrics/bw-ft"]   # both come from one true airspeed (FGAerodynamics.cpp:150-156)
    Y = np.array([evaluate(dict(zip(PROPS, row))) for row in X])
    out["aero_inputs"], out["aero_sums"] = X, Y
    out["aero_input_names"] = np.array(PROPS)

    # the three engine tables on a (mach, density altitude) sample, incl. beyond both axes
    e_names = [f[1]["name"] for f in find_all(eng, "function")]
    M = rng.uniform(-0.2, 3.0, 300)
    H = rng.uniform(-15000.0, 75000.0, 300)
    out["eng_inputs"] = np.stack([M, H], axis=1)
    for nm in e_names:
        _, rk, ck, v = read_table(next(find_all(next(f for f in find_all(eng, "function") if f[1]["name"] == nm), "table")))
        out["eng|%s" % nm] = np.array([lookup2(rk, ck, v, m, h) for m, h in zip(M, H)])
    np.savez_compressed(os.path.join(OUT, "f16_aero_check.npz"), **out)
    print(len(names), "tables;", n, "aero samples; engine functions:", e_names)


if __name__ == "__main__":
    main()
