#!/usr/bin/env python3
"""Extract the F-16 model DATA (coefficient tables and scalar constants) from the
reference's JSBSim XML files into a C header.

Inputs  (read-only, this container only):
  /root/reference/envs/JSBSim/data/aircraft/f16/f16.xml
  /root/reference/envs/JSBSim/data/engine/F100-PW-229.xml
Outputs (identical copies; the data is numbers only, no reference source text):
  oracle/f16_tables.h
  aircombat-selfplay_amd/csrc/f16_tables.h

Layout of the packed blob F16_TAB[] (doubles):
  1-D table  : x[NR] then y[NR]
  2-D table  : rowkeys[NR] then colkeys[NC] then values[NR*NC] (row-major)
For every table T the header defines T_<NAME>_OFF, T_<NAME>_NR, T_<NAME>_NC (NC = 0 for 1-D).
"""
import os
import sys
import xml.etree.ElementTree as ET

REF = os.environ.get("AC_REFERENCE_ROOT", "/root/reference")
F16 = os.path.join(REF, "envs/JSBSim/data/aircraft/f16/f16.xml")
ENG = os.path.join(REF, "envs/JSBSim/data/engine/F100-PW-229.xml")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUTS = [os.path.join(ROOT, "oracle", "f16_tables.h"),
        os.path.join(ROOT, "aircombat-selfplay_amd", "csrc", "f16_tables.h")]


def parse_table(tab):
    ivs = [(iv.get("lookup", "row"), iv.text.strip()) for iv in tab.findall("independentVar")]
    rows = [l.split() for l in tab.find("tableData").text.strip().split("\n") if l.strip()]
    if len(ivs) == 1:
        x = [float(a[0]) for a in rows]
        y = [float(a[1]) for a in rows]
        return dict(nr=len(x), nc=0, data=x + y, ivs=ivs)
    cols = [float(a) for a in rows[0]]
    rk = [float(a[0]) for a in rows[1:]]
    vals = []
    for a in rows[1:]:
        assert len(a) == len(cols) + 1
        vals += [float(v) for v in a[1:]]
    order = dict(ivs)
    assert ivs[0][0] == "row" and ivs[1][0] == "column", order
    return dict(nr=len(rk), nc=len(cols), data=rk + cols + vals, ivs=ivs)


def cname(s):
    return s.split("/")[-1].replace("-", "_").upper()


def main():
    root = ET.parse(F16).getroot()
    tables = []   # (name, table dict, comment)
    consts = []   # (name, value, comment)

    # ---- aerodynamics
    aero = root.find("aerodynamics")
    for el in aero:
        funcs = [el] if el.tag == "function" else list(el)
        for f in funcs:
            name = cname(f.get("name"))
            tab = next(f.iter("table"), None)
            props = [p.text.strip() for p in f.iter("property")]
            if tab is not None:
                t = parse_table(tab)
                tables.append((name, t, "f16.xml %s * table(%s)" % (" * ".join(props), ", ".join(i[1] for i in t["ivs"]))))
            else:
                v = float(next(f.iter("value")).text)
                consts.append(("F16_K_" + name, v, "f16.xml %s * value" % " * ".join(props)))

    # ---- flight control tables (scheduled gains)
    fc = root.find("flight_control")
    for sg in fc.iter("scheduled_gain"):
        name = "FCS_" + cname(sg.get("name"))
        t = parse_table(sg.find("table"))
        tables.append((name, t, "f16.xml scheduled_gain %s on %s" % (sg.get("name"), t["ivs"][0][1])))

    # ---- engine tables
    eng = ET.parse(ENG).getroot()
    for f in eng.findall("function"):
        name = "ENG_" + f.get("name").upper()
        t = parse_table(f.find("table"))
        tables.append((name, t, "F100-PW-229.xml %s(mach, density-altitude)" % f.get("name")))
    for key in ["milthrust", "maxthrust", "bypassratio", "tsfc", "atsfc", "idlen1", "idlen2", "maxn1", "maxn2",
                "augmented", "augmethod", "injected"]:
        consts.append(("F16_ENG_" + key.upper(), float(eng.find(key).text), "F100-PW-229.xml <%s>" % key))

    # ---- metrics / mass
    met = root.find("metrics")
    for key in ["wingarea", "wingspan", "chord"]:
        consts.append(("F16_" + key.upper(), float(met.find(key).text), "f16.xml metrics/%s" % key))
    for loc in met.findall("location"):
        for ax in "xyz":
            consts.append(("F16_%s_%s" % (loc.get("name"), ax.upper()), float(loc.find(ax).text),
                           "f16.xml metrics/location %s [in]" % loc.get("name")))
    mb = root.find("mass_balance")
    assert mb.get("negated_crossproduct_inertia") == "true"
    for key in ["ixx", "iyy", "izz", "ixy", "ixz", "iyz", "emptywt"]:
        consts.append(("F16_" + key.upper(), float(mb.find(key).text), "f16.xml mass_balance/%s" % key))
    cg = mb.find("location")
    for ax in "xyz":
        consts.append(("F16_CG_" + ax.upper(), float(cg.find(ax).text), "f16.xml mass_balance CG [in]"))
    for i, pm in enumerate(mb.findall("pointmass")):
        consts.append(("F16_PM%d_WEIGHT" % i, float(pm.find("weight").text), "f16.xml pointmass %s [lbs]" % pm.get("name")))
        for ax in "xyz":
            consts.append(("F16_PM%d_%s" % (i, ax.upper()), float(pm.find("location").find(ax).text),
                           "f16.xml pointmass %s [in]" % pm.get("name")))
    pr = root.find("propulsion")
    for i, tk in enumerate(pr.findall("tank")):
        consts.append(("F16_TANK%d_CONTENTS" % i, float(tk.find("contents").text), "f16.xml tank %d [lbs]" % i))
        consts.append(("F16_TANK%d_CAPACITY" % i, float(tk.find("capacity").text), "f16.xml tank %d [lbs]" % i))
        for ax in "xyz":
            consts.append(("F16_TANK%d_%s" % (i, ax.upper()), float(tk.find("location").find(ax).text),
                           "f16.xml tank %d [in]" % i))
    th = pr.find("engine").find("thruster").find("location")
    for ax in "xyz":
        consts.append(("F16_THRUSTER_" + ax.upper(), float(th.find(ax).text), "f16.xml thruster location [in]"))

    # ---- emit
    blob = []
    lines = []
    lines.append("/* GENERATED by tools/gen_f16_tables.py from the reference's f16.xml and F100-PW-229.xml.")
    lines.append(" * DATA ONLY (coefficient tables and scalar constants). Do not edit by hand. */")
    lines.append("#ifndef F16_TABLES_H")
    lines.append("#define F16_TABLES_H")
    lines.append("")
    for name, v, cm in consts:
        lines.append("#define %-24s (%r)  /* %s */" % (name, v, cm))
    lines.append("")
    for name, t, cm in tables:
        off = len(blob)
        blob += t["data"]
        lines.append("/* %s */" % cm)
        lines.append("#define T_%s_OFF %d" % (name, off))
        lines.append("#define T_%s_NR %d" % (name, t["nr"]))
        lines.append("#define T_%s_NC %d" % (name, t["nc"]))
    lines.append("")
    lines.append("#define F16_TAB_LEN %d" % len(blob))
    lines.append("static const double F16_TAB[F16_TAB_LEN] = {")
    for i in range(0, len(blob), 8):
        lines.append("  " + ", ".join("%r" % v for v in blob[i:i + 8]) + ",")
    lines.append("};")
    lines.append("")
    lines.append("#endif /* F16_TABLES_H */")
    text = "\n".join(lines) + "\n"
    for o in OUTS:
        os.makedirs(os.path.dirname(o), exist_ok=True)
        with open(o, "w") as f:
            f.write(text)
        print("wrote", o, len(blob), "doubles,", len(tables), "tables,", len(consts), "constants")


if __name__ == "__main__":
    sys.exit(main())
