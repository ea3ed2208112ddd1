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

    # ------------------------------------------------------------------ packed layout for the HIP kernel
    # Tables that share their breakpoint axes are interleaved so that one wavefront lane fetches every value it
    # needs at a breakpoint with one 16-byte LDS read (ds_read_b128) instead of one 4-byte read per table:
    #   PA1  [12 alpha][16]      : 16 one-dimensional alpha tables in four groups of four
    #   PAE  [12 alpha][5 de][4] : CDDh, CLDh, CmDh, pad            (alpha x elevator)
    #   PAB13[12 alpha][13 b][2] : Clb, Cnb                          (alpha x beta, 5-degree grid)
    #   PAB7 [12 alpha][7 b][4]  : Clda, Cldr, Cnda, Cndr            (alpha x beta, 10-degree grid)
    #   PM   [NM mach][12]       : the nine Mach tables resampled on the union of their breakpoints (a piecewise-linear
    #                              function with clamped ends is exactly representable on any finer grid)
    #   PENG [14 mach][8 alt][4] : idle, mil, aug thrust factors, pad (rows beyond a table's last Mach repeat it: clamp)
    T = {name: t for name, t, _ in tables}

    def t1(name):
        t = T[name]
        return t["data"][:t["nr"]], t["data"][t["nr"]:]

    def t2(name):
        t = T[name]
        nr, nc = t["nr"], t["nc"]
        return t["data"][:nr], t["data"][nr:nr + nc], [t["data"][nr + nc + r * nc:nr + nc + (r + 1) * nc] for r in range(nr)]

    def interp_clamped(x, y, key):
        if key <= x[0]:
            return y[0]
        if key >= x[-1]:
            return y[-1]
        for i in range(1, len(x)):
            if key <= x[i]:
                f = (key - x[i - 1]) / (x[i] - x[i - 1])
                return y[i - 1] + f * (y[i] - y[i - 1])

    alpha_axis = t1("CDDLEF")[0]
    groups1 = ["CDDLEF", "CDQ", "CDQ_DLEF", "CLDLEF", "CYP", "CYR", "CLP", "CLR", "CLQ", "CMQ", "CNP", "CNR",
               "CDDSB", "CLDSB", "CLQ_DSB", "CMDSB"]
    for g in groups1:
        assert t1(g)[0] == alpha_axis, g
    pack = []
    offs = {}

    def put(name, vals):
        while len(pack) % 4:
            pack.append(0.0)
        offs[name] = len(pack)
        pack.extend(vals)

    put("ALPHA_X", alpha_axis)
    de_axis = t2("CDDH")[1]
    b13_axis = t2("CLB")[1]
    b7_axis = t2("CLDA")[1]
    put("DE_X", de_axis)
    put("B13_X", b13_axis)
    put("B7_X", b7_axis)
    put("A1", [t1(g)[1][r] for r in range(12) for g in groups1])
    for nm in ("CDDH", "CLDH", "CMDH"):
        assert t2(nm)[0] == alpha_axis and t2(nm)[1] == de_axis
    put("AE", [v for r in range(12) for c in range(5) for v in (t2("CDDH")[2][r][c], t2("CLDH")[2][r][c], t2("CMDH")[2][r][c], 0.0)])
    for nm in ("CLB", "CNB"):
        assert t2(nm)[0] == alpha_axis and t2(nm)[1] == b13_axis
    put("AB13", [v for r in range(12) for c in range(13) for v in (t2("CLB")[2][r][c], t2("CNB")[2][r][c])])
    for nm in ("CLDA", "CLDR", "CNDA", "CNDR"):
        assert t2(nm)[0] == alpha_axis and t2(nm)[1] == b7_axis
    put("AB7", [v for r in range(12) for c in range(7) for v in (t2("CLDA")[2][r][c], t2("CLDR")[2][r][c], t2("CNDA")[2][r][c], t2("CNDR")[2][r][c])])
    mach_tabs = ["CDMACH", "CYB_M", "CLB_M", "CLDA_M", "CLDR_M", "CMA_M", "CNB_M", "CNDA_M", "CNDR_M"]
    mach_axis = sorted(set(x for m in mach_tabs for x in t1(m)[0]))
    put("MACH_X", mach_axis)
    put("M", [v for xm in mach_axis for v in ([interp_clamped(*t1(m), xm) for m in mach_tabs] + [0.0, 0.0, 0.0])])
    eng = {k: t2(k) for k in ("ENG_IDLETHRUST", "ENG_MILTHRUST", "ENG_AUGTHRUST")}
    alt_axis = eng["ENG_MILTHRUST"][1]
    assert all(e[1] == alt_axis for e in eng.values())
    assert alt_axis == [-10000.0 + 10000.0 * i for i in range(8)]
    eng_rows = 14
    for k, (rk, ck, vv) in eng.items():
        assert all(abs(rk[i] - 0.2 * i) < 1e-12 for i in range(len(rk))), k
    put("ENG", [v for r in range(eng_rows) for c in range(8) for v in (
        eng["ENG_IDLETHRUST"][2][min(r, 5)][c], eng["ENG_MILTHRUST"][2][min(r, 7)][c], eng["ENG_AUGTHRUST"][2][min(r, 13)][c], 0.0)])
    put("KCLGE_X", t1("KCLGE")[0])
    put("KCLGE_Y", t1("KCLGE")[1])
    while len(pack) % 4:
        pack.append(0.0)
    lines.append("/* ---- packed, axis-interleaved layout staged in LDS by the HIP kernel (see tools/gen_f16_tables.py) */")
    for k, v in offs.items():
        lines.append("#define P_%s_OFF %d" % (k, v))
    lines.append("#define P_MACH_N %d" % len(mach_axis))
    lines.append("#define P_A1_STRIDE 16")
    lines.append("#define P_M_STRIDE 12")
    lines.append("#define F16_PACK_LEN %d" % len(pack))
    lines.append("static const double F16_PACK[F16_PACK_LEN] = {")
    for i in range(0, len(pack), 8):
        lines.append("  " + ", ".join("%r" % v for v in pack[i:i + 8]) + ",")
    lines.append("};")
    # small FCS schedules as compile-time constants (looked up with compare/select chains, no memory access)
    for nm in ("FCS_AILERON_SPEED_COMPENSATED", "FCS_ELEVATOR_SCHEDULER", "FCS_YAW_RATE_NORM"):
        x, y = t1(nm)
        lines.append("#define C_%s_N %d" % (nm, len(x)))
        lines.append("#define C_%s_X {%s}" % (nm, ", ".join("%rf" % v for v in x)))
        lines.append("#define C_%s_Y {%s}" % (nm, ", ".join("%rf" % v for v in y)))
    for nm, ax in (("ALPHA", alpha_axis), ("DE", de_axis), ("B13", b13_axis), ("B7", b7_axis), ("MACH", mach_axis)):
        lines.append("#define C_%s_N %d" % (nm, len(ax)))
        lines.append("#define C_%s_X {%s}" % (nm, ", ".join("%rf" % v for v in ax)))
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
