#!/usr/bin/env python3
"""Generate aircombat-selfplay_amd/csrc/f16_split.hpp from tick() of f16_device.hpp: the same statements, cut at the section
comments into the pieces the three-wave step kernels run on different waves (dynamics wave: rates / velocity update, auxiliary, table
look-ups, assembly + accelerations; systems wave: FCS, propulsion, mass balance; kinematics wave: attitude / position / frame /
gravity / atmosphere of the coming tick). tick() stays the single source of the
arithmetic; re-run this script after editing it (the build checks that the generated file is current)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "aircombat-selfplay_amd", "csrc", "f16_device.hpp")
OUT = os.path.join(ROOT, "aircombat-selfplay_amd", "csrc", "f16_split.hpp")


def cut(txt, start, end):
    i = txt.index(start)
    j = txt.index(end, i)
    return txt[i:j]


def rep(txt, old, new):
    assert old in txt, old
    return txt.replace(old, new)


def generate():
    s = open(SRC).read()
    a = s.index("template <bool DT_ZERO, bool TANK_ARM_ORIGIN = false>\n__device__ __forceinline__ void tick_after_propagate(State& s, Derived& d, const Tab& T) {")
    body = s[a:s.index("template <bool DT_ZERO, bool TANK_ARM_ORIGIN = false>\n__device__ __forceinline__ void tick(State& s")]
    # the integration step lives in propagate(); together with the frame calls that open tick_after_propagate() it is the
    # "Propagate" section the pieces below are cut from
    pb = cut(s, "__device__ __forceinline__ void propagate(State& s) {\n", "\n}\n")
    pb = pb[pb.index("\n") + 1:]
    pb = rep(pb, "  constexpr float dt = 1.0f / 60.0f;\n", "")
    frames = cut(body, "  locate_fast(s, d);", "  // ---------------- Inertial")
    prop = "  // ---------------- Propagate (FGPropagate.cpp:218-290, :336-360)\n  if (!DT_ZERO) {\n" + "\n".join("  " + l if l.strip() else l for l in pb.split("\n")) + "\n  }\n" + frames
    grav = cut(body, "  // ---------------- Inertial", "  // ---------------- Atmosphere")
    fcs = cut(body, "  // ---------------- FCS", "  // ---------------- MassBalance")
    mass = cut(body, "  // ---------------- MassBalance", "  // ---------------- Auxiliary")
    aux = cut(body, "  // ---------------- Auxiliary", "  // ---------------- Propulsion")
    eng = cut(body, "  // ---------------- Propulsion", "  // ---------------- Aerodynamics")
    aero = cut(body, "  // ---------------- Aerodynamics", "  // ---------------- Accelerations")
    acc = cut(body, "  // ---------------- Accelerations", "  // publish this tick's auxiliary")
    i_c = acc.index("    // symmetric 3x3 inverse by cofactors")
    i_w = acc.index("    s.wdx = ")
    cof = acc[i_c:i_w]
    acc = acc[:i_c] + acc[i_w:]
    acc = rep(acc, "    float im_ = 1.0f / mass;\n", "")
    for old, new in (("    float c00 = ", "    c00 = "), (", c01 = ", "; c01 = "), (", c02 = ", "; c02 = "), ("    float c11 = ", "    c11 = "), (", c12 = ", "; c12 = "),
                     (", c22 = ", "; c22 = "), ("    float idet = ", "    idet = ")):
        cof = rep(cof, old, new)
    cof += "    im_ = 1.0f / mass;\n"
    pub = body[body.index("  // publish this tick's auxiliary"):]
    pub = pub[:pub.rindex("}")]

    # ---- dynamics, part 1: propagate .. gravity (DT_ZERO = false only: the suspended passes exist in the init kernels alone)
    prop = rep(prop, "  if (!DT_ZERO) {", "  {")
    prop = rep(prop, "  const float* Tb = d.T;\n", "")
    # the kinematics wave integrates attitude and position (they depend on LAST tick's rates and velocity only: explicit
    # integrators), the dynamics wave integrates rates and velocity with this tick's accelerations
    kin, kinq, lite = [], [], []
    for line in prop.split("\n"):
        t = line.strip()
        if t.startswith(("float qd", "float a0", "float rn", "s.q0 =")):
            kinq.append(line)
        elif t.startswith(("const float k =", "s.rx +=", "s.ry +=", "s.rz +=")):
            kin.append(line)
        elif t.startswith(("s.hv2x =", "s.hv1x =")):
            kin.append(line); lite.append(line)
        elif t.startswith(("s.wp =", "s.vx +=", "s.vy +=", "s.vz +=", "s.ha1x =", "s.ticks +=")):
            lite.append(line)
        else:
            assert t in ("", "{", "}", "locate_fast(s, d);", "body_frame(s, d);") or t.startswith("//"), t
    kin, kinq, lite = "\n".join(kin) + "\n", "\n".join(kinq) + "\n", "\n".join(lite) + "\n"
    bf = s[s.index("__device__ __forceinline__ void body_frame(const State& s, Derived& d) {"):]
    bf = bf[bf.index("{") + 2:bf.index("\n}\n")] + "\n"
    i_om = bf.index("  const float om = (float)kOmega;")
    bf_T, bf_rest = bf[:i_om], bf[i_om:]
    bf_T = rep(bf_T, "  float* T = d.T;\n", "")
    bf_rest = rep(bf_rest, "om * (float)s.ry", "om * ryf")
    bf_rest = rep(bf_rest, "om * (float)s.rx", "om * rxf")
    grav = rep(grav, "  float gx = kxy * rxf, gy = kxy * ryf, gz = kz * rzf;", "  gx = kxy * rxf; gy = kxy * ryf; gz = kz * rzf;")
    grav = rep(grav, "  float rxf = (float)s.rx, ryf = (float)s.ry, rzf = (float)s.rz;", "  const float rxf = (float)s.rx, ryf = (float)s.ry, rzf = (float)s.rz;")
    # ---- dynamics, part 2: atmosphere, mass balance, auxiliary
    mass = rep(mass, "  float mass = kLb2Slug * weight;", "  mass = kLb2Slug * weight;")
    mass = rep(mass, "  float cgx = ((float)(F16_EMPTYWT", "  cgx = ((float)(F16_EMPTYWT")
    mass = rep(mass, "  float cgy = ((float)F16_TANK0_Y", "  cgy = ((float)F16_TANK0_Y")
    mass = rep(mass, "  float cgz = ((float)(F16_EMPTYWT", "  cgz = ((float)(F16_EMPTYWT")
    mass = rep(mass, "  float Jxx = (float)F16_IXX, Jyy = (float)F16_IYY, Jzz = (float)F16_IZZ, Jxy = 0.0f, Jxz = (float)F16_IXZ, Jyz = 0.0f;",
               "  Jxx = (float)F16_IXX; Jyy = (float)F16_IYY; Jzz = (float)F16_IZZ; Jxy = 0.0f; Jxz = (float)F16_IXZ; Jyz = 0.0f;")
    mass = rep(mass, "const bool origin = TANK_ARM_ORIGIN && i >= 2;", "const bool origin = false;")
    aux = rep(aux, "  float vt = sqrtf(vt2);", "  vt = sqrtf(vt2);")
    aux = rep(aux, "  float alpha = 0.0f, beta = 0.0f, ca = 1.0f, sa = 0.0f, cb = 1.0f, sb = 0.0f;", "  alpha = 0.0f; beta = 0.0f; ca = 1.0f; sa = 0.0f; cb = 1.0f; sb = 0.0f;")
    aux = rep(aux, "  float qbar = 0.5f * A.rho * vt2;", "  qbar = 0.5f * A.rho * vt2;")
    aux = rep(aux, "  float mach = vt / A.a;", "  mach = vt / A.a;")
    aux = rep(aux, "  float vg = sqrtf(d.vn * d.vn + d.ve * d.ve);", "  vg = sqrtf(d.vn * d.vn + d.ve * d.ve);")
    aux = rep(aux, "  float qc = (mach > 0.0f) ? pitot_impact_pressure(mach, A.P) : 0.0f;", "  qc = (mach > 0.0f) ? pitot_impact_pressure(mach, A.P) : 0.0f;")
    aux = rep(aux, "  float npx = (s.bax + t1x + t2x) * (1.0f / kG0), npy = (s.bay + t1y + t2y) * (1.0f / kG0), npz = (s.baz + t1z + t2z) * (1.0f / kG0);",
              "  npx = (s.bax + t1x + t2x) * (1.0f / kG0); npy = (s.bay + t1y + t2y) * (1.0f / kG0); npz = (s.baz + t1z + t2z) * (1.0f / kG0);")
    # ---- systems: FCS
    fcs = rep(fcs, "  float aileron_rad = 0.375f * roll_cmd;", "  aileron_rad = 0.375f * roll_cmd;")
    fcs = rep(fcs, "  float flaperon_rad = 1.4324f *", "  flaperon_rad = 1.4324f *")
    fcs = rep(fcs, "  float cthcph = Tb[6] * d.d_eci[0] + Tb[7] * d.d_eci[1] + Tb[8] * d.d_eci[2];\n", "")
    fcs = rep(fcs, "  float elevator_rad = 0.436f * s.elev;", "  elevator_rad = 0.436f * s.elev;")
    fcs = rep(fcs, "  float rudder_rad = 0.524f * slew(", "  rudder_rad = 0.524f * slew(")
    fcs = rep(fcs, "  float lef_rad = (alpha_p > 0.0873f)", "  lef_rad = (alpha_p > 0.0873f)")
    fcs = rep(fcs, "  float throttle_pos = 2.0f * s.thr;", "  throttle_pos = 2.0f * s.thr;")
    fcs = rep(fcs, "(d.v <= 18.0f)", "(vbody <= 18.0f)")
    fcs = rep(fcs, "  float sb_rad = s.sbdeg * 0.01745329252f;", "  sb_rad = s.sbdeg * 0.01745329252f;")
    assert "Tb[" not in fcs and "d." not in fcs.replace("0.d", ""), "FCS must not touch Derived"
    # ---- systems: propulsion
    arm = eng[eng.index("  // thrust along body x through the structural origin"):]
    eng = eng[:eng.index("  // thrust along body x through the structural origin")]
    eng = rep(eng, "  engine_factors(T, mach, d.h_sl_ft, idle_f, mil_f, aug_f);", "  engine_factors(T, mach, h_sl_ft, idle_f, mil_f, aug_f);")
    eng = rep(eng, "  float thrust;\n", "")
    for old, new in (("phase == PH_TRIM && !DT_ZERO", "phase == PH_TRIM"), ("    if (DT_ZERO) phase = PH_TRIM;\n", ""), ("!starved && !DT_ZERO", "!starved")):
        eng = rep(eng, old, new)
    eng = eng.replace("dt)", "(1.0f / 60.0f))").replace("* dt /", "* (1.0f / 60.0f) /")
    assert "DT_ZERO" not in eng and " dt" not in eng, eng
    arm = rep(arm, "  float Mpy = tz * thrust, Mpz = -ty * thrust;", "  const float Mpy = tz * thrust, Mpz = -ty * thrust;")
    # ---- dynamics, part 3 / 4: aerodynamics cut between the table look-ups and the coefficient assembly
    i_l0 = aero.index("    const float Sw = (float)F16_WINGAREA")
    i_as = aero.index("    float p = d.p, q = d.q, r = d.r;")
    look, asm = aero[i_l0:i_as], aero[i_as:]
    asm = asm[:asm.rindex("  }")]
    for old, new in (("    const float Sw = (float)F16_WINGAREA, bw = (float)F16_WINGSPAN, cbar = (float)F16_CHORD;\n", ""),
                     ("    float i2v = ", "    const float i2v = "), ("    float bi2vel = bw * i2v, ci2vel = cbar * i2v;", "    bi2vel = bw * i2v; ci2vel = cbar * i2v;"),
                     ("    float qS = qbar * Sw;", "    qS = qbar * Sw;"),
                     ("    float4 g0 = lerp4(", "    g0 = lerp4("), ("    float4 g1 = lerp4(", "    g1 = lerp4("), ("    float4 g2 = lerp4(", "    g2 = lerp4("),
                     ("    float4 g3 = make_float4(", "    g3 = make_float4("), ("    float4 ge = lerp4(", "    ge = lerp4("),
                     ("    float clb = lerpf(", "    clb = lerpf("), ("    float cnb = lerpf(", "    cnb = lerpf("), ("    float4 g7 = lerp4(", "    g7 = lerp4("),
                     ("    float4 m0 = lerp4(", "    m0 = lerp4("), ("    float4 m1 = lerp4(", "    m1 = lerp4("), ("    float cndr_m = lerpf(", "    cndr_m = lerpf("),
                     ("    float kge = 1.0f;", "    kge = 1.0f;")):
        look = rep(look, old, new)
    head = '''// GENERATED by tools/gen_split_tick.py from tick() of f16_device.hpp — do not edit; edit tick() and re-run the script.
// The statements of one executive tick, cut at the section comments into the pieces the three-wave step kernels (split_kernel.hpp) run
// on different waves of a workgroup: the dynamics wave (rates / velocity | auxiliary | table look-ups | assembly + accelerations), the
// systems wave (FCS | propulsion | mass balance) and the kinematics wave (attitude, position, geodetic frame, gravity and atmosphere of
// the coming tick), which exchange their results per aircraft through LDS.
#pragma once

namespace f16 {
struct Surf { float aileron_rad, flaperon_rad, elevator_rad, rudder_rad, lef_rad, sb_rad, throttle_pos; };
struct DynVars {
  float gx, gy, gz;
  Atmos A;
  float mass, cgx, cgy, cgz, Jxx, Jyy, Jzz, Jxy, Jxz, Jyz;
  float vt, alpha, beta, ca, sa, cb, sb, qbar, mach, vg, qc, npx, npy, npz;
  float c00, c01, c02, c11, c12, c22, idet, im_;
  float bi2vel, ci2vel, qS, clb, cnb, cndr_m, kge;
  float4 g0, g1, g2, g3, ge, g7, m0, m1;
};
#define F16_DYN_REFS(k)                                                                                                          \\
  float &gx = k.gx, &gy = k.gy, &gz = k.gz; Atmos& A = k.A;                                                                      \\
  float &mass = k.mass, &cgx = k.cgx, &cgy = k.cgy, &cgz = k.cgz, &Jxx = k.Jxx, &Jyy = k.Jyy, &Jzz = k.Jzz, &Jxy = k.Jxy, &Jxz = k.Jxz, &Jyz = k.Jyz; \\
  float &c00 = k.c00, &c01 = k.c01, &c02 = k.c02, &c11 = k.c11, &c12 = k.c12, &c22 = k.c22, &idet = k.idet, &im_ = k.im_; \\
  float &vt = k.vt, &alpha = k.alpha, &beta = k.beta, &ca = k.ca, &sa = k.sa, &cb = k.cb, &sb = k.sb, &qbar = k.qbar, &mach = k.mach, &vg = k.vg,    \\
        &qc = k.qc, &npx = k.npx, &npy = k.npy, &npz = k.npz;                                                                    \\
  float &bi2vel = k.bi2vel, &ci2vel = k.ci2vel, &qS = k.qS, &clb = k.clb, &cnb = k.cnb, &cndr_m = k.cndr_m, &kge = k.kge; \\
  float4 &g0 = k.g0, &g1 = k.g1, &g2 = k.g2, &g3 = k.g3, &ge = k.ge, &g7 = k.g7, &m0 = k.m0, &m1 = k.m1;                     \\
  (void)gx; (void)gy; (void)gz; (void)A; (void)mass; (void)cgx; (void)cgy; (void)cgz; (void)Jxx; (void)Jyy; (void)Jzz; (void)Jxy; (void)Jxz; (void)Jyz; \\
  (void)c00; (void)c01; (void)c02; (void)c11; (void)c12; (void)c22; (void)idet; (void)im_; \\
  (void)vt; (void)alpha; (void)beta; (void)ca; (void)sa; (void)cb; (void)sb; (void)qbar; (void)mach; (void)vg; (void)qc; (void)npx; (void)npy; (void)npz; \\
  (void)bi2vel; (void)ci2vel; (void)qS; (void)clb; (void)cnb; (void)cndr_m; (void)kge; (void)g0; (void)g1; (void)g2; (void)g3; (void)ge; (void)g7; (void)m0; (void)m1

'''
    out = head
    out += "// dynamics wave, part 1: this tick's kinematics and gravity\n__device__ __forceinline__ void dyn_p1(State& s, Derived& d, DynVars& k) {\n  F16_DYN_REFS(k);\n  constexpr float dt = 1.0f / 60.0f;\n" + prop + grav + "}\n\n"
    out += ("// kinematics wave: position of the coming tick (from the velocities up to this tick), its geodetic frame and gravity ...\n"
            "struct KinOut { float T[9]; float h_sl_ft, n_eci[3], e_eci[2], d_eci[3], gx, gy, gz, rxf, ryf; Atmos A; };\n"
            "__device__ __forceinline__ void kin_position(State& s, KinOut& o) {\n  constexpr float dt = 1.0f / 60.0f;\n  Derived d;\n  float gx, gy, gz;\n"
            + kin + "  locate_fast(s, d);\n" + grav +
            "  o.h_sl_ft = d.h_sl_ft; o.gx = gx; o.gy = gy; o.gz = gz; o.rxf = (float)s.rx; o.ryf = (float)s.ry;\n"
            "#pragma unroll\n  for (int i = 0; i < 3; ++i) { o.n_eci[i] = d.n_eci[i]; o.d_eci[i] = d.d_eci[i]; }\n  o.e_eci[0] = d.e_eci[0]; o.e_eci[1] = d.e_eci[1];\n}\n"
            "// ... and its attitude (from this tick's rates), direction cosine matrix, and the atmosphere at the new altitude\n"
            "__device__ __forceinline__ void kin_attitude(State& s, KinOut& o) {\n  constexpr float dt = 1.0f / 60.0f;\n  float* T = o.T;\n"
            + kinq + bf_T + "  o.A = atmosphere(o.h_sl_ft);\n}\n\n")
    out += ("// dynamics wave, part 1 when the kinematics wave has prepared the tick: rates and velocity, then the body-frame quantities\n"
            "__device__ __forceinline__ void dyn_p1_lite(State& s, Derived& d, DynVars& k, const KinOut& o) {\n  constexpr float dt = 1.0f / 60.0f;\n"
            + lite +
            "  d.h_sl_ft = o.h_sl_ft; k.gx = o.gx; k.gy = o.gy; k.gz = o.gz; k.A = o.A;\n  const float rxf = o.rxf, ryf = o.ryf;\n  float* T = d.T;\n"
            "#pragma unroll\n  for (int i = 0; i < 9; ++i) T[i] = o.T[i];\n"
            "#pragma unroll\n  for (int i = 0; i < 3; ++i) { d.n_eci[i] = o.n_eci[i]; d.d_eci[i] = o.d_eci[i]; }\n  d.e_eci[0] = o.e_eci[0]; d.e_eci[1] = o.e_eci[1]; d.e_eci[2] = 0.0f;\n"
            + bf_rest + "}\n\n")
    out += "// dynamics wave, part 2: atmosphere and auxiliary (the mass properties in k come from the systems wave)\ntemplate <bool HAVE_ATMOSPHERE>\n__device__ __forceinline__ void dyn_p2(State& s, Derived& d, DynVars& k) {\n  F16_DYN_REFS(k);\n  if (!HAVE_ATMOSPHERE) A = atmosphere(d.h_sl_ft);\n" + aux + "}\n\n"
    out += ("// dynamics wave, part 3: every table look-up of the tick (the elevator and speedbrake deflections come from the systems wave)\n"
            "__device__ __forceinline__ void dyn_p3(const Derived& d, const Tab& T, DynVars& k, const Surf& sf) {\n  F16_DYN_REFS(k);\n"
            "  const float elevator_rad = sf.elevator_rad, sb_rad = sf.sb_rad;\n"
            "  const float Sw = (float)F16_WINGAREA, bw = (float)F16_WINGSPAN, cbar = (float)F16_CHORD;\n  {\n" + look + "  }\n}\n\n")
    out += ("// dynamics wave, part 4: coefficient assembly, forces, accelerations, publication\n"
            "__device__ __forceinline__ void dyn_p4(State& s, Derived& d, DynVars& k, const Surf& sf, float thrust) {\n  F16_DYN_REFS(k);\n"
            "  const float aileron_rad = sf.aileron_rad, flaperon_rad = sf.flaperon_rad, rudder_rad = sf.rudder_rad, lef_rad = sf.lef_rad, sb_rad = sf.sb_rad;\n"
            "  const float bw = (float)F16_WINGSPAN, cbar = (float)F16_CHORD;\n  const float* Tb = d.T;\n  (void)Tb;\n"
            + arm + "  float Fx, Fy, Fz, Mx, My, Mz;\n  {\n" + asm + "  }\n" + acc + pub + "}\n\n")
    out += ("// systems wave, part 1: the flight control system (inputs: last tick's auxiliary values in s, this tick's cos(theta)cos(phi) and body v)\n"
            "__device__ __forceinline__ void sys_fcs(State& s, float cthcph, float vbody, Surf& sf) {\n"
            "  float &aileron_rad = sf.aileron_rad, &flaperon_rad = sf.flaperon_rad, &elevator_rad = sf.elevator_rad, &rudder_rad = sf.rudder_rad,\n"
            "        &lef_rad = sf.lef_rad, &sb_rad = sf.sb_rad, &throttle_pos = sf.throttle_pos;\n" + fcs + "}\n\n")
    out += ("// systems wave: mass balance of the coming tick from the tanks the turbine just drew from, and the inverse inertia\n"
            "__device__ __forceinline__ void sys_mass(const State& s, DynVars& k) {\n  F16_DYN_REFS(k);\n" + mass + "  {\n" + cof + "  }\n}\n\n")
    out += ("// systems wave, part 2: turbine and fuel (inputs: this tick's Mach, dynamic pressure, atmosphere and altitude)\n"
            "__device__ __forceinline__ void sys_engine(State& s, const Tab& T, float mach, float qbar, const Atmos& A, float h_sl_ft, float throttle_pos, float& thrust) {\n"
            "  constexpr bool DT_ZERO = false;\n  (void)DT_ZERO;\n" + eng + "}\n\n}  // namespace f16\n")
    return out


if __name__ == "__main__":
    text = generate()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(OUT, len(text.splitlines()), "lines")
