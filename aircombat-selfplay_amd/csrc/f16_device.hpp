// MI355X (gfx950) device-side F-16 flight dynamics: one wavefront lane per aircraft.
//
// What this replaces: the per-aircraft `jsbsim.FGFDMExec.run()` call the reference makes from
// envs/JSBSim/core/simulatior.py:210-229 (AircraftSimulator.run) six times per env step, i.e. one JSBSim
// executive tick (data/src/FGFDMExec.cpp:407-431) of the F-16 model data/aircraft/f16/f16.xml.
//
// Formulation (deliberately not JSBSim's object graph):
//  * fp32 everywhere except the three ECI position words, the Earth-angle rotation and the ECEF->geodetic
//    reduction, which stay fp64 (|r| ~ 2e7 ft: fp32 would quantise position to ~2 ft).
//  * no Tl2b / Tec2b / Euler angles per tick: the FCS only needs cos(theta)cos(phi) = <body z, local down>,
//    ground speed comes from the ECI-relative velocity projected on the local north/east unit vectors, and
//    J2 gravity is evaluated directly in ECI (the formula is invariant under rotation about the polar axis).
//  * all 25 one-dimensional alpha tables and the 12-row two-dimensional tables share one alpha breakpoint
//    search per tick; tables live in LDS (staged once per workgroup) and are gathered with ds_read_b32.
//  * integrators are the reference's own: quaternion and inertial body rates rectangular Euler, inertial
//    velocity Adams-Bashforth-2, inertial position Adams-Bashforth-3 (data/src/models/FGPropagate.cpp:93-96).
#pragma once
#include <hip/hip_runtime.h>
#include "f16_tables.h"
#include "fp64_fast.hpp"

namespace f16 {

// ---- unit constants (published values; JSBSim keeps them in FGJSBBase.h)
constexpr float kFt2M = 0.3048f;
constexpr float kInch2Ft = 1.0f / 12.0f;
constexpr float kLb2Slug = 1.0f / 32.174049f;
constexpr float kKts2Fps = 1.68781f;
constexpr float kG0 = 32.174049f;  // 9.80665 / 0.3048
constexpr float kPi = 3.14159265358979f;
// Earth model, data/src/models/FGInertial.cpp:56-60
constexpr double kOmega = 0.00007292115;
constexpr double kGM = 14.0764417572E15;
constexpr double kJ2 = 1.08262982E-03;
constexpr double kA = 20925646.32546;
constexpr double kB = 20855486.5951;
// FCS components keep the executive's load-time dt of 1/120 s (FGFCSComponent.cpp:58, FGFDMExec.cpp:96,
// simulatior.py:165-169 calls set_dt only after load_model): PID derivative and actuator rate limits use it.
constexpr float kFcsDt = 1.0f / 120.0f;

// ---- engine state bits
enum : int { ENG_PHASE_MASK = 7, ENG_RUNNING = 8, ENG_CUTOFF = 16, ENG_STARVED = 32, ENG_AUG = 64 };
enum : int { PH_OFF = 0, PH_RUN = 1, PH_START = 3, PH_TRIM = 6 };

// Complete dynamic state of one aircraft: everything the next tick depends on.
struct State {
  double rx, ry, rz;                       // ECI position [ft]
  float vx, vy, vz;                        // ECI velocity [ft/s]
  float q0, q1, q2, q3;                    // ECI -> body quaternion
  float wp, wq, wr;                        // inertial body rates PQRi [rad/s]
  float hv1x, hv1y, hv1z, hv2x, hv2y, hv2z;  // inertial velocity one and two ticks ago (AB3 history)
  float ha1x, ha1y, ha1z;                  // inertial acceleration one tick ago (AB2 history)
  float wdx, wdy, wdz;                     // PQRi-dot of the last tick
  float aix, aiy, aiz;                     // inertial acceleration of the last tick
  float bax, bay, baz;                     // body specific force / mass of the last tick
  float da, de, dr, thr;                   // commands (clipped)
  float pin_r, pin_p, pin_y;               // PID previous inputs
  float pi_r, pi_p, pi_y;                  // PID integrator totals
  float tef, ail, elev, sbdeg;             // kinematic (rate-limited) outputs
  float alpha, mach, qc, vg;               // FGAuxiliary outputs of the last tick (the FCS runs before it); qc = pitot impact pressure [psf]
  float ap, aq, ar;                        // aero body rates of the last tick
  float npx, npy, npz;                     // pilot-station load factors of the last tick
  float n1, n2, n2norm, ff;                // turbine
  float tank0, tank1;                      // internal tanks [lb]
  int eng;                                 // phase | flags
  int ticks;                               // executive ticks since reset (sim time = ticks/60)
};

// What one tick exposes to the environment layer.
struct Derived {
  float h_sl_ft;                           // radius - sea-level radius [ft]
  float u, v, w;                           // body velocity [ft/s]
  float p, q, r;                           // body rates relative to ECEF [rad/s]
  float vn, ve, vd;                        // local NED velocity [ft/s]
  float n_eci[3], e_eci[3], d_eci[3];      // local north / east / down unit vectors in ECI
  float T[9];                              // ECI -> body
  float veci;                              // |v_eci|
  double X, Y, Z;                          // ECEF position [ft]
  double sLat64, cLat64, sLon64, cLon64;   // fp64 copies for the geodetic -> NED reduction of the env layer
};

// LDS copy of F16_PACK (tools/gen_f16_tables.py): tables that share breakpoint axes are interleaved so one lane fetches
// all values it needs at a breakpoint with a single 16-byte ds_read_b128.
struct Tab {
  const float* t;  // 16-byte aligned
  __device__ __forceinline__ float operator[](int i) const { return t[i]; }
  __device__ __forceinline__ float4 v4(int i) const { return *reinterpret_cast<const float4*>(t + i); }
  __device__ __forceinline__ float2 v2(int i) const { return *reinterpret_cast<const float2*>(t + i); }
};

__device__ __forceinline__ float clampf(float lo, float v, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float f) {
  return make_float4(lerpf(a.x, b.x, f), lerpf(a.y, b.y, f), lerpf(a.z, b.z, f), lerpf(a.w, b.w, f));
}

// Segment index of `key` on a breakpoint axis known at compile time: r in [1, N-1] with x[r-1] < key <= x[r] (clamped at
// both ends). Pure compare/add on immediates — no memory access, so the dependent LDS reads of a tick can all be issued in
// one batch. Equivalent to FGTable::GetValue's walk (data/src/math/FGTable.cpp:443-516): the interpolant is continuous at
// breakpoints, so which of two adjacent segments an exact breakpoint falls in does not change the value.
template <int N>
__device__ __forceinline__ int seg_index(const float (&x)[N], float key) {
  int r = 1;
#pragma unroll
  for (int i = 1; i < N - 1; ++i) r += (x[i] < key) ? 1 : 0;
  return r;
}
// clamped interpolation fraction on the segment whose end points were read from LDS
__device__ __forceinline__ float seg_frac(float x0, float x1, float key) { return clampf(0.0f, (key - x0) / (x1 - x0), 1.0f); }
// tiny schedule tables held as immediates (FCS gains): compare/select chain
template <int N>
__device__ __forceinline__ float tabc(const float (&x)[N], const float (&y)[N], float key) {
  float x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
#pragma unroll
  for (int i = 1; i < N - 1; ++i) {
    bool b = x[i] < key;
    x0 = b ? x[i] : x0; x1 = b ? x[i + 1] : x1; y0 = b ? y[i] : y0; y1 = b ? y[i + 1] : y1;
  }
  return lerpf(y0, y1, seg_frac(x0, x1, key));
}

// ---------------------------------------------------------------- standard atmosphere 1976 (geopotential layers)
// data/src/models/atmosphere/FGStandardAtmosphere.cpp:66-74,152-222,244-268 — only the three layers an F-16 can reach.
// x^y for x > 0 on the two transcendental units (v_log_f32, v_exp_f32; ~1 ulp each): the library pow spends >100 instructions
// on special cases none of these call sites can hit. Relative error <= 2e-7 for the exponents used here.
__device__ __forceinline__ float pow_pos(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
// atan2 for finite arguments, not both zero: odd minimax polynomial of degree 17 on [0, 1] (max error 8e-8 rad in fp32)
// after the usual min/max reduction.
__device__ __forceinline__ float atan2_fast(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = mn * __builtin_amdgcn_rcpf(mx);
  float t = a * a;
  float p = 0.0024489392526447773f;
  p = fmaf(p, t, -0.014364523813128471f); p = fmaf(p, t, 0.0397086925804615f); p = fmaf(p, t, -0.07227175682783127f);
  p = fmaf(p, t, 0.10494229197502136f); p = fmaf(p, t, -0.14159545302391052f); p = fmaf(p, t, 0.1998557597398758f);
  p = fmaf(p, t, -0.33332565426826477f); p = fmaf(p, t, 0.9999998807907104f);
  float r = p * a;
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < 0.0f) ? 3.14159265358979324f - r : r;
  return copysignf(r, y);
}

struct Atmos { float T, P, rho, a; };
__device__ __forceinline__ Atmos atmosphere(float h_ft) {
  const float Re = 20855531.5f;                 // 6356766 m in ft
  const float R = 1716.557158f;                 // Rstar/Mair = 8.31432*kgtoslug/(1.8*0.3048^2) / (28.9645*kgtoslug/1000)
  const float g0R = kG0 / R;
  float gp = h_ft * Re / (Re + h_ft);
  // the three layers as selects over one pow-shaped and one exp-shaped evaluation (no divergent branches in the tick)
  const bool tropo = gp < 36089.2388f, strato1 = gp < 65616.7979f;
  const float L0 = (389.97f - 518.67f) / 36089.2388f;                               // also JSBSim's extrapolation below sea level
  const float L2 = (411.57f - 389.97f) / (104986.8766f - 65616.7979f);
  const float Pb1 = 472.680579f, Pb2 = 114.344890f;                                   // layer-base pressures (:460-481)
  Atmos A;
  A.T = tropo ? 518.67f + L0 * gp : (strato1 ? 389.97f : 389.97f + L2 * (gp - 65616.7979f));
  const float base = tropo ? 518.67f : 389.97f;
  // P = Pb * (Tb / T)^(g0R / L)  <=>  exp2(-(g0R / L) * log2(T / Tb));  isothermal layer: exp(-g0R (gp - hb) / Tb)
  const float ex_pow = -(tropo ? g0R / L0 : g0R / L2) * __builtin_amdgcn_logf(A.T * (1.0f / base));
  const float ex_iso = (-g0R * 1.44269504088896341f / 389.97f) * (gp - 36089.2388f);
  const bool iso = !tropo && strato1;
  A.P = (tropo ? 2116.228f : (strato1 ? Pb1 : Pb2)) * __builtin_amdgcn_exp2f(iso ? ex_iso : ex_pow);
  A.rho = A.P / (R * A.T);
  A.a = sqrtf(1.4f * R * A.T);
  return A;
}

// Calibrated airspeed (data/src/FGJSBBase.cpp:245-296) = sea-level speed whose pitot impact pressure equals the measured
// one. The impact pressure qc = pt - p is cheap (one pow) and is what the per-tick code carries: the FCS only compares the
// calibrated airspeed against fixed thresholds, and vc is monotonic in qc, so "vc-kts lt V" <=> "qc < qc_sl(V)". The inverse
// (with its 10-step supersonic fixed point) runs once per env step when the observation needs vc itself.
__device__ __forceinline__ float pitot_impact_pressure(float mach, float p) {
  // subsonic (1 + 0.2 M^2)^3.5 and Rayleigh 166.92158 M^7 / (7 M^2 - 1)^2.5 with half-integer powers as products and one sqrt
  const float m2 = mach * mach;
  const float x = 1.0f + 0.2f * m2;
  const float sub = x * x * x * sqrtf(x);
  const float y = fmaxf(7.0f * m2 - 1.0f, 1.0f);
  const float sup = 166.92158009316827f * (m2 * m2 * m2 * mach) / (y * y * sqrtf(y));
  return p * ((mach < 1.0f) ? sub : sup) - p;
}
__device__ __forceinline__ float vcas_from_impact_pressure(float qc) {
  const float psl = 2116.228f, asl = 1116.448558f;  // sqrt(1.4 * R * 518.67)
  float A = qc / psl + 1.0f;
  float M = sqrtf(fmaxf(0.0f, 5.0f * (pow_pos(A, 1.0f / 3.5f) - 1.0f)));
  if (M > 1.0f) {
    for (int i = 0; i < 10; ++i) M = 0.8812848543473311f * sqrtf(A * pow_pos(1.0f - 1.0f / (7.0f * M * M), 2.5f));
  }
  return asl * M;
}
// sea-level impact pressures of the FCS airspeed thresholds: psl * ((1 + 0.2 (V kts / asl)^2)^3.5 - 1)
constexpr float kQc250 = 219.261788f, kQc20 = 1.35453246f, kQc10 = 0.338575078f, kQc5 = 0.0846401425f;

// Rate limiter through two detents at [-lim.., lim] (FGKinemat.cpp:99-170 specialised to a single segment).
__device__ __forceinline__ float slew(float out, float in, float lo, float hi, float rate) {
  in = clampf(lo, in, hi);
  float step = rate * kFcsDt;
  float d = in - out;
  return (fabsf(d) <= step) ? in : out + copysignf(step, d);
}
// The trailing-edge-flap kinematic has detents {-1, 0, 1} with times {3, 0, 3}: the (-1,0] segment has zero
// transit time (the output jumps to the input whenever the segment it starts in is that one), the (0,1] segment
// moves at 1/3 per second. Follows the detent search of FGKinemat.cpp:116-140 inside one FCS tick.
__device__ __forceinline__ float tef_kinematic(float out, float in) {
  in = clampf(-1.0f, in, 1.0f);
  float dt0 = kFcsDt;
  bool walking = true;
  // at most three segment hops per tick; written as straight-line selects (no divergent loop in the middle of the tick)
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const bool go = walking && dt0 > 0.0f && in != out;
    // segment index: 1 => [-1,0] (zero traverse time: jump), 2 => [0,1]
    const int ind = (in < out) ? ((0.0f < out) ? 2 : 1) : ((0.0f <= out) ? 2 : 1);
    const bool jump = ind == 1;
    const float rate = 1.0f / 3.0f;
    const float thisIn = clampf(0.0f, in, 1.0f);
    const float thisDt = fabsf((thisIn - out) / rate);
    const bool partial = dt0 < thisDt;
    const float moved = out + ((out < in) ? dt0 * rate : -dt0 * rate);
    const float out_n = jump ? in : (partial ? moved : thisIn);
    const float dt_n = jump ? dt0 : (partial ? 0.0f : dt0 - thisDt);
    out = go ? out_n : out;
    dt0 = go ? dt_n : dt0;
    walking = go && !jump;
  }
  return out;
}
// FGPID::Run (FGPID.cpp:154-204) with the Adams-Bashforth-2 integrator the f16 <ki> elements default to.
__device__ __forceinline__ float pid(float in, float& in_prev, float& itot, bool trig_zero, float kp, float ki, float kd) {
  float dval = (in - in_prev) * (1.0f / kFcsDt);
  if (trig_zero) itot += ki * kFcsDt * (1.5f * in - 0.5f * in_prev);
  in_prev = in;
  return kp * in + itot + kd * dval;
}
__device__ __forceinline__ float seek(float v, float target, float accel, float decel, float dt) {
  if (v > target) v = fmaxf(v - dt * decel, target);
  else if (v < target) v = fminf(v + dt * accel, target);
  return v;
}

// Where the vehicle is: altitude above the sea-level radius and the local north / east / down unit vectors in ECI.
// The local frame only needs the DIRECTION of the geodetic normal, so the per-tick form works in the inertial
// meridian plane (inertial longitude = atan2(ry, rx); rotating to ECEF and back by the Earth angle cancels) with
// Fukushima's one-step reduction (data/src/math/FGLocation.cpp:283-317) on coordinates normalised by the semi-major
// axis in fp32. Only |r| needs fp64 (2e7 ft against a sub-foot altitude budget); the sea-level radius
// a*ec/sqrt(1 - e2*cos^2(lat_gc)) (FGLocation.cpp:243-249) is expanded about b so that its fp32 part is < 0.01 ft.
__device__ __forceinline__ void locate_fast(const State& s, Derived& d) {
  double rr = s.rx * s.rx + s.ry * s.ry + s.rz * s.rz;
  double rad = sqrt(rr);
  float ia = (float)(1.0 / kA);
  float px = (float)s.rx * ia, py = (float)s.ry * ia, z = (float)s.rz * ia;   // O(1) coordinates
  float p2 = px * px + py * py;
  float irp = rsqrtf(p2);
  float p = p2 * irp;
  float cosLon = px * irp, sinLon = py * irp;
  const float ec = (float)(kB / kA), e2 = (float)(1.0 - (kB / kA) * (kB / kA));
  float s0 = fabsf(z), zc = ec * s0, c0 = ec * p, c02 = c0 * c0, s02 = s0 * s0, a02 = c02 + s02;
  float a0 = sqrtf(a02), a03 = a02 * a0;
  float s1 = zc * a03 + e2 * s02 * s0, c1 = p * a03 - e2 * c02 * c0, cs = e2 * c0 * s0;
  float b0 = 1.5f * cs * ((p * s0 - zc * c0) * a0 - cs);
  s1 = s1 * a03 - b0 * s0;
  float cc = ec * (c1 * a03 - b0 * c0);
  float inv = rsqrtf(s1 * s1 + cc * cc);
  float sinLat = copysignf(s1, z) * inv, cosLat = cc * inv;
  // sea-level radius: b * (1 - x)^(-1/2), x = e2 * cos^2(geocentric latitude) <= 0.0067
  float cg2 = p2 / (p2 + z * z);
  float x = e2 * cg2;
  float ser = x * (0.5f + x * (0.375f + x * (0.3125f + x * 0.2734375f)));
  d.h_sl_ft = (float)(rad - kB) - (float)kB * ser;
  d.n_eci[0] = -cosLon * sinLat; d.n_eci[1] = -sinLon * sinLat; d.n_eci[2] = cosLat;
  d.e_eci[0] = -sinLon; d.e_eci[1] = cosLon; d.e_eci[2] = 0.0f;
  d.d_eci[0] = -cosLon * cosLat; d.d_eci[1] = -sinLon * cosLat; d.d_eci[2] = -sinLat;
}
// fp64 form for the environment layer (geodetic -> NED of the aircraft position, once per env step): ECEF position,
// geodetic direction cosines and longitude cosines to full precision.
__device__ __forceinline__ void locate(const State& s, Derived& d) {
  // Earth position angle is tiny (< 0.1 rad over an episode): series in fp64 is exact to < 1e-15.
  double epa = kOmega * (double)s.ticks * (1.0 / 60.0);
  double e2 = epa * epa;
  double ce = 1.0 - e2 * (0.5 - e2 * (1.0 / 24.0 - e2 * (1.0 / 720.0)));
  double se = epa * (1.0 - e2 * (1.0 / 6.0 - e2 * (1.0 / 120.0 - e2 * (1.0 / 5040.0))));
  double X = ce * s.rx + se * s.ry, Y = -se * s.rx + ce * s.ry, Z = s.rz;
  d.X = X; d.Y = Y; d.Z = Z;
  double rxy2 = X * X + Y * Y;
  double rxy = fx::sqrt(rxy2), rad = fx::sqrt(rxy2 + Z * Z);
  const double ec = kB / kA, ec2 = ec * ec, ee = 1.0 - ec2, c = kA * ee;
  double s0 = fabs(Z), zc = ec * s0, c0 = ec * rxy, c02 = c0 * c0, s02 = s0 * s0, a02 = c02 + s02;
  double a0 = fx::sqrt(a02), a03 = a02 * a0;
  double s1 = zc * a03 + c * s02 * s0, c1 = rxy * a03 - c * c02 * c0, cs = c * c0 * s0;
  double b0 = 1.5 * cs * ((rxy * s0 - zc * c0) * a0 - cs);
  s1 = s1 * a03 - b0 * s0;
  double cc = ec * (c1 * a03 - b0 * c0);
  double inv = fx::rsqrt(s1 * s1 + cc * cc);
  double sinLat = (Z >= 0.0 ? s1 : -s1) * inv, cosLat = cc * inv;
  double cgc = rxy * fx::rcp(rad);
  double slr = kA * ec * fx::rsqrt(1.0 - ee * cgc * cgc);
  d.h_sl_ft = (float)(rad - slr);
  double irxy = fx::rcp(rxy);
  double cosLon = X * irxy, sinLon = Y * irxy;
  d.sLat64 = sinLat; d.cLat64 = cosLat; d.sLon64 = sinLon; d.cLon64 = cosLon;
  // local unit vectors in ECEF, rotated back by the Earth angle into ECI
  float cef = (float)ce, sef = (float)se;
  float sLa = (float)sinLat, cLa = (float)cosLat, sLo = (float)sinLon, cLo = (float)cosLon;
  float nx = -cLo * sLa, ny = -sLo * sLa, nz = cLa;
  float ex = -sLo, ey = cLo;
  float dx = -cLo * cLa, dy = -sLo * cLa, dz = -sLa;
  d.n_eci[0] = cef * nx - sef * ny; d.n_eci[1] = sef * nx + cef * ny; d.n_eci[2] = nz;
  d.e_eci[0] = cef * ex - sef * ey; d.e_eci[1] = sef * ex + cef * ey; d.e_eci[2] = 0.0f;
  d.d_eci[0] = cef * dx - sef * dy; d.d_eci[1] = sef * dx + cef * dy; d.d_eci[2] = dz;
}

__device__ __forceinline__ void body_frame(const State& s, Derived& d) {
  float q0 = s.q0, q1 = s.q1, q2 = s.q2, q3 = s.q3;
  float* T = d.T;
  T[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; T[1] = 2.0f * (q1 * q2 + q0 * q3); T[2] = 2.0f * (q1 * q3 - q0 * q2);
  T[3] = 2.0f * (q1 * q2 - q0 * q3); T[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; T[5] = 2.0f * (q2 * q3 + q0 * q1);
  T[6] = 2.0f * (q1 * q3 + q0 * q2); T[7] = 2.0f * (q2 * q3 - q0 * q1); T[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
  const float om = (float)kOmega;
  // velocity relative to the rotating Earth, in ECI axes: v - Omega x r
  float rvx = s.vx + om * (float)s.ry, rvy = s.vy - om * (float)s.rx, rvz = s.vz;
  d.u = T[0] * rvx + T[1] * rvy + T[2] * rvz;
  d.v = T[3] * rvx + T[4] * rvy + T[5] * rvz;
  d.w = T[6] * rvx + T[7] * rvy + T[8] * rvz;
  d.p = s.wp - om * T[2]; d.q = s.wq - om * T[5]; d.r = s.wr - om * T[8];
  d.vn = d.n_eci[0] * rvx + d.n_eci[1] * rvy + d.n_eci[2] * rvz;
  d.ve = d.e_eci[0] * rvx + d.e_eci[1] * rvy;
  d.vd = d.d_eci[0] * rvx + d.d_eci[1] * rvy + d.d_eci[2] * rvz;
  d.veci = sqrtf(s.vx * s.vx + s.vy * s.vy + s.vz * s.vz);
}

// F100-PW-229 thrust-factor tables (engine/F100-PW-229.xml:27-83): Mach rows are exact multiples of 0.2, altitude columns
// of 10000 ft (density altitude equals geometric altitude in the standard atmosphere the reference flies in); the
// idle / mil / aug factors are interleaved per grid point, rows past a table's last Mach repeat it (FGTable clamps).
__device__ __forceinline__ void engine_factors(const Tab& T, float mach, float h_ft, float& idle_f, float& mil_f, float& aug_f) {
  float mr = mach * 5.0f, hc = (h_ft + 10000.0f) * 1e-4f;
  int rm = max(1, min(13, (int)floorf(mr) + 1)), jc = max(1, min(7, (int)floorf(hc) + 1));
  float fm = clampf(0.0f, mr - (float)(rm - 1), 1.0f), fh = clampf(0.0f, hc - (float)(jc - 1), 1.0f);
  int base = P_ENG_OFF + ((rm - 1) * 8 + (jc - 1)) * 4;
  float4 e00 = T.v4(base), e01 = T.v4(base + 4), e10 = T.v4(base + 32), e11 = T.v4(base + 36);
  float4 e = lerp4(lerp4(e00, e10, fm), lerp4(e01, e11, fm), fh);
  idle_f = e.x; mil_f = e.y; aug_f = e.z;
}

// The integration step of FGPropagate (FGPropagate.cpp:218-290, :336-360): attitude and inertial rates by rectangular Euler,
// inertial position by Adams-Bashforth 3, inertial velocity by Adams-Bashforth 2 -- all explicit in the derivatives the LAST
// tick left in the state, so the pose after this tick is known before any of this tick's forces are (the cooperative kernel forms
// hand it to the weapons wave at this point).
__device__ __forceinline__ void propagate(State& s) {
  constexpr float dt = 1.0f / 60.0f;
  float qd0 = -0.5f * (s.q1 * s.wp + s.q2 * s.wq + s.q3 * s.wr);
  float qd1 = 0.5f * (s.q0 * s.wp - s.q3 * s.wq + s.q2 * s.wr);
  float qd2 = 0.5f * (s.q3 * s.wp + s.q0 * s.wq - s.q1 * s.wr);
  float qd3 = 0.5f * (-s.q2 * s.wp + s.q1 * s.wq + s.q0 * s.wr);
  float a0 = fmaf(dt, qd0, s.q0), a1 = fmaf(dt, qd1, s.q1), a2 = fmaf(dt, qd2, s.q2), a3 = fmaf(dt, qd3, s.q3);
  float rn = rsqrtf(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
  s.q0 = a0 * rn; s.q1 = a1 * rn; s.q2 = a2 * rn; s.q3 = a3 * rn;
  s.wp = fmaf(dt, s.wdx, s.wp); s.wq = fmaf(dt, s.wdy, s.wq); s.wr = fmaf(dt, s.wdz, s.wr);
  const float k = dt / 12.0f;
  s.rx += (double)(k * (23.0f * s.vx - 16.0f * s.hv1x + 5.0f * s.hv2x));
  s.ry += (double)(k * (23.0f * s.vy - 16.0f * s.hv1y + 5.0f * s.hv2y));
  s.rz += (double)(k * (23.0f * s.vz - 16.0f * s.hv1z + 5.0f * s.hv2z));
  s.hv2x = s.hv1x; s.hv2y = s.hv1y; s.hv2z = s.hv1z;
  s.hv1x = s.vx; s.hv1y = s.vy; s.hv1z = s.vz;
  s.vx += dt * (1.5f * s.aix - 0.5f * s.ha1x);
  s.vy += dt * (1.5f * s.aiy - 0.5f * s.ha1y);
  s.vz += dt * (1.5f * s.aiz - 0.5f * s.ha1z);
  s.ha1x = s.aix; s.ha1y = s.aiy; s.ha1z = s.aiz;
  s.ticks += 1;
}

// The rest of the tick, from the frames of the (new) pose to the accelerations the next tick integrates.
template <bool DT_ZERO, bool TANK_ARM_ORIGIN = false>
__device__ __forceinline__ void tick_after_propagate(State& s, Derived& d, const Tab& T) {
  constexpr float dt = DT_ZERO ? 0.0f : (1.0f / 60.0f);
  locate_fast(s, d);
  body_frame(s, d);
  const float* Tb = d.T;

  // ---------------- Inertial: WGS84 + J2 gravity, evaluated in ECI (FGInertial.cpp:193-213)
  float rxf = (float)s.rx, ryf = (float)s.ry, rzf = (float)s.rz;
  float r2 = rxf * rxf + ryf * ryf + rzf * rzf;
  float ir = rsqrtf(r2);
  float sl = rzf * ir;
  float adr = (float)kA * ir;
  float pre = 1.5f * (float)kJ2 * adr * adr;
  float gmr2 = (float)kGM * ir * ir;
  float kxy = -gmr2 * (1.0f + pre * (1.0f - 5.0f * sl * sl)) * ir;
  float kz = -gmr2 * (1.0f + pre * (3.0f - 5.0f * sl * sl)) * ir;
  float gx = kxy * rxf, gy = kxy * ryf, gz = kz * rzf;

  // ---------------- Atmosphere
  Atmos A = atmosphere(d.h_sl_ft);

  // ---------------- FCS (f16.xml:317-992; inputs from FGAuxiliary are last tick's)
  const float alpha_p = s.alpha, mach_p = s.mach, qc_p = s.qc;
  // flaps
  float tef_rad = (qc_p < kQc250) ? 0.349f : ((mach_p > 0.9f) ? -0.0349f : 0.0f);
  s.tef = tef_kinematic(s.tef, 2.864789f * tef_rad);
  // roll
  float roll_err = s.da - 0.31821f * s.ap;
  float roll_pid = pid(roll_err, s.pin_r, s.pi_r, qc_p < kQc20, 3.0f, 0.0005f, -0.00125f);
  float roll_cmd = clampf(-1.0f, roll_pid + s.da, 1.0f);
  float aileron_rad = 0.375f * roll_cmd;
  s.ail = slew(s.ail, roll_cmd, -1.0f, 1.0f, 2.0f / 0.3f);
  constexpr float kAilX[] = C_FCS_AILERON_SPEED_COMPENSATED_X, kAilY[] = C_FCS_AILERON_SPEED_COMPENSATED_Y;
  float ail_sc = s.ail * tabc(kAilX, kAilY, mach_p);
  float flaperon_rad = 1.4324f * (clampf(-1.0f, -s.tef - ail_sc, 1.0f) + clampf(-1.0f, s.tef - ail_sc, 1.0f));
  // pitch: cos(theta)cos(phi) is the projection of body z on local down
  float cthcph = Tb[6] * d.d_eci[0] + Tb[7] * d.d_eci[1] + Tb[8] * d.d_eci[2];
  float elev_lim = clampf(-1.0f, s.de, 0.44f);
  constexpr float kElX[] = C_FCS_ELEVATOR_SCHEDULER_X, kElY[] = C_FCS_ELEVATOR_SCHEDULER_Y;
  float elev_sched = elev_lim * tabc(kElX, kElY, alpha_p);
  float pitch_err = elev_sched + 6.2f * s.aq - 0.020f * (s.npz - cthcph);
  float g_pid = clampf(-1.0f, pid(pitch_err, s.pin_p, s.pi_p, qc_p < kQc5, 0.3f, 0.025f, 0.0f), 1.0f);
  float pitch_sched = clampf(-1.0f, elev_sched + 1.0472f * alpha_p + g_pid, 1.0f);
  s.elev = slew(s.elev, pitch_sched, -1.0f, 1.0f, 2.0f / 0.3f);
  float elevator_rad = 0.436f * s.elev;
  // yaw: the PID writes fcs/rudder-pos-norm, the kinematic re-reads that property as its own output
  constexpr float kYwX[] = C_FCS_YAW_RATE_NORM_X, kYwY[] = C_FCS_YAW_RATE_NORM_Y;
  float yaw_err = s.dr + s.ar * tabc(kYwX, kYwY, s.vg) + 0.25f * s.npy;
  float yaw_pid = clampf(-1.0f, pid(yaw_err, s.pin_y, s.pi_y, qc_p < kQc10, 0.1055f, 0.00001f, 0.00005f), 1.0f);
  float yaw_sched = clampf(-1.0f, s.dr + yaw_pid, 1.0f);
  float rudder_rad = 0.524f * slew(yaw_pid, yaw_sched, -1.0f, 1.0f, 2.0f / 0.4f);
  // gear stays down (FGFCS.cpp:81, never commanded): gear-pos-norm = 1, gear-wow = 0
  float lef_rad = (alpha_p > 0.0873f) ? 0.262f : ((mach_p > 0.9f) ? -0.0349f : 0.0f);
  float throttle_pos = 2.0f * s.thr;
  // speedbrake auto-deploy: alpha >= 53 deg and body v <= 18 ft/s; scheduler gain 0.71667 with the gear commanded down
  float sb_in = ((alpha_p * 57.29577951f >= 53.0f) && (d.v <= 18.0f)) ? 0.71667f * 60.0f : 0.0f;
  s.sbdeg = slew(s.sbdeg, sb_in, 0.0f, 60.0f, 60.0f);
  float sb_rad = s.sbdeg * 0.01745329252f;

  // ---------------- MassBalance (FGMassBalance.cpp:181-262); tanks 2/3 are empty external tanks
  float fuel = s.tank0 + s.tank1;
  float weight = (float)F16_EMPTYWT + (float)F16_PM0_WEIGHT + fuel;
  float mass = kLb2Slug * weight;
  float iw = 1.0f / weight;
  float cgx = ((float)(F16_EMPTYWT * F16_CG_X + F16_PM0_WEIGHT * F16_PM0_X) + (float)F16_TANK0_X * fuel) * iw;
  float cgy = ((float)F16_TANK0_Y * s.tank0 + (float)F16_TANK1_Y * s.tank1) * iw;
  float cgz = ((float)(F16_EMPTYWT * F16_CG_Z + F16_PM0_WEIGHT * F16_PM0_Z) + (float)F16_TANK0_Z * fuel) * iw;
  float Jxx = (float)F16_IXX, Jyy = (float)F16_IYY, Jzz = (float)F16_IZZ, Jxy = 0.0f, Jxz = (float)F16_IXZ, Jyz = 0.0f;
  {
    // parallel-axis terms of empty mass, pilot and the two tanks about the current CG (structural -> body arms)
    const float ms[4] = {kLb2Slug * (float)F16_EMPTYWT, kLb2Slug * (float)F16_PM0_WEIGHT, kLb2Slug * s.tank0, kLb2Slug * s.tank1};
    const float px[4] = {(float)F16_CG_X, (float)F16_PM0_X, (float)F16_TANK0_X, (float)F16_TANK1_X};
    const float py[4] = {(float)F16_CG_Y, (float)F16_PM0_Y, (float)F16_TANK0_Y, (float)F16_TANK1_Y};
    const float pz[4] = {(float)F16_CG_Z, (float)F16_PM0_Z, (float)F16_TANK0_Z, (float)F16_TANK1_Z};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool origin = TANK_ARM_ORIGIN && i >= 2;
      float x = kInch2Ft * ((origin ? 0.0f : cgx) - px[i]), y = kInch2Ft * (py[i] - (origin ? 0.0f : cgy)), z = kInch2Ft * ((origin ? 0.0f : cgz) - pz[i]);
      float m = ms[i];
      Jxx += m * (y * y + z * z); Jyy += m * (x * x + z * z); Jzz += m * (x * x + y * y);
      Jxy -= m * x * y; Jxz -= m * x * z; Jyz -= m * y * z;
    }
  }

  // ---------------- Auxiliary (FGAuxiliary.cpp:134-232)
  float muw = d.u * d.u + d.w * d.w, vt2 = muw + d.v * d.v;
  float vt = sqrtf(vt2);
  float alpha = 0.0f, beta = 0.0f, ca = 1.0f, sa = 0.0f, cb = 1.0f, sb = 0.0f;
  if (vt > 0.001f) {
    float suw = sqrtf(muw);
    beta = atan2_fast(d.v, suw);
    cb = suw / vt; sb = d.v / vt;
    if (muw >= 1e-6f) { alpha = atan2_fast(d.w, d.u); ca = d.u / suw; sa = d.w / suw; }
  }
  float qbar = 0.5f * A.rho * vt2;
  float mach = vt / A.a;
  float vg = sqrtf(d.vn * d.vn + d.ve * d.ve);
  float qc = (mach > 0.0f) ? pitot_impact_pressure(mach, A.P) : 0.0f;
  // pilot-station load factors from LAST tick's accelerations and the inertial rates (:205-217)
  float ex = kInch2Ft * (cgx - (float)F16_EYEPOINT_X), ey = kInch2Ft * ((float)F16_EYEPOINT_Y - cgy), ez = kInch2Ft * (cgz - (float)F16_EYEPOINT_Z);
  float t1x = s.wdy * ez - s.wdz * ey, t1y = s.wdz * ex - s.wdx * ez, t1z = s.wdx * ey - s.wdy * ex;
  float cx = s.wq * ez - s.wr * ey, cy = s.wr * ex - s.wp * ez, cz = s.wp * ey - s.wq * ex;
  float t2x = s.wq * cz - s.wr * cy, t2y = s.wr * cx - s.wp * cz, t2z = s.wp * cy - s.wq * cx;
  float npx = (s.bax + t1x + t2x) * (1.0f / kG0), npy = (s.bay + t1y + t2y) * (1.0f / kG0), npz = (s.baz + t1z + t2z) * (1.0f / kG0);

  // ---------------- Propulsion: F100-PW-229 turbine (FGTurbine.cpp:107-270,400-411), AugMethod 2
  float mil_f, idle_f, aug_f;
  engine_factors(T, mach, d.h_sl_ft, idle_f, mil_f, aug_f);
  float thrust;
  {
    const float MIL = (float)F16_ENG_MILTHRUST, MAXT = (float)F16_ENG_MAXTHRUST;
    const float IN1 = (float)F16_ENG_IDLEN1, IN2 = (float)F16_ENG_IDLEN2;
    const float N1f = (float)(F16_ENG_MAXN1 - F16_ENG_IDLEN1), N2f = (float)(F16_ENG_MAXN2 - F16_ENG_IDLEN2);
    const float idle_ff = 757.648518f;  // pow(milthrust, 0.2) * 107 (FGTurbine.cpp:522)
    float tp = throttle_pos, aug_cmd = 0.0f;
    if (tp > 1.0f) { aug_cmd = tp - 1.0f; tp = 1.0f; }
    int phase = s.eng & ENG_PHASE_MASK;
    bool running = s.eng & ENG_RUNNING, cutoff = s.eng & ENG_CUTOFF, starved = s.eng & ENG_STARVED, augm = s.eng & ENG_AUG;
    if (phase == PH_TRIM && !DT_ZERO) {
      if (running && !starved) { phase = PH_RUN; s.n2 = IN2 + tp * N2f; s.n1 = IN1 + tp * N1f; cutoff = false; }
      else { phase = PH_OFF; cutoff = true; }
    }
    if (qbar > 30.0f && !running && !cutoff && s.n2 > 15.0f) phase = PH_START;
    if (cutoff) phase = PH_OFF;
    if (DT_ZERO) phase = PH_TRIM;
    if (starved) phase = PH_OFF;
    float idle = MIL * idle_f, mil = (MIL - idle) * mil_f;
    thrust = 0.0f;
    if (phase == PH_RUN) {
      running = true;
      float dens_ratio = A.rho * (1.0f / 0.00237691175f);  // sea-level density 2116.228 / (R * 518.67)
      float n = fminf(1.0f, s.n2norm + 0.1f);
      float den = 1.0f / (1.0f + 3.0f * (1.0f - n) * (1.0f - n) * (1.0f - n) + (1.0f - dens_ratio));
      const float base = 90.0f / ((float)F16_ENG_BYPASSRATIO + 3.0f);
      s.n2 = seek(s.n2, IN2 + tp * N2f, base * den, 3.0f * base * den, dt);
      s.n1 = seek(s.n1, IN1 + tp * N1f, base * den, 2.4f * base * den, dt);
      s.n2norm = (s.n2 - IN2) / N2f;
      thrust = idle + mil * s.n2norm * s.n2norm;
      if (!augm) {
        float tsfc = (float)F16_ENG_TSFC * sqrtf(A.T * (1.0f / 389.7f)) * (0.84f + (1.0f - s.n2norm) * (1.0f - s.n2norm));
        s.ff = fmaxf(seek(s.ff, thrust * tsfc, 1000.0f, 10000.0f, dt), idle_ff);
      }
      if (aug_cmd > 0.0f) {
        augm = true;
        thrust += (MAXT * aug_f - thrust) * aug_cmd;
        s.ff = seek(s.ff, thrust * (float)F16_ENG_ATSFC, 5000.0f, 10000.0f, dt);
      } else augm = false;
      if (cutoff || starved) phase = PH_OFF;
    } else if (phase == PH_TRIM) {
      thrust = idle + mil * tp * tp;
      if (aug_cmd > 0.0f) thrust += (MAXT * aug_f - thrust) * aug_cmd;
    } else if (phase == PH_START) {
      if (s.n2 > 15.0f && !starved) {
        if (s.n2 < IN2) {
          s.n2 = seek(s.n2, IN2, 2.0f, s.n2 * 0.5f, dt);
          s.n1 = seek(s.n1, IN1, 1.4f, s.n1 * 0.5f, dt);
          s.ff = idle_ff * s.n2 / IN2;
          if (qbar < 30.0f) phase = PH_OFF;
        } else { phase = PH_RUN; running = true; }
      } else phase = PH_OFF;
    } else {  // Off(): spin down towards windmilling
      running = false;
      s.ff = seek(s.ff, 0.0f, 1000.0f, 10000.0f, dt);
      s.n1 = seek(s.n1, qbar * 0.1f, s.n1 * 0.5f + 0.1f, s.n1 * 0.5f, dt);
      s.n2 = seek(s.n2, qbar * (1.0f / 15.0f), s.n2 * 0.5f + 0.1f, s.n2 * 0.5f, dt);
      augm = false;
    }
    // ConsumeFuel (FGPropulsion.cpp:164-258): equal draw from every tank that still holds fuel
    int nfuel = (s.tank0 > 0.0f ? 1 : 0) + (s.tank1 > 0.0f ? 1 : 0);
    starved = (nfuel == 0);
    if (!starved && !DT_ZERO) {
      float need = s.ff * (1.0f / 3600.0f) * dt / (float)nfuel;
      if (s.tank0 > 0.0f) s.tank0 = (s.tank0 - need >= 0.0f) ? s.tank0 - need : 0.0f;
      if (s.tank1 > 0.0f) s.tank1 = (s.tank1 - need >= 0.0f) ? s.tank1 - need : 0.0f;
    }
    s.eng = phase | (running ? ENG_RUNNING : 0) | (cutoff ? ENG_CUTOFF : 0) | (starved ? ENG_STARVED : 0) | (augm ? ENG_AUG : 0);
  }
  // thrust along body x through the structural origin: arm from the CG
  float tx = kInch2Ft * (cgx - (float)F16_THRUSTER_X), ty = kInch2Ft * ((float)F16_THRUSTER_Y - cgy), tz = kInch2Ft * (cgz - (float)F16_THRUSTER_Z);
  (void)tx;
  float Mpy = tz * thrust, Mpz = -ty * thrust;

  // ---------------- Aerodynamics (FGAerodynamics.cpp:132-300; f16.xml:994-1925)
  float Fx, Fy, Fz, Mx, My, Mz;
  {
    const float Sw = (float)F16_WINGAREA, bw = (float)F16_WINGSPAN, cbar = (float)F16_CHORD;
    float i2v = (vt != 0.0f) ? 0.5f / vt : 0.0f;
    float bi2vel = bw * i2v, ci2vel = cbar * i2v;
    float qS = qbar * Sw;
    // segment indices from immediates, then every LDS read of the tick issued as ONE batch (breakpoints and table rows depend on
    // the indices only), then the interpolation: one LDS round trip per tick instead of one per table group
    constexpr float kAX[] = C_ALPHA_X, kDX[] = C_DE_X, kB13X[] = C_B13_X, kB7X[] = C_B7_X, kMX[] = C_MACH_X;
    const int ia = seg_index(kAX, alpha), ide = seg_index(kDX, elevator_rad), ib13 = seg_index(kB13X, beta),
              ib7 = seg_index(kB7X, beta), im = seg_index(kMX, mach);
    const int a1 = P_A1_OFF + (ia - 1) * P_A1_STRIDE;
    const int ae = P_AE_OFF + ((ia - 1) * 5 + (ide - 1)) * 4;
    const int ab13 = P_AB13_OFF + ((ia - 1) * 13 + (ib13 - 1)) * 2;
    const int ab7 = P_AB7_OFF + ((ia - 1) * 7 + (ib7 - 1)) * 4;
    const int am = P_M_OFF + (im - 1) * P_M_STRIDE;
    const float xa0 = T[P_ALPHA_X_OFF + ia - 1], xa1 = T[P_ALPHA_X_OFF + ia], xe0 = T[P_DE_X_OFF + ide - 1], xe1 = T[P_DE_X_OFF + ide];
    const float xb0 = T[P_B13_X_OFF + ib13 - 1], xb1 = T[P_B13_X_OFF + ib13], xc0 = T[P_B7_X_OFF + ib7 - 1], xc1 = T[P_B7_X_OFF + ib7];
    const float xm0 = T[P_MACH_X_OFF + im - 1], xm1 = T[P_MACH_X_OFF + im];
    const float4 ra0 = T.v4(a1), ra1 = T.v4(a1 + 16), ra2 = T.v4(a1 + 4), ra3 = T.v4(a1 + 20), ra4 = T.v4(a1 + 8), ra5 = T.v4(a1 + 24);
    const float4 re0 = T.v4(ae), re1 = T.v4(ae + 20), re2 = T.v4(ae + 4), re3 = T.v4(ae + 24);
    const float2 h00 = T.v2(ab13), h01 = T.v2(ab13 + 2), h10 = T.v2(ab13 + 26), h11 = T.v2(ab13 + 28);
    const float4 rb0 = T.v4(ab7), rb1 = T.v4(ab7 + 28), rb2 = T.v4(ab7 + 4), rb3 = T.v4(ab7 + 32);
    const float4 rm0 = T.v4(am), rm1 = T.v4(am + 12), rm2 = T.v4(am + 4), rm3 = T.v4(am + 16);
    const float rn0 = T[am + 8], rn1 = T[am + 20];
    __builtin_amdgcn_sched_barrier(0);
    const float fa = seg_frac(xa0, xa1, alpha), fde = seg_frac(xe0, xe1, elevator_rad), fb13 = seg_frac(xb0, xb1, beta),
                fb7 = seg_frac(xc0, xc1, beta), fmach = seg_frac(xm0, xm1, mach);
    float4 g0 = lerp4(ra0, ra1, fa);                           // CDDlef CDq CDq_Dlef CLDlef
    float4 g1 = lerp4(ra2, ra3, fa);                           // CYp CYr Clp Clr
    float4 g2 = lerp4(ra4, ra5, fa);                           // CLq Cmq Cnp Cnr
    float4 g3 = make_float4(0.f, 0.f, 0.f, 0.f);               // CDDsb CLDsb CLq_Dsb CmDsb: only with the speedbrake out
    if (sb_rad != 0.0f) g3 = lerp4(T.v4(a1 + 12), T.v4(a1 + 28), fa);
    float4 ge = lerp4(lerp4(re0, re1, fa), lerp4(re2, re3, fa), fde);      // CDDh CLDh CmDh
    float clb = lerpf(lerpf(h00.x, h10.x, fa), lerpf(h01.x, h11.x, fa), fb13);
    float cnb = lerpf(lerpf(h00.y, h10.y, fa), lerpf(h01.y, h11.y, fa), fb13);
    float4 g7 = lerp4(lerp4(rb0, rb1, fa), lerp4(rb2, rb3, fa), fb7);      // Clda Cldr Cnda Cndr
    float4 m0 = lerp4(rm0, rm1, fmach);                        // CDmach CYb_M Clb_M Clda_M
    float4 m1 = lerp4(rm2, rm3, fmach);                        // Cldr_M Cma_M Cnb_M Cnda_M
    float cndr_m = lerpf(rn0, rn1, fmach);                     // Cndr_M
    // hoverbmac > 1.1 everywhere above the 2500 m floor, kCLge = 1 there; the table only matters for very low floors
    // (altitude above the sea-level radius stands in for AGL; the reference-point offset is < 2 ft)
    float hb = d.h_sl_ft * (1.0f / (float)F16_WINGSPAN);
    float kge = 1.0f;
    if (hb < 1.1f) {
      constexpr float kGX[] = {0.0f, 0.1f, 0.15f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f, 1.0f, 1.1f};
      int ig = seg_index(kGX, hb);
      kge = lerpf(T[P_KCLGE_Y_OFF + ig - 1], T[P_KCLGE_Y_OFF + ig], seg_frac(T[P_KCLGE_X_OFF + ig - 1], T[P_KCLGE_X_OFF + ig], hb));
    }
    float p = d.p, q = d.q, r = d.r;
    float qc = q * ci2vel;
    float CD = ge.x + m0.x + lef_rad * g0.x + flaperon_rad * (float)F16_K_CDDFLAPS + (float)F16_K_CDGEAR + sb_rad * g3.x +
               qc * (g0.y + lef_rad * g0.z);
    float CY = beta * ((float)F16_K_CYB + m0.y) + aileron_rad * (float)F16_K_CYDA + rudder_rad * (float)F16_K_CYDR +
               bi2vel * (p * g1.x + r * g1.y);
    float CL = kge * (ge.y + lef_rad * g0.w + flaperon_rad * (float)F16_K_CLDFLAPS + sb_rad * g3.y + qc * g2.x) + qc * sb_rad * g3.z;
    float Cl = clb + beta * m0.z + bi2vel * (p * g1.z + r * g1.w) + aileron_rad * (g7.x + alpha * m0.w) + rudder_rad * (g7.y + alpha * m1.x);
    float Cm = ge.z + alpha * m1.y + sb_rad * g3.w + qc * g2.y;
    float Cn = cnb + beta * m1.z + bi2vel * (p * g2.z + r * g2.w) + aileron_rad * (m1.w + g7.z) + rudder_rad * (g7.w + alpha * cndr_m);
    float D = qS * CD, Y = qS * CY, L = qS * CL;
    // wind -> body: F = Tw2b * (-D, Y, -L)
    Fx = ca * cb * (-D) - ca * sb * Y + sa * L;
    Fy = sb * (-D) + cb * Y;
    Fz = sa * cb * (-D) - sa * sb * Y - ca * L;
    // moments about the aero reference point, transferred to the CG
    float ax = kInch2Ft * (cgx - (float)F16_AERORP_X), ay = kInch2Ft * ((float)F16_AERORP_Y - cgy), az = kInch2Ft * (cgz - (float)F16_AERORP_Z);
    Mx = qS * bw * Cl + (ay * Fz - az * Fy);
    My = qS * cbar * Cm + (az * Fx - ax * Fz);
    Mz = qS * bw * Cn + (ax * Fy - ay * Fx);
  }

  // ---------------- Accelerations (FGAccelerations.cpp:138-208)
  {
    float Tx = Fx + thrust, Ty = Fy, Tz = Fz;
    float Lm = Mx, Mm = My + Mpy, Nm = Mz + Mpz;
    // J * wi, wi x (J wi)   (J has negated products off the diagonal: J12 = -Ixy ... stored here as Jxy = J(1,2))
    float hx = Jxx * s.wp + Jxy * s.wq + Jxz * s.wr;
    float hy = Jxy * s.wp + Jyy * s.wq + Jyz * s.wr;
    float hz = Jxz * s.wp + Jyz * s.wq + Jzz * s.wr;
    float rx_ = Lm - (s.wq * hz - s.wr * hy), ry_ = Mm - (s.wr * hx - s.wp * hz), rz_ = Nm - (s.wp * hy - s.wq * hx);
    // symmetric 3x3 inverse by cofactors
    float c00 = Jyy * Jzz - Jyz * Jyz, c01 = Jxz * Jyz - Jxy * Jzz, c02 = Jxy * Jyz - Jxz * Jyy;
    float c11 = Jxx * Jzz - Jxz * Jxz, c12 = Jxy * Jxz - Jxx * Jyz, c22 = Jxx * Jyy - Jxy * Jxy;
    float idet = 1.0f / (Jxx * c00 + Jxy * c01 + Jxz * c02);
    s.wdx = (c00 * rx_ + c01 * ry_ + c02 * rz_) * idet;
    s.wdy = (c01 * rx_ + c11 * ry_ + c12 * rz_) * idet;
    s.wdz = (c02 * rx_ + c12 * ry_ + c22 * rz_) * idet;
    float im_ = 1.0f / mass;
    s.bax = Tx * im_; s.bay = Ty * im_; s.baz = Tz * im_;
    // inertial acceleration: Tb2i * a_body + g_eci
    s.aix = Tb[0] * s.bax + Tb[3] * s.bay + Tb[6] * s.baz + gx;
    s.aiy = Tb[1] * s.bax + Tb[4] * s.bay + Tb[7] * s.baz + gy;
    s.aiz = Tb[2] * s.bax + Tb[5] * s.bay + Tb[8] * s.baz + gz;
  }
  // publish this tick's auxiliary outputs for the next tick's FCS
  s.alpha = alpha; s.mach = mach; s.qc = qc; s.vg = vg;
  s.ap = d.p; s.aq = d.q; s.ar = d.r;
  s.npx = npx; s.npy = npy; s.npz = npz;
}

// One executive tick. DT_ZERO = true reproduces the two "integration suspended" passes of FGFDMExec::RunIC
// (data/src/FGFDMExec.cpp:636-669): nothing integrates, the turbine runs its Trim() branch, the FCS still steps.
// TANK_ARM_ORIGIN = true is the very first of those passes: FGMassBalance has not run yet, so the tank inertia the
// executive loads for it (FGFDMExec.cpp:572 before FGMassBalance::Run) is taken about the structural origin; that pass's
// angular acceleration reaches the FCS of the first real tick through the pilot-station load factor.
template <bool DT_ZERO, bool TANK_ARM_ORIGIN = false>
__device__ __forceinline__ void tick(State& s, Derived& d, const Tab& T) {
  if (!DT_ZERO) propagate(s);
  tick_after_propagate<DT_ZERO, TANK_ARM_ORIGIN>(s, d, T);
}

}  // namespace f16
