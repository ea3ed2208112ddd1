/* ORACLE — TEST INFRASTRUCTURE ONLY (see f16_fdm.h header for scope and parity status).
 *
 * Scalar float64 restatement of one JSBSim tick for the F-16 as the reference drives it.
 * "S/" below abbreviates /root/reference/envs/JSBSim/data/src/ ; "f16.xml" is
 * /root/reference/envs/JSBSim/data/aircraft/f16/f16.xml.
 *
 * Model order per tick (S/FGFDMExec.cpp:217-236,407-431): Propagate, Inertial, Atmosphere, FCS,
 * MassBalance, Auxiliary, Propulsion, Aerodynamics, (ground/external/buoyant: zero in flight),
 * Aircraft, Accelerations.  Every model reads what the models before it wrote THIS tick and what
 * the models after it wrote LAST tick (S/FGFDMExec.cpp:435-605).
 */
#include "f16_fdm.h"
#include "f16_tables.h"
#include <math.h>
#include <string.h>
#include <float.h>

/* ---- constants that live in the stripped JSBSim headers; standard published values */
#define FTTOM 0.3048
#define INCHTOFT (1.0 / 12.0)
#define SLUGTOLB 32.174049
#define LBTOSLUG (1.0 / 32.174049)
#define KGTOSLUG 0.06852168
#define KTSTOFPS 1.68781
#define RADTODEG (180.0 / M_PI)
#define DEGTORAD (M_PI / 180.0)
#define G0_FT (9.80665 / FTTOM) /* Inertial standard gravity, ft/s^2 */
/* S/models/FGInertial.cpp:56-60 */
#define OMEGA_E 0.00007292115
#define GM_E 14.0764417572E15
#define J2_E 1.08262982E-03
#define A_E 20925646.32546
#define B_E 20855486.5951
/* FCS components latch dt when the model is LOADED (S/models/flight_control/FGFCSComponent.cpp:58)
 * and the executive's default dT at that moment is 1/120 s (S/FGFDMExec.cpp:96); the reference
 * calls set_dt(1/60) only AFTER load_model (envs/JSBSim/core/simulatior.py:165-169), so every PID
 * and kinematic of the F-16 FCS keeps dt = 1/120 while the EOM integrate at 1/60. */
#define FCS_DT (1.0 / 120.0)

/* ------------------------------------------------------------------ small vector helpers */
static void cross3(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mv3(const double M[9], const double v[3], double o[3]) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mtv3(const double M[9], const double v[3], double o[3]) { /* M^T v */
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
  double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
  double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
static void mm3(const double A[9], const double B[9], double C[9]) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static void mt3(const double A[9], double T[9]) {
  double t[9] = {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]};
  memcpy(T, t, sizeof t);
}
static double clampd(double lo, double v, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double sgn(double v) { return v >= 0.0 ? 1.0 : -1.0; } /* FGJSBBase sign(): num>=0 ? 1 : -1 */

/* ------------------------------------------------------------------ FGTable (S/math/FGTable.cpp:443-516) */
double f16_tab1(int off, int nr, double key) {
  const double* x = &F16_TAB[off];
  const double* y = x + nr;
  if (key <= x[0]) return y[0];
  if (key >= x[nr - 1]) return y[nr - 1];
  int r = 1;
  while (r < nr - 1 && x[r] < key) r++;
  double span = x[r] - x[r - 1], f = 1.0;
  if (span != 0.0) { f = (key - x[r - 1]) / span; if (f > 1.0) f = 1.0; }
  return f * (y[r] - y[r - 1]) + y[r - 1];
}
double f16_tab2(int off, int nr, int nc, double rk, double ck) {
  const double* rx = &F16_TAB[off];
  const double* cx = rx + nr;
  const double* v = cx + nc;
  int r = 1, c = 1;
  while (r < nr - 1 && rx[r] < rk) r++;
  while (c < nc - 1 && cx[c] < ck) c++;
  double rf = (rk - rx[r - 1]) / (rx[r] - rx[r - 1]);
  double cf = (ck - cx[c - 1]) / (cx[c] - cx[c - 1]);
  rf = clampd(0.0, rf, 1.0);
  cf = clampd(0.0, cf, 1.0);
  double c1 = rf * (v[r * nc + c - 1] - v[(r - 1) * nc + c - 1]) + v[(r - 1) * nc + c - 1];
  double c2 = rf * (v[r * nc + c] - v[(r - 1) * nc + c]) + v[(r - 1) * nc + c];
  return c1 + cf * (c2 - c1);
}
#define TAB1(N, key) f16_tab1(T_##N##_OFF, T_##N##_NR, (key))
#define TAB2(N, rk, ck) f16_tab2(T_##N##_OFF, T_##N##_NR, T_##N##_NC, (rk), (ck))

/* ------------------------------------------------------------------ atmosphere
 * S/models/atmosphere/FGStandardAtmosphere.cpp:66-74 (table), :152-222 (pressure), :244-268 (temperature),
 * :494-521 (density altitude); S/models/FGAtmosphere.cpp:107-131. Standard day: no bias/gradient/humidity. */
static const double ATM_H[9] = {0.0, 36089.2388, 65616.7979, 104986.8766, 154199.4751, 167322.8346, 232939.6325, 278385.8268, 298556.4304};
static const double ATM_T[9] = {518.67, 389.97, 389.97, 411.57, 487.17, 487.17, 386.37, 336.5028, 336.5028};
#define ATM_SLP 2116.228
#define ATM_EARTH_R (6356766.0 / FTTOM)
static double atm_reng(void) {
  double Rstar = 8.31432 * KGTOSLUG / (1.8 * FTTOM * FTTOM);
  double Mair = 28.9645 * KGTOSLUG / 1000.0;
  return Rstar / Mair;
}
static double atm_lapse[8], atm_pb[9], atm_db[9];
static int atm_ready = 0;
/* CalculatePressureBreakpoints (FGStandardAtmosphere.cpp:461-481) for a temperature bias (SetTemperatureBias, :344-354; the
 * reference never sets one: bias = 0 gives StdPressureBreakpoints). The biased form exists so that the reference's own
 * TestDensityAltitude / TestPressureAltitude tables (delta-T = +-27 R rows) can pin the layer formulas. */
static void atm_breakpoints(double bias, double* pb) {
  double R = atm_reng();
  pb[0] = ATM_SLP;
  for (int b = 0; b < 8; b++) {
    double Tmb = ATM_T[b] + bias, dH = ATM_H[b + 1] - ATM_H[b], L = atm_lapse[b];
    if (L != 0.0) pb[b + 1] = pb[b] * pow(Tmb / (Tmb + L * dH), G0_FT / (R * L));
    else pb[b + 1] = pb[b] * exp(-G0_FT * dH / (R * Tmb));
  }
}
static void atm_init(void) {
  if (atm_ready) return;
  double R = atm_reng();
  for (int b = 0; b < 8; b++) atm_lapse[b] = (ATM_T[b + 1] - ATM_T[b]) / (ATM_H[b + 1] - ATM_H[b]);
  atm_breakpoints(0.0, atm_pb);
  for (int b = 0; b < 9; b++) atm_db[b] = atm_pb[b] / (R * ATM_T[b]);
  atm_ready = 1;
}
static double atm_temp_geopot(double gp) {
  if (gp < 0.0) return ATM_T[0] + gp * atm_lapse[0];
  if (gp <= ATM_H[0]) return ATM_T[0];
  if (gp >= ATM_H[8]) return ATM_T[8];
  int r = 1;
  while (r < 8 && ATM_H[r] < gp) r++;
  double f = (gp - ATM_H[r - 1]) / (ATM_H[r] - ATM_H[r - 1]);
  if (f > 1.0) f = 1.0;
  return f * (ATM_T[r] - ATM_T[r - 1]) + ATM_T[r - 1];
}
/* GetTemperature (:232-256), GetPressure (:186-214), Density = P / (Reng T) (FGAtmosphere.cpp:107-131, dry air),
 * CalculateDensityAltitude (:494-521), CalculatePressureAltitude (:525-553); `bias` = atmosphere/delta-T [R] */
void f16_atmosphere_bias(double h, double bias, double* T, double* P, double* rho, double* snd, double* dens_alt, double* press_alt) {
  atm_init();
  double R = atm_reng();
  double pbs[9];
  const double* pb = atm_pb;
  if (bias != 0.0) { atm_breakpoints(bias, pbs); pb = pbs; }
  double gp = h * ATM_EARTH_R / (ATM_EARTH_R + h);
  *T = atm_temp_geopot(gp) + bias;
  double base = ATM_H[0];
  int b;
  for (b = 0; b < 7; ++b) {
    double test = ATM_H[b + 1];
    if (gp < test) break;
    base = test;
  }
  double Tmb = atm_temp_geopot(base) + bias, dH = gp - base, L = atm_lapse[b];
  if (L != 0.0) *P = pb[b] * pow(Tmb / (Tmb + L * dH), G0_FT / (R * L));
  else *P = pb[b] * exp(-G0_FT * dH / (R * Tmb));
  *rho = *P / (R * *T);
  *snd = sqrt(1.4 * R * *T);
  /* CalculateDensityAltitude: inverted on the STANDARD day's breakpoints */
  int k = 0;
  for (; k < 7; k++) if (*rho >= atm_db[k + 1]) break;
  double Tk = ATM_T[k], Hk = ATM_H[k], Lk = atm_lapse[k], pk = atm_db[k], da;
  if (Lk != 0.0) da = Hk + (Tk / Lk) * (pow(*rho / pk, -1.0 / (1.0 + G0_FT / (R * Lk))) - 1.0);
  else da = Hk + (-R * Tk / G0_FT) * log(*rho / pk);
  *dens_alt = da * ATM_EARTH_R / (ATM_EARTH_R - da);
  if (press_alt) {
    int j = 0;
    for (; j < 7; j++) if (*P >= atm_pb[j + 1]) break;
    double Tj = ATM_T[j], Hj = ATM_H[j], Lj = atm_lapse[j], Pj = atm_pb[j], pa;
    if (Lj != 0.0) pa = Hj + (Tj / Lj) * (pow(*P / Pj, -R * Lj / G0_FT) - 1.0);
    else pa = Hj + (-R * Tj / G0_FT) * log(*P / Pj);
    *press_alt = pa * ATM_EARTH_R / (ATM_EARTH_R - pa);
  }
}
void f16_atmosphere(double h, double* T, double* P, double* rho, double* snd, double* dens_alt) {
  f16_atmosphere_bias(h, 0.0, T, P, rho, snd, dens_alt, 0);
}

/* S/FGJSBBase.cpp:245-296 */
static double pitot_total_pressure(double mach, double p) {
  if (mach < 0) return p;
  if (mach < 1) return p * pow(1 + 0.2 * mach * mach, 3.5);
  return p * 166.92158009316827 * pow(mach, 7.0) / pow(7 * mach * mach - 1, 2.5);
}
static double mach_from_impact_pressure(double qc, double p) {
  double A = qc / p + 1;
  double M = sqrt(5.0 * (pow(A, 1. / 3.5) - 1));
  if (M > 1.0)
    for (int i = 0; i < 10; i++) M = 0.8812848543473311 * sqrt(A * pow(1 - 1.0 / (7.0 * M * M), 2.5));
  return M;
}
double f16_pitot_qc(double mach, double p) { return pitot_total_pressure(mach, p) - p; }
double f16_vcas_from_qc(double qc) { return sqrt(1.4 * atm_reng() * 518.67) * mach_from_impact_pressure(qc, ATM_SLP); }
double f16_vcas_from_mach(double mach, double p) {
  double asl = sqrt(1.4 * atm_reng() * 518.67);
  double qc = pitot_total_pressure(mach, p) - p;
  return asl * mach_from_impact_pressure(qc, ATM_SLP);
}

/* ------------------------------------------------------------------ FGLocation (S/math/FGLocation.cpp:243-330) */
void f16_geodetic_from_ecef(const double r[3], double* lon, double* lat_gc, double* lat_geod, double* h_geod, double* radius) {
  const double a = A_E, ec = B_E / A_E, ec2 = ec * ec, e2 = 1.0 - ec2, c = a * e2;
  double rad = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  double rxy = sqrt(r[0] * r[0] + r[1] * r[1]);
  *radius = rad;
  *lon = (rxy == 0.0) ? 0.0 : atan2(r[1], r[0]);
  *lat_gc = atan2(r[2], rxy);
  double s0 = fabs(r[2]), zc = ec * s0, c0 = ec * rxy, c02 = c0 * c0, s02 = s0 * s0, a02 = c02 + s02;
  double a0 = sqrt(a02), a03 = a02 * a0;
  double s1 = zc * a03 + c * s02 * s0, c1 = rxy * a03 - c * c02 * c0, cs0c0 = c * c0 * s0;
  double b0 = 1.5 * cs0c0 * ((rxy * s0 - zc * c0) * a0 - cs0c0);
  s1 = s1 * a03 - b0 * s0;
  double cc = ec * (c1 * a03 - b0 * c0);
  *lat_geod = sgn(r[2]) * atan(s1 / cc);
  double s12 = s1 * s1, cc2 = cc * cc, norm = sqrt(s12 + cc2);
  *h_geod = (rxy * cc + s0 * s1 - a * sqrt(ec2 * s12 + cc2)) / norm;
}
static double sea_level_radius(double lat_gc) { /* FGLocation::GetSeaLevelRadius */
  const double ec = B_E / A_E, e2 = 1.0 - ec * ec;
  double cl = cos(lat_gc);
  return A_E * ec / sqrt(1.0 - e2 * cl * cl);
}
static void set_position_geodetic(double lon, double lat, double h, double r[3]) { /* FGLocation::SetPositionGeodetic */
  const double ec = B_E / A_E, e2 = 1.0 - ec * ec;
  double sl = sin(lat), cl = cos(lat), RN = A_E / sqrt(1.0 - e2 * sl * sl);
  r[0] = (RN + h) * cl * cos(lon);
  r[1] = (RN + h) * cl * sin(lon);
  r[2] = ((1 - e2) * RN + h) * sl;
}

/* ------------------------------------------------------------------ FGQuaternion (S/math/FGQuaternion.cpp) */
static void quat_from_euler(double phi, double tht, double psi, double q[4]) {
  double st = sin(0.5 * tht), sp = sin(0.5 * psi), sf = sin(0.5 * phi);
  double ct = cos(0.5 * tht), cp = cos(0.5 * psi), cf = cos(0.5 * phi);
  q[0] = cf * ct * cp + sf * st * sp;
  q[1] = sf * ct * cp - cf * st * sp;
  q[2] = cf * st * cp + sf * ct * sp;
  q[3] = cf * ct * sp - sf * st * cp;
}
static void quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n == 0.0 || fabs(n - 1.0) < 1e-10) return;
  double rn = 1.0 / n;
  q[0] *= rn; q[1] *= rn; q[2] *= rn; q[3] *= rn;
}
static void quat_to_T(const double q[4], double T[9]) {
  double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  T[0] = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; T[1] = 2.0 * (q1 * q2 + q0 * q3); T[2] = 2.0 * (q1 * q3 - q0 * q2);
  T[3] = 2.0 * (q1 * q2 - q0 * q3); T[4] = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; T[5] = 2.0 * (q2 * q3 + q0 * q1);
  T[6] = 2.0 * (q1 * q3 + q0 * q2); T[7] = 2.0 * (q2 * q3 - q0 * q1); T[8] = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
}
static void quat_from_T(const double m[9], double q[4]) { /* FGMatrix33::GetQuaternion (S/math/FGMatrix33.cpp:106-151), row-major here */
  double t[4] = {1.0 + m[0] + m[4] + m[8], 1.0 + m[0] - m[4] - m[8], 1.0 - m[0] + m[4] - m[8], 1.0 - m[0] - m[4] + m[8]};
  int idx = 0;
  for (int i = 1; i < 4; i++) if (t[i] > t[idx]) idx = i;
  /* data[] there is column-major: data[7]=m(2,3) data[5]=m(3,2) data[2]=m(3,1) data[6]=m(1,3) data[3]=m(1,2) data[1]=m(2,1) */
  double m23 = m[5], m32 = m[7], m31 = m[6], m13 = m[2], m12 = m[1], m21 = m[3];
  switch (idx) {
    case 0: q[0] = 0.5 * sqrt(t[0]); q[1] = 0.25 * (m23 - m32) / q[0]; q[2] = 0.25 * (m31 - m13) / q[0]; q[3] = 0.25 * (m12 - m21) / q[0]; break;
    case 1: q[1] = 0.5 * sqrt(t[1]); q[0] = 0.25 * (m23 - m32) / q[1]; q[2] = 0.25 * (m12 + m21) / q[1]; q[3] = 0.25 * (m31 + m13) / q[1]; break;
    case 2: q[2] = 0.5 * sqrt(t[2]); q[0] = 0.25 * (m31 - m13) / q[2]; q[1] = 0.25 * (m12 + m21) / q[2]; q[3] = 0.25 * (m23 + m32) / q[2]; break;
    default: q[3] = 0.5 * sqrt(t[3]); q[0] = 0.25 * (m12 - m21) / q[3]; q[1] = 0.25 * (m13 + m31) / q[3]; q[2] = 0.25 * (m23 + m32) / q[3]; break;
  }
}
static void quat_mul(const double a[4], const double b[4], double o[4]) { /* FGQuaternion operator* */
  double q0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double q1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double q2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double q3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
}
static void euler_from_T(const double m[9], double* phi, double* tht, double* psi) { /* FGMatrix33::GetEuler (S/math/FGMatrix33.cpp:159-190) */
  int lock = 0;
  if (m[2] <= -1.0) { *tht = 0.5 * M_PI; lock = 1; }
  else if (1.0 <= m[2]) { *tht = -0.5 * M_PI; lock = 1; }
  else *tht = asin(-m[2]);
  if (lock) { *phi = atan2(-m[7], m[4]); *psi = 0.0; }
  else {
    *phi = atan2(m[5], m[8]);
    double p = atan2(m[1], m[0]);
    if (p < 0.0) p += 2 * M_PI;
    *psi = p;
  }
}

/* ------------------------------------------------------------------ FCS components */
/* FGKinemat::Run (S/models/flight_control/FGKinemat.cpp:99-170), noscale absent => Input *= Detents.back() */
static int equal_to_roundoff(double a, double b) {
  double eps = 2.0 * DBL_EPSILON;
  return fabs(a - b) <= eps * fmax(fabs(a), fabs(b));
}
double f16_kinemat(double out, double in, const double* det, const double* tt, int n, double dt) {
  double dt0 = dt;
  in *= det[n - 1];
  in = clampd(det[0], in, det[n - 1]);
  while (dt0 > 0.0 && !equal_to_roundoff(in, out)) {
    int ind;
    for (ind = 1; (in < out) ? det[ind] < out : det[ind] <= out; ++ind)
      if (ind >= n - 1) { break; } /* guard: the original reads one past the end before testing */
    if (ind > n - 1) ind = n - 1;
    if (tt[ind] <= 0.0) { out = in; break; }
    double rate = (det[ind] - det[ind - 1]) / tt[ind];
    double this_in = clampd(det[ind - 1], in, det[ind]);
    double this_dt = fabs((this_in - out) / rate);
    if (dt0 < this_dt) {
      this_dt = dt0;
      if (out < in) out += this_dt * rate; else out -= this_dt * rate;
    } else out = this_in;
    dt0 -= this_dt;
  }
  return out;
}
/* FGPID::Run (S/models/flight_control/FGPID.cpp:154-204); <ki> without type => Adams-Bashforth-2 (:84-98) */
static double pid_run_dt(OrPid* p, double in, double trigger, double kp, double ki, double kd, int clip, double dt) {
  double dval = (in - p->in_prev) / dt;
  double i_delta = 0.0;
  if (fabs(trigger) < 0.000001) i_delta = 1.5 * in - 0.5 * p->in_prev;
  if (trigger < 0.0) p->i_total = 0.0;
  p->i_total += ki * dt * i_delta;
  double out = kp * in + p->i_total + kd * dval;
  p->in_prev2 = trigger < 0.0 ? 0.0 : p->in_prev;
  p->in_prev = in;
  if (clip) out = clampd(-1.0, out, 1.0);
  p->out = out;
  return out;
}
static double pid_run(OrPid* p, double in, double trigger, double kp, double ki, double kd, int clip) {
  return pid_run_dt(p, in, trigger, kp, ki, kd, clip, FCS_DT);
}
/* test hook: one FGPID::Run at a given dt (the reference's TestIntegrators.py runs its <pid> blocks at 0.005 s); st = {in_prev, in_prev2, i_total, out} */
double f16_test_pid(double* st, double in, double trigger, double kp, double ki, double kd, double dt) {
  OrPid p = {st[0], st[1], st[2], st[3]};
  double o = pid_run_dt(&p, in, trigger, kp, ki, kd, 0, dt);
  st[0] = p.in_prev; st[1] = p.in_prev2; st[2] = p.i_total; st[3] = p.out;
  return o;
}

/* f16.xml:317-992, channels in document order (S/models/FGFCS.cpp:153-178) */
static void fcs_run(F16State* s) {
  const double vc_kts = s->vc_fps / KTSTOFPS;
  const double alpha = s->alpha, mach = s->mach;
  /* Flaps channel f16.xml:325-358 */
  double tef_pos_rad = 0.0;
  if (vc_kts < 250) tef_pos_rad = 0.349;
  else if (mach > 0.9) tef_pos_rad = -0.0349;
  double tef_pos_norm = 2.864789 * tef_pos_rad;
  { static const double d[3] = {-1.0, 0.0, 1.0}, t[3] = {3.0, 0.0, 3.0};
    s->tef_control = f16_kinemat(s->tef_control, tef_pos_norm, d, t, 3, FCS_DT); }
  /* Roll channel :360-496 */
  double roll_rate_norm = 0.31821 * s->aero_pqr[0];
  double roll_trim_error = s->da_cmd - roll_rate_norm;
  double ail_trig = (vc_kts < 20.0) ? 0.0 : 1.0;
  double roll_pid = pid_run(&s->pid_roll, roll_trim_error, ail_trig, 3.0, 0.0005, -0.00125, 0);
  double roll_rate_command = clampd(-1.0, roll_pid + s->da_cmd, 1.0);
  s->aileron_pos_rad = 0.375 * roll_rate_command; /* aerosurface_scale, zero-centred (FGGain.cpp:138-170) */
  { static const double d[2] = {-1.0, 1.0}, t[2] = {0.3, 0.3};
    s->left_aileron_pos_norm = f16_kinemat(s->left_aileron_pos_norm, roll_rate_command, d, t, 2, FCS_DT); }
  double ail_sc = s->left_aileron_pos_norm * TAB1(FCS_AILERON_SPEED_COMPENSATED, mach);
  double left_flap = clampd(-1.0, -s->tef_control - ail_sc, 1.0);
  double right_flap = clampd(-1.0, s->tef_control - ail_sc, 1.0);
  s->flaperon_mix_rad = 1.4324 * (left_flap + right_flap);
  /* Pitch channel :498-682; attitude is Propagate's of THIS tick */
  double nz_corr = cos(s->tht) * cos(s->phi);
  double g_load_corrected = s->npilot[2] - nz_corr;
  double elev_lim = clampd(-1.0, s->de_cmd + 0.0, 0.44);
  double elev_sched = elev_lim * TAB1(FCS_ELEVATOR_SCHEDULER, alpha);
  double alpha_lim = 1.0472 * alpha;
  double pitch_rate_norm = 6.2 * s->aero_pqr[1];
  double g_load_norm = 0.020 * g_load_corrected;
  double pitch_trim_error = elev_sched + pitch_rate_norm - g_load_norm;
  double el_trig = (vc_kts < 5.0) ? 0.0 : 1.0;
  double g_pid = pid_run(&s->pid_pitch, pitch_trim_error, el_trig, 0.3, 0.025, 0.0, 1);
  double pitch_sched = clampd(-1.0, elev_sched + alpha_lim + g_pid, 1.0);
  { static const double d[2] = {-1.0, 1.0}, t[2] = {0.3, 0.3};
    s->elevator_pos_norm = f16_kinemat(s->elevator_pos_norm, pitch_sched, d, t, 2, FCS_DT); }
  s->elevator_pos_rad = 0.436 * s->elevator_pos_norm;
  /* Yaw channel :684-771 */
  double yaw_rate_norm = s->aero_pqr[2] * TAB1(FCS_YAW_RATE_NORM, s->vg);
  double yaw_load_norm = 0.25 * s->npilot[1];
  double yaw_trim_error = s->dr_cmd + yaw_rate_norm + yaw_load_norm;
  double rud_trig = (vc_kts < 10.0) ? 0.0 : 1.0;
  double yaw_pid = pid_run(&s->pid_yaw, yaw_trim_error, rud_trig, 0.1055, 0.00001, 0.00005, 1);
  s->rudder_pos_norm = yaw_pid; /* the PID's <output> is fcs/rudder-pos-norm (f16.xml:734) ... */
  double yaw_sched = clampd(-1.0, s->dr_cmd + 0.0 + yaw_pid, 1.0);
  { static const double d[2] = {-1.0, 1.0}, t[2] = {0.4, 0.4}; /* ... which the kinematic re-reads as its own Output (FGKinemat.cpp:106-107) */
    s->rudder_pos_norm = f16_kinemat(s->rudder_pos_norm, yaw_sched, d, t, 2, FCS_DT); }
  s->rudder_pos_rad = 0.524 * s->rudder_pos_norm;
  /* Landing gear :773-812 — gear defaults DOWN (FGFCS.cpp:81) and is never commanded */
  const double gear_wow = 0.0;
  { static const double d[2] = {0.0, 1.0}, t[2] = {0.0, 5.0};
    s->gear_pos_norm = f16_kinemat(s->gear_pos_norm, s->gear_cmd_norm, d, t, 2, FCS_DT); }
  /* Leading edge flap :814-858: fcs/lef-pos-rad is the switch output itself */
  double lef = 0.0;
  if (gear_wow == 1.0 && s->gear_pos_norm > 0) lef = -0.0349;
  else if (s->gear_pos_norm == 0 && alpha > 0.2618) lef = 0.436;
  else if (gear_wow == 0.0 && alpha > 0.0873) lef = 0.262;
  else if (mach > 0.9) lef = -0.0349;
  s->lef_pos_rad = lef;
  /* Throttle :860-868 */
  s->throttle_pos = 2.0 * s->throttle_cmd;
  /* Speedbrake :870-935; velocities/v-fps is Propagate's body v of THIS tick */
  double sb_lim = (alpha * RADTODEG >= 53 && s->uvw[1] <= 18) ? 1.0 : 0.0;
  double sb_init = (sb_lim == 1.0) ? 1.0 : 0.0;
  double sb_sched = sb_init * TAB1(FCS_SPEEDBRAKE_SCHEDULER, s->gear_cmd_norm);
  { static const double d[2] = {0.0, 60.0}, t[2] = {0.0, 1.0};
    s->speedbrake_pos_deg = f16_kinemat(s->speedbrake_pos_deg, sb_sched, d, t, 2, FCS_DT); }
  s->speedbrake_pos_rad = s->speedbrake_pos_deg * DEGTORAD;
}

/* test hook: ONE pass of the whole <flight_control> section. st[18] = the section's memory (three PIDs {in_prev, in_prev2, i_total, out},
 * tef-control, left-aileron-pos-norm, elevator-pos-norm, rudder-pos-norm, speedbrake-pos-deg, gear-pos-norm), updated in place;
 * in[17] in the order of tests/golden/make_f16_fcs_check.py IN_PROPS (calibrated airspeed in knots); out[16] in the order of its OUT_PROPS. */
void f16_test_fcs(double* st, const double* in, double* out) {
  F16State s;
  memset(&s, 0, sizeof s);
  OrPid* pid[3] = {&s.pid_roll, &s.pid_pitch, &s.pid_yaw};
  for (int i = 0; i < 3; i++) { pid[i]->in_prev = st[4 * i]; pid[i]->in_prev2 = st[4 * i + 1]; pid[i]->i_total = st[4 * i + 2]; pid[i]->out = st[4 * i + 3]; }
  s.tef_control = st[12]; s.left_aileron_pos_norm = st[13]; s.elevator_pos_norm = st[14]; s.rudder_pos_norm = st[15];
  s.speedbrake_pos_deg = st[16]; s.gear_pos_norm = st[17];
  s.da_cmd = in[0]; s.de_cmd = in[1]; s.dr_cmd = in[2]; s.throttle_cmd = in[3]; s.gear_cmd_norm = in[4];
  s.vc_fps = in[5] * KTSTOFPS; s.mach = in[6]; s.aero_pqr[0] = in[7]; s.aero_pqr[1] = in[8]; s.aero_pqr[2] = in[9];
  s.vg = in[10]; s.uvw[1] = in[11]; s.tht = in[12]; s.phi = in[13]; s.npilot[2] = in[14]; s.npilot[1] = in[15]; s.alpha = in[16];
  fcs_run(&s);
  for (int i = 0; i < 3; i++) { st[4 * i] = pid[i]->in_prev; st[4 * i + 1] = pid[i]->in_prev2; st[4 * i + 2] = pid[i]->i_total; st[4 * i + 3] = pid[i]->out; }
  st[12] = s.tef_control; st[13] = s.left_aileron_pos_norm; st[14] = s.elevator_pos_norm; st[15] = s.rudder_pos_norm;
  st[16] = s.speedbrake_pos_deg; st[17] = s.gear_pos_norm;
  out[0] = s.aileron_pos_rad; out[1] = s.elevator_pos_rad; out[2] = s.rudder_pos_rad; out[3] = s.lef_pos_rad; out[4] = s.flaperon_mix_rad;
  out[5] = s.speedbrake_pos_rad; out[6] = s.throttle_pos; out[7] = s.gear_pos_norm; out[8] = s.tef_control; out[9] = s.left_aileron_pos_norm;
  out[10] = s.elevator_pos_norm; out[11] = s.rudder_pos_norm; out[12] = s.pid_roll.out; out[13] = s.pid_pitch.out; out[14] = s.pid_yaw.out;
  out[15] = s.speedbrake_pos_deg;
}

/* ------------------------------------------------------------------ FGPropagate */
static void propagate_derived(F16State* s) { /* tail of FGPropagate::Run, S/models/FGPropagate.cpp:237-283 */
  double ce = cos(s->epa), se = sin(s->epa);
  double Ti2ec[9] = {ce, se, 0, -se, ce, 0, 0, 0, 1};
  memcpy(s->Ti2ec, Ti2ec, sizeof Ti2ec);
  mv3(Ti2ec, s->r_eci, s->r_ecef);
  f16_geodetic_from_ecef(s->r_ecef, &s->lon, &s->lat_gc, &s->lat_geod, &s->h_geod, &s->radius);
  s->h_sl = s->radius - sea_level_radius(s->lat_gc);
  /* Tec2l with geodetic sin/cos (FGLocation.cpp:319-324) */
  double rxy = sqrt(s->r_ecef[0] * s->r_ecef[0] + s->r_ecef[1] * s->r_ecef[1]);
  double sinLon = (rxy == 0.0) ? 0.0 : s->r_ecef[1] / rxy, cosLon = (rxy == 0.0) ? 1.0 : s->r_ecef[0] / rxy;
  double sinLat = sin(s->lat_geod), cosLat = cos(s->lat_geod);
  double Tec2l[9] = {-cosLon * sinLat, -sinLon * sinLat, cosLat, -sinLon, cosLon, 0.0, -cosLon * cosLat, -sinLon * cosLat, -sinLat};
  memcpy(s->Tec2l, Tec2l, sizeof Tec2l);
  double Ti2l[9], Tl2i[9];
  mm3(Tec2l, Ti2ec, Ti2l);
  mt3(Ti2l, Tl2i);
  quat_to_T(s->q_eci, s->Ti2b);
  mm3(s->Ti2b, Tl2i, s->Tl2b);
  double Tec2i[9];
  mt3(Ti2ec, Tec2i);
  mm3(s->Ti2b, Tec2i, s->Tec2b);
  const double om[3] = {0, 0, OMEGA_E};
  double oxr[3], vrel[3], omb[3];
  cross3(om, s->r_eci, oxr);
  for (int i = 0; i < 3; i++) vrel[i] = s->v_eci[i] - oxr[i];
  mv3(s->Ti2b, vrel, s->uvw);
  mv3(s->Ti2b, om, omb);
  for (int i = 0; i < 3; i++) s->pqr[i] = s->pqr_i[i] - omb[i];
  const double* q = s->q_eci; const double* w = s->pqr_i;
  s->qdot[0] = -0.5 * (q[1] * w[0] + q[2] * w[1] + q[3] * w[2]);
  s->qdot[1] = 0.5 * (q[0] * w[0] - q[3] * w[1] + q[2] * w[2]);
  s->qdot[2] = 0.5 * (q[3] * w[0] + q[0] * w[1] - q[1] * w[2]);
  s->qdot[3] = 0.5 * (-q[2] * w[0] + q[1] * w[1] + q[0] * w[2]);
  /* qAttitudeLocal = Tl2b.GetQuaternion(); Euler angles come from its matrix */
  double ql[4], Tl[9];
  quat_from_T(s->Tl2b, ql);
  quat_to_T(ql, Tl);
  euler_from_T(Tl, &s->phi, &s->tht, &s->psi);
  mtv3(s->Tl2b, s->uvw, s->vel_ned);
}
static void propagate_run(F16State* s, double dt) { /* S/models/FGPropagate.cpp:218-235,336-360 */
  if (dt > 0.0) {
    for (int i = 0; i < 4; i++) s->q_eci[i] += dt * s->qdot[i]; /* eRectEuler */
    quat_normalize(s->q_eci);
    for (int i = 0; i < 3; i++) s->pqr_i[i] += dt * s->pqridot[i]; /* eRectEuler */
    memcpy(s->hist_v[2], s->hist_v[1], sizeof s->hist_v[0]);
    memcpy(s->hist_v[1], s->hist_v[0], sizeof s->hist_v[0]);
    memcpy(s->hist_v[0], s->v_eci, sizeof s->hist_v[0]);
    for (int i = 0; i < 3; i++) /* eAdamsBashforth3 */
      s->r_eci[i] += (1 / 12.0) * dt * (23.0 * s->hist_v[0][i] - 16.0 * s->hist_v[1][i] + 5.0 * s->hist_v[2][i]);
    memcpy(s->hist_a[1], s->hist_a[0], sizeof s->hist_a[0]);
    memcpy(s->hist_a[0], s->uvwidot, sizeof s->hist_a[0]);
    for (int i = 0; i < 3; i++) /* eAdamsBashforth2 */
      s->v_eci[i] += dt * (1.5 * s->hist_a[0][i] - 0.5 * s->hist_a[1][i]);
  }
  s->epa += OMEGA_E * dt;
  propagate_derived(s);
}

/* ------------------------------------------------------------------ FGInertial::GetGravityJ2 (S/models/FGInertial.cpp:193-213) */
static void inertial_run(F16State* s) {
  double r = s->radius, sl = sin(s->lat_gc), adivr = A_E / r, pre = 1.5 * J2_E * adivr * adivr;
  double xy = 1.0 - 5.0 * sl * sl, z = 3.0 - 5.0 * sl * sl, g = GM_E / (r * r);
  s->grav_ecef[0] = -g * ((1.0 + pre * xy) * s->r_ecef[0] / r);
  s->grav_ecef[1] = -g * ((1.0 + pre * xy) * s->r_ecef[1] / r);
  s->grav_ecef[2] = -g * ((1.0 + pre * z) * s->r_ecef[2] / r);
}

/* ------------------------------------------------------------------ FGMassBalance::Run (S/models/FGMassBalance.cpp:181-262) */
static void struct_to_body(const double cg[3], const double r[3], double o[3]) {
  o[0] = INCHTOFT * (cg[0] - r[0]); o[1] = INCHTOFT * (r[1] - cg[1]); o[2] = INCHTOFT * (cg[2] - r[2]);
}
static void pointmass_inertia(const double cg[3], double mass_sl, const double r[3], double J[9]) {
  double v[3];
  struct_to_body(cg, r, v);
  double sv[3] = {mass_sl * v[0], mass_sl * v[1], mass_sl * v[2]};
  double xx = sv[0] * v[0], yy = sv[1] * v[1], zz = sv[2] * v[2];
  double xy = -sv[0] * v[1], xz = -sv[0] * v[2], yz = -sv[1] * v[2];
  J[0] += yy + zz; J[1] += xy; J[2] += xz;
  J[3] += xy; J[4] += xx + zz; J[5] += yz;
  J[6] += xz; J[7] += yz; J[8] += xx + yy;
}
static const double TANK_XYZ[4][3] = {{F16_TANK0_X, F16_TANK0_Y, F16_TANK0_Z}, {F16_TANK1_X, F16_TANK1_Y, F16_TANK1_Z},
                                      {F16_TANK2_X, F16_TANK2_Y, F16_TANK2_Z}, {F16_TANK3_X, F16_TANK3_Y, F16_TANK3_Z}};
static void massbalance_core(F16State* s, double pm0_w, double pm1_w) {
  const double base_cg[3] = {F16_CG_X, F16_CG_Y, F16_CG_Z};
  const double pm0[3] = {F16_PM0_X, F16_PM0_Y, F16_PM0_Z}, pm1[3] = {F16_PM1_X, F16_PM1_Y, F16_PM1_Z};
  /* in.TankInertia is loaded BEFORE Run (FGFDMExec.cpp:572) with the cg of the previous tick */
  double tankJ[9] = {0};
  double tanks_w = 0.0, tanks_m[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) {
    pointmass_inertia(s->cg, LBTOSLUG * s->tank[i], TANK_XYZ[i], tankJ);
    tanks_w += s->tank[i];
    for (int k = 0; k < 3; k++) tanks_m[k] += TANK_XYZ[i][k] * s->tank[i];
  }
  s->weight = F16_EMPTYWT + tanks_w + pm0_w + pm1_w;
  s->mass = LBTOSLUG * s->weight;
  for (int k = 0; k < 3; k++)
    s->cg[k] = (F16_EMPTYWT * base_cg[k] + pm0_w * pm0[k] + pm1_w * pm1[k] + tanks_m[k]) / s->weight;
  /* baseJ with negated_crossproduct_inertia="true" (FGMassBalance.cpp:91-112) */
  double J[9] = {F16_IXX, -F16_IXY, F16_IXZ, -F16_IXY, F16_IYY, -F16_IYZ, F16_IXZ, -F16_IYZ, F16_IZZ};
  pointmass_inertia(s->cg, LBTOSLUG * F16_EMPTYWT, base_cg, J);
  pointmass_inertia(s->cg, LBTOSLUG * pm0_w, pm0, J);
  pointmass_inertia(s->cg, LBTOSLUG * pm1_w, pm1, J);
  for (int k = 0; k < 9; k++) J[k] += tankJ[k];
  memcpy(s->J, J, sizeof J);
  double Ixx = J[0], Iyy = J[4], Izz = J[8], Ixy = -J[1], Ixz = -J[2], Iyz = -J[5];
  double k1 = Iyy * Izz - Iyz * Iyz, k2 = Iyz * Ixz + Ixy * Izz, k3 = Ixy * Iyz + Iyy * Ixz;
  double denom = 1.0 / (Ixx * k1 - Ixy * k2 - Ixz * k3);
  k1 *= denom; k2 *= denom; k3 *= denom;
  double k4 = (Izz * Ixx - Ixz * Ixz) * denom, k5 = (Ixy * Ixz + Iyz * Ixx) * denom, k6 = (Ixx * Iyy - Ixy * Ixy) * denom;
  double Ji[9] = {k1, k2, k3, k2, k4, k5, k3, k5, k6};
  memcpy(s->Jinv, Ji, sizeof Ji);
  memcpy(s->tankJ, tankJ, sizeof tankJ);
}
static void massbalance_run(F16State* s) { massbalance_core(s, F16_PM0_WEIGHT, F16_PM1_WEIGHT); }
/* test hook: FGMassBalance::Run for given tank contents [lbs] and point-mass weights [lbs]; the tank inertia is taken about cg_tanks
 * (the executive hands MassBalance the tank inertia computed with the previous pass's CG, FGFDMExec.cpp:572).
 * out = weight, cg[3], J[9], Jinv[9], tank inertia[9] */
void f16_test_massbalance(const double* tanks4, const double* pm2, const double* cg_tanks, double* out31) {
  F16State s;
  memset(&s, 0, sizeof s);
  for (int i = 0; i < 4; i++) s.tank[i] = tanks4[i];
  for (int i = 0; i < 3; i++) s.cg[i] = cg_tanks[i];
  massbalance_core(&s, pm2[0], pm2[1]);
  out31[0] = s.weight;
  for (int i = 0; i < 3; i++) out31[1 + i] = s.cg[i];
  for (int i = 0; i < 9; i++) { out31[4 + i] = s.J[i]; out31[13 + i] = s.Jinv[i]; out31[22 + i] = s.tankJ[i]; }
}

/* ------------------------------------------------------------------ FGAuxiliary::Run (S/models/FGAuxiliary.cpp:134-232) */
/* vPilotAccel = vBodyAccel + vPQRidot x r + vPQRi x (vPQRi x r), r = StructuralToBody(eye point) (:205-213); [in], [ft/s2], [rad/s] */
static void pilot_accel(const double cg[3], const double eye[3], const double body_accel[3], const double pqridot[3], const double pqri[3], double out[3]) {
  double r[3], t1[3], t2[3];
  struct_to_body(cg, eye, r);
  cross3(pqridot, r, t1);
  cross3(pqri, r, t2);
  cross3(pqri, t2, t2);
  for (int i = 0; i < 3; i++) out[i] = body_accel[i] + t1[i] + t2[i];
}
void f16_test_pilot_accel(const double* cg, const double* eye, const double* body_accel, const double* pqridot, const double* pqri, double* out3) {
  pilot_accel(cg, eye, body_accel, pqridot, pqri, out3);
}
/* wind -> body (S/models/FGAuxiliary.cpp:256-264) */
static void wind_to_body(double alpha, double beta, double Tw2b[9]) {
  double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
  Tw2b[0] = ca * cb; Tw2b[1] = -ca * sb; Tw2b[2] = -sa;
  Tw2b[3] = sb;      Tw2b[4] = cb;       Tw2b[5] = 0.0;
  Tw2b[6] = sa * cb; Tw2b[7] = -sa * sb; Tw2b[8] = ca;
}
static void auxiliary_run(F16State* s) {
  for (int i = 0; i < 3; i++) s->aero_pqr[i] = s->pqr[i];
  double u = s->uvw[0], v = s->uvw[1], w = s->uvw[2];
  double mUW = u * u + w * w, Vt2 = mUW + v * v;
  s->vt = sqrt(Vt2);
  s->alpha = s->beta = 0.0;
  if (s->vt > 0.001) {
    s->beta = atan2(v, sqrt(mUW));
    if (mUW >= 1E-6) s->alpha = atan2(w, u);
  }
  wind_to_body(s->alpha, s->beta, s->Tw2b);
  s->qbar = 0.5 * s->rho * Vt2;
  s->mach = s->vt / s->snd;
  s->vg = sqrt(s->vel_ned[0] * s->vel_ned[0] + s->vel_ned[1] * s->vel_ned[1]);
  s->vc_fps = (fabs(s->mach) > 0.0) ? f16_vcas_from_mach(s->mach, s->P) : 0.0;
  /* pilot station acceleration uses last tick's Accelerations and the inertial rates (:205-217) */
  const double eye[3] = {F16_EYEPOINT_X, F16_EYEPOINT_Y, F16_EYEPOINT_Z};
  double apilot[3];
  pilot_accel(s->cg, eye, s->body_accel, s->pqridot, s->pqr_i, apilot);
  for (int i = 0; i < 3; i++) s->npilot[i] = apilot[i] / G0_FT;
  /* hoverbmac: (AGL - (Tb2l*RPBody).z)/b, terrain elevation 0 => AGL = geodetic altitude */
  const double rp[3] = {F16_AERORP_X, F16_AERORP_Y, F16_AERORP_Z};
  double rpb[3], mac[3];
  struct_to_body(s->cg, rp, rpb);
  mtv3(s->Tl2b, rpb, mac);
  s->h_b_mac = (s->h_geod - mac[2]) / F16_WINGSPAN;
}

/* ------------------------------------------------------------------ FGTurbine (S/models/propulsion/FGTurbine.cpp:107-270,400-411) */
static double seek(double v, double target, double accel, double decel, double dt) {
  if (v > target) { v -= dt * decel; if (v < target) v = target; }
  else if (v < target) { v += dt * accel; if (v > target) v = target; }
  return v;
}
/* FGSpoolUp / FGSimplifiedTSFC live in the stripped FGTurbine.h; restated from the published JSBSim 1.1.x header */
static double spool_rate(const F16State* s, double factor, double dens_ratio) {
  double delay = factor * 90.0 / (F16_ENG_BYPASSRATIO + 3.0);
  double n = fmin(1.0, s->n2norm + 0.1);
  return delay / (1 + 3 * (1 - n) * (1 - n) * (1 - n) + (1 - dens_ratio));
}
static double turbine_calculate(F16State* s, double dt) {
  const double N1f = F16_ENG_MAXN1 - F16_ENG_IDLEN1, N2f = F16_ENG_MAXN2 - F16_ENG_IDLEN2;
  const double idle_ff = pow(F16_ENG_MILTHRUST, 0.2) * 107.0;
  double tp = s->throttle_pos, aug_cmd = 0.0;
  if (tp > 1.0) { aug_cmd = tp - 1.0; tp -= aug_cmd; }
  if (s->phase == TP_TRIM && dt > 0) {
    if (s->running && !s->starved) {
      s->phase = TP_RUN;
      s->n2 = F16_ENG_IDLEN2 + tp * N2f;
      s->n1 = F16_ENG_IDLEN1 + tp * N1f;
      s->cutoff = 0;
    } else { s->phase = TP_OFF; s->cutoff = 1; }
  }
  if (s->qbar > 30.0) { if (!s->running && !s->cutoff && s->n2 > 15.0) s->phase = TP_START; }
  if (s->cutoff && s->phase != TP_SPINUP) s->phase = TP_OFF;
  if (dt == 0) s->phase = TP_TRIM;
  if (s->starved) s->phase = TP_OFF;
  double idle = F16_ENG_MILTHRUST * TAB2(ENG_IDLETHRUST, s->mach, s->density_alt);
  double mil = (F16_ENG_MILTHRUST - idle) * TAB2(ENG_MILTHRUST, s->mach, s->density_alt);
  double dens_ratio = s->rho / (ATM_SLP / (atm_reng() * 518.67));
  double thrust = 0.0;
  switch (s->phase) {
    case TP_RUN: {
      s->running = 1;
      s->n2 = seek(s->n2, F16_ENG_IDLEN2 + tp * N2f, spool_rate(s, 1.0, dens_ratio), spool_rate(s, 3.0, dens_ratio), dt);
      s->n1 = seek(s->n1, F16_ENG_IDLEN1 + tp * N1f, spool_rate(s, 1.0, dens_ratio), spool_rate(s, 2.4, dens_ratio), dt);
      s->n2norm = (s->n2 - F16_ENG_IDLEN2) / N2f;
      thrust = idle + mil * s->n2norm * s->n2norm;
      if (!s->augmentation) {
        double tsfc = F16_ENG_TSFC * sqrt(s->T / 389.7) * (0.84 + (1 - s->n2norm) * (1 - s->n2norm));
        s->fuelflow_pph = seek(s->fuelflow_pph, thrust * tsfc, 1000.0, 10000.0, dt);
        if (s->fuelflow_pph < idle_ff) s->fuelflow_pph = idle_ff;
      }
      /* AugMethod 2 (:242-252) */
      if (aug_cmd > 0.0) {
        s->augmentation = 1;
        double tdiff = F16_ENG_MAXTHRUST * TAB2(ENG_AUGTHRUST, s->mach, s->density_alt) - thrust;
        thrust += tdiff * aug_cmd;
        s->fuelflow_pph = seek(s->fuelflow_pph, thrust * F16_ENG_ATSFC, 5000.0, 10000.0, dt);
      } else s->augmentation = 0;
      if (s->cutoff) s->phase = TP_OFF;
      if (s->starved) s->phase = TP_OFF;
    } break;
    case TP_TRIM: {
      double n2 = F16_ENG_IDLEN2 + tp * N2f, n2n = (n2 - F16_ENG_IDLEN2) / N2f;
      thrust = idle + mil * n2n * n2n;
      if (aug_cmd > 0.0) thrust += (F16_ENG_MAXTHRUST * TAB2(ENG_AUGTHRUST, s->mach, s->density_alt) - thrust) * aug_cmd;
    } break;
    case TP_START: { /* :290-316; Starter is never set, so this needs qbar > 30 */
      if (s->n2 > 15.0 && !s->starved) {
        if (s->n2 < F16_ENG_IDLEN2) {
          s->n2 = seek(s->n2, F16_ENG_IDLEN2, 2.0, s->n2 / 2.0, dt);
          s->n1 = seek(s->n1, F16_ENG_IDLEN1, 1.4, s->n1 / 2.0, dt);
          s->fuelflow_pph = idle_ff * s->n2 / F16_ENG_IDLEN2;
          if (s->qbar < 30.0) s->phase = TP_OFF;
        } else { s->phase = TP_RUN; s->running = 1; }
      } else s->phase = TP_OFF;
    } break;
    default: { /* Off(), :178-194 */
      s->running = 0;
      s->fuelflow_pph = seek(s->fuelflow_pph, 0, 1000.0, 10000.0, dt);
      s->n1 = seek(s->n1, s->qbar / 10.0, s->n1 / 2.0 + 0.1, s->n1 / 2.0, dt);
      s->n2 = seek(s->n2, s->qbar / 15.0, s->n2 / 2.0 + 0.1, s->n2 / 2.0, dt);
      s->augmentation = 0;
    } break;
  }
  s->thrust = thrust;
  return thrust;
}
/* test hook: ONE FGTurbine::Run pass in phase Run from (n1, n2, n2norm) with the throttle position and density ratio given,
 * through the very turbine_calculate() the tick uses. Pinned by the reference's TestTurbine.py (seek, default spool-up law,
 * N1 / N2 spool-down factors 2.4 / 3.0). io = {n1, n2, n2norm} in and out. */
void f16_test_turbine_run(double* io, double throttle_pos, double sigma, double dt) {
  F16State s;
  memset(&s, 0, sizeof s);
  s.phase = TP_RUN; s.running = 1;
  s.n1 = io[0]; s.n2 = io[1]; s.n2norm = io[2];
  s.throttle_pos = throttle_pos;
  s.T = 518.67; s.rho = sigma * (ATM_SLP / (atm_reng() * 518.67));
  s.tank[0] = s.tank[1] = 1000.0;
  turbine_calculate(&s, dt);
  io[0] = s.n1; io[1] = s.n2; io[2] = s.n2norm;
}
/* moment about the CG of a body-axis force acting at a structural location (FGForce::GetBodyForces, S/models/propulsion/FGForce.cpp) */
static void force_moment_about_cg(const double cg[3], const double loc[3], const double f[3], double m[3]) {
  double r[3];
  struct_to_body(cg, loc, r);
  cross3(r, f, m);
}
/* FGPropulsion::Run + ConsumeFuel (S/models/FGPropulsion.cpp:113-258), FGTank::Drain (S/models/propulsion/FGTank.cpp:281-294) */
static void propulsion_run(F16State* s, double dt) {
  double thrust = turbine_calculate(s, dt);
  int with_fuel = 0;
  for (int i = 0; i < 4; i++) if (s->tank[i] > 0.0) with_fuel++;
  s->starved = (with_fuel == 0);
  if (!s->starved) {
    double need = s->fuelflow_pph / 3600.0 * dt / with_fuel;
    for (int i = 0; i < 4; i++)
      if (s->tank[i] > 0.0) {
        if (s->tank[i] - need >= 0.0) s->tank[i] -= need; else s->tank[i] = 0.0;
      }
  }
  /* direct thruster along body x acting at the structural origin (f16.xml:259-270; FGForce.cpp GetBodyForces) */
  const double loc[3] = {F16_THRUSTER_X, F16_THRUSTER_Y, F16_THRUSTER_Z};
  s->f_prop[0] = thrust; s->f_prop[1] = 0; s->f_prop[2] = 0;
  force_moment_about_cg(s->cg, loc, s->f_prop, s->m_prop);
}
/* test hook: the moment about the CG (structural inches) of a body-axis force applied at the F-16 thruster's location, through the
 * function propulsion_run uses. Pinned by the reference's CheckMomentsUpdate.py:66-76 (a force at the structural origin). */
void f16_test_thruster_moment(const double* cg, const double* force3, double* m3) {
  const double loc[3] = {F16_THRUSTER_X, F16_THRUSTER_Y, F16_THRUSTER_Z};
  force_moment_about_cg(cg, loc, force3, m3);
}

/* ------------------------------------------------------------------ FGAerodynamics::Run (S/models/FGAerodynamics.cpp:132-300), f16.xml:994-1925 */
/* the six axis sums (DRAG, SIDE, LIFT wind axes; ROLL, PITCH, YAW body axes), each function = product of its properties and table */
static void aero_axis_sums(const F16State* s, double* o) {
  const double Sw = F16_WINGAREA, bw = F16_WINGSPAN, cbar = F16_CHORD;
  double twovel = 2 * s->vt, bi2vel = 0.0, ci2vel = 0.0;
  if (twovel != 0) { bi2vel = bw / twovel; ci2vel = cbar / twovel; }
  const double qS = s->qbar * Sw, a = s->alpha, b = s->beta, M = s->mach;
  const double p = s->aero_pqr[0], q = s->aero_pqr[1], r = s->aero_pqr[2];
  const double de = s->elevator_pos_rad, da = s->aileron_pos_rad, dr = s->rudder_pos_rad;
  const double lef = s->lef_pos_rad, fl = s->flaperon_mix_rad, sbk = s->speedbrake_pos_rad, gear = s->gear_pos_norm;
  const double kge = TAB1(KCLGE, s->h_b_mac);
  double D = 0, Y = 0, L = 0, l = 0, m = 0, n = 0;
  D += qS * TAB2(CDDH, a, de);
  D += qS * TAB1(CDMACH, M);
  D += qS * lef * TAB1(CDDLEF, a);
  D += qS * fl * F16_K_CDDFLAPS;
  D += qS * gear * F16_K_CDGEAR;
  D += qS * sbk * TAB1(CDDSB, a);
  D += qS * q * ci2vel * TAB1(CDQ, a);
  D += qS * q * ci2vel * lef * TAB1(CDQ_DLEF, a);
  Y += qS * b * F16_K_CYB;
  Y += qS * b * TAB1(CYB_M, M);
  Y += qS * da * F16_K_CYDA;
  Y += qS * dr * F16_K_CYDR;
  Y += qS * bi2vel * p * TAB1(CYP, a);
  Y += qS * bi2vel * r * TAB1(CYR, a);
  L += qS * kge * TAB2(CLDH, a, de);
  L += qS * lef * kge * TAB1(CLDLEF, a);
  L += qS * fl * kge * F16_K_CLDFLAPS;
  L += qS * kge * sbk * TAB1(CLDSB, a);
  L += qS * q * kge * ci2vel * TAB1(CLQ, a);
  L += qS * q * ci2vel * sbk * TAB1(CLQ_DSB, a);
  l += qS * bw * TAB2(CLB, a, b);
  l += qS * bw * b * TAB1(CLB_M, M);
  l += qS * bw * bi2vel * p * TAB1(CLP, a);
  l += qS * bw * bi2vel * r * TAB1(CLR, a);
  l += qS * bw * da * TAB2(CLDA, a, b);
  l += qS * bw * a * da * TAB1(CLDA_M, M);
  l += qS * bw * a * dr * TAB1(CLDR_M, M);
  l += qS * bw * dr * TAB2(CLDR, a, b);
  m += qS * cbar * TAB2(CMDH, a, de);
  m += qS * cbar * a * TAB1(CMA_M, M);
  m += qS * cbar * sbk * TAB1(CMDSB, a);
  m += qS * cbar * ci2vel * q * TAB1(CMQ, a);
  n += qS * bw * TAB2(CNB, a, b);
  n += qS * bw * b * TAB1(CNB_M, M);
  n += qS * bw * bi2vel * p * TAB1(CNP, a);
  n += qS * bw * bi2vel * r * TAB1(CNR, a);
  n += qS * bw * da * TAB1(CNDA_M, M);
  n += qS * bw * da * TAB2(CNDA, a, b);
  n += qS * bw * dr * TAB2(CNDR, a, b);
  n += qS * bw * a * dr * TAB1(CNDR_M, M);
  o[0] = D; o[1] = Y; o[2] = L; o[3] = l; o[4] = m; o[5] = n;
}
/* the axis sums -> body-axis force and moment about the CG: DRAG / SIDE / LIFT are wind-axis natives with drag and lift sign-flipped,
 * rotated by Tw2b (S/models/FGAerodynamics.cpp:205-216); the moments are taken at the aerodynamic reference point and moved to the CG
 * with r x F (:280) */
static void aero_frame(const double cg[3], const double rp[3], const double Tw2b[9], const double o[6], double f_aero[3], double m_aero[3]) {
  const double D = o[0], Y = o[1], L = o[2], l = o[3], m = o[4], n = o[5];
  double fw[3] = {-D, Y, -L};
  mv3(Tw2b, fw, f_aero);
  double rpb[3], mx[3];
  struct_to_body(cg, rp, rpb);
  cross3(rpb, f_aero, mx);
  m_aero[0] = l + mx[0]; m_aero[1] = m + mx[1]; m_aero[2] = n + mx[2];
}
static void aerodynamics_run(F16State* s) {
  double o[6];
  aero_axis_sums(s, o);
  const double rp[3] = {F16_AERORP_X, F16_AERORP_Y, F16_AERORP_Z};
  aero_frame(s->cg, rp, s->Tw2b, o, s->f_aero, s->m_aero);
}
/* test hook: aerodynamics_run's frame handling for given (alpha, beta), CG and reference point (structural inches) and the six axis
 * sums, through the very wind_to_body() / aero_frame() the tick uses. Pinned by the reference's TestAeroFuncFrame.py testAeroFrame. */
void f16_test_aero_frame(double alpha, double beta, const double* cg, const double* rp, const double* sums6, double* f3, double* m3) {
  double Tw2b[9];
  wind_to_body(alpha, beta, Tw2b);
  aero_frame(cg, rp, Tw2b, sums6, f3, m3);
}

/* test hook: the axis sums for given property values, in the order of tests/golden/make_f16_aero_check.py's PROPS (alpha, beta,
 * mach, qbar, bi2vel, ci2vel, p, q, r aero, elevator, aileron, rudder, lef, flaperon-mix, speedbrake [rad], gear, h/b) */
void f16_test_aero_sums(const double* in, double* out6) {
  F16State s;
  memset(&s, 0, sizeof s);
  s.alpha = in[0]; s.beta = in[1]; s.mach = in[2]; s.qbar = in[3];
  /* aerodynamics_run derives bi2vel = b / 2Vt and ci2vel = cbar / 2Vt from vt: hand it the vt that gives in[4], and check in[5] agrees */
  s.vt = F16_WINGSPAN / (2.0 * in[4]);
  s.aero_pqr[0] = in[6]; s.aero_pqr[1] = in[7]; s.aero_pqr[2] = in[8];
  s.elevator_pos_rad = in[9]; s.aileron_pos_rad = in[10]; s.rudder_pos_rad = in[11]; s.lef_pos_rad = in[12];
  s.flaperon_mix_rad = in[13]; s.speedbrake_pos_rad = in[14]; s.gear_pos_norm = in[15]; s.h_b_mac = in[16];
  aero_axis_sums(&s, out6);
}
/* ------------------------------------------------------------------ FGAccelerations::Run (S/models/FGAccelerations.cpp:138-208) */
static void accelerations_run(F16State* s) {
  double F[3], Mo[3];
  for (int i = 0; i < 3; i++) { F[i] = s->f_aero[i] + s->f_prop[i]; Mo[i] = s->m_aero[i] + s->m_prop[i]; }
  double Jw[3], wJw[3], rhs[3];
  mv3(s->J, s->pqr_i, Jw);
  cross3(s->pqr_i, Jw, wJw);
  for (int i = 0; i < 3; i++) rhs[i] = Mo[i] - wJw[i];
  mv3(s->Jinv, rhs, s->pqridot);
  for (int i = 0; i < 3; i++) s->body_accel[i] = F[i] / s->mass;
  const double om[3] = {0, 0, OMEGA_E};
  double omb[3], w2[3], c1[3], oxr[3], ooxr[3], t[3], gb[3];
  mv3(s->Ti2b, om, omb);
  for (int i = 0; i < 3; i++) w2[i] = s->pqr[i] + 2.0 * omb[i];
  cross3(w2, s->uvw, c1);
  cross3(om, s->r_eci, oxr);
  cross3(om, oxr, ooxr);
  mv3(s->Ti2b, ooxr, t);
  mv3(s->Tec2b, s->grav_ecef, gb);
  for (int i = 0; i < 3; i++) s->uvwdot[i] = s->body_accel[i] - c1[i] - t[i] + gb[i];
  double bi[3], gi[3];
  mtv3(s->Ti2b, s->body_accel, bi);       /* Tb2i * a */
  mtv3(s->Ti2ec, s->grav_ecef, gi);       /* Tec2i * g */
  for (int i = 0; i < 3; i++) s->uvwidot[i] = bi[i] + gi[i];
}

/* ------------------------------------------------------------------ one executive tick */
void f16_tick(F16State* s, double dt) {
  if (dt > 0.0) { s->sim_time += dt; s->ticks++; } /* IncrTime, FGFDMExec.cpp:196-203 */
  propagate_run(s, dt);
  inertial_run(s);
  f16_atmosphere(s->h_sl, &s->T, &s->P, &s->rho, &s->snd, &s->density_alt);
  fcs_run(s);
  massbalance_run(s);
  auxiliary_run(s);
  propulsion_run(s, dt);
  aerodynamics_run(s);
  accelerations_run(s);
}

void f16_refresh_derived(F16State* s) {
  s->sim_time = (double)s->ticks / 60.0;
  s->epa = OMEGA_E * s->sim_time;
  propagate_derived(s);
  massbalance_run(s); /* first call rebuilds the CG from the tanks, second the inertia about it */
  massbalance_run(s);
}

void f16_set_controls(F16State* s, double ail, double ele, double rud, double thr) {
  /* catalog.py:189-197 bounds applied by simulatior.py:307-311 */
  s->da_cmd = clampd(-1.0, ail, 1.0);
  s->de_cmd = clampd(-1.0, ele, 1.0);
  s->dr_cmd = clampd(-1.0, rud, 1.0);
  s->throttle_cmd = clampd(0.0, thr, 0.9);
}

void f16_default_init(F16Init* ic) {
  ic->lon_deg = 120.0; ic->lat_geod_deg = 60.0; ic->h_sl_ft = 20000; ic->psi_deg = 0.0;
  ic->u_fps = 800.0; ic->v_fps = 0; ic->w_fps = 0; ic->p = 0; ic->q = 0; ic->r = 0;
}

/* FGInitialCondition::SetAltitudeASLFtIC with lastLatitudeSet == setgeod (S/initialization/FGInitialCondition.cpp:749-823):
 * find the geodetic altitude whose radius minus the sea-level radius (at the resulting geocentric latitude) equals alt. */
static double geod_alt_from_asl(double geod_lat, double alt) {
  const double a = A_E, b = B_E, e2 = 1.0 - b * b / (a * a);
  double cg = cos(geod_lat), sg = sin(geod_lat), N = a / sqrt(1 - e2 * sg * sg);
  double n = e2, prev_n = 1.0;
  int iter = 0;
  if (cg > fabs(sg)) {
    double tg = sg / cg, x0 = N * e2 * cg, x = 0.0;
    while (fabs(n - prev_n) > 1E-15 && iter < 10) {
      double tl = (1 - n) * tg, c2 = 1. / (1. + tl * tl), slr = b / sqrt(1. - e2 * c2), R = slr + alt;
      x = R * sqrt(c2);
      prev_n = n; n = x0 / x; iter++;
    }
    return x / cg - N;
  }
  double ctg = cg / sg, z0 = N * e2 * sg, z = 0.0;
  while (fabs(n - prev_n) > 1E-15 && iter < 10) {
    double ctl = ctg / (1 - n), s2 = 1. / (1. + ctl * ctl), c2 = 1. - s2, slr = b / sqrt(1. - e2 * c2), R = slr + alt;
    z = R * sgn(ctl) * sqrt(s2);
    prev_n = n; n = z0 / (z0 + z); iter++;
  }
  return z / sg - N * (1 - e2);
}

/* AircraftSimulator.reload (envs/JSBSim/core/simulatior.py:152-190):
 *   load_model -> fresh FDM;  IC properties;  run_ic (FGFDMExec::RunIC, S/FGFDMExec.cpp:636-669: two executive passes with
 *   integration suspended, then InitializeDerivatives);  engine init_running (FGTurbine::InitRunning :604-616);
 *   propulsion.get_steady_state (FGPropulsion::GetSteadyState, S/models/FGPropulsion.cpp:262-310). */
void f16_reset(F16State* s, const F16Init* ic) {
  memset(s, 0, sizeof *s);
  atm_init();
  /* component initial values */
  s->gear_pos_norm = s->gear_cmd_norm = 1.0; /* FGFCS.cpp:81 */
  s->tank[0] = F16_TANK0_CONTENTS; s->tank[1] = F16_TANK1_CONTENTS; s->tank[2] = F16_TANK2_CONTENTS; s->tank[3] = F16_TANK3_CONTENTS;
  s->phase = TP_OFF; s->cutoff = 1; s->running = 0; /* FGTurbine::ResetToIC */
  /* FGMassBalance::Load leaves vXYZcg at (0,0,0) until the first Run (S/models/FGMassBalance.cpp:115-160); the tank-inertia
   * term loaded for the very first IC pass (FGFDMExec.cpp:572) is therefore taken about the structural origin. Kept as is:
   * that pass's angular acceleration feeds the pilot-station load factor the FCS sees on the first real tick. */
  /* ---- FGPropagate::SetInitialState (S/models/FGPropagate.cpp:150-190) */
  double lon = ic->lon_deg * DEGTORAD, lat = ic->lat_geod_deg * DEGTORAD, psi = ic->psi_deg * DEGTORAD;
  double hg = geod_alt_from_asl(lat, ic->h_sl_ft);
  set_position_geodetic(lon, lat, hg, s->r_ecef);
  s->epa = 0.0;
  memcpy(s->r_eci, s->r_ecef, sizeof s->r_eci);
  /* local->ECEF at this location */
  double lo, lgc, lgd, hgd, rad;
  f16_geodetic_from_ecef(s->r_ecef, &lo, &lgc, &lgd, &hgd, &rad);
  double rxy = sqrt(s->r_ecef[0] * s->r_ecef[0] + s->r_ecef[1] * s->r_ecef[1]);
  double sinLon = s->r_ecef[1] / rxy, cosLon = s->r_ecef[0] / rxy, sinLat = sin(lgd), cosLat = cos(lgd);
  double Tec2l[9] = {-cosLon * sinLat, -sinLon * sinLat, cosLat, -sinLon, cosLon, 0.0, -cosLon * cosLat, -sinLon * cosLat, -sinLat};
  double Ti2l[9], ql[4], qi2l[4];
  memcpy(Ti2l, Tec2l, sizeof Ti2l); /* Ti2ec = I at epa = 0 */
  quat_from_euler(0.0, 0.0, psi, ql);
  quat_normalize(ql);
  quat_from_T(Ti2l, qi2l);
  quat_mul(qi2l, ql, s->q_eci);   /* qAttitudeECI = Ti2l.GetQuaternion()*qAttitudeLocal */
  quat_to_T(s->q_eci, s->Ti2b);
  double uvw[3] = {ic->u_fps, ic->v_fps, ic->w_fps};
  const double om[3] = {0, 0, OMEGA_E};
  double omb[3], oxr[3], vi[3];
  mv3(s->Ti2b, om, omb);
  s->pqr_i[0] = ic->p + omb[0]; s->pqr_i[1] = ic->q + omb[1]; s->pqr_i[2] = ic->r + omb[2];
  mtv3(s->Ti2b, uvw, vi);
  cross3(om, s->r_eci, oxr);
  for (int i = 0; i < 3; i++) s->v_eci[i] = vi[i] + oxr[i];
  propagate_derived(s);
  /* ---- RunIC: Initialize() runs the executive once, RunIC runs it again, both with dT = 0 */
  f16_tick(s, 0.0);
  f16_tick(s, 0.0);
  /* InitializeDerivatives (S/models/FGPropagate.cpp:194-200) */
  for (int k = 0; k < 3; k++) memcpy(s->hist_v[k], s->v_eci, sizeof s->hist_v[0]);
  for (int k = 0; k < 2; k++) memcpy(s->hist_a[k], s->uvwidot, sizeof s->hist_a[0]);
  /* ---- engine.init_running(): ThrottlePos member is 0 at this point */
  s->cutoff = 0; s->running = 1;
  s->n2 = F16_ENG_IDLEN2; s->n1 = F16_ENG_IDLEN1;
  { double keep = s->throttle_pos; s->throttle_pos = 0.0; turbine_calculate(s, 0.0); s->throttle_pos = keep; }
  s->phase = TP_RUN;
  /* ---- propulsion.get_steady_state(): TotalDeltaT = 0.5, up to 6000 iterations, steady after 121 equal thrusts */
  {
    double last = -1.0, cur = 0.0;
    int steady = 0, cnt = 0, j = 0;
    while (!steady && j < 6000) {
      turbine_calculate(s, 0.5);
      last = cur; cur = s->thrust;
      if (fabs(last - cur) < 0.0001) { if (++cnt > 120) steady = 1; } else cnt = 0;
      j++;
    }
  }
  s->sim_time = 0.0; s->ticks = 0;
}
