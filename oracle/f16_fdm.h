/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU (plain C, float64) restatement of the reference's flight-dynamics path:
 * the JSBSim F-16 model as driven by envs/JSBSim/core/simulatior.py (AircraftSimulator).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 * The shipped product (aircombat-selfplay_amd/) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for the F-16 trajectory. The reference runs the third-party wheel jsbsim==1.1.6
 * (README.md:10); it is not installed here, its vendored sources under envs/JSBSim/data/src have no headers and cannot be
 * compiled, and the reference's tests hold no F-16 golden trajectory. This file restates the published algorithm from those
 * sources (file:line cited at each function) with the aircraft data of aircraft/f16/f16.xml. The generic blocks it is built from
 * ARE pinned by what the reference holds (tests/test_oracle_jsbsim_blocks.py): atmosphere, density / pressure altitude, the
 * <kinematic> and <pid> components, the turbine spool law, every table and the aerodynamic summation.
 */
#ifndef ORACLE_F16_FDM_H
#define ORACLE_F16_FDM_H

#ifdef __cplusplus
extern "C" {
#endif

/* turbine phases, FGTurbine.cpp:107-176 */
enum { TP_OFF = 0, TP_RUN = 1, TP_SPINUP = 2, TP_START = 3, TP_STALL = 4, TP_SEIZE = 5, TP_TRIM = 6 };

typedef struct {
  double in_prev, in_prev2, i_total, out;
} OrPid;

typedef struct {
  /* ---------------- FGPropagate state (FGPropagate.cpp:93-96,218-290) */
  double r_eci[3];      /* vInertialPosition [ft] */
  double v_eci[3];      /* vInertialVelocity [ft/s] */
  double q_eci[4];      /* qAttitudeECI */
  double pqr_i[3];      /* vPQRi [rad/s] */
  double epa;           /* earth position angle [rad] */
  double hist_v[3][3];  /* dqInertialVelocity[0..2] */
  double hist_a[2][3];  /* dqUVWidot[0..1] */
  /* derived each tick by Propagate */
  double r_ecef[3];
  double Ti2ec[9], Tec2l[9], Ti2b[9], Tl2b[9], Tec2b[9];
  double uvw[3], pqr[3], vel_ned[3], qdot[4];
  double phi, tht, psi;
  double lon, lat_gc, lat_geod, radius, h_sl, h_geod;
  /* ---------------- FGAccelerations outputs (consumed next tick) */
  double pqridot[3], uvwidot[3], uvwdot[3], body_accel[3];
  /* ---------------- FGInertial / FGStandardAtmosphere */
  double grav_ecef[3];
  double T, P, rho, snd, density_alt;
  /* ---------------- FGFCS (f16.xml:317-992) */
  double da_cmd, de_cmd, dr_cmd, throttle_cmd;
  OrPid pid_roll, pid_pitch, pid_yaw;
  double tef_control, left_aileron_pos_norm, elevator_pos_norm, rudder_pos_norm;
  double speedbrake_pos_deg, gear_pos_norm, gear_cmd_norm;
  double aileron_pos_rad, elevator_pos_rad, rudder_pos_rad, lef_pos_rad, flaperon_mix_rad;
  double speedbrake_pos_rad, throttle_pos;
  /* ---------------- FGAuxiliary outputs (FCS of the NEXT tick reads these) */
  double vt, alpha, beta, qbar, mach, vc_fps, vg;
  double aero_pqr[3];
  double npilot[3];
  double Tw2b[9];
  double h_b_mac;
  /* ---------------- FGMassBalance */
  double mass, weight, cg[3], last_cg[3], J[9], Jinv[9], tankJ[9];
  int have_last_cg;
  /* ---------------- FGPropulsion / FGTurbine / FGTank */
  double tank[4];
  double n1, n2, n2norm, fuelflow_pph, thrust;
  int phase, running, cutoff, starved, augmentation;
  /* ---------------- forces (body) */
  double f_aero[3], m_aero[3], f_prop[3], m_prop[3];
  /* ---------------- time */
  double sim_time;
  long ticks;
} F16State;

typedef struct {
  double lon_deg, lat_geod_deg, h_sl_ft, psi_deg, u_fps, v_fps, w_fps, p, q, r;
} F16Init;

void f16_default_init(F16Init* ic);                 /* simulatior.py:192-208 */
void f16_reset(F16State* s, const F16Init* ic);     /* simulatior.py:152-190 */
void f16_set_controls(F16State* s, double aileron, double elevator, double rudder, double throttle); /* simulatior.py:299-319 + catalog.py bounds */
void f16_tick(F16State* s, double dt);              /* FGFDMExec::Run, FGFDMExec.cpp:407-431; dt = 0 => integration suspended */

/* after overwriting the integrator state by hand (tests): rebuild every derived quantity the next tick reads */
void f16_refresh_derived(F16State* s);

/* table helpers exposed for unit tests */
double f16_tab1(int off, int nr, double key);
double f16_tab2(int off, int nr, int nc, double rkey, double ckey);
void f16_atmosphere(double h_ft, double* T, double* P, double* rho, double* snd, double* dens_alt);
/* the same with atmosphere/delta-T [R] (tests only: pins the layer formulas on the reference's TestDensityAltitude / TestPressureAltitude tables) */
void f16_atmosphere_bias(double h_ft, double bias_R, double* T, double* P, double* rho, double* snd, double* dens_alt, double* press_alt);
double f16_vcas_from_mach(double mach, double p);
double f16_pitot_qc(double mach, double p);   /* impact pressure pt - p */
double f16_vcas_from_qc(double qc);
void f16_geodetic_from_ecef(const double r[3], double* lon, double* lat_gc, double* lat_geod, double* h_geod, double* radius);
/* one FGTurbine::Run pass in phase Run (tests only); io = {n1, n2, n2norm} */
void f16_test_turbine_run(double* io, double throttle_pos, double sigma, double dt);
/* FGAerodynamics axis sums for given property values (tests only; order: tests/golden/make_f16_aero_check.py PROPS) */
void f16_test_aero_sums(const double* in17, double* out6);
/* one FGPID::Run at a given dt (tests only); st = {in_prev, in_prev2, i_total, out} */
double f16_test_pid(double* st, double in, double trigger, double kp, double ki, double kd, double dt);
double f16_kinemat(double out, double in, const double* detents, const double* times, int n, double dt);
/* one pass of the <flight_control> section / FGMassBalance::Run / the pilot-station acceleration with everything they read passed in
 * (tests only: tests/test_oracle_f16_wiring.py holds them to an independent generic reading of f16.xml) */
void f16_test_fcs(double* st18, const double* in17, double* out16);
void f16_test_massbalance(const double* tanks4, const double* pm2, const double* cg_tanks, double* out31);
void f16_test_pilot_accel(const double* cg, const double* eye, const double* body_accel, const double* pqridot, const double* pqri, double* out3);
void f16_test_aero_frame(double alpha, double beta, const double* cg, const double* rp, const double* sums6, double* f3, double* m3);
void f16_test_thruster_moment(const double* cg, const double* force3, double* m3);

#ifdef __cplusplus
}
#endif
#endif
