/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * Test glue: exports / imports the oracle's per-aircraft state in the product's documented state-vector
 * layout (the names returned by ac_state_field_name of include/aircombat.h), so that parity tests can
 * teacher-force both implementations from the same state. Nothing here is part of the reference's algorithm. */
#include "combat_env.h"
#include <math.h>
#include <string.h>

#define KTSTOFPS 1.68781
size_t or_env_sizeof(void) { return sizeof(OrEnv); }
size_t or_env_config_sizeof(void) { return sizeof(OrEnvConfig); }

int or_state_export(const OrEnv* e, int i, double* out, int n) {
  const OrAircraft* a = &e->ac[i];
  const F16State* s = &a->fdm;
  double v[96];
  int k = 0;
  for (int j = 0; j < 3; j++) v[k++] = s->r_eci[j];
  for (int j = 0; j < 3; j++) v[k++] = s->v_eci[j];
  for (int j = 0; j < 4; j++) v[k++] = s->q_eci[j];
  for (int j = 0; j < 3; j++) v[k++] = s->pqr_i[j];
  for (int j = 0; j < 3; j++) v[k++] = s->hist_v[0][j];
  for (int j = 0; j < 3; j++) v[k++] = s->hist_v[1][j];
  for (int j = 0; j < 3; j++) v[k++] = s->hist_a[0][j];
  for (int j = 0; j < 3; j++) v[k++] = s->pqridot[j];
  for (int j = 0; j < 3; j++) v[k++] = s->uvwidot[j];
  for (int j = 0; j < 3; j++) v[k++] = s->body_accel[j];
  v[k++] = s->da_cmd; v[k++] = s->de_cmd; v[k++] = s->dr_cmd; v[k++] = s->throttle_cmd;
  v[k++] = s->pid_roll.in_prev; v[k++] = s->pid_pitch.in_prev; v[k++] = s->pid_yaw.in_prev;
  v[k++] = s->pid_roll.i_total; v[k++] = s->pid_pitch.i_total; v[k++] = s->pid_yaw.i_total;
  v[k++] = s->tef_control; v[k++] = s->left_aileron_pos_norm; v[k++] = s->elevator_pos_norm; v[k++] = s->speedbrake_pos_deg;
  v[k++] = s->alpha; v[k++] = s->mach; v[k++] = (fabs(s->mach) > 0.0) ? f16_pitot_qc(s->mach, s->P) : 0.0; v[k++] = s->vg;
  for (int j = 0; j < 3; j++) v[k++] = s->aero_pqr[j];
  for (int j = 0; j < 3; j++) v[k++] = s->npilot[j];
  v[k++] = s->n1; v[k++] = s->n2; v[k++] = s->n2norm; v[k++] = s->fuelflow_pph;
  v[k++] = s->tank[0]; v[k++] = s->tank[1];
  /* task floats */
  v[k++] = a->bloods; v[k++] = a->pre_posture; v[k++] = a->pre_altitude; v[k++] = a->pre_event; v[k++] = a->pre_shoot;
  /* ints */
  v[k++] = (double)(s->phase | (s->running ? 8 : 0) | (s->cutoff ? 16 : 0) | (s->starved ? 32 : 0) | (s->augmentation ? 64 : 0));
  v[k++] = (double)s->ticks;
  v[k++] = a->status; v[k++] = a->die_flag; v[k++] = a->remaining_missiles; v[k++] = a->pre_remaining_missiles;
  v[k++] = a->shoot_action; v[k++] = a->last_missile; v[k++] = a->last_shoot_time;
  int bits = 0; for (int j = 0; j < a->lock_n && j < 16; j++) bits |= (a->lock_window[j] ? 1 : 0) << j;
  v[k++] = bits; v[k++] = a->lock_pos; v[k++] = e->current_step;
  for (int j = 0; j < n; j++) out[j] = j < k ? v[j] : 0.0;
  return k;
}

int or_state_import(OrEnv* e, int i, const double* v) {
  OrAircraft* a = &e->ac[i];
  F16State* s = &a->fdm;
  int k = 0;
  for (int j = 0; j < 3; j++) s->r_eci[j] = v[k++];
  for (int j = 0; j < 3; j++) s->v_eci[j] = v[k++];
  for (int j = 0; j < 4; j++) s->q_eci[j] = v[k++];
  for (int j = 0; j < 3; j++) s->pqr_i[j] = v[k++];
  for (int j = 0; j < 3; j++) s->hist_v[0][j] = v[k++];
  for (int j = 0; j < 3; j++) s->hist_v[1][j] = v[k++];
  for (int j = 0; j < 3; j++) s->hist_a[0][j] = v[k++];
  for (int j = 0; j < 3; j++) s->pqridot[j] = v[k++];
  for (int j = 0; j < 3; j++) s->uvwidot[j] = v[k++];
  for (int j = 0; j < 3; j++) s->body_accel[j] = v[k++];
  s->da_cmd = v[k++]; s->de_cmd = v[k++]; s->dr_cmd = v[k++]; s->throttle_cmd = v[k++];
  s->pid_roll.in_prev = v[k++]; s->pid_pitch.in_prev = v[k++]; s->pid_yaw.in_prev = v[k++];
  s->pid_roll.i_total = v[k++]; s->pid_pitch.i_total = v[k++]; s->pid_yaw.i_total = v[k++];
  s->tef_control = v[k++]; s->left_aileron_pos_norm = v[k++]; s->elevator_pos_norm = v[k++]; s->speedbrake_pos_deg = v[k++];
  s->alpha = v[k++]; s->mach = v[k++]; s->vc_fps = f16_vcas_from_qc(v[k++]); s->vg = v[k++];
  for (int j = 0; j < 3; j++) s->aero_pqr[j] = v[k++];
  for (int j = 0; j < 3; j++) s->npilot[j] = v[k++];
  s->n1 = v[k++]; s->n2 = v[k++]; s->n2norm = v[k++]; s->fuelflow_pph = v[k++];
  s->tank[0] = v[k++]; s->tank[1] = v[k++];
  a->bloods = v[k++]; a->pre_posture = v[k++]; a->pre_altitude = v[k++]; a->pre_event = v[k++]; a->pre_shoot = v[k++];
  int eng = (int)llround(v[k++]);
  s->phase = eng & 7; s->running = (eng >> 3) & 1; s->cutoff = (eng >> 4) & 1; s->starved = (eng >> 5) & 1; s->augmentation = (eng >> 6) & 1;
  s->ticks = (long)llround(v[k++]);
  a->status = (int)llround(v[k++]); a->die_flag = (int)llround(v[k++]); a->remaining_missiles = (int)llround(v[k++]);
  a->pre_remaining_missiles = (int)llround(v[k++]); a->shoot_action = (int)llround(v[k++]); a->last_missile = (int)llround(v[k++]);
  a->last_shoot_time = (int)llround(v[k++]);
  int bits = (int)llround(v[k++]); for (int j = 0; j < 16; j++) a->lock_window[j] = (bits >> j) & 1;
  a->lock_pos = (int)llround(v[k++]); e->current_step = (int)llround(v[k++]);
  f16_refresh_derived(s);
  /* refresh the wrapper's cached pose (simulatior.py:238-258) */
  or_env_refresh_cache(e, i);
  return k;
}

/* flat views for the Python harness */
void or_env_get_pose(const OrEnv* e, int i, double out[12]) {
  const OrAircraft* a = &e->ac[i];
  out[0] = a->geodetic[0]; out[1] = a->geodetic[1]; out[2] = a->geodetic[2];
  out[3] = a->posture[0]; out[4] = a->posture[1]; out[5] = a->posture[2];
  out[6] = a->velocity[0]; out[7] = a->velocity[1]; out[8] = a->velocity[2];
  out[9] = a->position[0]; out[10] = a->position[1]; out[11] = a->position[2];
}
int or_env_num_missiles(const OrEnv* e) { return e->n_msl; }
void or_env_get_missile(const OrEnv* e, int k, double out[14]) {
  const OrMissile* m = &e->msl[k];
  out[0] = m->status; out[1] = m->position[0]; out[2] = m->position[1]; out[3] = m->position[2];
  out[4] = m->velocity[0]; out[5] = m->velocity[1]; out[6] = m->velocity[2];
  out[7] = m->posture[1]; out[8] = m->posture[2]; out[9] = m->t; out[10] = m->m; out[11] = m->parent; out[12] = m->target; out[13] = m->geodetic[2];
}
int or_env_status(const OrEnv* e, int i) { return e->ac[i].status; }
void or_env_set_status(OrEnv* e, int i, int status) { e->ac[i].status = status; }
double or_env_bloods(const OrEnv* e, int i) { return e->ac[i].bloods; }
void or_env_set_bloods(OrEnv* e, int i, double b) { e->ac[i].bloods = b; }
const F16State* or_env_fdm(const OrEnv* e, int i) { return &e->ac[i].fdm; }

/* ---- golden-vector hooks: put an aircraft in a synthetic pose (what FakeAircraft.set_pose of tests/golden/make_golden.py
 * gives the reference code) without running the FDM. pose = lon, lat, alt_m, roll, pitch, yaw, vN, vE, vDown [m/s],
 * u, v, w [m/s], vc [m/s], npilot x y z, sim_time, status, bloods, extreme_flag */
void or_env_set_pose(OrEnv* e, int i, const double* p) {
  OrAircraft* a = &e->ac[i];
  F16State* s = &a->fdm;
  const double M2FT = 1.0 / 0.3048, D2R = M_PI / 180.0;
  s->lon = p[0] * D2R; s->lat_geod = p[1] * D2R; s->h_sl = p[2] * M2FT;
  s->phi = p[3]; s->tht = p[4]; s->psi = p[5];
  s->vel_ned[0] = p[6] * M2FT; s->vel_ned[1] = p[7] * M2FT; s->vel_ned[2] = p[8] * M2FT;
  s->uvw[0] = p[9] * M2FT; s->uvw[1] = p[10] * M2FT; s->uvw[2] = p[11] * M2FT;
  s->vc_fps = p[12] * M2FT;
  s->npilot[0] = p[13]; s->npilot[1] = p[14]; s->npilot[2] = p[15];
  s->sim_time = p[16];
  a->status = (int)p[17]; a->bloods = p[18];
  if (p[19] != 0.0) s->npilot[0] = 11.0; /* stands for detect/extreme-state = 1 supplied directly to the reference */
  s->pqr[0] = s->pqr[1] = s->pqr[2] = 0; s->v_eci[0] = 300; s->v_eci[1] = s->v_eci[2] = 0;
  or_env_refresh_cache(e, i);
}
void or_env_set_step(OrEnv* e, int step) { e->current_step = step; }
int or_env_add_missile(OrEnv* e, int parent, int target, int model, const double pos[3], const double vel[3]) {
  if (e->n_msl >= OR_MAX_MSL) return -1;
  OrMissile* m = &e->msl[e->n_msl];
  or_missile_init(m, model, 1.0 / e->cfg.sim_freq);
  or_missile_launch(m, e->ac[parent].geodetic, e->ac[parent].position, e->ac[parent].velocity, e->ac[parent].posture);
  m->parent = parent; m->target = target;
  for (int k = 0; k < 3; k++) { m->position[k] = pos[k]; m->velocity[k] = vel[k]; }
  return e->n_msl++;
}
void or_env_clear_missiles(OrEnv* e) { e->n_msl = 0; }
/* heading-task injection: targets and the quantities HeadingTask / HeadingReward / UnreachHeading read */
void or_env_heading_targets(OrEnv* e, double hdg_deg, double alt_ft, double u_mps, double check_time) {
  e->ac[0].target_heading_deg = hdg_deg; e->ac[0].target_altitude_ft = alt_ft; e->ac[0].target_velocities_u_mps = u_mps;
  e->ac[0].heading_check_time = check_time; e->heading_turn_counts = 0;
}
void or_env_heading_pose(OrEnv* e, double psi_deg, double h_ft, double u_mps, double roll, double pitch, double v_mps, double w_mps,
                         double vc_mps, double p, double q, double sim_time) {
  F16State* s = &e->ac[0].fdm;
  const double M2FT = 1.0 / 0.3048;
  s->psi = psi_deg * M_PI / 180.0; s->h_sl = h_ft; s->uvw[0] = u_mps * M2FT; s->uvw[1] = v_mps * M2FT; s->uvw[2] = w_mps * M2FT;
  s->phi = roll; s->tht = pitch; s->vc_fps = vc_mps * M2FT; s->pqr[0] = p; s->pqr[1] = q; s->pqr[2] = 0; s->sim_time = sim_time;
  s->npilot[0] = s->npilot[1] = 0; s->npilot[2] = -1; s->v_eci[0] = 300; s->v_eci[1] = s->v_eci[2] = 0;
  s->lon = 120.0 * M_PI / 180.0; s->lat_geod = 60.0 * M_PI / 180.0; s->vel_ned[0] = s->vel_ned[1] = s->vel_ned[2] = 0;
  or_env_refresh_cache(e, 0);
}
/* vertical speed (m/s, positive down) of the scripted pose: AltitudeReward reads it (altitude_reward.py:30) */
void or_env_heading_vdown(OrEnv* e, double vd_mps) {
  e->ac[0].fdm.vel_ned[2] = vd_mps / 0.3048;
  or_env_refresh_cache(e, 0);
}
void or_env_heading_get(const OrEnv* e, double out[5]) {
  out[0] = e->ac[0].target_heading_deg; out[1] = e->ac[0].target_altitude_ft; out[2] = e->ac[0].target_velocities_u_mps;
  out[3] = e->ac[0].heading_check_time; out[4] = e->heading_turn_counts;
}
void or_missile_raw_run(double* st /* [20] */, int model, const double tp[3], const double tv[3], int alive) {
  /* flat missile state for the fly-out golden test: status, pos3, vel3, theta, psi, t, m, dtheta, dphi, dist_prev, recede_count, alt */
  OrMissile m;
  or_missile_init(&m, model, 1.0 / 60.0);
  m.status = (int)st[0];
  for (int k = 0; k < 3; k++) { m.position[k] = st[1 + k]; m.velocity[k] = st[4 + k]; }
  m.posture[1] = st[7]; m.posture[2] = st[8]; m.t = st[9]; m.m = st[10]; m.dtheta = st[11]; m.dphi = st[12]; m.dist_prev = st[13];
  m.recede_count = (int)st[14]; m.geodetic[2] = st[15];
  or_missile_run(&m, tp, tv, alive, 1.0 / 60.0, 120.0, 60.0, 0.0);
  st[0] = m.status;
  for (int k = 0; k < 3; k++) { st[1 + k] = m.position[k]; st[4 + k] = m.velocity[k]; }
  st[7] = m.posture[1]; st[8] = m.posture[2]; st[9] = m.t; st[10] = m.m; st[11] = m.dtheta; st[12] = m.dphi; st[13] = m.dist_prev;
  st[14] = m.recede_count; st[15] = m.geodetic[2];
}

/* ---- scenario-task golden hooks */
void or_env_set_shoot4(OrEnv* e, int i, const int bits[4]) { for (int k = 0; k < 4; k++) e->ac[i].shoot4[k] = bits[k]; }
void or_env_task_step(OrEnv* e);             /* combat_env.c */
void or_env_run_projectiles(OrEnv* e, int substeps);
void or_env_get_counters(const OrEnv* e, int i, double out[6]) {
  const OrAircraft* a = &e->ac[i];
  out[0] = a->rem_gun; out[1] = a->rem_9m; out[2] = a->rem_120b; out[3] = a->rem_chaff; out[4] = a->bloods; out[5] = a->status;
}
void or_env_get_misc(const OrEnv* e, double out[3]) { out[0] = e->n_chaff; out[1] = (double)e->chaff_draws; out[2] = e->current_step; }

/* hierarchical tasks: the controller's GRU state and its last output for aircraft i */
void or_env_get_rnn(const OrEnv* e, int i, double* h, int* low_action) {
  for (int k = 0; k < 128; k++) h[k] = e->ac[i].rnn[k];
  if (low_action) for (int k = 0; k < 4; k++) low_action[k] = e->ac[i].low_action[k];
}
void or_env_set_rnn(OrEnv* e, int i, const double* h) { for (int k = 0; k < 128; k++) e->ac[i].rnn[k] = h[k]; }
void or_env_get_ctl_gaps(const OrEnv* e, int i, double* gaps4) { for (int k = 0; k < 4; k++) gaps4[k] = e->ac[i].ctl_gap[k]; }

/* scripted-opponent golden hooks: delta values and the 12 controller inputs of aircraft i chasing aircraft j / flying the schedule */
void or_pursue_delta(const OrAircraft* ego, const OrAircraft* tgt, double dv[3]);
void or_maneuver_delta(OrAircraft* a, double turn_interval, double time_interval, double dv[3]);
void or_baseline_observation(const OrAircraft* a, const double dv[3], double x[12]);
void or_env_pursue(OrEnv* e, int i, int j, double* dv, double* x) { or_pursue_delta(&e->ac[i], &e->ac[j], dv); or_baseline_observation(&e->ac[i], dv, x); }
void or_env_maneuver(OrEnv* e, int i, double turn_interval, double time_interval, double* dv, double* x) {
  or_maneuver_delta(&e->ac[i], turn_interval, time_interval, dv); or_baseline_observation(&e->ac[i], dv, x);
}
void or_env_get_ctl_in(const OrEnv* e, int i, double* x) { for (int k = 0; k < 12; k++) x[k] = e->ac[i].ctl_in[k]; }
