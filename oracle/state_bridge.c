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
  v[k++] = s->alpha; v[k++] = s->mach; v[k++] = s->vc_fps / KTSTOFPS; v[k++] = s->vg;
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
  s->alpha = v[k++]; s->mach = v[k++]; s->vc_fps = v[k++] * KTSTOFPS; s->vg = v[k++];
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
  extern void or_env_refresh_cache(OrEnv * e, int i);
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
