/* ORACLE — TEST INFRASTRUCTURE ONLY (see combat_env.h). "R/" = /root/reference/envs/JSBSim/ */
#include "combat_env.h"
#include "geodesy.h"
#include <math.h>
#include <string.h>

#define FT2M 0.3048
#define RAD2DEG (180.0 / M_PI)

static double clampd(double lo, double v, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double sign0(double v) { return (v > 0) - (v < 0); }

/* MultipleCombatEnv semantics (multiplecombat_env.py:119-182): MultipleCombatTask and everything built on it. The rule-based
 * MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-134) is OR_TASK_DODGE_MISSILE with more than two aircraft. */
static int nvn_env(const OrEnvConfig* c) {
  return c->task == OR_TASK_MULTICOMBAT || c->task == OR_TASK_SCENARIO_NVN || (c->task == OR_TASK_DODGE_MISSILE && c->n_aircraft > 2);
}
int or_env_obs_dim(int task) {
  switch (task) {
    case OR_TASK_HEADING: return 12;
    case OR_TASK_SINGLECOMBAT: return 15;
    default: return 21;
  }
}
int or_env_obs_dim_n(int task, int n_aircraft) {
  if (task == OR_TASK_MULTICOMBAT) return 9 + (n_aircraft - 1) * 6; /* multiplecombat_task.py:95-98 */
  if (task == OR_TASK_SCENARIO_NVN) return 9 + 6 * (n_aircraft / 2) + 6 * (n_aircraft / 2) + 6; /* scenario2_task.py:244-254 */
  if (task == OR_TASK_WVR || task == OR_TASK_MANEUVER) return 15; /* HierarchicalSingleCombatTask keeps SingleCombatTask's 15-value observation */
  return or_env_obs_dim(task);
}
/* ------------------------------------------------------------------ scripted opponents (model/baseline.py) */
static double in_range_rad(double a) { /* utils.py:114-119, Python % semantics */
  a = fmod(a, 2 * M_PI);
  if (a < 0) a += 2 * M_PI;
  if (a > M_PI) a -= 2 * M_PI;
  return a;
}
/* BaselineAgent.get_observation (baseline.py:45-63): delta values + own attitude / speeds */
static void baseline_observation(const OrAircraft* a, const double dv[3], double x[12]) {
  x[0] = dv[0] / 1000; x[1] = in_range_rad(dv[1]); x[2] = dv[2] / 340;
  x[3] = clampd(-500, a->fdm.h_sl * 0.3048, 26000) / 5000;
  x[4] = sin(a->fdm.phi); x[5] = cos(a->fdm.phi); x[6] = sin(a->fdm.tht); x[7] = cos(a->fdm.tht);
  x[8] = clampd(-700, a->fdm.uvw[0] * 0.3048, 700) / 340; x[9] = clampd(-700, a->fdm.uvw[1] * 0.3048, 700) / 340;
  x[10] = clampd(-700, a->fdm.uvw[2] * 0.3048, 700) / 340; x[11] = clampd(0, a->fdm.vc_fps * 0.3048, 1400) / 340;
}
/* PursueAgent.set_delta_value (baseline.py:85-104): climb to the target's height, turn onto it (2-D AO with side), match its speed */
void or_pursue_delta(const OrAircraft* ego, const OrAircraft* tgt, double dv[3]) {
  dv[0] = tgt->position[2] - ego->position[2];
  double vx = ego->velocity[0], vy = ego->velocity[1];
  double ev = hypot(vx, vy), dx = tgt->position[0] - ego->position[0], dy = tgt->position[1] - ego->position[1];
  double R = hypot(dx, dy), proj = dx * vx + dy * vy;
  double ao = acos(clampd(-1, proj / (R * ev + 1e-8), 1));
  double cr = vx * dy - vy * dx;
  dv[1] = ao * ((cr > 0) - (cr < 0));
  dv[2] = clampd(-700, tgt->fdm.uvw[0] * 0.3048, 700) - clampd(-700, ego->fdm.uvw[0] * 0.3048, 700);
}
/* ManeuverAgent('triangle').set_delta_value (baseline.py:114-155): heading schedule [pi/3, pi, -pi/3] x 100 every turn_interval
 * seconds relative to the heading latched at the first call; 6000 m, 243 m/s */
void or_maneuver_delta(OrAircraft* a, double turn_interval, double time_interval, double dv[3]) {
  static const double hl[3] = {M_PI / 3, M_PI, -M_PI / 3};
  double cur = a->fdm.psi;
  if (!a->man_init_set) { a->man_init_heading = cur; a->man_init_set = 1; }
  int i = 0;
  for (i = 0; i < 300; i++) if (a->man_step <= (i + 1) * turn_interval / time_interval) break;
  if (i >= 300) i = 299;
  dv[1] = a->man_init_heading + hl[i % 3] - cur;
  dv[0] = 6000 - clampd(-500, a->fdm.h_sl * 0.3048, 26000);
  dv[2] = 243 - clampd(-700, a->fdm.uvw[0] * 0.3048, 700);
  a->man_step += 1;
}
void or_baseline_observation(const OrAircraft* a, const double dv[3], double x[12]) { baseline_observation(a, dv, x); }

/* hierarchical action spaces: MultiDiscrete [3,5,3] (singlecombat_task.py:221-222), + [2,2,2,2] weapon bits for the scenario tasks */
int or_env_act_dim_h(int task, int hierarchical) {
  if (!hierarchical) return or_env_act_dim(task);
  if (task == OR_TASK_SHOOT_MISSILE) return 4;   /* HierarchicalSingleCombatShootTask: Tuple([3,5,3], Discrete(2)) (singlecombat_with_missile_task.py:221-223) */
  return (task == OR_TASK_SCENARIO1 || task == OR_TASK_SCENARIO_NVN) ? 7 : 3;
}
int or_env_act_dim(int task) {
  if (task == OR_TASK_SCENARIO1 || task == OR_TASK_SCENARIO_NVN) return 8; /* 4 low-level controls + [gun, AIM-9M, AIM-120B, chaff] */
  return task == OR_TASK_SHOOT_MISSILE ? 5 : 4;
}

void or_env_default_config(OrEnvConfig* c, int task) {
  memset(c, 0, sizeof *c);
  c->task = task;
  c->sim_freq = 60;
  c->agent_interaction_steps = 6;
  c->center_lon = 120.0; c->center_lat = 60.0; c->center_alt = 0.0;
  c->altitude_limit = 2500; c->acc_limit_x = c->acc_limit_y = c->acc_limit_z = 10.0;
  c->posture_scale = 15.0; c->posture_potential = 1;   /* R/configs/scenario1 YAMLs */
  c->altitude_scale = 1.0; c->event_scale = 1.0; c->event_potential = 1;
  c->heading_scale = 1.0; c->missile_posture_scale = 30.0; c->shoot_penalty_scale = 1.0;
  c->alt_safe = 4.0; c->alt_danger = 3.5; c->alt_kv = 0.2;
  c->max_attack_angle = 45; c->max_attack_distance = 14000; c->min_attack_interval = 25;
  c->relative_altitude_scale = 1.0; c->relative_altitude_KH = 1.0; c->gun_scale = 1.0; c->chaff_seed = 1;
  for (int i = 0; i < OR_MAX_AC; i++) f16_default_init(&c->init[i]);
  if (task == OR_TASK_MULTICOMBAT || task == OR_TASK_SCENARIO_NVN) {
    /* aircraft block of R/configs/scenario2/scenario2_nvn.yaml:15-62 (2v2) */
    for (int i = 0; i < 4; i++) c->num_missiles[i] = 2;
    c->n_aircraft = 4; c->n_ego = 2; c->max_steps = 9000;
    c->event_potential = 0; /* that YAML sets no EventDrivenReward_potential */
    c->init[1].lon_deg = 120.01;
    c->init[2].lat_geod_deg = 60.1; c->init[2].psi_deg = 180.0;
    c->init[3].lon_deg = 120.01; c->init[3].lat_geod_deg = 60.1; c->init[3].psi_deg = 180.0;
    c->min_attack_interval = 125;
    return;
  }
  if (task == OR_TASK_HEADING) {
    /* R/configs/singlecontrol/heading.yaml */
    c->n_aircraft = 1; c->n_ego = 1; c->max_steps = 10000;
    c->posture_scale = 1.0; c->posture_potential = 0; c->event_potential = 0;
    c->max_heading_increment = 180; c->max_altitude_increment = 7000; c->max_velocities_u_increment = 100; c->check_interval = 30;
  } else {
    /* aircraft block of R/configs/scenario1/WVR_selfplay.yaml:15-40 */
    c->n_aircraft = 2; c->n_ego = 1; c->max_steps = 9000;
    c->init[1].lon_deg = 120.5; c->init[1].lat_geod_deg = 60.1; c->init[1].psi_deg = 180.0;
    c->num_missiles[0] = c->num_missiles[1] = 2;
  }
}

/* ------------------------------------------------------------------ numpy PCG64 + Generator.uniform */
void or_env_seed_pcg64(OrEnv* e, uint64_t shi, uint64_t slo, uint64_t ihi, uint64_t ilo) {
  e->rng_state = ((unsigned __int128)shi << 64) | slo;
  e->rng_inc = ((unsigned __int128)ihi << 64) | ilo;
}
static uint64_t pcg64_next(OrEnv* e) {
  const unsigned __int128 mult = ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
  e->rng_state = e->rng_state * mult + e->rng_inc;
  uint64_t hi = (uint64_t)(e->rng_state >> 64), lo = (uint64_t)e->rng_state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
double or_env_uniform(OrEnv* e, double lo, double hi) {
  double u = (double)(pcg64_next(e) >> 11) * (1.0 / 9007199254740992.0);
  return lo + (hi - lo) * u;
}

/* The reference draws the decoy outcome from the GLOBAL, unseeded np.random (env_base.py:153): only statistical parity is
 * possible. Oracle and kernel both use this counter-based generator, keyed by WHAT is being tested (substep tick since reset,
 * missile = launcher + uid number, chaff = releaser + its release index) rather than by a sequential draw counter, so that the
 * result does not depend on the evaluation order of independent pairs. */
static double chaff_uniform(OrEnv* e, int tick, int msl_parent, int msl_num, int chaff_parent, int chaff_local) {
  uint64_t k = ((uint64_t)(uint32_t)tick << 32) | ((uint64_t)(msl_parent & 0xff) << 24) | ((uint64_t)(msl_num & 0xff) << 16) |
               ((uint64_t)(chaff_parent & 0xff) << 8) | (uint64_t)(chaff_local & 0xff);
  uint64_t z = e->cfg.chaff_seed * 0x9E3779B97F4A7C15ULL + k * 0xD1B54A32D192ED03ULL;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
  e->chaff_draws++;
  return (double)(z >> 40) * (1.0 / 16777216.0);
}

/* ------------------------------------------------------------------ utils.py:58-103 */
void or_get_AO_TA_R(const double ego[6], const double enm[6], int two_d, double out[4]) {
  double ev, nv, R, pe, pn;
  double dx = enm[0] - ego[0], dy = enm[1] - ego[1], dz = enm[2] - ego[2];
  if (two_d) {
    ev = sqrt(ego[3] * ego[3] + ego[4] * ego[4]);
    nv = sqrt(enm[3] * enm[3] + enm[4] * enm[4]);
    R = sqrt(dx * dx + dy * dy);
    pe = dx * ego[3] + dy * ego[4];
    pn = dx * enm[3] + dy * enm[4];
  } else {
    ev = norm3(ego + 3);
    nv = norm3(enm + 3);
    R = sqrt(dx * dx + dy * dy + dz * dz);
    pe = dx * ego[3] + dy * ego[4] + dz * ego[5];
    pn = dx * enm[3] + dy * enm[4] + dz * enm[5];
  }
  out[0] = acos(clampd(-1, pe / (R * ev + 1e-8), 1));
  out[1] = acos(clampd(-1, pn / (R * nv + 1e-8), 1));
  out[2] = R;
  out[3] = sign0(ego[3] * dy - ego[4] * dx);
}

/* posture_reward.py:58-75, versions v2 / v3 (the only ones the YAMLs select) */
double or_posture_orientation(double AO, double TA) {
  return 1 / (50 * AO / M_PI + 2) + 1.0 / 2 + fmin(atanh(1. - fmax(2 * TA / M_PI, 1e-4)) / (2 * M_PI), 0.) + 0.5;
}
double or_posture_range(double R) {
  return 1 * (R < 5) + (R >= 5) * clampd(0, -0.032 * R * R + 0.284 * R + 0.38, 1) + clampd(0, exp(-0.16 * R), 0.2);
}
double or_posture_reward(double AO, double TA, double R) { return or_posture_orientation(AO, TA) * or_posture_range(R); }
/* altitude_reward.py:20-40 */
double or_altitude_reward(double z, double vz, double safe, double danger, double kv) {
  double Pv = 0., PH = 0.;
  if (z <= safe) Pv = -clampd(0., vz / kv * (safe - z) / safe, 1.);
  if (z <= danger) PH = clampd(0., z / danger, 1.) - 1. - 1.;
  return Pv + PH;
}

/* ------------------------------------------------------------------ MissileSimulator, simulatior.py:393-608 */
void or_missile_init(OrMissile* m, int model, double dt) {
  memset(m, 0, sizeof *m);
  m->status = OR_MSL_INACTIVE;
  m->parent = m->target = -1;
  m->g = 9.81; m->dm = 6; m->v_min = 150;
  if (model == 0) { /* AIM-9L defaults :421-433 */
    m->t_max = 60; m->t_thrust = 3; m->Isp = 120; m->Length = 2.87; m->Diameter = 0.127; m->cD = 0.4; m->m0 = 84;
    m->K = 3; m->nyz_max = 30; m->Rc = 300;
  } else { /* the set both AIM_9M and AIM_120B carry :659-672, :696-709 */
    m->t_max = 27.22; m->t_thrust = 1.4; m->Isp = 1837; m->Length = 3.66; m->Diameter = 0.18; m->cD = 0.02; m->m0 = 152;
    m->K = 5; m->nyz_max = 50; m->Rc = 5;
  }
  m->recede_max = (int)(5 / dt);
}
void or_missile_launch(OrMissile* m, const double geodetic[3], const double position[3], const double velocity[3], const double rpy[3]) {
  memcpy(m->geodetic, geodetic, sizeof m->geodetic);
  memcpy(m->position, position, sizeof m->position);
  memcpy(m->velocity, velocity, sizeof m->velocity);
  m->posture[0] = 0; m->posture[1] = rpy[1]; m->posture[2] = rpy[2];
  m->t = 0; m->m = m->m0; m->dtheta = m->dphi = 0;
  m->status = OR_MSL_LAUNCHED;
  m->dist_prev = INFINITY;
  m->recede_count = 0; m->recede_len = 0;
}
int or_missile_run(OrMissile* m, const double tp[3], const double tv[3], int target_alive, double dt, double lon0, double lat0, double alt0) {
  m->t += dt;
  /* _guidance :556-576 */
  double xm = m->position[0], ym = m->position[1], zm = m->position[2];
  double dxm = m->velocity[0], dym = m->velocity[1], dzm = m->velocity[2];
  double vm = norm3(m->velocity);
  double theta_m = asin(dzm / vm);
  double xt = tp[0], yt = tp[1], zt = tp[2], dxt = tv[0], dyt = tv[1], dzt = tv[2];
  double Rxy = sqrt((xm - xt) * (xm - xt) + (ym - yt) * (ym - yt));
  double Rxyz = sqrt((xm - xt) * (xm - xt) + (ym - yt) * (ym - yt) + (zt - zm) * (zt - zm));
  double dbeta = ((dyt - dym) * (xt - xm) - (dxt - dxm) * (yt - ym)) / (Rxy * Rxy);
  double deps = ((dzt - dzm) * (Rxy * Rxy) - (zt - zm) * ((xt - xm) * (dxt - dxm) + (yt - ym) * (dyt - dym))) / (Rxyz * Rxyz * Rxy);
  double K = fmax(m->K * (m->t_max - m->t) / m->t_max, 0);
  double ny = clampd(-m->nyz_max, K * vm / m->g * cos(theta_m) * dbeta, m->nyz_max);
  double nz = clampd(-m->nyz_max, K * vm / m->g * deps + cos(theta_m), m->nyz_max);
  double distance = Rxyz;
  /* run :520-533 */
  int inc = distance > m->dist_prev;
  if (m->recede_len < m->recede_max) m->recede_len++;
  m->recede_count = inc ? m->recede_count + 1 : 0; /* sum(deque) >= maxlen <=> the last maxlen samples were all True */
  m->dist_prev = distance;
  if (distance < m->Rc && target_alive && m->status != OR_MSL_MISS) {
    m->status = OR_MSL_HIT;
  } else if (m->t > m->t_max || norm3(m->velocity) < m->v_min || m->recede_count >= m->recede_max || !target_alive) {
    m->status = OR_MSL_MISS;
  } else {
    /* _state_trans :578-608 */
    for (int i = 0; i < 3; i++) m->position[i] += dt * m->velocity[i];
    or_neu2lla(m->position[0], m->position[1], m->position[2], lon0, lat0, alt0, m->geodetic);
    double v = norm3(m->velocity);
    double theta = m->posture[1], phi = m->posture[2];
    double Isp = (m->t < m->t_thrust) ? m->Isp : 0;
    double T = m->g * Isp * m->dm;
    double S = M_PI * (m->Diameter / 2) * (m->Diameter / 2) + hypot(sin(m->dtheta), sin(m->dphi)) * m->Diameter * m->Length;
    double rho = 1.225 * exp(-m->geodetic[2] / 9300);
    double D = 0.5 * m->cD * S * rho * v * v;
    double nx = (T - D) / (m->m * m->g);
    double dv = m->g * (nx - sin(theta));
    m->dphi = m->g / v * (ny / cos(theta));
    m->dtheta = m->g / v * (nz - cos(theta));
    v += dt * dv; phi += dt * m->dphi; theta += dt * m->dtheta;
    m->velocity[0] = v * cos(theta) * cos(phi);
    m->velocity[1] = v * cos(theta) * sin(phi);
    m->velocity[2] = v * sin(theta);
    m->posture[0] = 0; m->posture[1] = theta; m->posture[2] = phi;
    if (m->t < m->t_thrust) m->m = m->m - dt * m->dm;
  }
  return m->status;
}

/* ------------------------------------------------------------------ AircraftSimulator wrapper */
static double in_range_deg(double a) { /* utils.py:106-111, Python % semantics */
  a = fmod(a, 360.0);
  if (a < 0) a += 360.0;
  if (a > 180) a -= 360;
  return a;
}
static double h_sl_m(const OrAircraft* a) { return clampd(-500, a->fdm.h_sl * FT2M, 26000); }        /* catalog.py:292-296 */
static double mps(double fps) { return clampd(-700, fps * FT2M, 700); }                                /* catalog.py:298-338 */
static double vc_mps(const OrAircraft* a) { return clampd(0, a->fdm.vc_fps * FT2M, 1400); }

static void update_properties(const OrEnv* e, OrAircraft* a) { /* simulatior.py:238-258 */
  a->geodetic[0] = a->fdm.lon * RAD2DEG;
  a->geodetic[1] = a->fdm.lat_geod * RAD2DEG;
  a->geodetic[2] = h_sl_m(a);
  or_lla2neu(a->geodetic[0], a->geodetic[1], a->geodetic[2], e->cfg.center_lon, e->cfg.center_lat, e->cfg.center_alt, a->position);
  a->posture[0] = a->fdm.phi; a->posture[1] = a->fdm.tht; a->posture[2] = a->fdm.psi;
  a->velocity[0] = mps(a->fdm.vel_ned[0]); a->velocity[1] = mps(a->fdm.vel_ned[1]); a->velocity[2] = mps(a->fdm.vel_ned[2]);
}
static void aircraft_reload(const OrEnv* e, OrAircraft* a, const F16Init* ic, int num_missiles) { /* simulatior.py:152-190 */
  int team = a->team;
  memset(a, 0, sizeof *a);
  a->team = team;
  a->bloods = 100; a->status = OR_ALIVE;
  a->remaining_missiles = num_missiles;
  a->last_missile = -1;
  f16_reset(&a->fdm, ic);
  update_properties(e, a);
}
static void aircraft_run(const OrEnv* e, OrAircraft* a) { /* simulatior.py:210-229 */
  if (a->status != OR_ALIVE) return;
  if (a->bloods <= 0) a->status = OR_SHOTDOWN;
  f16_tick(&a->fdm, 1.0 / e->cfg.sim_freq);
  update_properties(e, a);
}
void or_env_refresh_cache(OrEnv* e, int i) { update_properties(e, &e->ac[i]); }
static int extreme_state(const OrAircraft* a) { /* catalog.py:386-416 */
  const F16State* s = &a->fdm;
  int ev = norm3(s->v_eci) >= 1e10;
  int er = norm3(s->pqr) >= 1000;
  int ea = s->h_sl >= 1e10;
  double mx = fmax(fabs(s->npilot[0]), fmax(fabs(s->npilot[1]), fabs(s->npilot[2])));
  int eacc = mx > 1e1;
  return ea || er || ev || eacc;
}

void or_env_init(OrEnv* e, const OrEnvConfig* c) {
  memset(e, 0, sizeof *e);
  e->cfg = *c;
  e->obs_dim = or_env_obs_dim_n(c->task, c->n_aircraft) + (c->rwr ? 2 : 0);
  if (c->task == OR_TASK_SCENARIO_NVN && c->legacy_obs) e->obs_dim = 21;   /* multiplecombat_with_missile_task.py:30-31 */   /* scenario1_task.py:213-216, scenario2_task.py:403-413 */
  e->act_dim = or_env_act_dim_h(c->task, c->hierarchical);
  if (c->task == OR_TASK_MULTICOMBAT && c->legacy_obs) { /* hierarchical_multiplecombat_shoot (multiplecombat_with_missile_task.py:206-238) */
    e->obs_dim = 21;                      /* :180-183 */
    e->act_dim = c->hierarchical ? 4 : 5; /* Tuple([3,5,3], Discrete(2)) (:221-223) / MultipleCombatShootMissileTask's Tuple([41,41,41,30], Discrete(2)) (:176-178);
                                             the bit is stored (:204-206, :231) and never used (:215-216) */
  }
  for (int i = 0; i < c->n_aircraft; i++) e->ac[i].team = (i < c->n_ego) ? 0 : 1;
  e->mp_prev_missile = -1;
}

/* first alive missile aimed at aircraft i, in launch order (simulatior.py:321-325) */
static int missile_warning(const OrEnv* e, int i) {
  for (int k = 0; k < e->n_msl; k++)
    if (e->msl[k].target == i && e->msl[k].status == OR_MSL_LAUNCHED) return k;
  return -1;
}
static int first_enemy(const OrEnv* e, int i) {
  for (int k = 0; k < e->cfg.n_aircraft; k++) if (e->ac[k].team != e->ac[i].team) return k;
  return -1;
}

/* ------------------------------------------------------------------ observations */
static void obs_heading(const OrEnv* e, int i, double* o) { /* heading_task.py:67-100 */
  const OrAircraft* a = &e->ac[i];
  double d_alt = clampd(-40000, (a->target_altitude_ft - a->fdm.h_sl) * FT2M, 40000);
  double d_hdg = clampd(-180, in_range_deg(a->target_heading_deg - a->fdm.psi * RAD2DEG), 180);
  double d_vel = clampd(-1400, a->target_velocities_u_mps - mps(a->fdm.uvw[0]), 1400);
  o[0] = d_alt / 1000; o[1] = d_hdg / 180 * M_PI; o[2] = d_vel / 340; o[3] = h_sl_m(a) / 5000;
  o[4] = sin(a->fdm.phi); o[5] = cos(a->fdm.phi); o[6] = sin(a->fdm.tht); o[7] = cos(a->fdm.tht);
  o[8] = mps(a->fdm.uvw[0]) / 340; o[9] = mps(a->fdm.uvw[1]) / 340; o[10] = mps(a->fdm.uvw[2]) / 340; o[11] = vc_mps(a) / 340;
  for (int k = 0; k < 12; k++) o[k] = clampd(-10, o[k], 10);
}
static void feature6(const OrAircraft* a, double f[6]) {
  f[0] = a->position[0]; f[1] = a->position[1]; f[2] = a->position[2];
  f[3] = a->velocity[0]; f[4] = a->velocity[1]; f[5] = a->velocity[2];
}
static void obs_combat(const OrEnv* e, int i, double* o) { /* singlecombat_task.py:88-139, singlecombat_with_missile_task.py:31-99 */
  const OrAircraft* a = &e->ac[i];
  const OrAircraft* en = &e->ac[first_enemy(e, i)];
  if (nvn_env(&e->cfg)) { /* legacy layout: `target` = own index within the team (multiplecombat_with_missile_task.py:62-78) */
    int idx = (a->team == 0) ? i : i - e->cfg.n_ego;
    en = &e->ac[(a->team == 0) ? e->cfg.n_ego + idx : idx];
  }
  int dim = e->obs_dim;
  for (int k = 0; k < dim; k++) o[k] = 0;
  double ef[6], nf[6], r[4];
  feature6(a, ef); feature6(en, nf);
  o[0] = h_sl_m(a) / 5000;
  o[1] = sin(a->fdm.phi); o[2] = cos(a->fdm.phi); o[3] = sin(a->fdm.tht); o[4] = cos(a->fdm.tht);
  o[5] = mps(a->fdm.uvw[0]) / 340; o[6] = mps(a->fdm.uvw[1]) / 340; o[7] = mps(a->fdm.uvw[2]) / 340; o[8] = vc_mps(a) / 340;
  const int two_d = e->cfg.task == OR_TASK_SINGLECOMBAT || e->cfg.task == OR_TASK_WVR || e->cfg.task == OR_TASK_MANEUVER;
  or_get_AO_TA_R(ef, nf, two_d, r);
  o[9] = (mps(en->fdm.uvw[0]) - mps(a->fdm.uvw[0])) / 340;
  o[10] = (h_sl_m(en) - h_sl_m(a)) / 1000;
  o[11] = r[0]; o[12] = r[1]; o[13] = r[2] / 10000; o[14] = r[3];
  if (two_d) {
    for (int k = 0; k < 15; k++) o[k] = clampd(-10, o[k], 10);
    return;
  }
  int mk = e->cfg.rwr ? -1 : missile_warning(e, i);   /* Scenario1_RWR.get_obs: `missile_sim = None` (scenario1_task.py:298-300) */
  if (mk >= 0) {
    const OrMissile* m = &e->msl[mk];
    double mf[6] = {m->position[0], m->position[1], m->position[2], m->velocity[0], m->velocity[1], m->velocity[2]};
    or_get_AO_TA_R(ef, mf, 0, r);
    o[15] = (norm3(m->velocity) - mps(a->fdm.uvw[0])) / 340;
    o[16] = (mf[2] - h_sl_m(a)) / 1000;
    o[17] = r[0]; o[18] = r[1]; o[19] = r[2] / 10000; o[20] = r[3];
  }
}
static void obs_multicombat(const OrEnv* e, int i, double* o) { /* multiplecombat_task.py:105-135: partners then enemies, clipped */
  const OrAircraft* a = &e->ac[i];
  int dim = e->obs_dim;
  for (int k = 0; k < dim; k++) o[k] = 0;
  double ef[6], nf[6], r[4];
  feature6(a, ef);
  o[0] = h_sl_m(a) / 5000;
  o[1] = sin(a->fdm.phi); o[2] = cos(a->fdm.phi); o[3] = sin(a->fdm.tht); o[4] = cos(a->fdm.tht);
  o[5] = mps(a->fdm.uvw[0]) / 340; o[6] = mps(a->fdm.uvw[1]) / 340; o[7] = mps(a->fdm.uvw[2]) / 340; o[8] = vc_mps(a) / 340;
  int off = 8;
  for (int pass = 0; pass < 2; pass++) /* partners (same team) in order, then enemies in order (env_base.py:79-88) */
    for (int k = 0; k < e->cfg.n_aircraft; k++) {
      if (k == i) continue;
      int same = e->ac[k].team == a->team;
      if ((pass == 0) != same) continue;
      const OrAircraft* b = &e->ac[k];
      feature6(b, nf);
      or_get_AO_TA_R(ef, nf, 0, r);
      o[off + 1] = (mps(b->fdm.uvw[0]) - mps(a->fdm.uvw[0])) / 340;
      o[off + 2] = (h_sl_m(b) - h_sl_m(a)) / 1000;
      o[off + 3] = r[0]; o[off + 4] = r[1]; o[off + 5] = r[2] / 10000; o[off + 6] = r[3];
      off += 6;
    }
  for (int k = 0; k < dim; k++) o[k] = clampd(-10, o[k], 10);
}
static void obs_scenario_nvn(const OrEnv* e, int i, double* o) { /* scenario2_task.py:256-316: not clipped, missile block right after the enemies */
  const OrAircraft* a = &e->ac[i];
  int dim = e->obs_dim;
  for (int k = 0; k < dim; k++) o[k] = 0;
  double ef[6], nf[6], r[4];
  feature6(a, ef);
  o[0] = h_sl_m(a) / 5000;
  o[1] = sin(a->fdm.phi); o[2] = cos(a->fdm.phi); o[3] = sin(a->fdm.tht); o[4] = cos(a->fdm.tht);
  o[5] = mps(a->fdm.uvw[0]) / 340; o[6] = mps(a->fdm.uvw[1]) / 340; o[7] = mps(a->fdm.uvw[2]) / 340; o[8] = vc_mps(a) / 340;
  int off = 8;
  for (int pass = 0; pass < 2; pass++)
    for (int k = 0; k < e->cfg.n_aircraft; k++) {
      if (k == i) continue;
      int same = e->ac[k].team == a->team;
      if ((pass == 0) != same) continue;
      const OrAircraft* b = &e->ac[k];
      feature6(b, nf);
      or_get_AO_TA_R(ef, nf, 0, r);
      o[off + 1] = (mps(b->fdm.uvw[0]) - mps(a->fdm.uvw[0])) / 340;
      o[off + 2] = (h_sl_m(b) - h_sl_m(a)) / 1000;
      o[off + 3] = r[0]; o[off + 4] = r[1]; o[off + 5] = r[2] / 10000; o[off + 6] = r[3];
      off += 6;
    }
  int mk = missile_warning(e, i);
  if (mk >= 0) {
    const OrMissile* m = &e->msl[mk];
    double mf[6] = {m->position[0], m->position[1], m->position[2], m->velocity[0], m->velocity[1], m->velocity[2]};
    or_get_AO_TA_R(ef, mf, 0, r);
    o[off + 1] = (norm3(m->velocity) - mps(a->fdm.uvw[0])) / 340;
    o[off + 2] = (mf[2] - h_sl_m(a)) / 1000;
    o[off + 3] = r[0]; o[off + 4] = r[1]; o[off + 5] = r[2] / 10000; o[off + 6] = r[3];
  }
}
static void get_obs(const OrEnv* e, double* obs) {
  for (int i = 0; i < e->cfg.n_aircraft; i++) {
    if (e->cfg.task == OR_TASK_HEADING) obs_heading(e, i, obs + i * e->obs_dim);
    else if (e->cfg.task == OR_TASK_MULTICOMBAT && !e->cfg.legacy_obs) obs_multicombat(e, i, obs + i * e->obs_dim);
    else if (e->cfg.task == OR_TASK_SCENARIO_NVN && !e->cfg.legacy_obs) obs_scenario_nvn(e, i, obs + i * e->obs_dim);
    else obs_combat(e, i, obs + i * e->obs_dim);
  }
}

/* ------------------------------------------------------------------ rewards */
static double process(double r, double scale, int potential, double* pre) { /* reward_function_base.py:48-63 */
  r *= scale;
  if (potential) { double out = r - *pre; *pre = r; return out; }
  return r;
}
static double rw_posture(OrEnv* e, int i) { /* posture_reward.py:26-49 */
  double ef[6], nf[6], r[4], sum = 0;
  feature6(&e->ac[i], ef);
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    feature6(&e->ac[k], nf);
    or_get_AO_TA_R(ef, nf, 0, r);
    sum += or_posture_reward(r[0], r[1], r[2] / 1000);
  }
  return process(sum, e->cfg.posture_scale, e->cfg.posture_potential, &e->ac[i].pre_posture);
}
static double rw_altitude(OrEnv* e, int i) {
  const OrAircraft* a = &e->ac[i];
  double v = or_altitude_reward(a->position[2] / 1000, a->velocity[2] / 340, e->cfg.alt_safe, e->cfg.alt_danger, e->cfg.alt_kv);
  return process(v, e->cfg.altitude_scale, e->cfg.altitude_potential, &e->ac[i].pre_altitude);
}
static double rw_event(OrEnv* e, int i) { /* event_driven_reward.py:15-34 */
  double r = 0;
  if (e->ac[i].status == OR_SHOTDOWN) r -= 200; else if (e->ac[i].status == OR_CRASH) r -= 200;
  for (int k = 0; k < e->n_msl; k++) if (e->msl[k].parent == i && e->msl[k].status == OR_MSL_HIT) r += 200;
  return process(r, e->cfg.event_scale, e->cfg.event_potential, &e->ac[i].pre_event);
}
static double rw_shoot_penalty(OrEnv* e, int i) { /* shoot_penalty_reward.py:13-32 */
  double r = 0;
  if (e->ac[i].remaining_missiles == e->ac[i].pre_remaining_missiles - 1) r -= 30;
  e->ac[i].pre_remaining_missiles = e->ac[i].remaining_missiles;
  return process(r, e->cfg.shoot_penalty_scale, e->cfg.shoot_penalty_potential, &e->ac[i].pre_shoot);
}
static double rw_missile_posture(OrEnv* e, int i) { /* missile_posture_reward.py:18-46 */
  double reward = 0;
  int mk = missile_warning(e, i);
  if (mk >= 0) {
    const double* mv = e->msl[mk].velocity;
    const double* av = e->ac[i].velocity;
    if (e->mp_prev_missile < 0) e->mp_prev_missile = mk; /* previous_missile_v aliases that missile's live _velocity array */
    double v_dec = (norm3(e->msl[e->mp_prev_missile].velocity) - norm3(mv)) / 340 * e->cfg.missile_posture_scale;
    double angle = (mv[0] * av[0] + mv[1] * av[1] + mv[2] * av[2]) / (norm3(mv) * norm3(av));
    if (angle < 0) reward = angle / (fmax(v_dec, 0) + 1);
    else reward = angle * fmax(v_dec, 0);
  } else {
    e->mp_prev_missile = -1;
  }
  return reward;
}
#define FT2METERS 0.3048
/* the gun / geometry terms iterate over env.agents[agent_id].enemies in env order */
static double rw_combat_geometry(OrEnv* e, int i) { /* combat_geometry_reward.py:28-68: `i` there is never incremented and the
                                                     * prev lists are shared by all agents and never cleared between steps, so every
                                                     * enemy contributes -(AO_first - AO_ref) - (TA_first - TA_ref) */
  double ef[6], nf[6], r[4], sum = 0, AO0 = 0, TA0 = 0;
  int first = 1;
  feature6(&e->ac[i], ef);
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    feature6(&e->ac[k], nf);
    or_get_AO_TA_R(ef, nf, 0, r);
    if (first) { AO0 = r[0]; TA0 = r[1]; first = 0; }
    if (!e->cg_set) { e->cg_set = 1; e->cg_AO = r[0]; e->cg_TA = r[1]; }
    sum += -(AO0 - e->cg_AO) - (TA0 - e->cg_TA);
  }
  return sum * e->cfg.gun_scale;
}
static double rw_gun_wez(OrEnv* e, int i) { /* gun_WEZ_reward.py:28-55 */
  double ef[6], nf[6], r[4], sum = 0;
  feature6(&e->ac[i], ef);
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    feature6(&e->ac[k], nf);
    or_get_AO_TA_R(ef, nf, 0, r);
    if (r[2] >= 500 * FT2METERS && r[2] <= 3000 * FT2METERS && r[0] <= 1 * M_PI / 180) sum += 5 + 5 * (3000 * FT2METERS - r[2]) / (2500 * FT2METERS);
  }
  return sum * e->cfg.gun_scale;
}
static double rw_gun_behit(OrEnv* e, int i) { /* gun_behit_reward.py:27-54 */
  double ef[6], nf[6], r[4], sum = 0;
  feature6(&e->ac[i], ef);
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    feature6(&e->ac[k], nf);
    or_get_AO_TA_R(ef, nf, 0, r);
    if (r[2] >= 500 * FT2METERS && r[2] <= 3000 * FT2METERS && r[0] >= 179 * M_PI / 180) sum += -5;
  }
  return sum * e->cfg.gun_scale;
}
/* gun_WEZDOT_reward.py:33-77 and gun_targettail_reward.py:32-78 share one pattern: prev[j] read for enemy j is entry j of a
 * list that only ever grows; its first n_enemies entries are written by the first call after reset: prev[0] = d_0 and
 * prev[j] = d_{j-1} (of that first call), and stay fixed until the next reset */
static double rw_gun_track(OrEnv* e, int i, int tail) {
  double ef[6], nf[6], r[4], sum = 0, d[OR_MAX_AC];
  int n = 0;
  int* set = tail ? &e->tail_set : &e->wezdot_set;
  double* ref = tail ? e->tail_ref : e->wezdot_ref;
  feature6(&e->ac[i], ef);
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    feature6(&e->ac[k], nf);
    or_get_AO_TA_R(ef, nf, 0, r);
    double R = r[2], dd;
    if (!tail) {
      if (R >= 500 * FT2METERS && R <= 3000 * FT2METERS) dd = R * sin(r[0]);
      else dd = sqrt(R * R + pow(3000 * FT2METERS, 2) - 2 * R * (3000 * FT2METERS) * cos(r[0]));
    } else {
      if (R >= 3000 * FT2METERS && R <= 5000 * FT2METERS) dd = R * sin(r[1]);
      else if (R <= 3000 * FT2METERS) dd = sqrt(R * R + pow(3000 * FT2METERS, 2) - 2 * R * (3000 * FT2METERS) * cos(r[1]));
      else dd = sqrt(R * R + pow(5000 * FT2METERS, 2) - 2 * R * (5000 * FT2METERS) * cos(r[1]));
    }
    d[n] = dd;
    if (!*set) ref[n] = (n == 0) ? dd : d[n - 1];
    sum += -1.0 / 60 * tanh((d[n] - ref[n]) / sqrt(R));
    n++;
  }
  *set = 1;
  return sum * e->cfg.gun_scale;
}
static double rw_relative_altitude(OrEnv* e, int i) { /* relative_altitude_reward.py:18-32: enemies[0] only */
  int en = first_enemy(e, i);
  double v = fmin(e->cfg.relative_altitude_KH - fabs(e->ac[i].position[2] / 1000 - e->ac[en].position[2] / 1000), 0);
  return v * e->cfg.relative_altitude_scale;
}
static double rw_heading(OrEnv* e, int i) { /* heading_reward.py:18-71 */
  OrAircraft* a = &e->ac[i];
  double p = a->fdm.pqr[0], q = a->fdm.pqr[1];
  double roll_rate_r = 0, pitch_rate_r = 0;
  if (e->current_step > 1) { roll_rate_r = -fabs(p - a->last_roll_rate); pitch_rate_r = -fabs(q - a->last_pitch_rate); }
  double d_hdg = clampd(-180, in_range_deg(a->target_heading_deg - a->fdm.psi * RAD2DEG), 180);
  double d_alt = clampd(-40000, (a->target_altitude_ft - a->fdm.h_sl) * FT2M, 40000);
  double d_vel = clampd(-1400, a->target_velocities_u_mps - mps(a->fdm.uvw[0]), 1400);
  double hr = exp(-pow(d_hdg / 5.0, 2)), ar = exp(-pow(d_alt / 15.24, 2));
  double rr = exp(-pow(a->fdm.phi / 0.35, 2)), sr = exp(-pow(d_vel / 24, 2));
  double reward = pow(hr * ar * rr * sr, 1.0 / 4);
  if (e->current_step > 1) reward += roll_rate_r + pitch_rate_r;
  a->last_roll_rate = p; a->last_pitch_rate = q;
  return process(reward, e->cfg.heading_scale, e->cfg.heading_potential, &a->pre_heading);
}
static double task_reward_terms(OrEnv* e, int i) {
  switch (e->cfg.task) {
    case OR_TASK_HEADING: { double r = e->cfg.approach ? 0.0 : rw_heading(e, i); return r + rw_altitude(e, i); }  /* approach_task.py:20-22 */
    case OR_TASK_SCENARIO1: case OR_TASK_SCENARIO_NVN: { /* scenario1_task.py:13-25 / scenario2_task.py:228-240, list order */
      double r = rw_altitude(e, i); r += rw_combat_geometry(e, i); r += rw_event(e, i); r += rw_gun_behit(e, i);
      r += rw_gun_track(e, i, 1); r += rw_gun_track(e, i, 0); r += rw_gun_wez(e, i); r += rw_posture(e, i);
      r += rw_relative_altitude(e, i); r += rw_missile_posture(e, i);
      return r + rw_shoot_penalty(e, i); /* task.remaining_missiles never changes in the Scenario tasks: always 0 */
    }
    case OR_TASK_WVR: { /* WVR_task.py:21-30 */
      double r = rw_posture(e, i); r += rw_altitude(e, i); r += rw_event(e, i); r += rw_combat_geometry(e, i); r += rw_gun_behit(e, i);
      r += rw_gun_track(e, i, 1); r += rw_gun_wez(e, i); return r + rw_gun_track(e, i, 0);
    }
    case OR_TASK_MANEUVER: { /* singlecombat_task.py:267-277 */
      double r = rw_altitude(e, i); r += rw_combat_geometry(e, i); r += rw_event(e, i); r += rw_gun_behit(e, i);
      r += rw_gun_track(e, i, 1); r += rw_gun_track(e, i, 0); r += rw_gun_wez(e, i); r += rw_posture(e, i);
      return r + rw_relative_altitude(e, i);
    }
    case OR_TASK_MULTICOMBAT: /* same three terms, multiplecombat_task.py:27-31 */
    case OR_TASK_SINGLECOMBAT: { double r = rw_altitude(e, i); r += rw_posture(e, i); return r + rw_event(e, i); }
    case OR_TASK_DODGE_MISSILE: { double r = rw_posture(e, i); r += rw_missile_posture(e, i); r += rw_altitude(e, i); return r + rw_event(e, i); }
    default: { double r = rw_posture(e, i); r += rw_altitude(e, i); r += rw_event(e, i); return r + rw_shoot_penalty(e, i); }
  }
}
static double get_reward(OrEnv* e, int i) { /* singlecombat_task.py:190-195; heading task uses BaseTask.get_reward */
  if (e->cfg.task == OR_TASK_HEADING) return task_reward_terms(e, i);
  if (nvn_env(&e->cfg)) /* multiplecombat_task.py:147-151: only while alive */
    return e->ac[i].status == OR_ALIVE ? task_reward_terms(e, i) : 0.0;
  if (e->ac[i].die_flag) return 0.0;
  e->ac[i].die_flag = e->ac[i].status != OR_ALIVE;
  return task_reward_terms(e, i);
}
static void reward_reset(OrEnv* e) { /* reward_function_base.py:20-32 — each potential term seeds pre_rewards via get_reward */
  for (int i = 0; i < e->cfg.n_aircraft; i++) {
    OrAircraft* a = &e->ac[i];
    a->pre_posture = a->pre_altitude = a->pre_event = a->pre_heading = a->pre_shoot = 0;
    a->pre_remaining_missiles = e->cfg.num_missiles[i];
  }
  e->mp_prev_missile = -1;
  e->cg_set = e->wezdot_set = e->tail_set = 0;
  int t = e->cfg.task;
  /* reset order = reward_functions list order of the task */
  if (t == OR_TASK_HEADING) {
    if (e->cfg.heading_potential && !e->cfg.approach) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_heading(e, i);
    if (e->cfg.altitude_potential) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_altitude(e, i);
    return;
  }
  if (e->cfg.altitude_potential) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_altitude(e, i);
  if (e->cfg.posture_potential) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_posture(e, i);
  if (e->cfg.event_potential) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_event(e, i);
  if (t == OR_TASK_SHOOT_MISSILE && e->cfg.shoot_penalty_potential) for (int i = 0; i < e->cfg.n_aircraft; i++) rw_shoot_penalty(e, i);
}

/* ------------------------------------------------------------------ terminations (first that fires wins, task_base.py:88-112) */
static int t_low_altitude(OrEnv* e, int i, int* code) { /* low_altitude.py:15-34 */
  if (h_sl_m(&e->ac[i]) <= e->cfg.altitude_limit) { e->ac[i].status = OR_CRASH; *code = OR_DONE_LOW_ALTITUDE; return 1; }
  return 0;
}
static int t_extreme(OrEnv* e, int i, int* code) { /* extreme_state.py:14-33 */
  if (extreme_state(&e->ac[i])) { e->ac[i].status = OR_CRASH; *code = OR_DONE_EXTREME_STATE; return 1; }
  return 0;
}
static int t_overload(OrEnv* e, int i, int* code) { /* overload.py:18-46 */
  const F16State* s = &e->ac[i].fdm;
  if (s->sim_time > 10 && (fabs(s->npilot[0]) > e->cfg.acc_limit_x || fabs(s->npilot[1]) > e->cfg.acc_limit_y || fabs(s->npilot[2] + 1) > e->cfg.acc_limit_z)) {
    e->ac[i].status = OR_CRASH; *code = OR_DONE_OVERLOAD; return 1;
  }
  return 0;
}
static int t_safe_return(OrEnv* e, int i, int* code) { /* safe_return.py:15-50 */
  if (e->ac[i].status == OR_SHOTDOWN) { *code = OR_DONE_SHOTDOWN; return 1; }
  if (e->ac[i].status == OR_CRASH) { *code = OR_DONE_CRASHED; return 1; }
  int enemies_dead = 1, no_missile = 1;
  for (int k = 0; k < e->cfg.n_aircraft; k++) if (e->ac[k].team != e->ac[i].team && e->ac[k].status == OR_ALIVE) enemies_dead = 0;
  for (int k = 0; k < e->n_msl; k++) if (e->msl[k].target == i && e->msl[k].status == OR_MSL_LAUNCHED) no_missile = 0;
  if (enemies_dead && no_missile) { *code = OR_DONE_MISSION_COMPLETE; return 1; }
  return 0;
}
static int t_timeout(OrEnv* e, int i, int* code) { /* timeout.py:14-32 */
  (void)i;
  if (e->current_step >= e->cfg.max_steps) { *code = OR_DONE_TIMEOUT; return 1; }
  return 0;
}
static int t_unreach_heading(OrEnv* e, int i, int* code) { /* unreach_heading.py:22-65 */
  static const double inc_size[15] = {0.2, 0.4, 0.6, 0.8, 1.0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  OrAircraft* a = &e->ac[i];
  int done = 0;
  double check_time = a->heading_check_time;
  if (a->fdm.sim_time >= check_time) {
    double d_hdg = clampd(-180, in_range_deg(a->target_heading_deg - a->fdm.psi * RAD2DEG), 180);
    if (fabs(d_hdg) > 10) done = 1;
    else {
      double delta = inc_size[e->heading_turn_counts];
      double dh = or_env_uniform(e, -delta, delta) * e->cfg.max_heading_increment;
      double da = or_env_uniform(e, -delta, delta) * e->cfg.max_altitude_increment;
      double dv = or_env_uniform(e, -delta, delta) * e->cfg.max_velocities_u_increment;
      double nh = fmod(a->target_heading_deg + dh + 360, 360);
      if (nh < 0) nh += 360;
      a->target_heading_deg = clampd(0, nh, 360);
      a->target_altitude_ft = clampd(-1400, a->target_altitude_ft + da, 85000);
      a->target_velocities_u_mps = clampd(-700, a->target_velocities_u_mps + dv, 700);
      a->heading_check_time = clampd(0, check_time + e->cfg.check_interval, 1000000);
      e->heading_turn_counts += 1;
    }
  }
  if (done) *code = OR_DONE_UNREACH_HEADING;
  return done;
}
static int get_termination(OrEnv* e, int i, int* code) {
  if (e->cfg.task == OR_TASK_HEADING && e->cfg.approach) /* approach_task.py:23-28 */
    return t_low_altitude(e, i, code) || t_extreme(e, i, code) || t_overload(e, i, code) || t_timeout(e, i, code);
  if (e->cfg.task == OR_TASK_HEADING) /* heading_task.py:20-26 */
    return t_unreach_heading(e, i, code) || t_extreme(e, i, code) || t_overload(e, i, code) || t_low_altitude(e, i, code) || t_timeout(e, i, code);
  if (e->cfg.task == OR_TASK_WVR) /* WVR_task.py:31-36: no SafeReturn */
    return t_low_altitude(e, i, code) || t_extreme(e, i, code) || t_overload(e, i, code) || t_timeout(e, i, code);
  if (nvn_env(&e->cfg)) /* multiplecombat_task.py:33-39 */
    return t_safe_return(e, i, code) || t_extreme(e, i, code) || t_overload(e, i, code) || t_low_altitude(e, i, code) || t_timeout(e, i, code);
  /* singlecombat_task.py:34-40 */
  return t_low_altitude(e, i, code) || t_extreme(e, i, code) || t_overload(e, i, code) || t_safe_return(e, i, code) || t_timeout(e, i, code);
}

/* ------------------------------------------------------------------ task.step */
/* env.add_temp_simulator(sim): self._tempsims[sim.uid] = sim (env_base.py:93-95). A uid seen before keeps its dict position
 * and the earlier missile drops out of the dict: it is never run() again but stays in its parent's launch_missiles and its
 * target's under_missiles lists. key < 0: unique uid (tasks whose uids never collide). */
static int new_missile_keyed(OrEnv* e, int parent, int target, int model, int key) {
  if (e->n_msl >= OR_MAX_MSL) return -1;
  int k = e->n_msl++;
  OrMissile* m = &e->msl[k];
  or_missile_init(m, model, 1.0 / e->cfg.sim_freq);
  m->parent = parent; m->target = target; m->model = model;
  or_missile_launch(m, e->ac[parent].geodetic, e->ac[parent].position, e->ac[parent].velocity, e->ac[parent].posture);
  m->key = key; m->in_sims = 1; m->sim_pos = -1;
  if (key >= 0)
    for (int j = 0; j < k; j++)
      if (e->msl[j].in_sims && e->msl[j].key == key) { m->sim_pos = e->msl[j].sim_pos; e->msl[j].in_sims = 0; }
  if (m->sim_pos < 0) m->sim_pos = e->n_sim_keys++;
  return k;
}
static int new_missile(OrEnv* e, int parent, int target, int model) { return new_missile_keyed(e, parent, target, model, -1); }
static int missile_done(const OrEnv* e, int k) { return k < 0 || e->msl[k].status == OR_MSL_HIT || e->msl[k].status == OR_MSL_MISS; }
/* get_target (scenario1_task.py:139-145): the FARTHEST enemy, dead or alive */
static int farthest_enemy(const OrEnv* e, int i) {
  int best = -1; double bd = -1;
  for (int k = 0; k < e->cfg.n_aircraft; k++) {
    if (e->ac[k].team == e->ac[i].team) continue;
    double d[3] = {e->ac[k].position[0] - e->ac[i].position[0], e->ac[k].position[1] - e->ac[i].position[1], e->ac[k].position[2] - e->ac[i].position[2]};
    double n = norm3(d);
    if (n > bd) { bd = n; best = k; }
  }
  return best;
}
/* a2a_launch_available (scenario1_task.py:104-138) */
static void a2a_available(const OrEnv* e, int i, int tgt, int avail[3]) {
  avail[0] = avail[1] = avail[2] = 0;
  if (e->ac[tgt].status != OR_ALIVE) return;
  const OrAircraft* a = &e->ac[i];
  double t[3] = {e->ac[tgt].position[0] - a->position[0], e->ac[tgt].position[1] - a->position[1], e->ac[tgt].position[2] - a->position[2]};
  double dist = norm3(t);
  double dot = t[0] * a->velocity[0] + t[1] * a->velocity[1] + t[2] * a->velocity[2];
  double ang = RAD2DEG * acos(clampd(-1, dot / (dist * norm3(a->velocity) + 1e-8), 1));
  if (dist / 1000 < 3 && ang < 5) avail[0] = 1;
  if (dist / 1000 < 37 && ang < 90) avail[1] = 1;
  if (dist / 1000 < 7 && ang < 90) avail[2] = 1;
}
static void scenario_weapons(OrEnv* e) { /* scenario1_task.py:61-103 == scenario2_task.py:73-115 */
  for (int i = 0; i < e->cfg.n_aircraft; i++) {
    OrAircraft* a = &e->ac[i];
    int alive = a->status == OR_ALIVE;
    int f_gun = alive && a->shoot4[0] && a->rem_gun > 0;
    int f_9m = alive && a->shoot4[1] && a->rem_9m > 0;
    int f_120 = alive && a->shoot4[2] && a->rem_120b > 0;
    int f_chaff = alive && a->shoot4[3] && a->rem_chaff > 0;
    int avail[3];
    if (f_gun && missile_done(e, a->last_missile)) {
      int tg = farthest_enemy(e, i);
      a2a_available(e, i, tg, avail);
      if (avail[0]) { e->ac[tg].bloods -= 5; a->rem_gun -= 1; }
    }
    if (f_120 && missile_done(e, a->last_missile)) {
      int tg = farthest_enemy(e, i);
      a2a_available(e, i, tg, avail);
      if (avail[1]) { a->last_missile = new_missile_keyed(e, i, tg, 1, i * 100 + a->rem_120b); a->rem_120b -= 1; }
    }
    if (f_9m && missile_done(e, a->last_missile)) {
      int tg = farthest_enemy(e, i);
      a2a_available(e, i, tg, avail);
      if (avail[2]) { a->last_missile = new_missile_keyed(e, i, tg, 1, i * 100 + a->rem_9m); a->rem_9m -= 1; }
    }
    if (f_chaff && (a->last_chaff < 0 || e->chaff[a->last_chaff].status == 1)) {
      /* one chaff per missile in env._tempsims (done ones included) that targets this aircraft within 1000 m */
      for (int pos = 0; pos < e->n_sim_keys; pos++)
        for (int k = 0; k < e->n_msl; k++) {
          const OrMissile* m = &e->msl[k];
          if (!m->in_sims || m->sim_pos != pos || m->target != i) continue;
          double d[3] = {a->position[0] - m->position[0], a->position[1] - m->position[1], a->position[2] - m->position[2]};
          if (norm3(d) < 1000 && e->n_chaff < OR_MAX_CHAFF) {
            int c = e->n_chaff++;
            memcpy(e->chaff[c].pos, a->position, sizeof e->chaff[c].pos);
            e->chaff[c].t = 0; e->chaff[c].status = 0; e->chaff[c].parent = i;
            a->last_chaff = c;
            a->rem_chaff -= 1;
          }
        }
    }
  }
}
static void wvr_gun(OrEnv* e) { /* WVR_task.py:62-76 */
  for (int i = 0; i < e->cfg.n_aircraft; i++) {
    int tg = farthest_enemy(e, i);
    const OrAircraft* a = &e->ac[i];
    double t[3] = {e->ac[tg].position[0] - a->position[0], e->ac[tg].position[1] - a->position[1], e->ac[tg].position[2] - a->position[2]};
    double dist = norm3(t);
    double dot = t[0] * a->velocity[0] + t[1] * a->velocity[1] + t[2] * a->velocity[2];
    double ang = RAD2DEG * acos(clampd(-1, dot / (dist * norm3(a->velocity) + 1e-8), 1));
    if (dist / 1000 < 3 && ang < 5) e->ac[tg].bloods -= 5;
  }
}
static void task_step(OrEnv* e) {
  int t = e->cfg.task;
  if (t == OR_TASK_HEADING || t == OR_TASK_MULTICOMBAT) return;
  if (e->cfg.use_artillery) { /* singlecombat_task.py:162-188 */
    for (int i = 0; i < e->cfg.n_aircraft; i++) {
      double ef[6], nf[6], r[4];
      feature6(&e->ac[i], ef);
      for (int k = 0; k < e->cfg.n_aircraft; k++) {
        if (e->ac[k].team == e->ac[i].team || e->ac[k].status != OR_ALIVE) continue;
        feature6(&e->ac[k], nf);
        or_get_AO_TA_R(ef, nf, 0, r);
        double AO = r[0], Rk = r[2] / 1000, of = 0, df = 0;
        if (AO >= 0 && AO <= 0.5236) of = 1 - AO / 0.5236; else if (AO >= -0.5236 && AO <= 0) of = 1 + AO / 0.5236;
        if (Rk <= 1) df = 1; else if (Rk > 1 && Rk <= 3) df = (3 - Rk) / 2.;
        e->ac[k].bloods -= of * df;
      }
    }
  }
  if (t == OR_TASK_SCENARIO1 || t == OR_TASK_SCENARIO_NVN) { scenario_weapons(e); return; }
  if (t == OR_TASK_WVR || t == OR_TASK_MANEUVER) { wvr_gun(e); return; }   /* Maneuver_curriculum.step: the same rule through a2a_launch_available (:290-297) */
  if (t == OR_TASK_DODGE_MISSILE) { /* singlecombat_with_missile_task.py:108-124 == multiplecombat_with_missile_task.py:127-145 (target: agent.enemies[0]) */
    for (int i = 0; i < e->cfg.n_aircraft; i++) {
      OrAircraft* a = &e->ac[i];
      int en = first_enemy(e, i);
      double tg[3] = {e->ac[en].position[0] - a->position[0], e->ac[en].position[1] - a->position[1], e->ac[en].position[2] - a->position[2]};
      double dist = norm3(tg);
      double dot = tg[0] * a->velocity[0] + tg[1] * a->velocity[1] + tg[2] * a->velocity[2];
      double ang = RAD2DEG * acos(clampd(-1, dot / (dist * norm3(a->velocity) + 1e-8), 1));
      int maxlen = a->lock_n;
      a->lock_window[a->lock_pos % maxlen] = ang < e->cfg.max_attack_angle;
      a->lock_pos++;
      int filled = a->lock_pos < maxlen ? a->lock_pos : maxlen, sum = 0;
      for (int k = 0; k < filled; k++) sum += a->lock_window[k];
      int interval = e->current_step - a->last_shoot_time;
      int shoot = a->status == OR_ALIVE && sum >= maxlen && dist <= e->cfg.max_attack_distance && a->remaining_missiles > 0 && interval >= e->cfg.min_attack_interval;
      if (shoot) {
        new_missile(e, i, en, 0);
        a->remaining_missiles -= 1;
        a->last_shoot_time = e->current_step;
      }
    }
  } else if (t == OR_TASK_SHOOT_MISSILE) { /* :194-204 */
    for (int i = 0; i < e->cfg.n_aircraft; i++) {
      OrAircraft* a = &e->ac[i];
      int shoot = a->status == OR_ALIVE && a->shoot_action && a->remaining_missiles > 0;
      int prev_done = a->last_missile < 0 || e->msl[a->last_missile].status == OR_MSL_HIT || e->msl[a->last_missile].status == OR_MSL_MISS;
      if (shoot && prev_done) {
        a->last_missile = new_missile(e, i, first_enemy(e, i), 0);
        a->remaining_missiles -= 1;
      }
    }
  }
}

/* missiles, chaff and the decoy test of one substep (env_base.py:142-154) */
static void run_projectiles_once(OrEnv* e, double dt) {
  const OrEnvConfig* c = &e->cfg;
  e->sub_tick++;
  for (int pos = 0; pos < e->n_sim_keys; pos++) /* env._tempsims in dict order */
    for (int k = 0; k < e->n_msl; k++) {
      OrMissile* m = &e->msl[k];
      if (!m->in_sims || m->sim_pos != pos) continue;
      /* MissileSimulator.run is called for every entry, whatever its status (simulatior.py:520-533) */
      OrAircraft* tg = &e->ac[m->target];
      int st = or_missile_run(m, tg->position, tg->velocity, tg->status == OR_ALIVE, dt, c->center_lon, c->center_lat, c->center_alt);
      if (st == OR_MSL_HIT && tg->status == OR_ALIVE) tg->status = OR_SHOTDOWN;
    }
  for (int k = 0; k < e->n_chaff; k++) { /* ChaffSimulator.run (simulatior.py:377-381) */
    e->chaff[k].t += dt;
    if (e->chaff[k].t > 20) e->chaff[k].status = 1;
  }
  for (int pos = 0; pos < e->n_sim_keys; pos++) /* decoy test (env_base.py:146-154) */
    for (int k = 0; k < e->n_msl; k++) {
      OrMissile* m = &e->msl[k];
      if (!m->in_sims || m->sim_pos != pos) continue;
      if (m->status == OR_MSL_HIT || m->status == OR_MSL_MISS) continue;
      for (int q = 0; q < e->n_chaff; q++) {
        if (e->chaff[q].status == 1) continue;
        double d[3] = {e->chaff[q].pos[0] - m->position[0], e->chaff[q].pos[1] - m->position[1], e->chaff[q].pos[2] - m->position[2]};
        if (norm3(d) <= 300) {
          int local = 0;
          for (int z = 0; z < q; z++) if (e->chaff[z].parent == e->chaff[q].parent) local++;
          if (chaff_uniform(e, e->sub_tick, m->parent, m->key % 100, e->chaff[q].parent, local) < 0.85) m->status = OR_MSL_MISS;
        }
      }
    }
}
void or_env_run_projectiles(OrEnv* e, int substeps) {
  for (int i = 0; i < e->cfg.n_aircraft; i++) /* AircraftSimulator.run: bloods <= 0 -> shotdown (simulatior.py:220-222) */
    if (e->ac[i].status == OR_ALIVE && e->ac[i].bloods <= 0) e->ac[i].status = OR_SHOTDOWN;
  for (int s = 0; s < substeps; s++) run_projectiles_once(e, 1.0 / e->cfg.sim_freq);
}
void or_env_task_step(OrEnv* e) { task_step(e); }

/* ------------------------------------------------------------------ reset / step */
void or_env_reset(OrEnv* e, double* obs) {
  const OrEnvConfig* c = &e->cfg;
  e->current_step = 0;
  e->n_msl = 0; e->n_sim_keys = 0; e->n_chaff = 0; e->sub_tick = 0; e->chaff_draws = 0;
  if (c->task == OR_TASK_HEADING) { /* singlecontrol_env.py:24-49 */
    double hdg = or_env_uniform(e, 0., 180.), alt = or_env_uniform(e, 14000., 30000.), u = or_env_uniform(e, 400., 1200.);
    F16Init ic = c->init[0];
    ic.psi_deg = clampd(0, hdg, 360); ic.h_sl_ft = clampd(-1400, alt, 85000); ic.u_fps = u;
    aircraft_reload(e, &e->ac[0], &ic, 0);
    e->ac[0].target_heading_deg = clampd(0, hdg, 360);
    e->ac[0].target_altitude_ft = clampd(-1400, alt, 85000);
    e->ac[0].target_velocities_u_mps = clampd(-700, u * 0.3048, 700);
    e->ac[0].heading_check_time = 0;
    e->heading_turn_counts = 0;
  } else {
    for (int i = 0; i < c->n_aircraft; i++) aircraft_reload(e, &e->ac[i], &c->init[i], c->num_missiles[i]);
  }
  or_env_task_reset(e);
  get_obs(e, obs);
}

/* task.reset(env): bookkeeping + reward-function resets (task_base.py:54-62 and the task subclasses) */
void or_env_task_reset(OrEnv* e) {
  const OrEnvConfig* c = &e->cfg;
  for (int i = 0; i < c->n_aircraft; i++) {
    OrAircraft* a = &e->ac[i];
    a->die_flag = 0;
    a->last_shoot_time = -c->min_attack_interval;
    a->remaining_missiles = c->num_missiles[i];   /* singlecombat_with_missile_task.py:101-106, multiplecombat_with_missile_task.py:119-125 */
    a->lock_n = (int)(1 / ((double)c->agent_interaction_steps / c->sim_freq));
    if (a->lock_n > 16) a->lock_n = 16;
    a->lock_pos = 0;
    a->shoot_action = 0;
    a->last_missile = -1;
    a->last_chaff = -1;
    a->rem_gun = a->rem_9m = a->rem_120b = a->rem_chaff = c->num_missiles[i];
    for (int k = 0; k < 4; k++) a->shoot4[k] = 0;
    for (int k = 0; k < 128; k++) a->rnn[k] = 0;   /* singlecombat_task.py:258-262; BaselineAgent.reset (baseline.py:37-38,133-136) */
    a->man_step = 0; a->man_init_set = 0; a->man_init_heading = 0;
    for (int k = 0; k < 4; k++) a->low_action[k] = 0;
  }
  reward_reset(e);
}

static void decode_action(const OrEnv* e, int i, const double* act, double out[4]) {
  (void)i;
  if (e->cfg.task == OR_TASK_HEADING) { /* heading_task.py:102-110 */
    out[0] = act[0] * 2. / (41 - 1.) - 1.; out[1] = act[1] * 2. / (41 - 1.) - 1.; out[2] = act[2] * 2. / (41 - 1.) - 1.;
    out[3] = act[3] * 0.5 / (30 - 1.) + 0.4;
  } else { /* singlecombat_task.py:141-153 */
    out[0] = act[0] / 20 - 1.; out[1] = act[1] / 20 - 1.; out[2] = act[2] / 20 - 1.; out[3] = act[3] / 58 + 0.4;
  }
}

void or_env_step(OrEnv* e, const double* actions, double* obs, double* rew, uint8_t* done, int32_t* info) {
  const OrEnvConfig* c = &e->cfg;
  e->current_step += 1;
  double pre_obs[OR_MAX_AC * 64];
  if (c->hierarchical) get_obs(e, pre_obs);   /* normalize_action reads get_obs of the state BEFORE the step (singlecombat_task.py:231) */
  for (int i = 0; i < c->n_aircraft; i++) {
    const double* act = actions + i * e->act_dim;
    double u[4];
    if (c->hierarchical) {
      /* HierarchicalSingleCombatTask.normalize_action (singlecombat_task.py:223-256): [3,5,3] -> 12 controller inputs -> 4 indices */
      static const double d_alt[3] = {0.1, 0, -0.1}, d_vel[3] = {0.05, 0, -0.05};
      static const double d_hdg[5] = {-M_PI / 6, -M_PI / 12, 0, M_PI / 12, M_PI / 6};
      double x[12], low[4];
      const int scripted = c->use_baseline && e->ac[i].team == 1;
      if (scripted) {
        /* singlecombat_task.py:224-228 / scenario1_task.py:41-49 / scenario2_task.py:49-58: enemy k is flown by baseline agent k,
         * which chases aircraft k of the whole list (PursueAgent.get_action(env, task, idx)) */
        double dv[3];
        if (c->use_baseline == 2) or_maneuver_delta(&e->ac[i], 30.0, (double)c->agent_interaction_steps / c->sim_freq, dv);
        else or_pursue_delta(&e->ac[i], &e->ac[i - c->n_ego], dv);
        baseline_observation(&e->ac[i], dv, x);
      } else {
        x[0] = (e->ac[i].geodetic[2] < 3500) ? d_alt[0] : d_alt[(int)act[0]];   /* :235-239: below 3500 m always climb */
        x[1] = d_hdg[(int)act[1]];
        x[2] = d_vel[(int)act[2]];
        for (int k = 0; k < 9; k++) x[3 + k] = pre_obs[i * e->obs_dim + k];
      }
      for (int k = 0; k < 12; k++) e->ac[i].ctl_in[k] = x[k];
      {
        double lg[153];
        static const int off[5] = {0, 41, 82, 123, 153};
        or_actor_forward(x, e->ac[i].rnn, e->ac[i].low_action, lg);
        for (int hd = 0; hd < 4; hd++) {   /* best minus second best of the head */
          double best = lg[off[hd] + e->ac[i].low_action[hd]], second = -1e300;
          for (int j = off[hd]; j < off[hd + 1]; j++) if (j != off[hd] + e->ac[i].low_action[hd] && lg[j] > second) second = lg[j];
          e->ac[i].ctl_gap[hd] = best - second;
        }
      }
      for (int k = 0; k < 4; k++) low[k] = e->ac[i].low_action[k];
      if (scripted) { for (int k = 0; k < 4; k++) e->ac[i].shoot4[k] = c->use_artillery ? 1 : 0; }
      else if (c->task == OR_TASK_SCENARIO_NVN || (c->task == OR_TASK_SCENARIO1 && e->ac[i].team == 0))
        for (int k = 0; k < 4; k++) e->ac[i].shoot4[k] = act[3 + k] != 0;
      if (c->task == OR_TASK_SHOOT_MISSILE) e->ac[i].shoot_action = act[3] != 0;   /* self._shoot_action[agent_id] = action[-1] (:228) */
      decode_action(e, i, low, u);
      f16_set_controls(&e->ac[i].fdm, u[0], u[1], u[2], u[3]);
      continue;
    }
    if (c->task == OR_TASK_SHOOT_MISSILE) e->ac[i].shoot_action = act[4] != 0; /* :182-184 */
    /* Scenario1 only refreshes the ego team's weapon bits (scenario1_task.py:44-45); the NvN tasks refresh both (scenario2_task.py:58-61) */
    if (c->task == OR_TASK_SCENARIO_NVN || (c->task == OR_TASK_SCENARIO1 && e->ac[i].team == 0))
      for (int k = 0; k < 4; k++) e->ac[i].shoot4[k] = act[4 + k] != 0;
    decode_action(e, i, act, u);
    f16_set_controls(&e->ac[i].fdm, u[0], u[1], u[2], u[3]);
  }
  double dt = 1.0 / c->sim_freq;
  for (int s = 0; s < c->agent_interaction_steps; s++) { /* env_base.py:139-154 */
    for (int i = 0; i < c->n_aircraft; i++) aircraft_run(e, &e->ac[i]);
    run_projectiles_once(e, dt);
  }
  task_step(e);
  or_env_evaluate(e, obs, rew, done, info);
}

/* tail of BaseEnv.step (env_base.py:157-173): observations, then terminations for every agent, then rewards */
void or_env_evaluate(OrEnv* e, double* obs, double* rew, uint8_t* done, int32_t* info) {
  const OrEnvConfig* c = &e->cfg;
  int code = OR_DONE_NONE;
  get_obs(e, obs);
  if (nvn_env(c)) { /* MultipleCombatEnv.step, multiplecombat_env.py:160-182: rewards, team mean, then dones */
    double sum[2] = {0, 0}; int cnt[2] = {0, 0};
    for (int i = 0; i < c->n_aircraft; i++) { rew[i] = get_reward(e, i); sum[e->ac[i].team] += rew[i]; cnt[e->ac[i].team]++; }
    for (int i = 0; i < c->n_aircraft; i++) rew[i] = sum[e->ac[i].team] / cnt[e->ac[i].team];
    for (int i = 0; i < c->n_aircraft; i++) done[i] = (uint8_t)get_termination(e, i, &code);
    int alld = 1;
    for (int i = 0; i < c->n_aircraft; i++) alld = alld && done[i];
    if (info) { info[0] = e->current_step; info[1] = code; info[2] = e->heading_turn_counts; info[3] = alld; }
    return;
  }
  for (int i = 0; i < c->n_aircraft; i++) done[i] = (uint8_t)get_termination(e, i, &code);
  for (int i = 0; i < c->n_aircraft; i++) rew[i] = get_reward(e, i);
  int all = 1;
  for (int i = 0; i < c->n_aircraft; i++) all = all && done[i];
  if (info) { info[0] = e->current_step; info[1] = code; info[2] = e->heading_turn_counts; info[3] = all; }
}
