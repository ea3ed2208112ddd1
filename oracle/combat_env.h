/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU (plain C, float64) restatement of the reference's per-env step orchestration:
 *   envs/JSBSim/envs/env_base.py:98-173 (reset/step), envs/JSBSim/core/simulatior.py (aircraft wrapper,
 *   MissileSimulator), envs/JSBSim/tasks/{heading_task,singlecombat_task,singlecombat_with_missile_task}.py,
 *   envs/JSBSim/reward_functions/, envs/JSBSim/termination_conditions/, envs/JSBSim/utils/utils.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never does.
 * The Python parts restated here are pinned by tests/golden/ vectors generated from the reference's own
 * Python (see tests/golden/make_golden.py); the FDM underneath is "parity unpinned" (see f16_fdm.h).
 */
#ifndef ORACLE_COMBAT_ENV_H
#define ORACLE_COMBAT_ENV_H
#include <stdint.h>
#include "f16_fdm.h"
#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAX_AC 8
#define OR_MAX_MSL 64

enum { OR_TASK_HEADING = 0, OR_TASK_SINGLECOMBAT = 1, OR_TASK_DODGE_MISSILE = 2, OR_TASK_SHOOT_MISSILE = 3,
       OR_TASK_MULTICOMBAT = 4 /* MultipleCombatTask (multiplecombat_task.py:15-151) under MultipleCombatEnv.step */,
       OR_TASK_SCENARIO1 = 5   /* Scenario1 weapon rules + 11 rewards, 1v1 (scenario1_task.py:11-145), low-level control */,
       OR_TASK_SCENARIO_NVN = 6 /* Scenario2_NvN / Scenario3_NvN (scenario2_task.py:14-316), low-level control */,
       OR_TASK_WVR = 7         /* WVRTask gun-only 1v1 (WVR_task.py:10-90), low-level control */,
       OR_TASK_MANEUVER = 8    /* Maneuver_curriculum (singlecombat_task.py:264-359): WVR's gun, nine reward terms, ordinary 1v1 terminations */ };
#define OR_MAX_CHAFF 64
enum { OR_ALIVE = 0, OR_CRASH = 1, OR_SHOTDOWN = 2 };
enum { OR_MSL_INACTIVE = -1, OR_MSL_LAUNCHED = 0, OR_MSL_HIT = 1, OR_MSL_MISS = 2 };
/* done reason codes written to info (first condition that fired for the LAST agent evaluated, like info['done_condition']) */
enum { OR_DONE_NONE = 0, OR_DONE_LOW_ALTITUDE = 1, OR_DONE_EXTREME_STATE = 2, OR_DONE_OVERLOAD = 3, OR_DONE_SHOTDOWN = 4,
       OR_DONE_CRASHED = 5, OR_DONE_MISSION_COMPLETE = 6, OR_DONE_TIMEOUT = 7, OR_DONE_UNREACH_HEADING = 8 };

typedef struct {
  int task;
  int n_aircraft;                  /* aircraft in the env; first n_ego are team A (ego), the rest team B */
  int n_ego;
  int sim_freq;                    /* 60 */
  int agent_interaction_steps;     /* 6 */
  int max_steps;
  double center_lon, center_lat, center_alt;
  double altitude_limit;           /* m */
  double acc_limit_x, acc_limit_y, acc_limit_z;
  F16Init init[OR_MAX_AC];
  int num_missiles[OR_MAX_AC];
  /* rewards (reward_function_base.py:14-15): scale / potential per term */
  double posture_scale; int posture_potential;
  double altitude_scale; int altitude_potential;
  double event_scale; int event_potential;
  double heading_scale; int heading_potential;
  double missile_posture_scale;
  double shoot_penalty_scale; int shoot_penalty_potential;
  double alt_safe, alt_danger, alt_kv;       /* AltitudeReward */
  /* rule-based launch (singlecombat_with_missile_task.py:19-21) */
  double max_attack_angle, max_attack_distance; int min_attack_interval;
  /* heading task (unreach_heading.py:27-31) */
  double max_heading_increment, max_altitude_increment, max_velocities_u_increment, check_interval;
  int use_artillery;
  double relative_altitude_scale, relative_altitude_KH;   /* RelativeAltitudeReward_scale / _KH */
  double gun_scale;                                       /* common scale of the CombatGeometry / Gun* terms (all default 1) */
  uint64_t chaff_seed;                                    /* counter-based stand-in for the global np.random of env_base.py:153 */
  int legacy_obs;     /* Scenario2 / Scenario3 (not _NvN): MultipleCombatShootMissileTask's 21-value observation against the enemy with the
                         same index in its team (multiplecombat_with_missile_task.py:30-117) */
  int rwr;            /* *_RWR task variants: two reserved zero slots appended to the observation; Scenario1_RWR also blanks its missile block */
  int use_baseline;   /* the enemy team is flown by a scripted BaselineAgent (singlecombat_task.py:19-27): 0 none, 1 pursue, 2 maneuver('triangle') */
  int hierarchical;   /* Hierarchical* / Scenario* tasks as shipped: action = [3,5,3] (+4 weapon bits) through the low-level controller */
  int approach;       /* OR_TASK_HEADING only: ApproachTask (approach_task.py:9-120) = the heading env without HeadingReward and without
                         UnreachHeading; terminations LowAltitude, ExtremeState, Overload, Timeout in that order */
} OrEnvConfig;

typedef struct {
  F16State fdm;
  int status;
  double bloods;
  double geodetic[3];   /* lon deg, lat deg, alt m (clipped) */
  double position[3];   /* N, E, U m */
  double posture[3];    /* roll, pitch, yaw rad */
  double velocity[3];   /* vN, vE, vDOWN m/s (clipped) */
  int team;
  /* task bookkeeping */
  int die_flag;
  int remaining_missiles;
  int last_shoot_time;
  int lock_window[16]; int lock_n, lock_pos;   /* deque(maxlen=int(1/time_interval)) */
  int shoot_action;
  int last_missile;     /* index of agent_last_shot_missile, -1 = none */
  /* scenario weapon rules (scenario1_task.py:50-103) */
  int rem_gun, rem_9m, rem_120b, rem_chaff, shoot4[4], last_chaff;
  /* heading task extras */
  double target_heading_deg, target_altitude_ft, target_velocities_u_mps, heading_check_time;
  double last_roll_rate, last_pitch_rate;
  /* reward memory */
  double pre_posture, pre_altitude, pre_event, pre_heading, pre_shoot;
  int pre_remaining_missiles;
  double rnn[128];      /* _inner_rnn_states[agent_id] of the hierarchical tasks (singlecombat_task.py:258-262) */
  int low_action[4];    /* last output of the low-level controller */
  int man_step, man_init_set; double man_init_heading;   /* ManeuverAgent.step / init_heading (baseline.py:133-136) */
  double ctl_in[12];    /* last controller input vector (test hook) */
  double ctl_gap[4];    /* top-two logit gap of each of the four heads at the last controller call (test hook: an argmax that differs on
                           the fp32 device must sit on a near-tie here) */
} OrAircraft;

typedef struct {
  int status;
  int parent, target;
  double geodetic[3], position[3], velocity[3], posture[3];
  double t, m, dtheta, dphi, dist_prev;
  int recede_count;     /* consecutive "distance increased" samples (deque of maxlen int(5/dt)) */
  int recede_len, recede_max;
  int key, sim_pos, in_sims, model;   /* env._tempsims bookkeeping: dict key (uid), dict position, still in the dict */
  /* parameters (simulatior.py:421-433) */
  double g, t_max, t_thrust, Isp, Length, Diameter, cD, m0, dm, K, nyz_max, Rc, v_min;
} OrMissile;

typedef struct {
  OrEnvConfig cfg;
  OrAircraft ac[OR_MAX_AC];
  OrMissile msl[OR_MAX_MSL];
  int n_msl;
  int n_sim_keys;
  struct { double pos[3]; double t; int status; int parent; } chaff[OR_MAX_CHAFF];
  int n_chaff;
  uint64_t chaff_draws;
  int sub_tick;          /* projectile substeps since reset (key of the decoy draw) */
  /* reward memories shared by every agent of the env (CombatGeometry / GunWEZDOT / GunTargetTail keep module-level lists) */
  int cg_set; double cg_AO, cg_TA;
  int wezdot_set; double wezdot_ref[OR_MAX_AC];
  int tail_set; double tail_ref[OR_MAX_AC];
  int current_step;
  int heading_turn_counts;
  /* MissilePostureReward shared memory: index of the missile whose velocity array was aliased, -1 = None */
  int mp_prev_missile;
  /* numpy Generator(PCG64) state mirror for env.np_random */
  unsigned __int128 rng_state, rng_inc;
  int obs_dim, act_dim;
} OrEnv;

int or_env_obs_dim(int task);   /* for OR_TASK_MULTICOMBAT use or_env_obs_dim_n */
int or_env_obs_dim_n(int task, int n_aircraft);
int or_env_act_dim(int task);
int or_env_act_dim_h(int task, int hierarchical);
void or_env_default_config(OrEnvConfig* c, int task);
void or_env_init(OrEnv* e, const OrEnvConfig* c);
/* seed with the raw PCG64 state/inc as numpy reports them (np.random.PCG64(seed).state['state']) */
void or_env_seed_pcg64(OrEnv* e, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi, uint64_t inc_lo);
double or_env_uniform(OrEnv* e, double lo, double hi);
void or_env_reset(OrEnv* e, double* obs /* [n_aircraft][obs_dim] */);
void or_env_step(OrEnv* e, const double* actions /* [n_aircraft][act_dim] */, double* obs, double* rew, uint8_t* done,
                 int32_t* info /* [4]: current_step, done_code, heading_turn_counts, all_done */);

void or_env_task_reset(OrEnv* e);
/* low-level controller (lowlevel_actor.c) */
int or_actor_load(const char* path);
int or_actor_loaded(void);
void or_actor_forward(const double* x, double* h, int* act, double* logits);
void or_env_refresh_cache(OrEnv* e, int i);   /* AircraftSimulator._update_properties from the current FDM outputs */
void or_env_evaluate(OrEnv* e, double* obs, double* rew, uint8_t* done, int32_t* info);

/* exposed pieces for golden-vector tests */
double or_posture_orientation(double AO, double TA);
double or_posture_range(double R_km);
void or_get_AO_TA_R(const double ego[6], const double enm[6], int two_d, double out[4]);
double or_posture_reward(double AO, double TA, double R_km);
double or_altitude_reward(double ego_z_km, double ego_vz_mh, double safe, double danger, double kv);
void or_missile_init(OrMissile* m, int model /*0 AIM-9L, 1 AIM-120B parameter set*/, double dt);
void or_missile_launch(OrMissile* m, const double geodetic[3], const double position[3], const double velocity[3], const double rpy[3]);
/* one MissileSimulator.run(); returns new status; target_alive in/out semantics as the reference (HIT => caller shoots down target) */
int or_missile_run(OrMissile* m, const double tgt_pos[3], const double tgt_vel[3], int target_alive, double dt,
                   double lon0, double lat0, double alt0);

#ifdef __cplusplus
}
#endif
#endif
