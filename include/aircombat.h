/* aircombat.h — C ABI of the MI355X-native vectorised air-combat step().
 *
 * Drop-in boundary for the reference's VecEnv hot path. Each entry point names the reference interface it
 * replaces ("R/" = junghoseong/aircombat-selfplay). Plain pointers and sizes only; the caller owns every buffer.
 * All functions return 0 on success or a negative error code; ac_last_error() gives the message.
 * Blocking unless the name says _async. One handle drives one GPU; handles are not thread-safe.
 *
 * Non-finite state. A NaN / Inf in an aircraft's integrator state (or its reward) terminates that aircraft like the reference's
 * ExtremeState condition (R/envs/JSBSim/core/catalog.py:386-416, whose `>=` tests a NaN would pass unnoticed) and makes every call that
 * completes a step -- ac_step, ac_step_host_wait / ac_step_host, ac_sync, ac_step_timed_device -- return -1 with
 * "JSBSim failed. Non-finite state or reward in env E, agent A" in ac_last_error(), until ac_reset clears it: the reference's
 * RuntimeError("JSBSim failed.") (R/envs/JSBSim/core/simulatior.py:223-225) in place of its pdb NaN trap (R/envs/JSBSim/envs/env_base.py:277-281).
 *
 * Array conventions (E = n_envs, A = n_agents, row-major, agents ordered ego team first then enemy team,
 * exactly like BaseEnv._pack, R/envs/JSBSim/envs/env_base.py:269-283):
 *   actions  float32 [E][A][act_dim]   integer-valued indices, as the runners hand them over (act_dim = ac_act_dim(): control
 *                                      indices, or the [3,5,3] choice (+ weapon bits) when cfg.hierarchical)
 *   obs      float32 [E][A][obs_dim]
 *   rewards  float32 [E][A]
 *   dones    uint8   [E][A]
 *   info     int32   [E][4] = {current_step, done_code, heading_turn_counts, episode_was_reset}
 */
#ifndef AIRCOMBAT_H
#define AIRCOMBAT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AC_MAX_AGENTS 8
#define AC_MAX_MISSILES_PER_AGENT 4

/* task semantics, R/envs/JSBSim/tasks/ */
enum {
  AC_TASK_HEADING = 0,        /* heading_task.py:9-110 HeadingTask under SingleControlEnv (BASELINE C1): 1 aircraft, obs 12, act [41,41,41,30],
                                 randomised reset and UnreachHeading draws from the env's numpy Generator(PCG64), see ac_seed_envs */
  AC_TASK_SINGLECOMBAT = 1,   /* singlecombat_task.py:16-207 SingleCombatTask: obs 15, act [41,41,41,30] */
  AC_TASK_DODGE_MISSILE = 2,  /* singlecombat_with_missile_task.py:12-124 rule-based launch from the lock window, MissilePostureReward: obs 21, act 4 */
  AC_TASK_SHOOT_MISSILE = 3,  /* singlecombat_with_missile_task.py:147-204 learned shoot bit: obs 21, act 5 */
  AC_TASK_SCENARIO1 = 5,      /* scenario1_task.py:11-145 (1v1): gun / AIM-120B / AIM-9M / chaff rules, 11 reward terms; obs 21;
                                 act 8 = [41,41,41,30] + [gun, AIM-9M, AIM-120B, chaff] (low-level control; the controller net is row N1) */
  AC_TASK_SCENARIO_NVN = 6,   /* scenario2_task.py / scenario3_task.py *_NvN (2v2, 4v4) under MultipleCombatEnv.step: obs 9+6A+6, act 8 */
  AC_TASK_WVR = 7,            /* WVR_task.py:10-90 WVRTask (1v1): 15-value observation, unlimited gun, eight reward terms, no SafeReturn; act 4 (or [3,5,3]) */
  AC_TASK_MANEUVER = 8,       /* singlecombat_task.py:264-359 Maneuver_curriculum (1v1): WVR's gun, nine reward terms, the ordinary 1v1 terminations */
  AC_TASK_MULTICOMBAT = 4     /* multiplecombat_task.py:15-151 MultipleCombatTask under MultipleCombatEnv.step (NvN, n_agents 4 or 8):
                                 obs 9+6*(A-1), act [41,41,41,30]; share_obs is obs flattened per env (env_base.py:183-189) */
};
/* AircraftSimulator status, R/envs/JSBSim/core/simulatior.py:93-95 */
enum { AC_ALIVE = 0, AC_CRASH = 1, AC_SHOTDOWN = 2 };
/* which termination condition fired (info['done_condition'] of R/envs/JSBSim/termination_conditions/) */
enum { AC_DONE_NONE = 0, AC_DONE_LOW_ALTITUDE = 1, AC_DONE_EXTREME_STATE = 2, AC_DONE_OVERLOAD = 3, AC_DONE_SHOTDOWN = 4,
       AC_DONE_CRASHED = 5, AC_DONE_MISSION_COMPLETE = 6, AC_DONE_TIMEOUT = 7, AC_DONE_UNREACH_HEADING = 8 };

/* init_state block of a scenario YAML (R/envs/JSBSim/configs/, keys ic_*), defaults of simulatior.py:192-208 */
typedef struct ac_init_state {
  double lon_deg, lat_geod_deg, h_sl_ft, psi_deg, u_fps, v_fps, w_fps, p_rad_sec, q_rad_sec, r_rad_sec;
} ac_init_state_t;

/* Scalars of one scenario YAML as parsed by parse_config (R/envs/JSBSim/utils/utils.py:7-23) */
typedef struct ac_config {
  int32_t task;
  int32_t n_agents;                 /* aircraft per env */
  int32_t n_ego;                    /* first n_ego aircraft are team A */
  int32_t sim_freq;                 /* 60 */
  int32_t agent_interaction_steps;  /* 6 */
  int32_t max_steps;
  double center_lon, center_lat, center_alt;   /* battle_field_center */
  double altitude_limit;            /* m, LowAltitude */
  double acc_limit_x, acc_limit_y, acc_limit_z;
  ac_init_state_t init[AC_MAX_AGENTS];
  int32_t num_missiles[AC_MAX_AGENTS];
  double posture_scale;  int32_t posture_potential;
  double altitude_scale; int32_t altitude_potential;
  double event_scale;    int32_t event_potential;
  double missile_posture_scale;
  double shoot_penalty_scale; int32_t shoot_penalty_potential;
  double alt_safe, alt_danger, alt_kv;
  double max_attack_angle, max_attack_distance; int32_t min_attack_interval;
  int32_t use_artillery;
  /* HeadingTask only: HeadingReward_scale / _potential (reward_function_base.py:14-15), UnreachHeading limits (unreach_heading.py:27-31) */
  double heading_scale; int32_t heading_potential;
  double max_heading_increment, max_altitude_increment, max_velocities_u_increment, check_interval;
  int32_t legacy_obs;               /* Scenario2 / Scenario3 (the non-_NvN classes, scenario2_task.py:14-157): AC_TASK_SCENARIO_NVN rules with the 21-value
                                       observation of MultipleCombatShootMissileTask against the enemy of the same team index */
  int32_t rwr;                      /* *_RWR variants of the scenario tasks: obs_dim + 2 reserved zero slots; Scenario1_RWR also blanks the
                                       missile block of its observation (scenario1_task.py:213-314, scenario2_task.py:385-476) */
  int32_t use_baseline;             /* scripted enemy team (`use_baseline: true`, `baseline_type`, singlecombat_task.py:19-27, model/baseline.py):
                                       0 none, 1 PursueAgent, 2 ManeuverAgent('triangle'); the enemy rows of `actions` are ignored */
  int32_t hierarchical;             /* Hierarchical* / Scenario* tasks as shipped: actions are MultiDiscrete [3,5,3] (+ the four weapon
                                       bits) and go through the low-level controller (singlecombat_task.py:209-262); 0 = control indices */
  int32_t approach;                 /* AC_TASK_HEADING only: ApproachTask (`task: approach`, tasks/approach_task.py:9-120): the same env, reset
                                       draws and observation, reward = AltitudeReward alone, terminations LowAltitude, ExtremeState,
                                       Overload, Timeout (no UnreachHeading: the targets stay at their reset values) */
} ac_config_t;

typedef struct ac_env ac_env_t;

/* Number of doubles in the per-aircraft state vector of ac_get_state / ac_set_state, and the field names. */
#define AC_STATE_LEN 128
const char* ac_state_field_name(int i);

/* replaces SubprocVecEnv.__init__ (R/envs/env_wrappers.py:231-267): builds E envs on one GPU, runs every
 * aircraft's initial-condition pass (AircraftSimulator.reload, simulatior.py:152-190) on the device */
int ac_create(const ac_config_t* cfg, int32_t n_envs, int32_t device_id, uint64_t seed, ac_env_t** out);
/* replaces SubprocVecEnv.close (env_wrappers.py:300-310) */
int ac_destroy(ac_env_t* h);
int ac_obs_dim(const ac_env_t* h);
int ac_act_dim(const ac_env_t* h);
int ac_num_envs(const ac_env_t* h);
int ac_num_agents(const ac_env_t* h);

/* replaces SubprocVecEnv.reset (env_wrappers.py:284-290) -> obs[E][A][obs_dim] into a HOST buffer */
int ac_reset(ac_env_t* h, float* obs);
/* replaces SubprocVecEnv.step = step_async + step_wait (env_wrappers.py:269-282) with HOST buffers.
 * Envs whose agents are all done are reset inside the call and return the reset observation with the
 * terminal reward/done, like worker() does (env_wrappers.py:191-204). */
int ac_step(ac_env_t* h, const float* actions, float* obs, float* rewards, uint8_t* dones, int32_t* info);

/* Zero-copy form of the same step for the VecEnv shim: the library owns up to AC_HOST_SETS sets of page-locked host buffers that are mapped into
 * the device (rows padded to a multiple of 64 aircraft); the step kernel reads the actions of set `set` straight from host memory
 * and writes obs / rewards / dones / info of the step into the same set -- no copy commands. The caller fills the action buffer,
 * calls ac_step_host_async (= SubprocVecEnv.step_async, R/envs/env_wrappers.py:269-273) and ac_step_host_wait (= step_wait,
 * :275-282); alternating the two sets keeps the arrays of one step valid while the next one runs. The device buffers of
 * ac_device_buffers are written as well. ac_reset(h, obs) may be pointed at a set's obs buffer. */
/* `info` of a set is ONE packed word per env (the four-word rows of ac_step / ac_device_buffers are 11 % of a step's bytes across PCIe
 * otherwise): current_step in bits 0-15, done_code in bits 16-23, heading_turn_counts in bits 24-30, episode_was_reset in bit 31. */
#define AC_INFO_STEP(w) ((int32_t)((uint32_t)(w) & 0xFFFFu))
#define AC_INFO_DONE_CODE(w) ((int32_t)(((uint32_t)(w) >> 16) & 0xFFu))
#define AC_INFO_TURN_COUNTS(w) ((int32_t)(((uint32_t)(w) >> 24) & 0x7Fu))
#define AC_INFO_WAS_RESET(w) ((int32_t)((uint32_t)(w) >> 31))
#define AC_HOST_SETS 8   /* sets 0 .. 7, each allocated by its first ac_host_buffers call */
int ac_host_buffers(ac_env_t* h, int32_t set, float** actions, float** obs, float** rewards, uint8_t** dones, int32_t** info);
/* SubprocVecEnv.step_wait hands the caller arrays it owns for good (np.stack, env_wrappers.py:276-282). The shim gets the same without a
 * copy by handing out a set's arrays only while nobody holds that set's previous ones; a set still held when the VecEnv closes is
 * detached (ac_destroy no longer frees it, the handle can no longer step into it) and freed by its holder with ac_host_set_free. */
int ac_host_set_detach(ac_env_t* h, int32_t set);
void ac_host_set_free(void* actions, void* obs, void* rewards, void* dones, void* info);
int ac_step_host_async(ac_env_t* h, int32_t set);
int ac_step_host_wait(ac_env_t* h);
int ac_step_host(ac_env_t* h, int32_t set);   /* both in one call: VecEnv.step (env_wrappers.py:30-42) */

/* Device-resident variant of the same step: d_actions is a DEVICE pointer (or NULL to use the
 * handle's own action buffer); results stay in the handle's device buffers; asynchronous on the handle's stream. */
int ac_step_async_device(ac_env_t* h, const float* d_actions);
int ac_device_buffers(ac_env_t* h, float** d_actions, float** d_obs, float** d_rewards, uint8_t** d_dones, int32_t** d_info);
void* ac_stream(ac_env_t* h);   /* hipStream_t the kernels are launched on (created non-blocking: NOT ordered against other streams) */
int ac_sync(ac_env_t* h);
/* Ordering against the caller's streams without a host sync: ac_order_after makes the steps launched from now on wait for the work
 * already queued on `producer_stream` (the policy that wrote the actions); ac_order_before makes `consumer_stream` wait for the
 * steps launched so far (whoever reads obs / rewards / dones next). NULL = the device's default stream. */
int ac_order_after(ac_env_t* h, void* producer_stream);
int ac_order_before(ac_env_t* h, void* consumer_stream);

/* test/render access, mirrors env.agents[uid] property reads and env.agents[uid].crash() (R/tests/test_jsbsim.py:147-186) */
int ac_get_state(ac_env_t* h, int32_t env, int32_t agent, double* out /* [AC_STATE_LEN] */);
int ac_set_state(ac_env_t* h, int32_t env, int32_t agent, const double* in /* [AC_STATE_LEN] */);
int ac_set_status(ac_env_t* h, int32_t env, int32_t agent, int32_t status);
/* lon deg, lat deg, alt m, roll, pitch, yaw rad, vN, vE, vDown m/s, N, E, U m  (BaseSimulator getters, simulatior.py:47-61) */
int ac_get_entity(ac_env_t* h, int32_t env, int32_t agent, double out[12]);
/* missile k of an agent: status, N,E,U, vN,vE,vU, theta, psi, t, mass, model (0 AIM-9L, 1 AIM-120B, 2 AIM-9M) (MissileSimulator, simulatior.py:393-608) */
int ac_get_missile(ac_env_t* h, int32_t env, int32_t agent, int32_t k, double out[12]);

/* Order-independent 64-bit digest of all aircraft states on the device (sum over aircraft of a per-field hash): E envs in the
 * same state give E times the digest of one env (mod 2^64). Test / profiling aid: one read-only pass over the state arrays with
 * the step kernel's access pattern, (4*63 + 4*12 + 8*3) bytes per aircraft; no reference counterpart. */
int ac_state_checksum(ac_env_t* h, uint64_t* out);

/* Number of munitions with status LAUNCHED over the whole handle (len of the live entries of env._tempsims, R/envs/JSBSim/envs/env_base.py:142-143,
 * summed over envs): the bench prices SURVEY 8(d)'s 192 algorithmic bytes per live missile-step with it. 0 for the tasks without munitions. */
int ac_munitions_in_flight(ac_env_t* h, int32_t* count);

/* AC_TASK_HEADING: replaces env.seed(seed) -> gymnasium seeding.np_random(seed) (R/envs/JSBSim/envs/env_base.py:252-258): the four
 * 64-bit words (state_hi, state_lo, inc_hi, inc_lo) of numpy's PCG64 bit generator for every env, [E][4]. The reference seeds env i
 * with seed + 1000 i (scripts/train/train_jsbsim.py:33); resets and UnreachHeading then draw exactly numpy's stream on the device. */
int ac_seed_envs(ac_env_t* h, const uint64_t* states);
/* test access: sim_time, target heading deg / altitude ft / speed m/s, next check time, heading_turn_counts, last p, last q */
int ac_get_heading_state(ac_env_t* h, int32_t env, double out[8]);

/* Optional: page-lock a caller-owned host buffer that is handed to ac_step / ac_reset repeatedly, so that the copies run as
 * direct DMA instead of through a staging buffer (the caller still owns the memory; unpin before freeing it). The reference has
 * no counterpart: its workers pickle arrays through pipes (env_wrappers.py:182-229). */
int ac_pin_host_buffer(ac_env_t* h, void* ptr, int64_t bytes);
int ac_unpin_host_buffer(ac_env_t* h, void* ptr);

/* Low-level controller of the hierarchical tasks: replaces BaselineActor() + load_state_dict(baseline_model.pt) of
 * HierarchicalSingleCombatTask.__init__ (R/envs/JSBSim/tasks/singlecombat_task.py:211-219). `weights` = the 137753 float32 of
 * aircombat-selfplay_amd/data/baseline_actor.f32 (layout in tools/export_baseline_actor.py). Must be called once before the first
 * step of a handle created with cfg.hierarchical. */
int ac_load_controller(ac_env_t* h, const float* weights, int64_t n);
/* test access to _inner_rnn_states[agent] (float[128]) and the controller's last output (float[act_low]: 4 control indices (+ bits)) */
int ac_get_controller_state(ac_env_t* h, int32_t env, int32_t agent, float* hidden, float* low_action);
int ac_set_controller_state(ac_env_t* h, int32_t env, int32_t agent, const float* hidden);
/* Host-side check of the arithmetic behind the controller's GEMMs (no GPU, no handle): every fp32 weight and activation is taken apart
 * into two fp16 pieces, hi = fp16(x), lo = fp16(x - hi), |x - hi - lo| <= 2^-22 |x|, and the products run on the fp16 matrix path
 * (controller_pieces.hpp). Writes the two pieces of x[0..n) as float32 values (each with at most 11 significant bits). */
int ac_split_f16x2(const float* x, int64_t n, float* hi, float* lo);
/* Device self-test (needs the GPU, no handle): the closed form the NvN kernels use for MissilePostureReward's agent-by-agent walk over its
 * shared remembered missile (missile_posture_reward.py:18-46) against the round-by-round walk, for every combination of agent states
 * of a 2v2 and a 4v4 env. *mismatches receives the number of combinations that differ (0 = the two agree everywhere). */
int ac_selftest_missile_walk(int32_t device_id, int32_t* mismatches);

/* timing helper for the bench: average device milliseconds per step kernel over the last n ac_step* calls, measured
 * with HIP events on the handle's stream */
int ac_timing_begin(ac_env_t* h);
int ac_timing_end(ac_env_t* h, float* total_ms);
/* one device-resident step (as ac_step_async_device + ac_sync) with HIP events around its two kernels: the low-level controller of the
 * hierarchical tasks (0 for the control-index form) and the step kernel, each in milliseconds. For the bench's per-kernel rooflines. */
int ac_step_timed_device(ac_env_t* h, const float* d_actions, float* controller_ms, float* step_ms);

const char* ac_last_error(void);
const char* ac_version(void);

#ifdef __cplusplus
}
#endif
#endif
