/* aircombat_buffer.h -- C ABI of the device-resident rollout buffer: the step on the far side of the
 * env path. Replaces, behind the same operations, the numpy arrays of
 *   ReplayBuffer        R/algorithms/utils/buffer.py:26-268
 *   SharedReplayBuffer  R/algorithms/utils/buffer.py:270-448
 * (R = the reference repository). Everything stays in HBM: the env kernels' outputs are inserted device-to-device, the
 * return / GAE recurrence, the advantage normalisation and the mini-batch gather run as HIP kernels, and a policy living on the
 * same GPU reads the batches in place. Same library as aircombat.h (libaircombat_hip.so); errors: functions return 0 on
 * success, -1 on failure with the message in ac_last_error(). Blocking calls; one caller thread per handle.
 *
 * Array layout (all float32, time-major exactly like the reference's numpy arrays, N = n_envs * n_agents columns):
 *   OBS [T+1][N][obs_dim]   SHARE_OBS [T+1][N][share_obs_dim]   ACTIONS [T][N][act_dim]   REWARDS [T][N]
 *   MASKS, BAD_MASKS, ACTIVE_MASKS [T+1][N]   LOGP [T][N][logp_dim]   VALUES, RETURNS [T+1][N]
 *   RNN_ACTOR, RNN_CRITIC [T+1][N][hidden_layers * hidden_size]   ADVANTAGES [T][N] (filled by ac_buffer_advantages)
 */
#ifndef AIRCOMBAT_BUFFER_H
#define AIRCOMBAT_BUFFER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ac_buffer ac_buffer_t;

/* constructor arguments of both reference classes (buffer.py:36-47, :272-285) */
typedef struct {
  int32_t buffer_size;            /* args.buffer_size = T */
  int32_t n_envs;                 /* args.n_rollout_threads */
  int32_t n_agents;
  int32_t obs_dim;
  int32_t share_obs_dim;          /* 0: ReplayBuffer; > 0: SharedReplayBuffer (adds SHARE_OBS, ACTIVE_MASKS; insert ignores bad_masks, :343) */
  int32_t act_dim;                /* get_shape_from_space(act_space) */
  int32_t logp_dim;               /* 1 for ReplayBuffer (:64), act_dim for SharedReplayBuffer (:302) */
  int32_t hidden_layers, hidden_size;
  int32_t use_gae, use_proper_time_limits;
  double gamma, gae_lambda;
} ac_buffer_config_t;

enum {
  AC_BUF_OBS = 0, AC_BUF_SHARE_OBS, AC_BUF_ACTIONS, AC_BUF_REWARDS, AC_BUF_MASKS, AC_BUF_BAD_MASKS, AC_BUF_ACTIVE_MASKS, AC_BUF_LOGP,
  AC_BUF_VALUES, AC_BUF_RETURNS, AC_BUF_RNN_ACTOR, AC_BUF_RNN_CRITIC, AC_BUF_ADVANTAGES, AC_BUF_NFIELDS
};

/* arguments of insert() (buffer.py:77-111, :313-343): one [N][dim] slice per pointer; NULL = argument not given (allowed for
 * bad_masks, share_obs, active_masks). obs / masks / rnn states / share_obs / bad_masks / active_masks go to slot step+1,
 * actions / rewards / action_log_probs / value_preds to slot step. */
typedef struct {
  const float *obs, *actions, *rewards, *masks, *action_log_probs, *value_preds, *rnn_states_actor, *rnn_states_critic;
  const float *bad_masks, *share_obs, *active_masks;
} ac_buffer_step_t;

/* one mini-batch of recurrent_generator (buffer.py:237-268, :418-448): [chunk_len * n_chunks][dim] per-step arrays laid out
 * step-major (row l * n_chunks + j = step l of chunk j) and [n_chunks][hidden_layers * hidden_size] RNN states of each chunk's
 * first step. NULL pointers are skipped. */
typedef struct {
  float *obs, *share_obs, *actions, *masks, *active_masks, *action_log_probs, *advantages, *returns, *value_preds;
  float *rnn_states_actor, *rnn_states_critic;
} ac_buffer_batch_t;

ac_buffer_t* ac_buffer_create(const ac_buffer_config_t* cfg, int device_id);   /* NULL on failure; arrays zero / masks one (:51-69) */
void ac_buffer_destroy(ac_buffer_t* b);

/* insert(); on_device != 0: the pointers are device pointers on the buffer's GPU (e.g. the env handle's obs / reward buffers) */
int ac_buffer_insert(ac_buffer_t* b, const ac_buffer_step_t* step, int on_device);
int ac_buffer_step_index(const ac_buffer_t* b);                               /* self.step */
int ac_buffer_after_update(ac_buffer_t* b);                                   /* buffer.py:113-119, :345-348 */
int ac_buffer_clear(ac_buffer_t* b);                                          /* buffer.py:121-132 */

/* compute_returns(next_value) (buffer.py:134-167), next_value = [N] floats; the recurrence runs in the reference's float32
 * operation order, one lane per (env, agent) column, so the results are bit-identical to the numpy code */
int ac_buffer_compute_returns(ac_buffer_t* b, const float* next_value, int on_device);
/* the `advantages` property (buffer.py:72-75): (returns - values - mean) / (std + 1e-5) over all T*N entries -> AC_BUF_ADVANTAGES */
int ac_buffer_advantages(ac_buffer_t* b);

/* one mini-batch: chunk c covers rows [c*chunk_len, (c+1)*chunk_len) of the column-major sequence view (row = column*T + t,
 * the reference's _cast, buffer.py:33-34); `chunks` are host int32 (torch.randperm(...)[i*mb:(i+1)*mb] in the reference) */
int ac_buffer_minibatch(ac_buffer_t* b, const int32_t* chunks, int32_t n_chunks, int32_t chunk_len, const ac_buffer_batch_t* out, int on_device);

/* raw access: device pointer + float count of a field (for torch views), whole-field read-back, one time slot written from host */
int ac_buffer_device_ptr(ac_buffer_t* b, int32_t field, float** ptr, int64_t* n_floats);
int ac_buffer_read(ac_buffer_t* b, int32_t field, float* host_out);
int ac_buffer_write_slot(ac_buffer_t* b, int32_t field, int32_t t, const float* host_in);

/* bench helper: device milliseconds of the last ac_buffer_compute_returns kernel (HIP events on the buffer's stream) */
int ac_buffer_last_kernel_ms(ac_buffer_t* b, float* ms);

#ifdef __cplusplus
}
#endif
#endif
