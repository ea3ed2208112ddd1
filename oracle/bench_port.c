/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * Timed loop of the CPU restatement for bench.py's cpu_baseline leg ("kind": "port"): E envs behind the reference's
 * VecEnv semantics (auto-reset when every agent of an env is done, envs/env_wrappers.py:191-204), uniform random integer
 * actions regenerated every step, single thread. Returns the agent-steps executed. */
#include "combat_env.h"
#include <stdlib.h>
#include <stdint.h>
#include <time.h>

static uint64_t splitmix(uint64_t* s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

long or_bench_run(const OrEnvConfig* cfg, int n_envs, int steps, uint64_t seed, double* seconds, long* episodes) {
  OrEnv* envs = (OrEnv*)malloc(sizeof(OrEnv) * (size_t)n_envs);
  if (!envs) return -1;
  const int A = cfg->n_aircraft, od = or_env_obs_dim(cfg->task), ad = or_env_act_dim(cfg->task);
  double* obs = (double*)malloc(sizeof(double) * A * od);
  double* act = (double*)malloc(sizeof(double) * A * ad);
  double rew[OR_MAX_AC]; uint8_t done[OR_MAX_AC]; int32_t info[4];
  for (int e = 0; e < n_envs; e++) { or_env_init(&envs[e], cfg); or_env_reset(&envs[e], obs); }
  static const int nvec[5] = {41, 41, 41, 30, 20};
  uint64_t s = seed;
  long eps = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int k = 0; k < steps; k++) {
    for (int e = 0; e < n_envs; e++) {
      for (int i = 0; i < A; i++)
        for (int j = 0; j < ad; j++) {
          uint64_t r = splitmix(&s);
          act[i * ad + j] = (j < 4) ? (double)(r % nvec[j]) : (double)((r % nvec[4]) == 0); /* shoot bit ~ Bernoulli(0.05) */
        }
      or_env_step(&envs[e], act, obs, rew, done, info);
      if (info[3]) { or_env_reset(&envs[e], obs); eps++; }
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  if (episodes) *episodes = eps;
  free(envs); free(obs); free(act);
  return (long)steps * n_envs * A;
}

/* A block of envs stepped by one call (the unit a worker process of oracle/subproc_vec_env.py owns): VecEnv semantics with
 * float32 arrays in the product's layout -- actions [n][A][act_dim], obs [n][A][obs_dim], rewards [n][A], dones [n][A],
 * info [n][4] -- and the worker's auto-reset (envs/env_wrappers.py:191-204). */
typedef struct { int n, A, od, ad; OrEnvConfig cfg; OrEnv* envs; double* obs; double* act; } OrBlock;

OrBlock* or_block_create(const OrEnvConfig* cfg, int n_envs, uint64_t chaff_seed0) {
  OrBlock* b = (OrBlock*)calloc(1, sizeof(OrBlock));
  if (!b) return 0;
  b->n = n_envs; b->A = cfg->n_aircraft; b->cfg = *cfg;
  b->envs = (OrEnv*)malloc(sizeof(OrEnv) * (size_t)n_envs);
  for (int e = 0; e < n_envs; e++) {
    OrEnvConfig c = *cfg;
    c.chaff_seed = chaff_seed0 + (uint64_t)e;
    or_env_init(&b->envs[e], &c);
  }
  b->od = b->envs[0].obs_dim; b->ad = b->envs[0].act_dim;
  b->obs = (double*)malloc(sizeof(double) * b->A * b->od);
  b->act = (double*)malloc(sizeof(double) * b->A * b->ad);
  return b;
}
void or_block_destroy(OrBlock* b) { if (b) { free(b->envs); free(b->obs); free(b->act); free(b); } }
int or_block_obs_dim(const OrBlock* b) { return b->od; }
int or_block_act_dim(const OrBlock* b) { return b->ad; }
void or_block_reset(OrBlock* b, float* obs) {
  const int row = b->A * b->od;
  for (int e = 0; e < b->n; e++) {
    or_env_reset(&b->envs[e], b->obs);
    for (int k = 0; k < row; k++) obs[(size_t)e * row + k] = (float)b->obs[k];
  }
}
void or_block_step(OrBlock* b, const float* actions, float* obs, float* rew, uint8_t* done, int32_t* info) {
  const int A = b->A, row = A * b->od, arow = A * b->ad;
  double r[OR_MAX_AC]; uint8_t d[OR_MAX_AC];
  for (int e = 0; e < b->n; e++) {
    for (int k = 0; k < arow; k++) b->act[k] = (double)actions[(size_t)e * arow + k];
    or_env_step(&b->envs[e], b->act, b->obs, r, d, &info[(size_t)e * 4]);
    if (info[(size_t)e * 4 + 3]) or_env_reset(&b->envs[e], b->obs);
    for (int k = 0; k < row; k++) obs[(size_t)e * row + k] = (float)b->obs[k];
    for (int i = 0; i < A; i++) { rew[(size_t)e * A + i] = (float)r[i]; done[(size_t)e * A + i] = d[i]; }
  }
}
