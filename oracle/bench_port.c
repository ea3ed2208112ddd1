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
