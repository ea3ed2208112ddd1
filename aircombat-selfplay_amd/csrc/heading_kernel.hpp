// HeadingTask (BASELINE config C1) on the device: one aircraft per env, randomised reset, target-following reward and the
// UnreachHeading curriculum with the env's own numpy Generator(PCG64).
// Reference: envs/JSBSim/envs/singlecontrol_env.py:24-49 (reset draws heading U(0,180) deg, altitude U(14000,30000) ft, speed
// U(400,1200) ft/s from env.np_random), tasks/heading_task.py:9-110 (obs 12, act [41,41,41,30]),
// reward_functions/heading_reward.py:18-71 + altitude_reward.py, termination_conditions/unreach_heading.py:22-65 then
// ExtremeState, Overload, LowAltitude, Timeout (heading_task.py:20-26), env order of BaseEnv.step (dones, then rewards).
// Included by aircombat.hip.
#pragma once

enum { HD_sim_time, HD_tgt_hdg, HD_tgt_alt, HD_tgt_vel, HD_check_time, NHD };   // fp64 [field][N]
enum { HF_last_p, HF_last_q, HF_pre_heading, NHF };                              // fp32
struct HeadingPtrs {
  double* HD; float* HF; int* HI;          // HI: heading_turn_counts
  unsigned long long* HR;                  // PCG64 state_hi, state_lo, inc_hi, inc_lo  [4][N]
};
struct HeadingCfg {
  ac_init_state_t ic;                      // the YAML's init_state; heading / altitude / speed are overwritten by the draws
  double max_heading_increment, max_altitude_increment, max_velocities_u_increment, check_interval;
  float heading_scale; int heading_pot;
  int approach;                            // ApproachTask (approach_task.py): no HeadingReward, no UnreachHeading, 1v1-style termination order
};

// numpy's PCG64 (128-bit LCG, XSL-RR output) and Generator.uniform -> random() = (next64 >> 11) * 2^-53
struct Pcg { unsigned long long hi, lo, ihi, ilo; };
__device__ __forceinline__ unsigned long long pcg64_next(Pcg& g) {
  const unsigned long long MH = 0x2360ED051FC65DA4ULL, ML = 0x4385DF649FCCF645ULL;
  // state = state * mult + inc  (mod 2^128)
  unsigned long long lo = g.lo * ML;
  unsigned long long hi = __umul64hi(g.lo, ML) + g.hi * ML + g.lo * MH;
  unsigned long long nlo = lo + g.ilo;
  hi += g.ihi + (nlo < lo ? 1ULL : 0ULL);
  g.lo = nlo; g.hi = hi;
  unsigned long long x = hi ^ nlo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}
__device__ __forceinline__ double pcg_uniform(Pcg& g, double lo, double hi) {
  double u = (double)(pcg64_next(g) >> 11) * (1.0 / 9007199254740992.0);
  return lo + (hi - lo) * u;
}
__device__ __forceinline__ float in_range_deg_f(float a) {   // utils.py:106-111 with Python's % semantics
  a = fmodf(a, 360.0f);
  if (a < 0.0f) a += 360.0f;
  if (a > 180.0f) a -= 360.0f;
  return a;
}

struct HeadingState {
  double sim_time, tgt_hdg, tgt_alt, tgt_vel, check_time;
  float last_p, last_q, pre_heading;
  int turn_counts;
  Pcg rng;
};
__device__ __forceinline__ void load_heading(const HeadingPtrs& H, int N, int n, HeadingState& x) {
  x.sim_time = H.HD[HD_sim_time * N + n]; x.tgt_hdg = H.HD[HD_tgt_hdg * N + n]; x.tgt_alt = H.HD[HD_tgt_alt * N + n];
  x.tgt_vel = H.HD[HD_tgt_vel * N + n]; x.check_time = H.HD[HD_check_time * N + n];
  x.last_p = H.HF[HF_last_p * N + n]; x.last_q = H.HF[HF_last_q * N + n]; x.pre_heading = H.HF[HF_pre_heading * N + n];
  x.turn_counts = H.HI[n];
  x.rng.hi = H.HR[0 * N + n]; x.rng.lo = H.HR[1 * N + n]; x.rng.ihi = H.HR[2 * N + n]; x.rng.ilo = H.HR[3 * N + n];
}
__device__ __forceinline__ void store_heading(const HeadingPtrs& H, int N, int n, const HeadingState& x) {
  H.HD[HD_sim_time * N + n] = x.sim_time; H.HD[HD_tgt_hdg * N + n] = x.tgt_hdg; H.HD[HD_tgt_alt * N + n] = x.tgt_alt;
  H.HD[HD_tgt_vel * N + n] = x.tgt_vel; H.HD[HD_check_time * N + n] = x.check_time;
  H.HF[HF_last_p * N + n] = x.last_p; H.HF[HF_last_q * N + n] = x.last_q; H.HF[HF_pre_heading * N + n] = x.pre_heading;
  H.HI[n] = x.turn_counts;
  H.HR[0 * N + n] = x.rng.hi; H.HR[1 * N + n] = x.rng.lo; H.HR[2 * N + n] = x.rng.ihi; H.HR[3 * N + n] = x.rng.ilo;
}

// heading_reward.py:18-71 (raw value; scale / potential applied by the caller)
__device__ __forceinline__ float heading_reward_raw(const Props& pr, const Derived& d, const HeadingState& x, int cur_step, float& p_out, float& q_out) {
  float psi_deg = atan2f(pr.m12, pr.m11) * 57.29577951f;
  if (psi_deg < 0.0f) psi_deg += 360.0f;
  const float d_hdg = clampf(-180.0f, in_range_deg_f((float)x.tgt_hdg - psi_deg), 180.0f);
  const float d_alt = clampf(-40000.0f, ((float)x.tgt_alt - d.h_sl_ft) * f16::kFt2M, 40000.0f);
  const float d_vel = clampf(-1400.0f, (float)x.tgt_vel - pr.ub, 1400.0f);
  const float phi = atan2f(pr.sphi, pr.cphi);
  const float e = -((d_hdg / 5.0f) * (d_hdg / 5.0f) + (d_alt / 15.24f) * (d_alt / 15.24f) + (phi / 0.35f) * (phi / 0.35f) + (d_vel / 24.0f) * (d_vel / 24.0f));
  float reward = __expf(0.25f * e);   // (product of the four Gaussians) ^ (1/4)
  if (cur_step > 1) reward += -fabsf(d.p - x.last_p) - fabsf(d.q - x.last_q);
  p_out = d.p; q_out = d.q;
  return reward;
}
__device__ __forceinline__ void heading_obs(const Props& pr, const Derived& d, const HeadingState& x, float* o) {   // heading_task.py:67-100
  float psi_deg = atan2f(pr.m12, pr.m11) * 57.29577951f;
  if (psi_deg < 0.0f) psi_deg += 360.0f;
  o[0] = clampf(-40000.0f, ((float)x.tgt_alt - d.h_sl_ft) * f16::kFt2M, 40000.0f) / 1000.0f;
  o[1] = clampf(-180.0f, in_range_deg_f((float)x.tgt_hdg - psi_deg), 180.0f) * (f16::kPi / 180.0f);
  o[2] = clampf(-1400.0f, (float)x.tgt_vel - pr.ub, 1400.0f) / 340.0f;
  o[3] = pr.alt_m / 5000.0f;
  o[4] = pr.sphi; o[5] = pr.cphi; o[6] = pr.stht; o[7] = pr.ctht;
  o[8] = pr.ub / 340.0f; o[9] = pr.vb / 340.0f; o[10] = pr.wb / 340.0f; o[11] = pr.vc / 340.0f;
#pragma unroll
  for (int k = 0; k < 12; ++k) o[k] = clampf(-10.0f, o[k], 10.0f);
}

// SingleControlEnv.reset (singlecontrol_env.py:24-49) + HeadingTask.reset: three draws, reload, targets, reward memories
__device__ void heading_reset(const HeadingCfg& hc, const DevCfg& c, const Tab& T, State& s, Derived& d, Task& t, Props& pr, HeadingState& x, float* ob) {
  const double hdg = pcg_uniform(x.rng, 0.0, 180.0), alt = pcg_uniform(x.rng, 14000.0, 30000.0), u = pcg_uniform(x.rng, 400.0, 1200.0);
  ac_init_state_t ic = hc.ic;
  ic.psi_deg = hdg; ic.h_sl_ft = alt; ic.u_fps = u;
  initial_state(ic, T, s, d);
  t = Task{};
  t.bloods = 100.0f; t.status = AC_ALIVE; t.last_missile = -1;
  x.sim_time = 0.0;
  x.tgt_hdg = hdg; x.tgt_alt = alt; x.tgt_vel = fmin(fmax(u * 0.3048, -700.0), 700.0);
  x.check_time = 0.0; x.turn_counts = 0;
  x.last_p = 0.0f; x.last_q = 0.0f; x.pre_heading = 0.0f;
  f16::locate(s, d);
  make_props(s, d, c, pr);
  // RewardFunction.reset (reward_function_base.py:20-32): potential terms seed their memory with one evaluation, in list order
  if (hc.heading_pot && !hc.approach) {
    float p, q;
    x.pre_heading = heading_reward_raw(pr, d, x, 0, p, q) * hc.heading_scale;
    x.last_p = p; x.last_q = q;
  }
  if (c.altitude_pot) t.pre_altitude = altitude_raw(pr, c) * c.altitude_scale;
  heading_obs(pr, d, x, ob);
}

// WPE: waves per SIMD the build is sized for (1: 512 registers, nothing in scratch -- every grid up to 1024 workgroups and every reset launch;
// 2: 256 registers for the saturating grids beyond, where the in-kernel initial-condition procedure of an episode end keeps values in scratch)
template <bool SPLIT, int WPE = 1>
__global__ __launch_bounds__(SPLIT ? 192 : 64, SPLIT ? 1 : WPE) void step_kernel_heading(DevPtrs P, DevCfg c, HeadingPtrs H, HeadingCfg hc, int reset_only) {
  constexpr int OBS = 12;
  __shared__ __attribute__((aligned(16))) float lds_out[64 * OBS];
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  __shared__ __attribute__((aligned(16))) char split_lds[SPLIT ? sizeof(SplitLds) : 16];
  SplitLds& L = *reinterpret_cast<SplitLds*>(split_lds);
  const Tab T{lds_tab};
  const int N = c.N;
  const int l = threadIdx.x & 63;
  const int n = blockIdx.x * 64 + l;
  const bool live = n < N;
  const int nn = live ? n : N - 1;
  State s; Task t; Derived d; Props pr; HeadingState x;
  float ob[OBS];
  TableCopy<SPLIT ? 192 : 64> tc;               // table loads, the action row and the state behind them: one HBM round trip
  tc.issue(P.tab);
  const float4 a4 = load_controls(P.actions + (size_t)nn * c.act_dim, c.act_dim);   // (a reset never reads it: the row is there all the same)
  load_heading(H, N, nn, x);
  // (the one-wave form runs two waves per SIMD on 256 registers each: there the state is asked for once the tables have left the
  // registers, and the neighbouring wave covers the second round trip)
  if (SPLIT && !reset_only) load_state(P.F, P.I, P.D, N, nn, s, t);
  tc.commit(lds_tab);
  if (!SPLIT && !reset_only) load_state(P.F, P.I, P.D, N, nn, s, t);
  if (reset_only) {   // VecEnv.reset(): every env draws a new episode
    heading_reset(hc, c, T, s, d, t, pr, x, ob);
    if (live) {
      store_state(P.F, P.I, P.D, N, n, s, t);
      store_heading(H, N, n, x);
    }
    emit_outputs(P, lds_out, OBS, l, ob, 0.0f, false, 1, 0, 0, 0, 0);
    return;
  }
  t.cur_step += 1;
  // heading_task.py:102-110: a * 2 / (41 - 1) - 1 and a * 0.5 / (30 - 1) + 0.4, then the property bounds (catalog.py:189-197)
  s.da = clampf(-1.0f, a4.x * (2.0f / 40.0f) - 1.0f, 1.0f);
  s.de = clampf(-1.0f, a4.y * (2.0f / 40.0f) - 1.0f, 1.0f);
  s.dr = clampf(-1.0f, a4.z * (2.0f / 40.0f) - 1.0f, 1.0f);
  s.thr = clampf(0.0f, a4.w * (0.5f / 29.0f) + 0.4f, 0.9f);
  if (SPLIT && split_helper_wave(s, t, T, L, l, c.substeps)) return;
  bool have_pose = false;
  int nrun_split = 0;
  const bool split_located = SPLIT && dynamics_wave_ticks(s, t, d, T, L, l, c.substeps, nrun_split);
  for (int k = 0; k < nrun_split; ++k) x.sim_time += 1.0 / 60.0;
  for (int sub = 0; sub < c.substeps && !SPLIT; ++sub) {
    if (t.status == AC_ALIVE) {
      if (t.bloods <= 0.0f) t.status = AC_SHOTDOWN;
      f16::tick<false>(s, d, T);
      x.sim_time += 1.0 / 60.0;   // FGFDMExec::IncrTime: the fp64 running sum the check times are compared with
      have_pose = true;
    }
  }
  if (!split_located) {
    f16::locate(s, d);
    if (!have_pose) f16::body_frame(s, d);
  }
  make_props(s, d, c, pr);
  heading_obs(pr, d, x, ob);

  // ---- terminations, first that fires wins (heading_task.py:20-26)
  bool done = false;
  bool nonfinite = false;
  int code = AC_DONE_NONE;
  {
    // UnreachHeading (unreach_heading.py:22-65): at each check time either give up or draw the next targets
    const double inc_size[15] = {0.2, 0.4, 0.6, 0.8, 1.0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    if (!hc.approach && x.sim_time >= x.check_time) {
      float psi_deg = atan2f(pr.m12, pr.m11) * 57.29577951f;
      if (psi_deg < 0.0f) psi_deg += 360.0f;
      const float d_hdg = clampf(-180.0f, in_range_deg_f((float)x.tgt_hdg - psi_deg), 180.0f);
      if (fabsf(d_hdg) > 10.0f) { done = true; code = AC_DONE_UNREACH_HEADING; }
      else {
        const double delta = inc_size[min(x.turn_counts, 14)];
        const double dh = pcg_uniform(x.rng, -delta, delta) * hc.max_heading_increment;
        const double da = pcg_uniform(x.rng, -delta, delta) * hc.max_altitude_increment;
        const double dv = pcg_uniform(x.rng, -delta, delta) * hc.max_velocities_u_increment;
        double nh = fmod(x.tgt_hdg + dh + 360.0, 360.0);
        if (nh < 0.0) nh += 360.0;
        x.tgt_hdg = fmin(fmax(nh, 0.0), 360.0);
        x.tgt_alt = fmin(fmax(x.tgt_alt + da, -1400.0), 85000.0);
        x.tgt_vel = fmin(fmax(x.tgt_vel + dv, -700.0), 700.0);
        x.check_time = fmin(fmax(x.check_time + hc.check_interval, 0.0), 1000000.0);
        x.turn_counts += 1;
      }
    }
    const float np_max = fmaxf(fabsf(s.npx), fmaxf(fabsf(s.npy), fabsf(s.npz)));
    const float pqr = sqrtf(d.p * d.p + d.q * d.q + d.r * d.r);
    nonfinite = nonfinite_probe(d.veci, pqr, d.h_sl_ft, np_max);
    if (!done) {
      const bool extreme = nonfinite || (d.veci >= 1e10f) || (pqr >= 1000.0f) || (d.h_sl_ft >= 1e10f) || (np_max > 10.0f);
      const bool overload = (s.ticks >= kTickOverload) && (fabsf(s.npx) > c.acc_x || fabsf(s.npy) > c.acc_y || fabsf(s.npz + 1.0f) > c.acc_z);
      const bool low = pr.alt_m <= c.altitude_limit;
      if (hc.approach && low) { t.status = AC_CRASH; code = AC_DONE_LOW_ALTITUDE; done = true; }   // approach_task.py:23-28: LowAltitude first
      else if (extreme) { t.status = AC_CRASH; code = AC_DONE_EXTREME_STATE; done = true; }
      else if (overload) { t.status = AC_CRASH; code = AC_DONE_OVERLOAD; done = true; }
      else if (low) { t.status = AC_CRASH; code = AC_DONE_LOW_ALTITUDE; done = true; }
      else if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; }
    }
  }
  // ---- rewards: HeadingReward + AltitudeReward (heading_task.py:14-18), BaseTask.get_reward (no death latch)
  float reward;
  {
    float r_h = 0.0f;
    if (!hc.approach) {
      float p, q;
      r_h = heading_reward_raw(pr, d, x, t.cur_step, p, q);
      x.last_p = p; x.last_q = q;
      r_h = potential(r_h, hc.heading_scale, hc.heading_pot, x.pre_heading);
    }
    reward = r_h + potential(altitude_raw(pr, c), c.altitude_scale, c.altitude_pot, t.pre_altitude);
  }
  const int step_out = t.cur_step, turns_out = x.turn_counts;
  if (done) heading_reset(hc, c, T, s, d, t, pr, x, ob);   // worker auto-reset (env_wrappers.py:191-204): the reset observation goes out
  if (live) {
    store_state(P.F, P.I, P.D, N, n, s, t);
    store_heading(H, N, n, x);
  }
  emit_outputs(P, lds_out, OBS, l, ob, poison_if(nonfinite, reward), done, 1, step_out, code, turns_out, done ? 1 : 0);
}
