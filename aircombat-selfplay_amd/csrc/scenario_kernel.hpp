// Scenario tasks on the device: Scenario1 (1v1) and Scenario2_NvN / Scenario3_NvN (2v2, 4v4) with the reference's weapon
// rules (gun, AIM-120B, AIM-9M, chaff + decoy) and its eleven reward terms, driven by low-level controls
// (4 control indices + 4 weapon bits; the hierarchical controller net of the shipped YAMLs is SURVEY row N1).
// Reference: envs/JSBSim/tasks/scenario1_task.py:11-145, scenario2_task.py:14-316 (Scenario3 is identical with 8 agents),
// envs/JSBSim/envs/env_base.py:115-173 (1v1 order), multiplecombat_env.py:119-182 (NvN order), core/simulatior.py:327-608,
// reward_functions/*.py. Included by aircombat.hip after the shared helpers.
//
// Lane layout: the A aircraft of an env in A adjacent lanes (ego team first). Every aircraft owns two munition slots — the
// two uids "agent+2", "agent+1" that env._tempsims can hold for it (an AIM-9M launched after an AIM-120B with the same
// remaining-count reuses the uid and replaces the dict entry) — and two chaff release events. What another aircraft
// needs DURING the substeps (target pose and status) is fetched from the owning lane with __shfl; what the aircraft of an env
// exchange after the last substep (poses, munition entries, gun hits, chaff counts, the missile warning) and the chaff clouds go
// through LDS rows of the environment wave (xp_* / cl_* below).
#pragma once

// extension state of the scenario tasks, SoA [field][N]. The counters and flags the reference keeps in Python attributes (remaining rounds per
// weapon, the four weapon bits, the chaff bookkeeping, MissilePostureReward's remembered missile, ...) are 14 small integers: they live bit-packed in
// TWO words per aircraft, and every part of the record is written back only when it has changed -- the words when their value differs from what was
// loaded, the ten shared reward references when they are (re)set, a chaff cloud while it exists -- instead of 32 words in and 32 words out on
// every env step (round 2: 256 of the scenario kernels' ~1400 B per aircraft-step).
enum { XI_w0, XI_w1, NXI };
//  w0: rem_gun 0-3 | rem_9m 4-7 | rem_120b 8-11 | bits 12-15 | mp_prev 16-20 | ref_set 21-23 | orphan_hits 24-27
//  w1: last_chaff + 1 0-1 | n_ch 2-3 | ch_status0 4 | ch_status1 5 | ch_mult0 6-10 | ch_mult1 11-15 | rem_chaff 16-23 (signed: a release event takes
//      one round per qualifying incoming missile and can take the count below zero, scenario1_task.py:97-103)
// (a cloud's age keeps counting after it has dissolved, like ChaffSimulator.run's; nothing reads it then, and it is no longer written back)
enum { XF_cg_AO, XF_cg_TA, XF_wez0, XF_wez1, XF_wez2, XF_wez3, XF_tail0, XF_tail1, XF_tail2, XF_tail3,
       XF_c0x, XF_c0y, XF_c0z, XF_c0t, XF_c1x, XF_c1y, XF_c1z, XF_c1t, NXF };
enum { NXI_UNPACKED = 14 };   // ac_get_state still reports the integers one by one (x_rem_gun ... x_n_ch)

struct Ext {
  int rem_gun, rem_9m, rem_120b, rem_chaff, bits, last_chaff, orphan_hits, mp_prev, ref_set;
  int ch_status[2], ch_mult[2], n_ch;
  float cg_AO, cg_TA, wez[4], tail[4];
  float cx[2], cy[2], cz[2], ct[2];
};
struct ExtLoaded { int w0, w1, ref_set; bool cloud[2]; };   // what the record held when it was loaded (store_ext writes the differences)
__host__ __device__ __forceinline__ int ext_pack0(const Ext& x) {
  return (x.rem_gun & 15) | ((x.rem_9m & 15) << 4) | ((x.rem_120b & 15) << 8) | ((x.bits & 15) << 12) | ((x.mp_prev & 31) << 16) | ((x.ref_set & 7) << 21) |
         ((x.orphan_hits & 15) << 24);
}
__host__ __device__ __forceinline__ int ext_pack1(const Ext& x) {
  return ((x.last_chaff + 1) & 3) | ((x.n_ch & 3) << 2) | ((x.ch_status[0] & 1) << 4) | ((x.ch_status[1] & 1) << 5) | ((x.ch_mult[0] & 31) << 6) | ((x.ch_mult[1] & 31) << 11) |
         ((x.rem_chaff & 255) << 16);
}
__host__ __device__ __forceinline__ void ext_unpack(int w0, int w1, Ext& x) {
  x.rem_gun = w0 & 15; x.rem_9m = (w0 >> 4) & 15; x.rem_120b = (w0 >> 8) & 15;
  x.bits = (w0 >> 12) & 15; x.mp_prev = (w0 >> 16) & 31; x.ref_set = (w0 >> 21) & 7; x.orphan_hits = (w0 >> 24) & 15;
  x.last_chaff = (w1 & 3) - 1; x.n_ch = (w1 >> 2) & 3; x.ch_status[0] = (w1 >> 4) & 1; x.ch_status[1] = (w1 >> 5) & 1;
  x.ch_mult[0] = (w1 >> 6) & 31; x.ch_mult[1] = (w1 >> 11) & 31;
  x.rem_chaff = (int)((unsigned)w1 << 8) >> 24;           // sign-extended 8 bits
}
__device__ __forceinline__ void load_ext(const float* XF, const int* XI, int N, int n, Ext& x, ExtLoaded& was) {
  AC_LANE_INDEX(n);
  was.w0 = AC_AT(XI, XI_w0); was.w1 = AC_AT(XI, XI_w1);
  x.cg_AO = AC_AT(XF, XF_cg_AO); x.cg_TA = AC_AT(XF, XF_cg_TA);
#pragma unroll
  for (int k = 0; k < 4; ++k) { x.wez[k] = AC_AT(XF, XF_wez0 + k); x.tail[k] = AC_AT(XF, XF_tail0 + k); }
  x.cx[0] = AC_AT(XF, XF_c0x); x.cy[0] = AC_AT(XF, XF_c0y); x.cz[0] = AC_AT(XF, XF_c0z); x.ct[0] = AC_AT(XF, XF_c0t);
  x.cx[1] = AC_AT(XF, XF_c1x); x.cy[1] = AC_AT(XF, XF_c1y); x.cz[1] = AC_AT(XF, XF_c1z); x.ct[1] = AC_AT(XF, XF_c1t);
  ext_unpack(was.w0, was.w1, x);
  was.ref_set = x.ref_set;
#pragma unroll
  for (int q = 0; q < 2; ++q) was.cloud[q] = q < x.n_ch && x.ch_status[q] == 0;
}
// `all` : the whole record (reset of an episode / of the handle)
__device__ __forceinline__ void store_ext(float* XF, int* XI, int N, int n, const Ext& x, const ExtLoaded& was, bool all) {
  AC_LANE_INDEX(n);
  const int w0 = ext_pack0(x), w1 = ext_pack1(x);
  if (all || w0 != was.w0) AC_AT(XI, XI_w0) = w0;
  if (all || w1 != was.w1) AC_AT(XI, XI_w1) = w1;
  if (all || x.ref_set != was.ref_set) {   // the shared reward references: written by the first evaluation after a reset, never again
    AC_AT(XF, XF_cg_AO) = x.cg_AO; AC_AT(XF, XF_cg_TA) = x.cg_TA;
#pragma unroll
    for (int k = 0; k < 4; ++k) { AC_AT(XF, XF_wez0 + k) = x.wez[k]; AC_AT(XF, XF_tail0 + k) = x.tail[k]; }
  }
  // a chaff cloud: its release point and its age, while it exists (released this step, or ageing since an earlier one)
  if (all || was.cloud[0] || (x.n_ch > 0 && x.ch_status[0] == 0)) { AC_AT(XF, XF_c0x) = x.cx[0]; AC_AT(XF, XF_c0y) = x.cy[0]; AC_AT(XF, XF_c0z) = x.cz[0]; AC_AT(XF, XF_c0t) = x.ct[0]; }
  if (all || was.cloud[1] || (x.n_ch > 1 && x.ch_status[1] == 0)) { AC_AT(XF, XF_c1x) = x.cx[1]; AC_AT(XF, XF_c1y) = x.cy[1]; AC_AT(XF, XF_c1z) = x.cz[1]; AC_AT(XF, XF_c1t) = x.ct[1]; }
}
__device__ __forceinline__ Ext fresh_ext(int num) {
  Ext x{};
  x.rem_gun = x.rem_9m = x.rem_120b = x.rem_chaff = num;
  x.last_chaff = -1;
  x.ch_status[0] = x.ch_status[1] = 1;  // no cloud
  return x;
}

__device__ __forceinline__ MslParam aim120b() {  // the set both AIM_9M and AIM_120B carry, simulatior.py:659-672,696-709
  constexpr int kb = first_tick_not_below(1.4), kt = first_tick_above(27.22);
  return MslParam{9.81f, 27.22f, 1.4f, 1837.0f, 3.66f, 0.18f, 0.02f, 152.0f, 6.0f, 5.0f, 50.0f, 5.0f, 150.0f, 300, kb, kt};
}
// decoy draw keyed by what is tested (substep, missile = launcher + uid number, chaff = releaser + release index): the
// reference uses the global unseeded np.random (env_base.py:153), so only statistical parity with it is possible; the tests'
// CPU checker implements the same keyed generator so that both sides can be compared draw for draw
__device__ __forceinline__ float decoy_uniform(unsigned long long seed, int tick, int mp, int mn, int cp, int cl) {
  unsigned long long k = ((unsigned long long)(unsigned)tick << 32) | ((unsigned long long)(mp & 0xff) << 24) |
                         ((unsigned long long)(mn & 0xff) << 16) | ((unsigned long long)(cp & 0xff) << 8) | (unsigned long long)(cl & 0xff);
  unsigned long long z = seed * 0x9E3779B97F4A7C15ULL + k * 0xD1B54A32D192ED03ULL;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
constexpr float kFt2M = 0.3048f;

template <int A>
struct ScenarioDims {
  static constexpr bool MULTI = A > 2;
  static constexpr int OBS = MULTI ? 9 + 6 * (A / 2) + 6 * (A / 2) + 6 : 21;
  static constexpr int NE = A / 2;   // enemies per agent (teams are equal-sized in every shipped scenario)
};

// Relative-geometry quantities of every reward term towards the enemies of this lane, in enemy order.
struct EnemyGeo { float AO, TA, R, cAO, cTA; };
// the distances of the two gun-track reward terms to one enemy (scenario1_task.py rewards; sin / cos of the two angles: the angles are
// acos of these very cosines, utils.py:74-77)
__device__ __forceinline__ void gun_track_distances(float R, float cA, float cT, float& dwez, float& dtail) {
  const float r3 = 3000.0f * kFt2M, r5 = 5000.0f * kFt2M;
  const float sA = sqrtf((1.0f - cA) * (1.0f + cA)), sT = sqrtf((1.0f - cT) * (1.0f + cT));
  dwez = (R >= 500.0f * kFt2M && R <= r3) ? R * sA : sqrtf(R * R + r3 * r3 - 2.0f * R * r3 * cA);
  dtail = (R >= r3 && R <= r5) ? R * sT : ((R <= r3) ? sqrtf(R * R + r3 * r3 - 2.0f * R * r3 * cT) : sqrtf(R * R + r5 * r5 - 2.0f * R * r5 * cT));
}

// FORM: FORM_ONE = one wave per 64 aircraft does everything; FORM_SPLIT = the FDM ticks in the three-wave form (split_kernel.hpp; used
// for the gun-only tasks, which have nothing to fly between ticks); FORM_PAIR = a flight wave and an environment wave
// (pair_kernel.hpp; every task with munitions, at every batch size).
// FORM_QUAD = the three waves of FORM_SPLIT flying the ticks under the environment wave of FORM_PAIR (pair_kernel.hpp, "the quad form"),
// up to one workgroup per CU: the environment wave spreads a substep's munition work over the tick's three barrier gaps.
enum { FORM_ONE = 0, FORM_SPLIT = 1, FORM_PAIR = 2, FORM_QUAD = 3 };
// MissilePostureReward's single shared `previous_missile_v` (missile_posture_reward.py:18-46) is walked agent by agent in env order: an
// evaluating agent WITHOUT an incoming missile clears the remembered missile id, one WITH a missile sets it if it is clear and then
// scores against it; agents that do not evaluate are skipped. `prev_mine` = the remembered id at this lane's turn (meaningful for an
// evaluating lane with a missile), `prev_out` = what the env remembers after its last agent (identical in every lane of the env).
struct MpWalk { int prev_mine, prev_out; };
// the walk as the reference makes it, round by round (the self-test's yardstick: ac_selftest)
template <int A>
__device__ __forceinline__ MpWalk mp_walk_sequential(bool evaluates, int inc_id, int mp_prev, int slot, int base) {
  int prev = mp_prev, mine = 0;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const int ev_i = __shfl((int)evaluates, base + i), id_i = __shfl(inc_id, base + i);
    if (!ev_i) continue;
    if (id_i) {
      if (!prev) prev = id_i;
      if (slot == i) mine = prev;
    } else prev = 0;
  }
  return MpWalk{mine, prev};
}
// the same in closed form: the remembered missile at agent i's turn is the incoming missile of the first missile-holding evaluator after
// the last clearing evaluator before i (or the one carried over from the previous step if nobody has cleared it yet) -- two ballots
// and two fetches instead of A rounds of dependent cross-lane fetches
__device__ __forceinline__ MpWalk mp_walk_closed(bool evaluates, int inc_id, int mp_prev, int lane, int base, unsigned long long env_mask) {
  const bool holds = evaluates && inc_id != 0;
  const unsigned long long EV = __ballot(evaluates) & env_mask;
  const unsigned long long HM = __ballot(holds) & env_mask;          // evaluators with an incoming missile
  const unsigned long long NM = EV & ~HM;                            // evaluators without: they clear
  auto after = [](int p) { return p < 0 ? ~0ull : (p >= 63 ? 0ull : ~((2ull << p) - 1ull)); };   // lanes above p (p = -1 .. 63)
  const unsigned long long clr = NM & ~after(lane - 1);              // clearing evaluators before me
  const int last_clr = clr ? 63 - __clzll((long long)clr) : base - 1;
  const unsigned long long run = HM & after(last_clr) & ~after(lane);   // holders after the last clearing one, up to and including me
  const int src = (!clr && mp_prev != 0) ? -1 : (run ? (int)__ffsll((long long)run) - 1 : -2);   // -1: carried over, -2: nothing remembered
  const int src_id = __shfl(inc_id, src >= 0 ? src : lane);
  MpWalk w;
  w.prev_mine = (src == -1) ? mp_prev : (src >= 0 ? src_id : 0);
  w.prev_out = mp_prev;
  if (EV) {   // (env-uniform)
    const int last_ev = 63 - __clzll((long long)EV);
    const int carried = __shfl(w.prev_mine, last_ev);                // (meaningful when that lane holds a missile)
    w.prev_out = ((HM >> last_ev) & 1ull) ? carried : 0;
  }
  return w;
}
// Every combination of (does not evaluate | evaluates without a missile | evaluates with missile id a | ... id b) over the A agents of an
// env, with and without a remembered missile carried in: the closed form against the round-by-round walk. One env per A lanes.
template <int A>
__global__ void mp_walk_selftest_kernel(int* mismatches, unsigned long long combos) {
  const int lane = threadIdx.x & 63, slot = lane % A, base = lane - slot;
  const unsigned long long env = (unsigned long long)blockIdx.x * (64 / A) + lane / A;
  const unsigned long long combo = env < combos ? env : 0;
  const int st = (int)((combo >> (2 * slot)) & 3ull);
  const int carry = (int)((combo >> (2 * A)) & 1ull) ? 3 : 0;
  const bool evaluates = st != 0;
  const int inc_id = st == 0 ? slot + 1 : (st == 1 ? 0 : (st == 2 ? slot + 1 : ((slot + 3) % (2 * A)) + 1));   // (a non-evaluating agent may well have a missile coming)
  const unsigned long long env_mask = ((A == 64) ? ~0ull : ((1ull << A) - 1ull)) << base;
  const MpWalk a = mp_walk_sequential<A>(evaluates, inc_id, carry, slot, base);
  const MpWalk b = mp_walk_closed(evaluates, inc_id, carry, lane, base, env_mask);
  const bool holds = evaluates && inc_id != 0;
  if (env < combos && ((holds && a.prev_mine != b.prev_mine) || a.prev_out != b.prev_out)) atomicAdd(mismatches, 1);
  if (env < combos && holds && a.prev_mine != 0 && a.prev_mine != inc_id) atomicAdd(mismatches + 1, 1);   // cases where the remembered missile is another agent's
}

// the value of the other lane of an aligned lane pair (quad_perm [1,0,3,2]: v_mov_b32_dpp, full rate, all lanes active at the call sites)
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float pair_swap(float v) { return __int_as_float(pair_swap(__float_as_int(v))); }
__device__ __forceinline__ double pair_swap(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)pair_swap((int)(unsigned)b), hi = (unsigned)pair_swap((int)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// (cycle stamps of the wave that runs the environment layer: wave 3 in the quad form, wave 0 otherwise)
#define AC_CLKE(i) AC_CLKW(QUAD ? 3 : 0, i)
// DODGE: the rule-based MultipleCombatDodgeMissileTask on the NvN machinery (its own instantiations: the scenario builds do not carry its code).
template <int A, int WPE, int FORM = FORM_ONE, bool DODGE = false>
__global__ __launch_bounds__(FORM == FORM_SPLIT ? 192 : (FORM == FORM_PAIR ? 128 : (FORM == FORM_QUAD ? 256 : 64)), WPE) void step_kernel_scenario(DevPtrs P, DevCfg c, float* XF, int* XI, const float* tXF, const int* tXI) {
  using SD = ScenarioDims<A>;
  // forms launch_step reaches: the gun-only 1v1 tasks in the three-wave form, everything else in the pair form, the 1v1 scenario also in the quad form
  static_assert(FORM != FORM_ONE && (FORM == FORM_PAIR || A == 2), "no launch path for this (A, form)");
  constexpr bool SPLIT = FORM == FORM_SPLIT, QUAD = FORM == FORM_QUAD, PAIR = FORM == FORM_PAIR || QUAD;   // (the quad form's environment wave runs the pair form's code)
  constexpr bool MULTI = SD::MULTI;
  // pair form, who makes the fp64 geodetic reduction of each tick's pose (1.4 k cycles): THIS wave, from the raw ECI pose the flight wave posts
  // (the 1v1 missile tasks' arrangement). The flight wave's substep is 7.4 k cycles without it; this wave's is 1.4 k + 2.4 k per dict entry in
  // flight + 1.2 k, i.e. even with two entries per aircraft it is no longer than the flight wave's. (Measured before the munition update was
  // lightened -- 3.1 k per entry -- the same switch lost: 2v2 34.2 -> 36.9 us; after it: 33.9 -> 32.3 us, 4v4 37.5 -> 37.2 us.) The 256-register
  // builds for grids beyond one wave per SIMD keep the reduction on the flight wave: there this wave already spills.
  constexpr bool RAWP = WPE == 1;
  constexpr int OBS = SD::OBS;
  constexpr int NE = SD::NE;
  constexpr int MS = 2;  // munition slots (uids) per aircraft
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  __shared__ __attribute__((aligned(16))) float lds_out[64 * (OBS + 2)];
  // NvN pair form: what the rewards need of the geometry towards each enemy (AO, TA, R, the two gun-track distances, the posture term) and
  // the first enemy's altitude, computed by the flight wave after its last tick
  __shared__ float geo_lds[(SD::MULTI && FORM == FORM_PAIR) ? (SD::NE * 6 + 1) * 64 : 1];
  __shared__ float4 cl_pos[64 * 2];   // the chaff clouds of the workgroup's aircraft (position; live flag and multiplicity below)
  __shared__ int2 cl_meta[64 * 2];
  // What the aircraft of an env need of each other after the last substep (weapon rules, chaff rule, missile warning) goes through rows of
  // LDS owned by the environment wave: a lane posts its pose / status / munition entries once and reads the rows of the lanes it needs,
  // and the two rules that COUNT over the env's munitions (gun damage, chaff releases) and the one that takes a minimum over them (the
  // first incoming missile in launch order) are turned around -- the owner of an entry adds into / takes the minimum with its target's
  // cell -- instead of every lane walking all A x 2 entries with dependent cross-lane fetches (round 3: 9.4 k of the 4v4 kernel's 82 k cycles).
  __shared__ float xp_pose[3][64];     // final NEU position, fp32
  __shared__ int xp_status[64];        // status when the weapons are evaluated
  __shared__ int xp_cnt[2][64];        // [0] gun hits taken this step, [1] dict munitions within chaff range as this aircraft sees the dict at its turn
  __shared__ unsigned xp_inc[64];      // launch-order key of the first live munition aimed at this aircraft
  __shared__ float xp_mun[2 * 7][64];  // munition slot k: position, velocity (fp32) and speed, rows 7 k .. 7 k + 6
  __shared__ int xp_hit[64];           // during the substeps: hit by a munition this substep (set by the munition's owner, cleared by the aircraft)
  __shared__ __attribute__((aligned(16))) char split_lds[SPLIT ? sizeof(SplitLds) : (QUAD ? sizeof(QuadLds) : (PAIR ? sizeof(PairLds) : 16))];
  SplitLds& L = *reinterpret_cast<SplitLds*>(split_lds);
  QuadLds& LQ = *reinterpret_cast<QuadLds*>(split_lds);
  PairLds& LP = QUAD ? LQ.P : *reinterpret_cast<PairLds*>(split_lds);
  const Tab T{lds_tab};
  const int N = c.N;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 64 + lane;
  const bool live = n < N;
  const int slot = lane % A;
  const int nn = live ? n : (N - A + slot);
  const int base = lane - slot;
  const int n_ego = c.n_ego;
  const int team = slot < n_ego ? 0 : 1;
  const int e_first = team == 0 ? n_ego : 0;   // my enemies are slots e_first .. e_first + NE - 1
  // The environment wave's loads are two dependent round trips to HBM: (1) with the table pack, everything it can ask for up front --
  // the action row, the tick count and the status word of every munition slot; (2) the task bookkeeping and the munition slots that
  // are in use, in one batch.
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool flight_role = FORM == FORM_PAIR && wave == 1;
  const bool fdm_role = QUAD && wave != 3;      // quad form: waves 0..2 = dynamics, systems, kinematics; wave 3 = the environment wave
  const float* act = P.actions + (size_t)nn * c.act_dim;
  State s; Task t; Derived d; Props pr; Ext x;
  float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = a4;
  int mst[MS] = {MSL_INACTIVE, MSL_INACTIVE};
  if (fdm_role) {   // each FDM wave asks for the state fields its share of the tick reads (load_flight_role), and the action row
    s = State{}; t = Task{};
    if (wave == 0) { load_flight_role<0>(P.F, P.I, P.D, N, nn, s); t.status = state_word(P.F, SW_status, N, nn); }
    else if (wave == 1) load_flight_role<1>(P.F, P.I, P.D, N, nn, s);
    else { load_flight_role<2>(P.F, P.I, P.D, N, nn, s); t.status = state_word(P.F, SW_status, N, nn); }
  } else if (!flight_role) {
    // (pair / quad forms: only the wave that flies reads the action row -- a row in mapped host memory, ac_step_host, takes ~5 k cycles
    // across PCIe and the CU returns loads in issue order ACROSS its waves, so a row asked for first holds every state load of the workgroup
    // back -- and posts the weapon bits, which this wave needs after the last substep, with its final values)
    if (!PAIR) {
      a4 = load_controls(act, c.act_dim);
      if (c.act_dim == 8) b4 = load_controls(act + 4, c.act_dim);
    }
#pragma unroll
    for (int k = 0; k < MS; ++k) mst[k] = P.MI[((size_t)k * NMI + MI_status) * (size_t)N + nn];
    if (PAIR) s = State{};
  }
  stage_tables<SPLIT ? 192 : (QUAD ? 256 : (PAIR ? 128 : 64))>(lds_tab, P.tab);
  AC_CLKE(0);
  ActionFetch arow;       // quad form: the systems wave's action row (control indices; the weapon bits go on to the environment wave through LDS)
  if (QUAD && fdm_role) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // every load the compiler knows of has landed (the state, one round trip like the tables)
    if (wave == 1) arow.issue(act, c.act_dim);
  }
  if (QUAD && wave == 0) { quad_dynamics_wave(P, c, T, LQ, lane, n, live, s, t); return; }
  if (fdm_role && split_helper_wave<true, true>(s, t, T, LQ.S, lane, c.substeps, nullptr, &c, &arow)) return;
  // The NvN observation row (scenario2_task.py:256-316: ego 9, partners, enemies; the missile block after them is the environment
  // wave's), not clipped, written straight into this lane's row of the output staging buffer (a block's place in the row is a
  // run-time index: an LDS address, not a select chain over 63 registers). In the pair form of the NvN tasks the FLIGHT wave builds it
  // after its last tick, while the environment wave runs the weapon rules, rewards and terminations; one more barrier hands the rows over.
  constexpr bool ROWS_BY_FLIGHT = MULTI && FORM == FORM_PAIR;
  auto build_rows = [&](const Props& q) {
    float* orow = lds_out + lane * c.obs_dim;
    for (int k = 0; k < c.obs_dim; ++k) orow[k] = 0.0f;
    orow[0] = q.alt_m / 5000.0f;
    orow[1] = q.sphi; orow[2] = q.cphi; orow[3] = q.stht; orow[4] = q.ctht;
    orow[5] = q.ub / 340.0f; orow[6] = q.vb / 340.0f; orow[7] = q.wb / 340.0f; orow[8] = q.vc / 340.0f;
    const int n_mine = team == 0 ? n_ego : A - n_ego;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      Enemy E = gather_pose(q, base + j);
      if (j == slot) continue;
      Geo g = ao_ta_r<false>(q.n, q.e, q.u, q.vn, q.ve, q.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
      const int team_j = j < n_ego ? 0 : 1;
      const int idx = (team_j == team) ? (j - (team == 0 ? 0 : n_ego)) - (j > slot ? 1 : 0) : (n_mine - 1) + (j - e_first);
      float* blk = orow + 9 + idx * 6;
      blk[0] = (E.ub - q.ub) / 340.0f; blk[1] = (E.alt - q.alt_m) / 1000.0f; blk[2] = g.AO; blk[3] = g.TA; blk[4] = g.R / 10000.0f; blk[5] = g.side;
    }
  };
  if (flight_role) {   // (it waits for the environment wave's first flags anyway: its state load hides there)
    PairFlightIn in;
    pair_flight_load(P, c, nn, in);
    auto rows_tail = [&](const Props& q) {
      if (ROWS_BY_FLIGHT) {
        // geometry towards my enemies, for the environment wave's reward terms (every term iterates agent.enemies in env order)
#pragma unroll
        for (int qe = 0; qe < NE; ++qe) {
          const Enemy E = gather_pose(q, base + e_first + qe);
          const Geo g = ao_ta_r<false>(q.n, q.e, q.u, q.vn, q.ve, q.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
          float dw, dt;
          gun_track_distances(g.R, g.cAO, g.cTA, dw, dt);
          float* gq = geo_lds + (qe * 6) * 64 + lane;
          gq[0] = g.AO; gq[64] = g.TA; gq[128] = g.R; gq[192] = dw; gq[256] = dt; gq[320] = posture_fn(g.AO, g.TA, g.R * 0.001f);
          if (qe == 0) geo_lds[(NE * 6) * 64 + lane] = E.u;
        }
        wg_sync();                                  // the geometry is in LDS
        if (live) store_flight(P.F, P.I, P.D, c.N, n, in.s);   // (LATE_STORE: ordered before the environment wave's episode reset by the barrier below)
        if (!c.legacy_obs) build_rows(q);
        wg_sync();                                  // the rows are in LDS (the environment wave adds the missile block, or the reset rows)
        if (!c.legacy_obs) {
          wg_sync();                                // the rows are final: this wave sends them while the environment wave stores its state
          emit_obs_rows(P, lds_out, c.obs_dim, lane);
        }
      }
    };
    pair_flight_wave<RAWP, decltype(rows_tail), ROWS_BY_FLIGHT>(P, c, T, LP, lane, n, live, in, rows_tail);   // (rows_tail ends in two barriers)
    return;
  }

  if (PAIR) load_task(P.F, P.I, N, nn, t);   // the environment wave owns the task bookkeeping; of the flight state it only needs the tick count (Earth angle)
  else load_state(P.F, P.I, P.D, N, nn, s, t);
  const Task t_was = t;                       // (what the record holds in HBM: only the fields that end the step different are written back)
  ExtLoaded x_was;
  load_ext(XF, XI, N, nn, x, x_was);
  MslD ms[MS];
#pragma unroll
  for (int k = 0; k < MS; ++k) {
    // a slot that has not been used since the last reset holds zeros and MSL_INACTIVE: only its status word is read, and it is written
    // back only once it has been launched or reset
    if (mst[k] != MSL_INACTIVE) load_msl(P.MD, P.MI, N, nn, k, ms[k]);
    else { ms[k] = MslD{}; ms[k].status = MSL_INACTIVE; }
  }
  int msl_was_active = 0;
#pragma unroll
  for (int k = 0; k < MS; ++k) msl_was_active |= (ms[k].status != MSL_INACTIVE) << k;
  const bool late_bits = PAIR && c.act_dim == 8;   // (the wave that flies read the row and posts the bits with its final values)
  int msl_moved = 0;   // bit k: slot k took a state transition this step (flew a substep, was launched, was reset)

  // ---- actions: 4 control indices + [gun, AIM-9M, AIM-120B, chaff] (scenario1_task.py:33-48: Scenario1 only refreshes the ego
  // team's bits, its other team flies the scripted baseline with bits 0; scenario2_task.py:58-61 refreshes both teams)
  t.cur_step += 1;
  s.da = clampf(-1.0f, a4.x / 20.0f - 1.0f, 1.0f);
  s.de = clampf(-1.0f, a4.y / 20.0f - 1.0f, 1.0f);
  s.dr = clampf(-1.0f, a4.z / 20.0f - 1.0f, 1.0f);
  s.thr = clampf(0.0f, a4.w / 58.0f + 0.4f, 0.9f);
  const bool maneuver = !MULTI && c.task == AC_TASK_MANEUVER;   // Maneuver_curriculum (singlecombat_task.py:264-359)
  const bool gun_only = !MULTI && (c.task == AC_TASK_WVR || maneuver);   // WVRTask (WVR_task.py:10-90) / Maneuver_curriculum: no weapon bits
  const bool wvr = gun_only && !maneuver;
  // MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-145): the NvN env with a rule-based launch of the base-class missile at
  // enemies[0], no gun, no chaff, the paired-enemy 21-value observation (c.legacy_obs) and four reward terms
  static_assert(!DODGE || (SD::MULTI && FORM == FORM_PAIR), "the rule-based NvN task runs the pair form");
  constexpr bool dodge = DODGE;
  auto decode_bits = [&](float bx, float by, float bz, float bw) {
    if (!gun_only && !dodge && (MULTI || team == 0)) x.bits = (bx != 0.0f ? 1 : 0) | (by != 0.0f ? 2 : 0) | (bz != 0.0f ? 4 : 0) | (bw != 0.0f ? 8 : 0);
  };
  if (!late_bits) decode_bits(b4.x, b4.y, b4.z, b4.w);

  const MslParam MP = dodge ? aim9l() : aim120b();   // (MissileSimulator's own parameters, simulatior.py:421-433, for the rule-based task)
  bool have_pose = false;
  // Munitions only come into being in the weapons stage after the substeps, so whether this env has anything to fly during them
  // is known up front. With no missile entry and no chaff cloud in the env, the fp64 pose of the intermediate substeps is needed
  // by nobody and only the last substep computes it.
  const unsigned long long env_mask = ((A == 64) ? ~0ull : ((1ull << A) - 1ull)) << base;
  bool mine = x.n_ch > 0;
#pragma unroll
  for (int k = 0; k < MS; ++k) mine = mine || ms[k].status != MSL_INACTIVE;
  const bool env_has_munitions = (__ballot(mine) & env_mask) != 0;
  const bool env_has_clouds = (__ballot(x.n_ch > 0 && (x.ch_status[0] == 0 || x.ch_status[1] == 0)) & env_mask) != 0;
  if (env_has_clouds) {
#pragma unroll
    for (int q = 0; q < 2; ++q) { cl_pos[lane * 2 + q] = make_float4(x.cx[q], x.cy[q], x.cz[q], 0.0f); cl_meta[lane * 2 + q] = make_int2(0, x.ch_mult[q]); }
    wave_lds_fence();
  }
  if (SPLIT && split_helper_wave(s, t, T, L, lane, c.substeps)) return;
  xp_hit[lane] = 0;
  wave_lds_fence();
  int last_tick = -1;   // three-wave form: the last substep this aircraft flew
  AC_CLKE(1);
  int quad_nrun = 0;
  for (int sub = 0; sub < c.substeps; ++sub) {
    AC_CLKE(2 + 8 * sub);
    // quad form: the substep's munition work in three parts, one per barrier gap of the tick the other three waves fly meanwhile
    // (the barriers sit outside every per-env condition: all four waves meet at each of them)
    bool fly = true;
    if (QUAD) {
      quad_substep_begin<false>(t, LQ, lane, sub, env_has_munitions, 0, quad_nrun, pr, c);   // (B1 inside)
      fly = env_has_munitions;
    } else if (PAIR) {
      pair_substep<RAWP>(t, LP, lane, sub, env_has_munitions, pr, c);
      AC_CLKE(3 + 8 * sub);
      if (!env_has_munitions) continue;     // nothing to fly: the pose is only needed after the last substep
      AC_CLKE(4 + 8 * sub);
    } else {
    if (SPLIT) {
      if (dynamics_wave_tick(s, t, d, T, L, lane, sub)) { have_pose = true; last_tick = sub; }
    } else if (t.status == AC_ALIVE) {
      if (t.bloods <= 0.0f) t.status = AC_SHOTDOWN;
      f16::tick<false>(s, d, T);
      have_pose = true;
    }
    // (three-wave form with nothing in flight: the last substep's pose is taken after the loop, where the kinematics wave's fp64
    //  geodetic reduction is available; the block below would only compute that pose)
    if (!env_has_munitions && (sub + 1 < c.substeps || SPLIT)) continue;
    f16::locate(s, d);
    if (!have_pose) { f16::body_frame(s, d); have_pose = true; }
    make_props(s, d, c, pr);
    }
    const int tick_id = (t.cur_step - 1) * c.substeps + sub + 1;
    // ---- missiles: every dict entry is run(), finished ones included (env_base.py:142-143)
    int hit_tgt[MS] = {-1, -1};
    double tx[MS], ty[MS], tz[MS], tvx[MS], tvy[MS], tvz[MS];
    bool talive[MS];
    int hit_pos[MS];   // dict position of a missile that is inside its fuse radius of a live target this substep (else INT_MAX)
    auto fly_slot = [&](int k) {
      hit_tgt[k] = -1;
      if (ms[k].status != MSL_INACTIVE) {
        if (missile_run(ms[k], MP, tx[k], ty[k], tz[k], tvx[k], tvy[k], tvz[k], talive[k], c)) msl_moved |= 1 << k;
        if (ms[k].status == MSL_HIT && talive[k]) hit_tgt[k] = ms[k].order & 15;
      }
    };
    if (fly) {
    // 1v1: both munition slots of an aircraft can only be aimed at the other aircraft of the pair, the neighbouring lane -- one
    // quad-permute (a VALU move, no LDS crossbar trip) per value for both slots instead of a ds_bpermute per value and slot
    double pn = 0, pe = 0, pu = 0; float pvn = 0, pve = 0, pvd = 0; int pst = 0;
    if (A == 2) {
      pn = pair_swap(pr.n64); pe = pair_swap(pr.e64); pu = pair_swap(pr.u64);
      pvn = pair_swap(pr.vn); pve = pair_swap(pr.ve); pvd = pair_swap(pr.vd); pst = pair_swap(t.status);
    }
#pragma unroll
    for (int k = 0; k < MS; ++k) {
      const int tg = ms[k].order & 15;               // target slot lives in the low bits of `order`
      const bool used = ms[k].status != MSL_INACTIVE;
      if (A == 2) {                                   // (an unused slot's values are never looked at)
        tx[k] = pn; ty[k] = pe; tz[k] = pu; tvx[k] = (double)pvn; tvy[k] = (double)pve; tvz[k] = (double)pvd;
        talive[k] = pst == AC_ALIVE;
      } else {
        const int src = base + (used ? tg : slot);
        tx[k] = __shfl(pr.n64, src); ty[k] = __shfl(pr.e64, src); tz[k] = __shfl(pr.u64, src);
        tvx[k] = (double)__shfl(pr.vn, src); tvy[k] = (double)__shfl(pr.ve, src); tvz[k] = (double)__shfl(pr.vd, src);
        talive[k] = __shfl(t.status, src) == AC_ALIVE;
      }
      hit_pos[k] = 0x7fffffff;
      if (MULTI && used && talive[k] && ms[k].status != MSL_MISS) {   // the fuse test of run(), exactly as missile_run makes it
        const double ddx = tx[k] - ms[k].px, ddy = ty[k] - ms[k].py, ddz = tz[k] - ms[k].pz;
        const double Rxy2 = ddx * ddx + ddy * ddy, R2 = Rxy2 + ddz * ddz;
        if (fx::sqrt(R2) < (double)MP.Rc) hit_pos[k] = ms[k].dpos;
      }
    }
    if (MULTI) {
      // env._tempsims is walked ONCE per substep in dict order (env_base.py:142-143) and a hit kills its target on the spot
      // (simulatior.py:525-527): every missile later in the dict aimed at the same aircraft already sees a dead target in this
      // substep -- it turns MISS without moving, and it cannot hit as well. Only the earliest hitter of a target counts.
      const bool any_hit = hit_pos[0] != 0x7fffffff || hit_pos[1] != 0x7fffffff;
      if (__ballot(any_hit) & env_mask) {
        int first_on_me = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < A; ++j)
#pragma unroll
          for (int k = 0; k < MS; ++k) {
            const int hp = __shfl(hit_pos[k], base + j), htg = __shfl(ms[k].order & 15, base + j);
            if (hp != 0x7fffffff && htg == slot && hp < first_on_me) first_on_me = hp;
          }
#pragma unroll
        for (int k = 0; k < MS; ++k) {
          const int fp = __shfl(first_on_me, base + (ms[k].order & 15));
          if (ms[k].status != MSL_INACTIVE && fp < ms[k].dpos) talive[k] = false;
        }
      }
    }
    fly_slot(0);
    }
    if (QUAD) wg_sync();                                   // B2 of the tick
    if (fly) {
    fly_slot(1);
    AC_CLKE(5 + 8 * sub);
    // a hit grounds its target (simulatior.py:525-527): the munition's owner flags the target's cell
#pragma unroll
    for (int k = 0; k < MS; ++k) if (hit_tgt[k] >= 0) xp_hit[base + hit_tgt[k]] = 1;
    wave_lds_fence();
    if (xp_hit[lane]) { xp_hit[lane] = 0; if (t.status == AC_ALIVE) t.status = AC_SHOTDOWN; }
    }
    if (QUAD) wg_sync();                                   // B3 of the tick
    if (!fly) continue;
    AC_CLKE(6 + 8 * sub);
    // ---- chaff clouds age (ChaffSimulator.run, simulatior.py:377-381), then the decoy test (env_base.py:146-154). Clouds only come
    // into being in the weapons stage after the substeps: an env without a live cloud at the start of the step has none during it.
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < x.n_ch) { x.ct[q] += 1.0f / 60.0f; if (x.ct[q] > 20.0f) x.ch_status[q] = 1; }
    if (env_has_clouds) {
      // the env's clouds sit in LDS (position and multiplicity, once per step); which of them are live is two ballots, and a missile
      // in flight walks the LIVE clouds of its env only (most steps: none or one of the A x 2 records)
      constexpr unsigned amask_cl = (A >= 32) ? ~0u : ((1u << A) - 1u);
      const unsigned live0 = (unsigned)(__ballot(0 < x.n_ch && x.ch_status[0] == 0) >> base) & amask_cl;
      const unsigned live1 = (unsigned)(__ballot(1 < x.n_ch && x.ch_status[1] == 0) >> base) & amask_cl;
      bool mine_flying = false;
#pragma unroll
      for (int k = 0; k < MS; ++k) mine_flying = mine_flying || ms[k].status == MSL_LAUNCHED;
      if (mine_flying) {
        float mpx[MS], mpy[MS], mpz[MS];
#pragma unroll
        for (int k = 0; k < MS; ++k) { mpx[k] = (float)ms[k].px; mpy[k] = (float)ms[k].py; mpz[k] = (float)ms[k].pz; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned todo = q == 0 ? live0 : live1;          // (the draws are keyed by what is tested: the order of the walk does not matter)
          while (todo) {
            const int j = __ffs(todo) - 1;
            todo &= todo - 1u;
            const float4 cp = cl_pos[(base + j) * 2 + q];
            const int cm = cl_meta[(base + j) * 2 + q].y;
            const int cbase = (q == 0) ? 0 : cl_meta[(base + j) * 2].y;   // release index of the first chaff of this event
#pragma unroll
            for (int k = 0; k < MS; ++k) {
              if (ms[k].status != MSL_LAUNCHED) continue;
              float dx = cp.x - mpx[k], dy = cp.y - mpy[k], dz = cp.z - mpz[k];
              if (dx * dx + dy * dy + dz * dz <= 300.0f * 300.0f) {
                for (int m = 0; m < cm; ++m)   // one draw per chaff of the event; the missile stays "not done" only until one succeeds
                  if (ms[k].status == MSL_LAUNCHED && decoy_uniform(c.chaff_seed + (unsigned long long)(nn / A), tick_id, slot, MS - k, j, cbase + m) < 0.85f) ms[k].status = MSL_MISS;
              }
            }
          }
        }
      }
    }
  }
  AC_CLKE(60);
  const bool split_located = SPLIT && dynamics_wave_finish(s, d, L, lane, last_tick, c.substeps);   // (+ the helper waves' fields)
  if (PAIR) {
    if (QUAD) wg_sync();                           // (quad form: the helper waves hand their fields to the dynamics wave here)
    wg_sync();                                     // the flight wave has posted its final values and stored the flight state
    AC_CLKE(61);
    pair_read_final(LP, lane, s, d, pr);
  } else if (c.substeps == 0) { f16::locate(s, d); f16::body_frame(s, d); make_props(s, d, c, pr); }
  else if (SPLIT && !env_has_munitions) {
    if (!split_located) f16::locate(s, d);
    if (!have_pose) { f16::body_frame(s, d); have_pose = true; }
    make_props(s, d, c, pr);
  }

  AC_CLKE(62);
  // ---- weapons (scenario1_task.py:61-103). The reference walks the agents one after another in env order; what an agent decides
  // depends on the others only through the chaff rule, which counts the dict's missiles aimed at it (so it sees the launches of the
  // agents before it, and entries those launches replaced are gone). Gun damage lands on bloods, which nobody reads until the next
  // substep. So every lane decides gun / missiles for itself at once, and the chaff count reconstructs the dict as agent `slot` saw it.
  if (late_bits) {
    const int pk = (int)(QUAD ? LQ.S.M[mail::F_BITS][lane] : LP.FIN[pair::FIN_BITS][lane]);
    decode_bits((float)(pk & 1), (float)(pk & 2), (float)(pk & 4), (float)(pk & 8));
  }
  {
    const float hv = sqrtf(pr.vn * pr.vn + pr.ve * pr.ve + pr.vd * pr.vd);
    // farthest enemy (get_target, :139-145): poses and statuses do not change while the weapons are evaluated
    int tg = e_first; float bd = -1.0f, tdx = 0, tdy = 0, tdz = 0; int tg_status = AC_ALIVE;
    xp_pose[0][lane] = pr.n; xp_pose[1][lane] = pr.e; xp_pose[2][lane] = pr.u; xp_status[lane] = t.status;
    xp_cnt[0][lane] = 0; xp_cnt[1][lane] = 0; xp_inc[lane] = 0xffffffffu;
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int src = base + e_first + q;
      float ex = xp_pose[0][src] - pr.n, ey = xp_pose[1][src] - pr.e, ez = xp_pose[2][src] - pr.u;
      int est = xp_status[src];
      float dd = sqrtf(ex * ex + ey * ey + ez * ez);
      if (dd > bd) { bd = dd; tg = e_first + q; tdx = ex; tdy = ey; tdz = ez; tg_status = est; }
    }
    if (dodge) {   // the rule aims at agent.enemies[0] (:129), not at the farthest enemy
      const int src = base + e_first;
      tdx = xp_pose[0][src] - pr.n; tdy = xp_pose[1][src] - pr.e; tdz = xp_pose[2][src] - pr.u;
      bd = sqrtf(tdx * tdx + tdy * tdy + tdz * tdz); tg = e_first; tg_status = xp_status[src];
    }
    const float ang = 57.29577951f * acos_fast(clampf(-1.0f, (tdx * pr.vn + tdy * pr.ve + tdz * pr.vd) / (bd * hv + 1e-8f), 1.0f));
    AC_CLKE(80);
    // my dict entries before this step's launches (what an agent that acts before me still sees of them)
    int old_st[MS], old_tg[MS]; float old_x[MS], old_y[MS], old_z[MS];
#pragma unroll
    for (int k = 0; k < MS; ++k) {
      old_st[k] = ms[k].status; old_tg[k] = ms[k].order & 15;
      old_x[k] = (float)ms[k].px; old_y[k] = (float)ms[k].py; old_z[k] = (float)ms[k].pz;
    }
    float gun_dmg = 0.0f; int gun_tgt = -1;
    int launched_k = -1;   // the slot this aircraft launched into this step (one launch per step at most: the second rule needs the first missile done)
    auto launch_into = [&](int want_k, int want_model) {   // MissileSimulator.launch (:497-514) into the dict entry of uid "agent + (MS - k)"
      float tht = asinf(pr.stht);
      float psi = atan2f(pr.m12, pr.m11);
      if (psi < 0.0f) psi += 2.0f * f16::kPi;
#pragma unroll
      for (int q = 0; q < MS; ++q)
        if (q == want_k) {
          if (ms[q].status == MSL_HIT) x.orphan_hits += 1;   // a replaced entry that was HIT stays is_success forever (never run again)
          if (ms[q].status == MSL_INACTIVE) ms[q].dpos = (t.cur_step << 4) | slot;   // a uid launched again keeps its place in the dict
          ms[q].model = want_model;
          ms[q].px = pr.n64; ms[q].py = pr.e64; ms[q].pz = pr.u64; ms[q].vx = pr.vn; ms[q].vy = pr.ve; ms[q].vz = pr.vd;
          ms[q].theta = tht; ms[q].psi = psi; ms[q].t = 0.0; ms[q].m = MP.m0; ms[q].dth = 0.0; ms[q].dph = 0.0;
          ms[q].dprev = INFINITY; ms[q].recede = 0; ms[q].status = MSL_LAUNCHED;
          ms[q].order = (t.cur_step << 8) | (slot << 4) | tg;    // launch order (step, agent) and target slot
        }
      t.last_missile = want_k;
      launched_k = want_k;
      msl_moved |= 1 << want_k;
    };
    if (gun_only) {
      // WVR_task.py:62-76 / singlecombat_task.py:290-297: every aircraft, dead or alive, drains 5 blood from its farthest enemy inside 3 km and 5 deg, every step
      if (bd * 0.001f < 3.0f && ang < 5.0f) { gun_dmg = 5.0f; gun_tgt = tg; }
    } else if (dodge) {
      // multiplecombat_with_missile_task.py:127-145 (the 1v1 rule of singlecombat_with_missile_task.py:108-124): the window is updated by every
      // aircraft, dead ones included; a launch needs a full window, the distance gate, a round left, the interval and a live shooter
      const int len = c.lock_len;                       // deque(maxlen = int(1 / time_interval))
      const unsigned bit = 1u << (unsigned)(t.lock_pos % len);
      t.lock_bits = (ang < c.max_attack_angle) ? (t.lock_bits | (int)bit) : (t.lock_bits & ~(int)bit);
      t.lock_pos += 1;
      const bool locked = __popc((unsigned)t.lock_bits & ((1u << len) - 1u)) >= len;
      if (t.status == AC_ALIVE && locked && bd <= c.max_attack_distance && t.remaining > 0 && (t.cur_step - t.last_shoot_time) >= c.min_attack_interval) {
        launch_into(MS - t.remaining, 0);               // uid = agent_id + str(remaining): 2 -> the first entry, 1 -> the second
        t.remaining -= 1;
        t.last_shoot_time = t.cur_step;
      }
    } else if (t.status == AC_ALIVE) {
      const bool talive = tg_status == AC_ALIVE;
      const bool av_gun = talive && bd * 0.001f < 3.0f && ang < 5.0f;
      const bool av_120 = talive && bd * 0.001f < 37.0f && ang < 90.0f;
      const bool av_9m = talive && bd * 0.001f < 7.0f && ang < 90.0f;
      auto last_done = [&]() {
        if (t.last_missile < 0) return true;
        bool dn = true;
#pragma unroll
        for (int k = 0; k < MS; ++k) if (k == t.last_missile) dn = ms[k].status == MSL_HIT || ms[k].status == MSL_MISS;
        return dn;
      };
      int want_k = -1, want_model = 0;
      if ((x.bits & 1) && x.rem_gun > 0 && last_done() && av_gun) { gun_dmg = 5.0f; gun_tgt = tg; x.rem_gun -= 1; }
      if ((x.bits & 4) && x.rem_120b > 0 && last_done() && av_120) { want_k = MS - x.rem_120b; want_model = 0; x.rem_120b -= 1; }
      else if ((x.bits & 2) && x.rem_9m > 0 && last_done() && av_9m) { want_k = MS - x.rem_9m; want_model = 1; x.rem_9m -= 1; }   // (after an AIM-120B launch the last missile is in flight: no AIM-9M in the same step)
      if (want_k >= 0) launch_into(want_k, want_model);
    }
    AC_CLKE(81);
    // gun damage lands on the target's blood right away (:70-73): 5 per shooter, counted into the target's cell (bloods are multiples of 5
    // below 100: n subtractions of 5 and one of 5 n are the same number)
    if (gun_tgt >= 0) atomicAdd(&xp_cnt[0][base + gun_tgt], 1);
    AC_CLKE(82);
    // chaff (:97-103): one release per dict missile (done ones included) aimed at this agent within 1000 m, as the dict stands when
    // the agent acts: entries of agents up to and including itself already hold this step's launches, those of the agents after it are
    // still what they were. The OWNER of an entry tells its target: an entry that was not launched this step counts for its target;
    // one launched this step counts for its (new) target if that one acts at or after the launcher, and the entry it replaced counts
    // for ITS target if that one acts before the launcher.
    if (!gun_only && !dodge) {
      auto in_range = [&](int r, float mx, float my, float mz) {
        const float dx = xp_pose[0][base + r] - mx, dy = xp_pose[1][base + r] - my, dz = xp_pose[2][base + r] - mz;
        return sqrtf(dx * dx + dy * dy + dz * dz) < 1000.0f;
      };
#pragma unroll
      for (int k = 0; k < MS; ++k) {
        if (launched_k == k) {
          if (tg >= slot && in_range(tg, pr.n, pr.e, pr.u)) atomicAdd(&xp_cnt[1][base + tg], 1);
          if (old_st[k] != MSL_INACTIVE && old_tg[k] < slot && in_range(old_tg[k], old_x[k], old_y[k], old_z[k])) atomicAdd(&xp_cnt[1][base + old_tg[k]], 1);
        } else if (old_st[k] != MSL_INACTIVE && in_range(old_tg[k], old_x[k], old_y[k], old_z[k])) atomicAdd(&xp_cnt[1][base + old_tg[k]], 1);
      }
    }
    // the munition entries as they stand after the launches: the missile warning and MissilePostureReward read them below
#pragma unroll
    for (int k = 0; k < MS; ++k) {
      if (ms[k].status == MSL_LAUNCHED) {
        atomicMin(&xp_inc[base + (ms[k].order & 15)], ((unsigned)(ms[k].order >> 4) << 1) | (unsigned)k);   // launch order (step, launcher), then the slot
        xp_mun[7 * k + 0][lane] = (float)ms[k].px; xp_mun[7 * k + 1][lane] = (float)ms[k].py; xp_mun[7 * k + 2][lane] = (float)ms[k].pz;
        xp_mun[7 * k + 3][lane] = (float)ms[k].vx; xp_mun[7 * k + 4][lane] = (float)ms[k].vy; xp_mun[7 * k + 5][lane] = (float)ms[k].vz;
      }
      xp_mun[7 * k + 6][lane] = (float)sqrt(ms[k].vx * ms[k].vx + ms[k].vy * ms[k].vy + ms[k].vz * ms[k].vz);
    }
    wave_lds_fence();
    t.bloods -= 5.0f * (float)xp_cnt[0][lane];
    if (!gun_only && !dodge) {
      const int lc_status = (x.last_chaff & 1) ? x.ch_status[1] : x.ch_status[0];
      const bool can = t.status == AC_ALIVE && (x.bits & 8) && x.rem_chaff > 0 && (x.last_chaff < 0 || lc_status == 1);
      const int n_rel = can ? xp_cnt[1][lane] : 0;
      if (n_rel > 0 && x.n_ch < 2) {
        const bool second = x.n_ch == 1;
        x.cx[0] = second ? x.cx[0] : pr.n; x.cy[0] = second ? x.cy[0] : pr.e; x.cz[0] = second ? x.cz[0] : pr.u;
        x.ct[0] = second ? x.ct[0] : 0.0f; x.ch_status[0] = second ? x.ch_status[0] : 0; x.ch_mult[0] = second ? x.ch_mult[0] : n_rel;
        x.cx[1] = second ? pr.n : x.cx[1]; x.cy[1] = second ? pr.e : x.cy[1]; x.cz[1] = second ? pr.u : x.cz[1];
        x.ct[1] = second ? 0.0f : x.ct[1]; x.ch_status[1] = second ? 0 : x.ch_status[1]; x.ch_mult[1] = second ? n_rel : x.ch_mult[1];
        x.last_chaff = x.n_ch; x.n_ch += 1; x.rem_chaff -= n_rel;
      }
    }
  }

  AC_CLKE(63);
  // ---- geometry towards my enemies (every reward term iterates agent.enemies in env order)
  EnemyGeo eg[NE];
  float dwez[NE], dtail[NE], posture_e[NE];   // per enemy: the distances of the two gun-track terms, the posture term
  float e_u0 = 0.0f;
  if (ROWS_BY_FLIGHT) {
    wg_sync();                                     // the flight wave has posted them
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const float* gq = geo_lds + (q * 6) * 64 + lane;
      eg[q].AO = gq[0]; eg[q].TA = gq[64]; eg[q].R = gq[128]; eg[q].cAO = 0.0f; eg[q].cTA = 0.0f;
      dwez[q] = gq[192]; dtail[q] = gq[256]; posture_e[q] = gq[320];
    }
    e_u0 = geo_lds[(NE * 6) * 64 + lane];
  } else {
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      Enemy E = gather_pose(pr, base + e_first + q);
      Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
      eg[q].AO = g.AO; eg[q].TA = g.TA; eg[q].R = g.R; eg[q].cAO = g.cAO; eg[q].cTA = g.cTA;
      gun_track_distances(g.R, g.cAO, g.cTA, dwez[q], dtail[q]);
      posture_e[q] = posture_fn(g.AO, g.TA, g.R * 0.001f);
      if (q == 0) e_u0 = E.u;
    }
  }
  AC_CLKE(83);
  // ---- my first alive incoming missile in launch order (check_missile_warning), and whether any is alive
  Incoming inc{false, 0, 0, 0, 0, 0, 0};
  int inc_id = 0;    // 1 + launcher*MS + slot of that missile
  {
    // the minimum taken above finds it (launch order, launcher and slot sit in the key), then its six floats come from its owner's rows
    const unsigned key = xp_inc[lane];
    const int bj = (int)(key >> 1) & 15, bk = key == 0xffffffffu ? -1 : (int)(key & 1u);
    if (bk >= 0) {
      const float* mrow = &xp_mun[7 * bk][base + bj];
      inc.px = mrow[0]; inc.py = mrow[64]; inc.pz = mrow[128]; inc.vx = mrow[192]; inc.vy = mrow[256]; inc.vz = mrow[320];
    }
    inc.any = bk >= 0;
    inc_id = bk >= 0 ? 1 + bj * MS + bk : 0;
  }
  int my_hits = x.orphan_hits;
#pragma unroll
  for (int k = 0; k < MS; ++k) if (ms[k].status == MSL_HIT) my_hits += 1;

  AC_CLKE(64);
  // ---- observation
  float ob[OBS];
  if (!MULTI) {
    Enemy E = gather_pose(pr, base + e_first);
    const Incoming none{false, 0, 0, 0, 0, 0, 0};
    if (gun_only) observe_1v1<AC_TASK_SINGLECOMBAT>(pr, E, none, ob);   // HierarchicalSingleCombatTask keeps the clipped 15-value, 2-D observation
    else observe_1v1<AC_TASK_SHOOT_MISSILE>(pr, E, c.rwr ? none : inc, ob);   // Scenario1 keeps the 21-value layout (scenario1_task.py:31-32); Scenario1_RWR blanks the missile block (:298-300)
  } else if (c.legacy_obs) {
    // Scenario2 / Scenario3 (not _NvN) keep MultipleCombatShootMissileTask's 21 values against the enemy with the same index in
    // its team (multiplecombat_with_missile_task.py:30-117)
#pragma unroll
    for (int k = 0; k < OBS; ++k) ob[k] = 0.0f;
    Enemy E = gather_pose(pr, base + e_first + (slot - (team == 0 ? 0 : n_ego)));
    observe_1v1<AC_TASK_SHOOT_MISSILE>(pr, E, inc, ob);
  } else {
    // (the row itself: build_rows above -- by this wave, or by the flight wave in the pair form)
    if (!ROWS_BY_FLIGHT) build_rows(pr);
  }
  constexpr bool OBS_IN_LDS = MULTI;   // (the legacy 21-value form of the NvN tasks still goes through ob[] below)
  const bool row_direct = OBS_IN_LDS && !c.legacy_obs;

  AC_CLKE(65);
  // ---- terminations of the 1v1 family come BEFORE the rewards (env_base.py:159-171)
  bool done = false;
  int code = AC_DONE_NONE, last_code = AC_DONE_NONE;
  float np_max = fmaxf(fabsf(s.npx), fmaxf(fabsf(s.npy), fabsf(s.npz)));
  float pqr = sqrtf(d.p * d.p + d.q * d.q + d.r * d.r);
  const bool nonfinite = nonfinite_probe(d.veci, pqr, d.h_sl_ft, np_max);
  const bool extreme = nonfinite || (d.veci >= 1e10f) || (pqr >= 1000.0f) || (d.h_sl_ft >= 1e10f) || (np_max > 10.0f);
  const bool overload = (s.ticks >= kTickOverload) &&
                        (fabsf(s.npx) > c.acc_x || fabsf(s.npy) > c.acc_y || fabsf(s.npz + 1.0f) > c.acc_z);
  const bool low = pr.alt_m <= c.altitude_limit;
  // (Agents are walked in env order and agent i sees the final status of the agents before it, the not-yet-evaluated status of those
  // after it. The statuses travel as one ballot per round -- a status only matters as "alive or not" to the others -- instead of two
  // cross-lane fetches per round, whose latency was serialised by the walk: 5.6 k of the 4v4 kernel's 91 k cycles.)
  auto terminations = [&]() {
    // The walk itself only has to carry who is still alive: an agent's own checks change its status in one way (a crash condition
    // makes it CRASH), and what it sees of the others is whether any enemy is alive at its turn. Three ballots give every lane the
    // env's inputs as A-bit masks (alive, crash condition, no missile coming); every lane then runs the A rounds on those bits in its
    // own registers and keeps what happened at its own turn -- the rounds used to be A dependent ballots. The messages are assigned afterwards.
    const int st0 = t.status;
    const bool crash_cond = low || extreme || overload;
    constexpr unsigned amask = (A >= 32) ? ~0u : ((1u << A) - 1u);
    const unsigned al0 = (unsigned)(__ballot(st0 == AC_ALIVE) >> base) & amask;
    const unsigned ccb = (unsigned)(__ballot(crash_cond) >> base) & amask;
    const unsigned nob = (unsigned)(__ballot(!inc.any) >> base) & amask;
    const unsigned team0 = (1u << n_ego) - 1u, team1 = amask & ~team0;
    unsigned alive = al0;
    bool enemies_dead = false, crash_mine = false;
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const bool ed = (alive & (i < n_ego ? team1 : team0)) == 0;
      // NvN: SafeReturn comes first, so only an aircraft that is still flying and has no mission-complete reaches the crash checks;
      // 1v1 family: the crash checks come first and apply whatever the status was
      const bool ci = (ccb >> i) & 1u;
      const bool crash_now = MULTI ? (((al0 >> i) & 1u) && !(ed && ((nob >> i) & 1u)) && ci) : ci;
      if (crash_now) alive &= ~(1u << i);
      if (i == slot) { enemies_dead = ed; crash_mine = crash_now; }
    }
    if (crash_mine) t.status = AC_CRASH;
    AC_CLKE(84);
    if (MULTI) {   // SafeReturn, ExtremeState, Overload, LowAltitude, Timeout (multiplecombat_task.py:33-39)
      if (st0 == AC_SHOTDOWN) { code = AC_DONE_SHOTDOWN; done = true; }
      else if (st0 == AC_CRASH) { code = AC_DONE_CRASHED; done = true; }
      else if (enemies_dead && !inc.any) { code = AC_DONE_MISSION_COMPLETE; done = true; }
      else if (extreme) { code = AC_DONE_EXTREME_STATE; done = true; }
      else if (overload) { code = AC_DONE_OVERLOAD; done = true; }
      else if (low) { code = AC_DONE_LOW_ALTITUDE; done = true; }
      else if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; }
    } else {       // LowAltitude, ExtremeState, Overload, SafeReturn, Timeout (singlecombat_task.py:34-40)
      if (low) { code = AC_DONE_LOW_ALTITUDE; done = true; }
      else if (extreme) { code = AC_DONE_EXTREME_STATE; done = true; }
      else if (overload) { code = AC_DONE_OVERLOAD; done = true; }
      else if (wvr) { if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; } }   // WVR_task.py:31-36: no SafeReturn
      else if (st0 == AC_SHOTDOWN) { code = AC_DONE_SHOTDOWN; done = true; }
      else if (st0 == AC_CRASH) { code = AC_DONE_CRASHED; done = true; }
      else if (enemies_dead && !inc.any) { code = AC_DONE_MISSION_COMPLETE; done = true; }
      else if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; }
    }
    AC_CLKE(85);
    // info['done_condition'] keeps the message of the last agent (in env order) that has one
    const unsigned long long coded = __ballot(code != AC_DONE_NONE) & env_mask;
    const int last = coded ? 63 - __clzll((long long)coded) : lane;
    last_code = __shfl(code, last);
  };
  if (!MULTI) terminations();

  AC_CLKE(70);
  // ---- rewards: eleven terms in list order (scenario1_task.py:13-25); which agents evaluate them differs by family
  const bool evaluates = MULTI ? (t.status == AC_ALIVE) : !t.die_flag;   // multiplecombat_task.py:147-151 / singlecombat_task.py:190-195
  if (!MULTI && !t.die_flag) t.die_flag = (t.status != AC_ALIVE) ? 1 : 0;
  // the shared reference lists are written by the first agent that evaluates after a reset
  const unsigned long long ev_mask = __ballot(evaluates) & (((A == 64) ? ~0ull : ((1ull << A) - 1ull)) << base);
  const int first_lane = ev_mask ? (__ffsll((long long)ev_mask) - 1) : base;
  if (ev_mask && !(x.ref_set & 1)) {   // first evaluation since reset: everybody copies the first evaluator's values
    x.cg_AO = __shfl(eg[0].AO, first_lane); x.cg_TA = __shfl(eg[0].TA, first_lane);
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const float w_src = dwez[q == 0 ? 0 : q - 1], t_src = dtail[q == 0 ? 0 : q - 1];
      const float w = __shfl(w_src, first_lane);
      const float tl = __shfl(t_src, first_lane);
      x.wez[q] = w; x.tail[q] = tl;
    }
    x.ref_set = 7;
  }
  AC_CLKE(71);
  float own = 0.0f;
  // MissilePostureReward's shared remembered missile, in closed form (mp_walk_closed above)
  float r_mp = 0.0f;
  {
    const bool holds = evaluates && inc_id != 0;
    const MpWalk walk = mp_walk_closed(evaluates, inc_id, x.mp_prev, lane, base, env_mask);
    const int prev_mine = walk.prev_mine;
    // speeds of the remembered missile and of my incoming missile (the speed row of the owner's slot)
    const int pid = holds ? prev_mine : 1, cid = holds ? inc_id : 1;
    const float v_prev = xp_mun[7 * ((pid - 1) % MS) + 6][base + ((pid - 1) / MS)], v_cur = xp_mun[7 * ((cid - 1) % MS) + 6][base + ((cid - 1) / MS)];
    if (holds) {
      float v_dec = (v_prev - v_cur) / 340.0f * c.missile_posture_scale;
      float va = sqrtf(pr.vn * pr.vn + pr.ve * pr.ve + pr.vd * pr.vd);
      float ang = (inc.vx * pr.vn + inc.vy * pr.ve + inc.vz * pr.vd) / (v_cur * va);
      r_mp = (ang < 0.0f) ? ang / (fmaxf(v_dec, 0.0f) + 1.0f) : ang * fmaxf(v_dec, 0.0f);
    }
    x.mp_prev = walk.prev_out;
  }
  AC_CLKE(72);
  if (evaluates) {
    float r_alt = potential(altitude_raw(pr, c), c.altitude_scale, c.altitude_pot, t.pre_altitude);
    float cg = 0.0f, behit = 0.0f, tailr = 0.0f, wezdot = 0.0f, wez = 0.0f, posture = 0.0f;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const float R = eg[q].R;
      cg += -(eg[0].AO - x.cg_AO) - (eg[0].TA - x.cg_TA);
      const bool band = R >= 500.0f * kFt2M && R <= 3000.0f * kFt2M;
      if (band && eg[q].AO >= 179.0f * f16::kPi / 180.0f) behit += -5.0f;
      if (band && eg[q].AO <= f16::kPi / 180.0f) wez += 5.0f + 5.0f * (3000.0f * kFt2M - R) / (2500.0f * kFt2M);
      float isr = rsqrtf(R);
      tailr += -(1.0f / 60.0f) * tanh_fast((dtail[q] - x.tail[q]) * isr);
      wezdot += -(1.0f / 60.0f) * tanh_fast((dwez[q] - x.wez[q]) * isr);
      posture += posture_e[q];
    }
    float ev = ((t.status != AC_ALIVE) ? -200.0f : 0.0f) + 200.0f * (float)my_hits;
    float r_ev = potential(ev, c.event_scale, c.event_pot, t.pre_event);
    float r_pos = potential(posture, c.posture_scale, c.posture_pot, t.pre_posture);
    float r_ra = fminf(1.0f - fabsf(pr.u * 0.001f - e_u0 * 0.001f), 0.0f);
    own = r_alt + cg + r_ev + behit + tailr + wezdot + wez + r_pos + (wvr ? 0.0f : r_ra + (maneuver ? 0.0f : r_mp));
    if (dodge) own = r_pos + r_mp + r_alt + r_ev;   // multiplecombat_with_missile_task.py:23-28   // ShootPenalty never fires: remaining_missiles is constant; WVR has eight terms (WVR_task.py:20-29), Maneuver_curriculum nine
  }
  AC_CLKE(73);
  float reward = own;
  if (MULTI) {   // team mean (multiplecombat_env.py:170-175), then the terminations
    float tsum = 0.0f;
#pragma unroll
    for (int j = 0; j < A; ++j) { float rj = __shfl(own, base + j); if ((j < n_ego ? 0 : 1) == team) tsum += rj; }
    reward = tsum / (float)(team == 0 ? n_ego : A - n_ego);
    AC_CLKE(74);
    terminations();
  }

  AC_CLKE(66);
  if (ROWS_BY_FLIGHT) wg_sync();                    // the flight wave has built the observation rows
  if (row_direct && inc.any) {                       // the missile block directly after the enemies
    float* orow = lds_out + lane * c.obs_dim;
    Geo gm = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, inc.px, inc.py, inc.pz, inc.vx, inc.vy, inc.vz);
    float* blk = orow + 9 + 6 * (A - 1);
    blk[0] = (sqrtf(inc.vx * inc.vx + inc.vy * inc.vy + inc.vz * inc.vz) - pr.ub) / 340.0f;
    blk[1] = (inc.pz - pr.alt_m) / 1000.0f;
    blk[2] = gm.AO; blk[3] = gm.TA; blk[4] = gm.R / 10000.0f; blk[5] = gm.side;
  }
  constexpr unsigned amask_env = (A >= 32) ? ~0u : ((1u << A) - 1u);
  const bool all_done = ((unsigned)(__ballot(done) >> base) & amask_env) == amask_env;
  int step_out = t.cur_step;
  if (all_done) {
    load_state(P.tF, P.tI, P.tD, A, slot, s, t);
    x = fresh_ext(c.num_missiles[slot]);
    (void)tXF; (void)tXI;
#pragma unroll
    for (int k = 0; k < MS; ++k) { ms[k] = MslD{}; ms[k].status = MSL_INACTIVE; }
    msl_moved = (1 << MS) - 1;
    const float* tobs = P.tF + (size_t)NSW * A + slot * OBS;
    if (row_direct) { float* orow = lds_out + lane * c.obs_dim; for (int k = 0; k < OBS; ++k) orow[k] = tobs[k]; }
    else {
#pragma unroll
      for (int k = 0; k < OBS; ++k) ob[k] = tobs[k];
    }
  }
  if (ROWS_BY_FLIGHT && row_direct) wg_sync();          // the rows in LDS are final: the flight wave sends them
  if (live) {
    if (!PAIR) store_flight(P.F, P.I, P.D, N, n, s);
    else if (all_done) store_flight(P.F, P.I, P.D, N, n, s);   // the flight wave has stored the flown state (ordered before this by the barrier): an episode reset overwrites it
    // NvN kernels (flight-wave bound: this wave has the time): the task record and the munition slots are written back only where they
    // changed -- 1390 -> 1305 B (2v2) and 1460 -> 1360 B (4v4) of HBM traffic per aircraft-step. The 1v1 kernel is bound by THIS wave and
    // keeps the plain stores (the tests cost it 0.3 us for 4 % of its traffic).
    if (MULTI) store_task_changed(P.F, P.I, N, n, t, t_was);
    else store_task(P.F, P.I, N, n, t);
    store_ext(XF, XI, N, n, x, x_was, all_done);
#pragma unroll
    for (int k = 0; k < MS; ++k) {
      // a slot is written back whole when it moved this step (flew a substep, was launched into, was reset); one that stands still -- a
      // finished entry the reference keeps run()-ning -- only advances its clock, `dprev`, the receding count and maybe its status
      const bool was = (msl_was_active >> k) & 1;
      if (!MULTI) { if (ms[k].status != MSL_INACTIVE || was) store_msl(P.MD, P.MI, N, n, k, ms[k]); }
      else if (((msl_moved >> k) & 1) || (ms[k].status != MSL_INACTIVE && !was)) store_msl(P.MD, P.MI, N, n, k, ms[k]);
      else if (was) store_msl_clock(P.MD, P.MI, N, n, k, ms[k]);
    }
  }
  AC_CLKE(67);
  // (WVR uses the first 15 of the 21 slots; the *_RWR variants append two reserved zero slots)
  reward = poison_if(nonfinite, reward);
  if (row_direct && ROWS_BY_FLIGHT) emit_scalars(P, lane, reward, done, A, step_out, last_code, 0, all_done ? 1 : 0);   // (the rows: the flight wave, above)
  else if (row_direct) emit_rows(P, lds_out, c.obs_dim, lane, reward, done, A, step_out, last_code, 0, all_done ? 1 : 0);
  else emit_outputs(P, lds_out, c.obs_dim, lane, ob, reward, done, A, step_out, last_code, 0, all_done ? 1 : 0);
  AC_CLKE(68);
}

// reset template for the scenario tasks: same initial-condition pass, scenario observation layout, potential seeds
template <int A>
__global__ __launch_bounds__(64) void init_kernel_scenario(InitArgs ia, DevCfg c, const float* tab, float* tF, int* tI, double* tD) {
  using SD = ScenarioDims<A>;
  constexpr int OBS = SD::OBS;
  constexpr int NE = SD::NE;
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  stage_tables(lds_tab, tab);
  const Tab T{lds_tab};
  const int slot = threadIdx.x % A;
  const int base = (threadIdx.x & 63) - slot;
  const int team = slot < c.n_ego ? 0 : 1;
  const int e_first = team == 0 ? c.n_ego : 0;
  State s; Derived d; Task t{}; Props pr;
  initial_state(ia.ic[slot], T, s, d);
  t.bloods = 100.0f; t.status = AC_ALIVE;
  t.remaining = c.num_missiles[slot]; t.pre_remaining = c.num_missiles[slot];
  t.last_missile = -1; t.last_shoot_time = -c.min_attack_interval;
  f16::locate(s, d);
  make_props(s, d, c, pr);
  float ob[OBS];
  Incoming inc{false, 0, 0, 0, 0, 0, 0};
  float posture = 0.0f;
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    Enemy E = gather_pose(pr, base + e_first + q);
    Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
    posture += posture_fn(g.AO, g.TA, g.R * 0.001f);
  }
  if (A == 2) {
    Enemy E = gather_pose(pr, base + e_first);
    for (int k = 0; k < OBS; ++k) ob[k] = 0.0f;
    if (c.task == AC_TASK_WVR || c.task == AC_TASK_MANEUVER) observe_1v1<AC_TASK_SINGLECOMBAT>(pr, E, inc, ob);
    else observe_1v1<AC_TASK_SHOOT_MISSILE>(pr, E, inc, ob);
  } else if (c.legacy_obs) {
    for (int k = 0; k < OBS; ++k) ob[k] = 0.0f;
    Enemy E = gather_pose(pr, base + e_first + (slot - (team == 0 ? 0 : c.n_ego)));
    observe_1v1<AC_TASK_SHOOT_MISSILE>(pr, E, inc, ob);
  } else {
    for (int k = 0; k < OBS; ++k) ob[k] = 0.0f;
    ob[0] = pr.alt_m / 5000.0f;
    ob[1] = pr.sphi; ob[2] = pr.cphi; ob[3] = pr.stht; ob[4] = pr.ctht;
    ob[5] = pr.ub / 340.0f; ob[6] = pr.vb / 340.0f; ob[7] = pr.wb / 340.0f; ob[8] = pr.vc / 340.0f;
    const int n_mine = team == 0 ? c.n_ego : A - c.n_ego;
    for (int j = 0; j < A; ++j) {
      Enemy E = gather_pose(pr, base + j);
      if (j == slot) continue;
      Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
      const int team_j = j < c.n_ego ? 0 : 1;
      int idx = (team_j == team) ? (j - (team == 0 ? 0 : c.n_ego)) - (j > slot ? 1 : 0) : (n_mine - 1) + (j - e_first);
      ob[9 + idx * 6 + 0] = (E.ub - pr.ub) / 340.0f; ob[9 + idx * 6 + 1] = (E.alt - pr.alt_m) / 1000.0f;
      ob[9 + idx * 6 + 2] = g.AO; ob[9 + idx * 6 + 3] = g.TA; ob[9 + idx * 6 + 4] = g.R / 10000.0f; ob[9 + idx * 6 + 5] = g.side;
    }
  }
  if (c.altitude_pot) t.pre_altitude = altitude_raw(pr, c) * c.altitude_scale;
  if (c.posture_pot) t.pre_posture = posture * c.posture_scale;
  if (threadIdx.x < A) {
    store_state(tF, tI, tD, A, slot, s, t);
    float* tobs = tF + (size_t)NSW * A + slot * OBS;
    for (int k = 0; k < OBS; ++k) tobs[k] = ob[k];
  }
}

__global__ void reset_ext_kernel(DevCfg c, float* XF, int* XI) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= c.N) return;
  Ext x = fresh_ext(c.num_missiles[n % c.A]);
  store_ext(XF, XI, c.N, n, x, ExtLoaded{}, true);
}
