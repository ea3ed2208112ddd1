// libaircombat_hip.so — MI355X (gfx950) vectorised air-combat step() behind the C ABI of include/aircombat.h.
//
// One kernel launch advances every aircraft of every env by one env step (6 FDM ticks), runs the missile
// engine, builds observations, rewards and terminations, and resets finished episodes — the work the reference
// spreads over SubprocVecEnv workers (envs/env_wrappers.py:182-320), BaseEnv.step (envs/JSBSim/envs/env_base.py:115-173),
// AircraftSimulator/MissileSimulator (envs/JSBSim/core/simulatior.py) and the task/reward/termination classes.
//
// Layout in HBM: struct-of-arrays, lane-major. Field f of aircraft n lives at F[f*N + n] (N = E*A), so the 64
// lanes of a wavefront read 256 contiguous bytes per field (one coalesced request). The A aircraft of one env
// occupy adjacent lanes (A in {1,2,4,8} divides 64), so all pairwise combat geometry is exchanged with
// __shfl inside the wavefront; no LDS or global traffic is needed for it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../include/aircombat.h"
#include "../../include/aircombat_buffer.h"
#include "clk_stamps.hpp"
#include "f16_device.hpp"
#include "f16_split.hpp"

using f16::State;
using f16::Derived;
using f16::Tab;
using f16::clampf;

// ------------------------------------------------------------------------------------------------ state layout
#define AC_F_FIELDS(X)                                                                                           \
  X(vx) X(vy) X(vz) X(q0) X(q1) X(q2) X(q3) X(wp) X(wq) X(wr) X(hv1x) X(hv1y) X(hv1z) X(hv2x) X(hv2y) X(hv2z)   \
  X(ha1x) X(ha1y) X(ha1z) X(wdx) X(wdy) X(wdz) X(aix) X(aiy) X(aiz) X(bax) X(bay) X(baz) X(da) X(de) X(dr)       \
  X(thr) X(pin_r) X(pin_p) X(pin_y) X(pi_r) X(pi_p) X(pi_y) X(tef) X(ail) X(elev) X(sbdeg) X(alpha) X(mach)      \
  X(qc) X(vg) X(ap) X(aq) X(ar) X(npx) X(npy) X(npz) X(n1) X(n2) X(n2norm) X(ff) X(tank0) X(tank1)
#define AC_TF_FIELDS(X) X(bloods) X(pre_posture) X(pre_altitude) X(pre_event) X(pre_shoot)
#define AC_TI_FIELDS(X)                                                                                  \
  X(status) X(die_flag) X(remaining) X(pre_remaining) X(shoot_action) X(last_missile) X(last_shoot_time) \
  X(lock_bits) X(lock_pos) X(cur_step)

enum {
#define X(n) FF_##n,
  AC_F_FIELDS(X) AC_TF_FIELDS(X)
#undef X
  NF
};
enum {
  FI_eng, FI_ticks,
#define X(n) FI_##n,
  AC_TI_FIELDS(X)
#undef X
  NI
};
enum { ND = 3 };
// ---- storage. The names above are the EXTERNAL state vector (ac_get_state / ac_set_state and what the tests exchange through them). In HBM an aircraft's
// 75 32-bit words live in 19 groups of four: group g of aircraft n is the 16 bytes at S4[g * N + n], so every state access of a wave is one
// dwordx4 instruction moving one contiguous 1 KB (round 3 moved a dword per lane per instruction: 173 loads + 90 stores per workgroup in the
// BASELINE kernel, ~28 issue cycles each). Groups are cut by who reads them in the three-wave form -- 0-3 what the systems wave owns,
// 4-6 the air data it reads (with the two flight ints), 6-10 what the kinematics wave reads, 11-14 the dynamics wave's alone -- and 15-18
// are the task record (the environment layer's: group 15 changes every step, 16-18 a few times per episode). The fp64 position: (rx, ry)
// as one 16-byte pair at D2[n], rz at D[2 N + n].
enum : int {
  SW_tef = 0, SW_pin_r, SW_pin_p, SW_pin_y,
  SW_pi_r, SW_pi_p, SW_pi_y, SW_ail,
  SW_elev, SW_sbdeg, SW_n1, SW_n2,
  SW_n2norm, SW_ff, SW_tank0, SW_tank1,
  SW_alpha, SW_mach, SW_qc, SW_vg,
  SW_ap, SW_aq, SW_ar, SW_npy,
  SW_npz, SW_npx, SW_eng, SW_ticks,
  SW_vx, SW_vy, SW_vz, SW_q0,
  SW_q1, SW_q2, SW_q3, SW_wp,
  SW_wq, SW_wr, SW_hv1x, SW_hv1y,
  SW_hv1z, SW_hv2x, SW_hv2y, SW_hv2z,
  SW_ha1x, SW_ha1y, SW_ha1z, SW_wdx,
  SW_wdy, SW_wdz, SW_aix, SW_aiy,
  SW_aiz, SW_bax, SW_bay, SW_baz,
  SW_da, SW_de, SW_dr, SW_thr,
  SW_pre_posture, SW_pre_altitude, SW_pre_event, SW_cur_step,          // group 15
  SW_bloods, SW_status, SW_die_flag, SW_pre_shoot,                     // group 16
  SW_remaining, SW_pre_remaining, SW_shoot_action, SW_last_missile,    // group 17
  SW_last_shoot_time, SW_lock_bits, SW_lock_pos, SW_pad,               // group 18 (the pad word stays 0)
  NSW
};
enum { NSG = NSW / 4, SG_TASK0 = 15 };
static_assert(NSW == 76 && SW_pre_posture == 4 * SG_TASK0, "19 groups: 15 flight, 4 task");
// storage word of every external field, in the external order
static const int kSlotF[NF] = {
#define X(n) SW_##n,
    AC_F_FIELDS(X) AC_TF_FIELDS(X)
#undef X
};
static const int kSlotI[NI] = {SW_eng, SW_ticks,
#define X(n) SW_##n,
                               AC_TI_FIELDS(X)
#undef X
};
// missile slot fields
enum { MF_px, MF_py, MF_pz, MF_vx, MF_vy, MF_vz, MF_theta, MF_psi, MF_t, MF_m, MF_dth, MF_dph, MF_dprev, NMF };
enum { MI_status, MI_recede, MI_order, NMI };
enum { MSL_INACTIVE = -1, MSL_LAUNCHED = 0, MSL_HIT = 1, MSL_MISS = 2 };

static const char* kStateNames[AC_STATE_LEN] = {
    "rx", "ry", "rz",
#define X(n) #n,
    AC_F_FIELDS(X) AC_TF_FIELDS(X) "eng", "ticks", AC_TI_FIELDS(X)
#undef X
    // read-only tail, present for the scenario tasks only
    "x_rem_gun", "x_rem_9m", "x_rem_120b", "x_rem_chaff", "x_bits", "x_last_chaff", "x_orphan_hits", "x_mp_prev", "x_ref_set",
    "x_ch_status0", "x_ch_mult0", "x_ch_status1", "x_ch_mult1", "x_n_ch",
    "x_cg_AO", "x_cg_TA", "x_wez0", "x_wez1", "x_wez2", "x_wez3", "x_tail0", "x_tail1", "x_tail2", "x_tail3",
    "x_c0x", "x_c0y", "x_c0z", "x_c0t", "x_c1x", "x_c1y", "x_c1z", "x_c1t",
};

struct Task {  // per-aircraft task bookkeeping
#define X(n) float n;
  AC_TF_FIELDS(X)
#undef X
#define X(n) int n;
  AC_TI_FIELDS(X)
#undef X
};

// R = float for the 300 m proximity fuse of the 1v1 missile task; R = double for the scenario tasks, whose AIM-120B / AIM-9M
// fuse radius is 5 m against ~25 m of relative travel per tick: hit-or-miss then hinges on centimetres of a 300-tick
// integration, and the reference integrates in Python floats (fp64).
template <typename R>
struct MslT {
  R px, py, pz, vx, vy, vz, theta, psi, t, m, dth, dph, dprev;
  int status, recede, order;
  int dpos;   // position of the munition's uid in env._tempsims (the dict is walked in insertion order, and a uid that is launched again
              // keeps its place): (env step << 4 | agent) of the FIRST launch into this slot since the reset; kept in the upper bits of
              // the `recede` word in HBM
  int model;  // 0 = AIM-120B (or the 1v1 tasks' AIM-9L), 1 = AIM-9M: the same parameter set, a different name in the ACMI record
};
using Msl = MslT<float>;
using MslD = MslT<double>;

// Device-side scenario constants (passed by value to the kernels).
struct DevCfg {
  int task, A, n_ego, substeps, max_steps, obs_dim, act_dim, msl_slots;
  int N;                      // aircraft lanes = E*A
  float altitude_limit, acc_x, acc_y, acc_z;
  float posture_scale, altitude_scale, event_scale, missile_posture_scale, shoot_penalty_scale;
  int posture_pot, altitude_pot, event_pot, shoot_pot;
  float alt_safe, alt_danger, alt_kv;
  float max_attack_angle, max_attack_distance;
  int min_attack_interval, use_artillery, lock_len;
  int rwr;                    // *_RWR variants: obs_dim carries two extra zero slots
  int legacy_obs;             // Scenario2 / Scenario3 (not _NvN): 21-value observation against the paired enemy
  int tobs;                   // observation slots per aircraft in the reset template (the kernel family's own layout)
  int num_missiles[AC_MAX_AGENTS];
  // battle-field origin for pymap3d-style geodetic <-> NED (metres, WGS84)
  double P0x, P0y, P0z, sLat0, cLat0, sLon0, cLon0;
  float rm0, rn0, h0;         // meridional / prime-vertical radius of curvature at the battle-field centre (+ its height), that height
  unsigned long long chaff_seed;  // base of the keyed decoy draw (the env index is added on the device)
};

struct DevPtrs {
  float* F; int* I; double* D;           // live state, SoA [field][N]
  float* MF; int* MI;                    // missiles, SoA [slot][field][N]
  double* MD;                            // scenario tasks: the same layout in fp64 instead of MF
  float* H;                              // hierarchical tasks: GRU state of the low-level controller, [128][N] (else null)
  int* man_step; float* man_h0;          // scripted ManeuverAgent opponents: step counter and latched heading, [N]
  const float* tF; const int* tI; const double* tD;  // reset template, [field][A]
  const float* tab;                      // F16_PACK as fp32 in HBM (staged to LDS per workgroup)
  const float* actions;                  // [N][act_dim]
  float* obs; float* rew; uint8_t* done; int* info;        // outputs, rows padded to whole workgroups (64 aircraft)
  float* obs2; float* rew2; uint8_t* done2; int* info2;    // second copy of the outputs (pinned host memory mapped into the device:
                                                           // ac_step_host), or null
  int* err;                              // one word of page-locked host memory, 0 = healthy: (own state probe << 30) | (0x3fffffff - index) of the aircraft
                                         // whose state or reward went non-finite, merged with a system-scope max (emit_scalars)
};

// ------------------------------------------------------------------------------------------------ non-finite guard
// The reference traps NaN in BaseEnv._pack (env_base.py:277-281, a pdb prompt) and raises RuntimeError("JSBSim failed.") when the FDM gives up
// (simulatior.py:223-225); ExtremeState is meant to end an episode before that (catalog.py:386-416) but its tests are all `>=`, which a NaN
// fails. Here: a NaN / Inf anywhere in the quantities ExtremeState looks at (inertial speed, body rates, altitude, pilot-station load
// factors: every integrator state reaches one of them within a tick) counts as an extreme state, so the aircraft terminates, and poisons
// the step's reward, which emit_scalars reports through P.err; the host then fails the step with the aircraft's index.
__device__ __forceinline__ bool nonfinite_probe(float veci, float pqr, float h_sl_ft, float np_max) {
  return !(fabsf(veci + pqr + h_sl_ft + np_max) < INFINITY);
}
// (a NaN with payload 1: emit_scalars tells "this aircraft's own state probe fired" from a reward that merely inherited a NaN -- the
// opponent's posture term of a NaN pose is NaN as well -- and reports the aircraft that is actually at fault)
__device__ __forceinline__ float poison_if(bool bad, float reward) { return bad ? __int_as_float(0x7fc00001) : reward; }

// ------------------------------------------------------------------------------------------------ device helpers
// Field f of lane n at byte offset (f*N + n)*sizeof from ONE wave-uniform base pointer: the offset is a 32-bit lane value (one VALU
// add per field), so each access is `global_load/store v, v_off, s[base]` instead of per-field 64-bit VALU address arithmetic
// (~4 VALU per field before). Arrays stay below 4 GB (ac_create checks).
#define AC_LANE_INDEX(n) const unsigned un_ = (unsigned)(n), uN_ = (unsigned)N
#define AC_AT(base, f) \
  (*(decltype(base))((char*)(base) + (size_t)((unsigned)(f) * (uN_ * (unsigned)sizeof(*(base))) + un_ * (unsigned)sizeof(*(base)))))
// group g of lane n of the state storage (16 bytes), and one word of it (byte offsets stay below 2^32: 19 groups x 16 B x 2^23 aircraft)
#define AC_GRP(S, g) (*reinterpret_cast<float4*>((char*)(S) + (size_t)((unsigned)(g) * (uN_ * 16u) + un_ * 16u)))
#define AC_WORD(S, w) (*reinterpret_cast<int*>((char*)(S) + (size_t)((unsigned)((w) >> 2) * (uN_ * 16u) + un_ * 16u + 4u * (unsigned)((w) & 3))))
__device__ __forceinline__ int state_word(const float* S, int w, int N, int n) { AC_LANE_INDEX(n); return AC_WORD(S, w); }
// the flight groups that hold four floats (group 6 = npz, npx and the two ints is written out by hand)
#define AC_FLIGHT_GROUPS(G)                                                                                                       \
  G(0, tef, pin_r, pin_p, pin_y) G(1, pi_r, pi_p, pi_y, ail) G(2, elev, sbdeg, n1, n2) G(3, n2norm, ff, tank0, tank1)               \
  G(4, alpha, mach, qc, vg) G(5, ap, aq, ar, npy) G(7, vx, vy, vz, q0) G(8, q1, q2, q3, wp) G(9, wq, wr, hv1x, hv1y)                \
  G(10, hv1z, hv2x, hv2y, hv2z) G(11, ha1x, ha1y, ha1z, wdx) G(12, wdy, wdz, aix, aiy) G(13, aiz, bax, bay, baz) G(14, da, de, dr, thr)
static_assert(SW_alpha == 16 && SW_npz == 24 && SW_vx == 28 && SW_ha1x == 44 && SW_da == 56, "AC_FLIGHT_GROUPS follows the SW_ order");
__device__ __forceinline__ void load_position(const double* D, int N, int n, State& s) {
  const double2 xy = *reinterpret_cast<const double2*>((const char*)D + (size_t)((unsigned)n * 16u));
  s.rx = xy.x; s.ry = xy.y;
  s.rz = D[2 * (size_t)N + n];
}
__device__ __forceinline__ void store_position(double* D, int N, int n, const State& s) {
  *reinterpret_cast<double2*>((char*)D + (size_t)((unsigned)n * 16u)) = make_double2(s.rx, s.ry);
  D[2 * (size_t)N + n] = s.rz;
}
// the flight-dynamics part of an aircraft's state (what the FDM tick reads and writes) ...
// ROLE -1: all of it. Three-wave form: each wave asks only for what its share of the tick reads (the 64 aircraft's state would otherwise
// cross the L2 -> CU path three times at the start of every step). ROLE 0 = dynamics wave: everything except the groups the systems wave
// owns and hands over at the end (dynamics_wave_finish); 1 = systems wave: what sys_mass / sys_fcs / sys_engine read; 2 = kinematics
// wave: what kin_position / kin_attitude / locate read.
constexpr bool ac_role_reads(int role, int g) {
  return role < 0 || (role == 0 ? g >= 4 : (role == 1 ? g <= 6 : (g >= 6 && g <= 10)));
}
template <int ROLE>
__device__ __forceinline__ void load_flight_role(const float* F, const int* I, const double* D, int N, int n, State& s) {
  (void)I;
  AC_LANE_INDEX(n);
#define G(g, a, b, c, d) if (ac_role_reads(ROLE, g)) { const float4 v = AC_GRP(F, g); s.a = v.x; s.b = v.y; s.c = v.z; s.d = v.w; }
  AC_FLIGHT_GROUPS(G)
#undef G
  if (ac_role_reads(ROLE, 6)) {
    const float4 v = AC_GRP(F, 6);
    s.npz = v.x; s.npx = v.y; s.eng = __float_as_int(v.z); s.ticks = __float_as_int(v.w);
  }
  if (ROLE != 1) load_position(D, N, n, s);
}
__device__ __forceinline__ void load_flight(const float* F, const int* I, const double* D, int N, int n, State& s) { load_flight_role<-1>(F, I, D, N, n, s); }
__device__ __forceinline__ void store_flight(float* F, int* I, double* D, int N, int n, const State& s) {
  (void)I;
  AC_LANE_INDEX(n);
#define G(g, a, b, c, d) AC_GRP(F, g) = make_float4(s.a, s.b, s.c, s.d);
  AC_FLIGHT_GROUPS(G)
#undef G
  AC_GRP(F, 6) = make_float4(s.npz, s.npx, __int_as_float(s.eng), __int_as_float(s.ticks));
  store_position(D, N, n, s);
}
// ... and the task bookkeeping (what the environment layer reads and writes): groups 15-18
__device__ __forceinline__ void load_task(const float* F, const int* I, int N, int n, Task& t) {
  (void)I;
  AC_LANE_INDEX(n);
  const float4 a = AC_GRP(F, SG_TASK0), b = AC_GRP(F, SG_TASK0 + 1), c = AC_GRP(F, SG_TASK0 + 2), d = AC_GRP(F, SG_TASK0 + 3);
  t.pre_posture = a.x; t.pre_altitude = a.y; t.pre_event = a.z; t.cur_step = __float_as_int(a.w);
  t.bloods = b.x; t.status = __float_as_int(b.y); t.die_flag = __float_as_int(b.z); t.pre_shoot = b.w;
  t.remaining = __float_as_int(c.x); t.pre_remaining = __float_as_int(c.y); t.shoot_action = __float_as_int(c.z); t.last_missile = __float_as_int(c.w);
  t.last_shoot_time = __float_as_int(d.x); t.lock_bits = __float_as_int(d.y); t.lock_pos = __float_as_int(d.z);
}
__device__ __forceinline__ void store_task_group0(float* F, int N, int n, const Task& t) {
  AC_LANE_INDEX(n);
  AC_GRP(F, SG_TASK0) = make_float4(t.pre_posture, t.pre_altitude, t.pre_event, __int_as_float(t.cur_step));
}
__device__ __forceinline__ void store_task_rare(float* F, int N, int n, const Task& t) {
  AC_LANE_INDEX(n);
  AC_GRP(F, SG_TASK0 + 1) = make_float4(t.bloods, __int_as_float(t.status), __int_as_float(t.die_flag), t.pre_shoot);
  AC_GRP(F, SG_TASK0 + 2) = make_float4(__int_as_float(t.remaining), __int_as_float(t.pre_remaining), __int_as_float(t.shoot_action), __int_as_float(t.last_missile));
  AC_GRP(F, SG_TASK0 + 3) = make_float4(__int_as_float(t.last_shoot_time), __int_as_float(t.lock_bits), __int_as_float(t.lock_pos), 0.0f);
}
__device__ __forceinline__ void store_task(float* F, int* I, int N, int n, const Task& t) {
  (void)I;
  store_task_group0(F, N, n, t);
  store_task_rare(F, N, n, t);
}
__device__ __forceinline__ void load_state(const float* F, const int* I, const double* D, int N, int n, State& s, Task& t) {
  load_flight(F, I, D, N, n, s);
  load_task(F, I, N, n, t);
}
__device__ __forceinline__ void store_state(float* F, int* I, double* D, int N, int n, const State& s, const Task& t) {
  store_flight(F, I, D, N, n, s);
  store_task(F, I, N, n, t);
}
template <typename R>
__device__ __forceinline__ void load_msl(const R* MF, const int* MI, int N, int n, int slot, MslT<R>& m) {
  AC_LANE_INDEX(n);
  const R* f = MF + (size_t)slot * NMF * N;
  const int* i = MI + (size_t)slot * NMI * N;
  m.px = AC_AT(f, MF_px); m.py = AC_AT(f, MF_py); m.pz = AC_AT(f, MF_pz);
  m.vx = AC_AT(f, MF_vx); m.vy = AC_AT(f, MF_vy); m.vz = AC_AT(f, MF_vz);
  m.theta = AC_AT(f, MF_theta); m.psi = AC_AT(f, MF_psi); m.t = AC_AT(f, MF_t); m.m = AC_AT(f, MF_m);
  m.dth = AC_AT(f, MF_dth); m.dph = AC_AT(f, MF_dph); m.dprev = AC_AT(f, MF_dprev);
  m.status = AC_AT(i, MI_status); m.order = AC_AT(i, MI_order);
  const int rw = AC_AT(i, MI_recede);
  m.recede = rw & 511; m.model = (rw >> 9) & 1; m.dpos = rw >> 10;
}
template <typename R>
__device__ __forceinline__ void store_msl(R* MF, int* MI, int N, int n, int slot, const MslT<R>& m) {
  AC_LANE_INDEX(n);
  R* f = MF + (size_t)slot * NMF * N;
  int* i = MI + (size_t)slot * NMI * N;
  AC_AT(f, MF_px) = m.px; AC_AT(f, MF_py) = m.py; AC_AT(f, MF_pz) = m.pz;
  AC_AT(f, MF_vx) = m.vx; AC_AT(f, MF_vy) = m.vy; AC_AT(f, MF_vz) = m.vz;
  AC_AT(f, MF_theta) = m.theta; AC_AT(f, MF_psi) = m.psi; AC_AT(f, MF_t) = m.t; AC_AT(f, MF_m) = m.m;
  AC_AT(f, MF_dth) = m.dth; AC_AT(f, MF_dph) = m.dph; AC_AT(f, MF_dprev) = m.dprev;
  AC_AT(i, MI_status) = m.status; AC_AT(i, MI_recede) = (m.dpos << 10) | (m.model << 9) | m.recede; AC_AT(i, MI_order) = m.order;
}

// a munition entry that did not move this step: its clock, `dprev`, status and receding count are all that changed
template <typename R>
__device__ __forceinline__ void store_msl_clock(R* MF, int* MI, int N, int n, int slot, const MslT<R>& m) {
  AC_LANE_INDEX(n);
  R* f = MF + (size_t)slot * NMF * N;
  int* i = MI + (size_t)slot * NMI * N;
  AC_AT(f, MF_t) = m.t; AC_AT(f, MF_dprev) = m.dprev;
  AC_AT(i, MI_status) = m.status; AC_AT(i, MI_recede) = (m.dpos << 10) | (m.model << 9) | m.recede;
}
// the task record: the step counter and the three potentials (group 15) change every step and are always written; the other eleven words
// (status, blood, the launch bookkeeping: groups 16-18) change a few times per episode and are written together, under ONE test, when any
// of them differs from what was loaded (`was`)
__device__ __forceinline__ void store_task_changed(float* F, int* I, int N, int n, const Task& t, const Task& was) {
  (void)I;
  store_task_group0(F, N, n, t);
  const bool rare = __float_as_int(t.bloods) != __float_as_int(was.bloods) || __float_as_int(t.pre_shoot) != __float_as_int(was.pre_shoot) ||
                    t.status != was.status || t.die_flag != was.die_flag || t.remaining != was.remaining || t.pre_remaining != was.pre_remaining ||
                    t.shoot_action != was.shoot_action || t.last_missile != was.last_missile || t.last_shoot_time != was.last_shoot_time ||
                    t.lock_bits != was.lock_bits || t.lock_pos != was.lock_pos;
  if (rare) store_task_rare(F, N, n, t);
}

// task.reset() of the hierarchical tasks clears _inner_rnn_states (singlecombat_task.py:258-262)
__device__ __forceinline__ void zero_controller_state(const DevPtrs& P, int N, int n, bool live) {
  if (P.H && live) {
    for (int k = 0; k < 128; ++k) P.H[(size_t)k * N + n] = 0.0f;
    P.man_step[n] = 0; P.man_h0[n] = 0.0f;   // BaselineAgent.reset (baseline.py:37-38,133-136)
  }
}

// What the Python wrapper caches after every JSBSim run (simulatior.py:238-258) plus the clipped unit
// conversions of the catalogue (catalog.py:292-338).
struct Props {
  float n, e, u;            // NEU position about the battle-field centre [m]
  double n64, e64, u64;     // the same before rounding (missile targets of the scenario tasks)
  float vn, ve, vd;         // m/s, clipped to +-700
  float alt_m;              // clipped to [-500, 26000]
  float ub, vb, wb, vc;     // body velocities and calibrated airspeed [m/s], clipped
  float sphi, cphi, stht, ctht;
  float m11, m12;           // for the heading angle
};
__device__ __forceinline__ float mps(float fps) { return clampf(-700.0f, fps * f16::kFt2M, 700.0f); }

// the part a munition's guidance reads: NEU position (fp64 and rounded), NED velocity, altitude
__device__ __forceinline__ void make_pose(const Derived& d, const DevCfg& c, Props& p) {
  p.alt_m = clampf(-500.0f, d.h_sl_ft * f16::kFt2M, 26000.0f);
  p.vn = mps(d.vn); p.ve = mps(d.ve); p.vd = mps(d.vd);
  // LLA2NEU(lon, lat_geod, h_sl_m): the reference feeds the sea-level altitude to pymap3d.geodetic2ned as if it
  // were ellipsoidal height (simulatior.py:240-245, utils.py:30-41). fp64: differences of 6.4e6 m ECEF coordinates.
  const double a = 6378137.0, b = 6356752.314245179;  // pymap3d WGS84: a, a*(1-1/298.257223563)
  double sl = d.sLat64, cl = d.cLat64, so = d.sLon64, co = d.cLon64;
  double Nn = a * a * fx::rsqrt(a * a * cl * cl + b * b * sl * sl);
  double h = (double)p.alt_m;
  double x = (Nn + h) * cl * co, y = (Nn + h) * cl * so, z = (Nn * (b / a) * (b / a) + h) * sl;
  double dx = x - c.P0x, dy = y - c.P0y, dz = z - c.P0z;
  double t = c.cLon0 * dx + c.sLon0 * dy;
  p.e64 = -c.sLon0 * dx + c.cLon0 * dy; p.u64 = c.cLat0 * t + c.sLat0 * dz; p.n64 = -c.sLat0 * t + c.cLat0 * dz;
  p.e = (float)p.e64; p.u = (float)p.u64; p.n = (float)p.n64;
}
__device__ __forceinline__ void make_props(const State& s, const Derived& d, const DevCfg& c, Props& p) {
  make_pose(d, c, p);
  p.ub = mps(d.u); p.vb = mps(d.v); p.wb = mps(d.w);
  p.vc = clampf(0.0f, f16::vcas_from_impact_pressure(s.qc) * f16::kFt2M, 1400.0f);
  // Euler sines/cosines from Tl2b = Ti2b * Ti2l^T  (only the five entries GetEuler reads)
  const float* T = d.T;
  float m13 = T[0] * d.d_eci[0] + T[1] * d.d_eci[1] + T[2] * d.d_eci[2];
  float m23 = T[3] * d.d_eci[0] + T[4] * d.d_eci[1] + T[5] * d.d_eci[2];
  float m33 = T[6] * d.d_eci[0] + T[7] * d.d_eci[1] + T[8] * d.d_eci[2];
  p.m11 = T[0] * d.n_eci[0] + T[1] * d.n_eci[1] + T[2] * d.n_eci[2];
  p.m12 = T[0] * d.e_eci[0] + T[1] * d.e_eci[1] + T[2] * d.e_eci[2];
  p.stht = clampf(-1.0f, -m13, 1.0f);
  float ct = sqrtf(fmaxf(0.0f, 1.0f - p.stht * p.stht));
  p.ctht = ct;
  float ic = (ct > 1e-12f) ? 1.0f / sqrtf(m23 * m23 + m33 * m33) : 0.0f;
  p.sphi = (ct > 1e-12f) ? m23 * ic : 0.0f;
  p.cphi = (ct > 1e-12f) ? m33 * ic : 1.0f;
}

// utils.py:58-103. v = (vN, vE, vDOWN) against an (N, E, UP) position, exactly as the reference mixes them.
struct Geo { float AO, TA, R, side, cAO, cTA; };   // (cAO / cTA: the clipped cosines the angles were taken from)
// acos for an argument already clipped to [-1, 1]: atan2(sqrt((1 - x)(1 + x)), x) on the polynomial atan2 (8e-8 rad); the product
// form keeps the sine exact to an ulp where the angle is ill-conditioned (x -> +-1)
__device__ __forceinline__ float acos_fast(float x) { return f16::atan2_fast(sqrtf((1.0f - x) * (1.0f + x)), x); }
// tanh on the transcendental unit (|error| < 1e-6 absolute), atanh for x in (-1, 1)
__device__ __forceinline__ float tanh_fast(float x) {
  const float t = __expf(2.0f * clampf(-10.0f, x, 10.0f));
  return (t - 1.0f) / (t + 1.0f);
}
__device__ __forceinline__ float atanh_fast(float x) { return 0.5f * __logf((1.0f + x) / (1.0f - x)); }
template <bool TWO_D>
__device__ __forceinline__ Geo ao_ta_r(float ex, float ey, float ez, float evx, float evy, float evz,
                                       float nx, float ny, float nz, float nvx, float nvy, float nvz) {
  float dx = nx - ex, dy = ny - ey, dz = nz - ez;
  float ev, nv, R, pe, pn;
  if (TWO_D) {
    ev = sqrtf(evx * evx + evy * evy); nv = sqrtf(nvx * nvx + nvy * nvy); R = sqrtf(dx * dx + dy * dy);
    pe = dx * evx + dy * evy; pn = dx * nvx + dy * nvy;
  } else {
    ev = sqrtf(evx * evx + evy * evy + evz * evz); nv = sqrtf(nvx * nvx + nvy * nvy + nvz * nvz);
    R = sqrtf(dx * dx + dy * dy + dz * dz);
    pe = dx * evx + dy * evy + dz * evz; pn = dx * nvx + dy * nvy + dz * nvz;
  }
  Geo g;
  g.cAO = clampf(-1.0f, pe / (R * ev + 1e-8f), 1.0f);
  g.cTA = clampf(-1.0f, pn / (R * nv + 1e-8f), 1.0f);
  g.AO = acos_fast(g.cAO);
  g.TA = acos_fast(g.cTA);
  g.R = R;
  float cr = evx * dy - evy * dx;
  g.side = (cr > 0.0f) ? 1.0f : ((cr < 0.0f) ? -1.0f : 0.0f);
  return g;
}
// posture_reward.py:58-75 (orientation v2, range v3)
__device__ __forceinline__ float posture_fn(float AO, float TA, float Rkm) {
  // TA = pi exactly (the target flying straight away along the line of sight) cannot happen in the reference: its float64 cosine
  // stays above -1 by the 1e-8 in the denominator (utils.py:74-76) and atanh bottoms out near -8.5. In fp32 the cosine does round
  // to -1 once the true angle is within 3.5e-4 rad of pi, and atanh(-1) would be -inf: the argument is kept one ulp inside
  // (atanh = -8.66, the reference's own floor), so a reward is never infinite.
  float x = fmaxf(1.0f - fmaxf(2.0f * TA / f16::kPi, 1e-4f), -0.99999994f);
  float orn = 1.0f / (50.0f * AO / f16::kPi + 2.0f) + 0.5f + fminf(atanh_fast(x) / (2.0f * f16::kPi), 0.0f) + 0.5f;
  float rng = (Rkm < 5.0f ? 1.0f : 0.0f) + (Rkm >= 5.0f ? clampf(0.0f, -0.032f * Rkm * Rkm + 0.284f * Rkm + 0.38f, 1.0f) : 0.0f) +
              clampf(0.0f, __expf(-0.16f * Rkm), 0.2f);
  return orn * rng;
}
// altitude_reward.py:20-40
__device__ __forceinline__ float altitude_fn(float z_km, float vz_mh, const DevCfg& c) {
  float Pv = 0.0f, PH = 0.0f;
  if (z_km <= c.alt_safe) Pv = -clampf(0.0f, vz_mh / c.alt_kv * (c.alt_safe - z_km) / c.alt_safe, 1.0f);
  if (z_km <= c.alt_danger) PH = clampf(0.0f, z_km / c.alt_danger, 1.0f) - 2.0f;
  return Pv + PH;
}
__device__ __forceinline__ float potential(float r, float scale, int pot, float& pre) {
  r *= scale;
  if (pot) { float o = r - pre; pre = r; return o; }
  return r;
}

// ------------------------------------------------------------------------------------------------ missile engine
// The reference accumulates the missile clock in a Python float (`self._t += self.dt`, simulatior.py:522) and compares it with
// t_thrust / t_max; 84 additions of 1/60 land a hair ABOVE 1.4, 180 of them a hair below 3.0, so which tick the motor burns
// out on is a property of that double-precision sum. The thresholds are therefore turned into tick counts at compile time with
// the same sum, and the device keeps an exact tick count.
constexpr int first_tick_not_below(double limit) {   // smallest k with t_k >= limit, t_k = k-fold sum of 1/60 in fp64
  double t = 0.0; int k = 0;
  do { t += 1.0 / 60.0; ++k; } while (t < limit);
  return k;
}
constexpr int first_tick_above(double limit) {       // smallest k with t_k > limit
  double t = 0.0; int k = 0;
  do { t += 1.0 / 60.0; ++k; } while (!(t > limit));
  return k;
}
// Overload only counts after "simulation/sim-time-sec > 10" (overload.py:30): JSBSim's clock is the fp64 running sum of dT
// (FGFDMExec.cpp:196-203), and 600 additions of 1/60 already exceed 10, so the rule switches on at tick 600, not 601.
constexpr int kTickOverload = first_tick_above(10.0);
struct MslParam { float g, t_max, t_thrust, Isp, Length, Diameter, cD, m0, dm, K, nyz_max, Rc, v_min; int recede_max, k_burnout, k_timeout; };
__device__ __forceinline__ MslParam aim9l() {  // simulatior.py:421-433
  constexpr int kb = first_tick_not_below(3.0), kt = first_tick_above(60.0);
  return MslParam{9.81f, 60.0f, 3.0f, 120.0f, 2.87f, 0.127f, 0.4f, 84.0f, 6.0f, 3.0f, 30.0f, 300.0f, 150.0f, 300, kb, kt};
}
// geodetic height of an NEU point (utils.py:44-55 -> pymap3d.ned2geodetic): ENU -> ECEF offset in fp64, then the
// closed-form height of Fukushima's reduction (sub-millimetre, like pymap3d's You-2000 form)
__device__ __forceinline__ double neu_height64(double n, double e, double u, const DevCfg& c) {
  double t = c.cLat0 * u - c.sLat0 * n;
  double dz = c.sLat0 * u + c.cLat0 * n;
  double dx = c.cLon0 * t - c.sLon0 * e;
  double dy = c.sLon0 * t + c.cLon0 * e;
  double X = c.P0x + dx, Y = c.P0y + dy, Z = c.P0z + dz;
  const double a = 6378137.0, b = 6356752.314245179, ec = b / a, ec2 = ec * ec, cc0 = a * (1.0 - ec2);
  double rxy = sqrt(X * X + Y * Y);
  double s0 = fabs(Z), zc = ec * s0, c0 = ec * rxy, c02 = c0 * c0, s02 = s0 * s0, a02 = c02 + s02;
  double a0 = sqrt(a02), a03 = a02 * a0;
  double s1 = zc * a03 + cc0 * s02 * s0, c1 = rxy * a03 - cc0 * c02 * c0, cs = cc0 * c0 * s0;
  double b0 = 1.5 * cs * ((rxy * s0 - zc * c0) * a0 - cs);
  s1 = s1 * a03 - b0 * s0;
  double cc = ec * (c1 * a03 - b0 * c0);
  double s12 = s1 * s1, cc2 = cc * cc, norm = sqrt(s12 + cc2);
  return ((rxy * cc + s0 * s1 - a * sqrt(ec2 * s12 + cc2)) / norm);
}
__device__ __forceinline__ float neu_height(float n, float e, float u, const DevCfg& c) { return (float)neu_height64(n, e, u, c); }
// A missile only uses its geodetic height for the air density 1.225 exp(-h / 9300): the local-curvature form
// h0 + u + n^2 / 2(M0 + h) + e^2 / 2(N0 + h) is within 4 cm of the exact reduction over +-60 km of the battle-field centre (3.5 mm
// within 40 km), i.e. a density error below 5e-6 -- smaller than what the fp32 flight model's centimetres of target position do to
// an intercept -- and costs a handful of operations instead of ~160 fp64 ones with four square roots (neu_height64 above).
__device__ __forceinline__ float missile_height(float n, float e, float u, const DevCfg& c) {
  return c.h0 + u + n * n / (2.0f * (c.rm0 + u)) + e * e / (2.0f * (c.rn0 + u));
}
__device__ __forceinline__ double missile_height(double n, double e, double u, const DevCfg& c) {
  return (double)c.h0 + u + n * n * fx::rcp(2.0 * ((double)c.rm0 + u)) + e * e * fx::rcp(2.0 * ((double)c.rn0 + u));
}
// MissileSimulator.run (simulatior.py:520-533) with _guidance (:556-576) and _state_trans (:578-608).
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_sin(float x) { return __sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_exp(float x) { return __expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ void m_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void m_sincos(double x, double* s, double* c) { sincos(x, s, c); }
template <typename R>
__device__ __forceinline__ R m_clamp(R lo, R x, R hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <typename R>
__device__ __forceinline__ void missile_run(MslT<R>& m, const MslParam& P, R tx, R ty, R tz, R tvx, R tvy, R tvz,
                                            bool target_alive, const DevCfg& c) {
  const R dt = (R)(1.0 / 60.0);
  const int k = (int)rint((double)m.t * 60.0) + 1;   // ticks since launch, exact (the stored clock is always k/60 rounded once)
  m.t = (R)k * dt;
  const bool burning = k < P.k_burnout;              // reference: t < t_thrust
  const R g = (R)P.g, t_max = (R)P.t_max, nyz_max = (R)P.nyz_max;
  R vm = m_sqrt(m.vx * m.vx + m.vy * m.vy + m.vz * m.vz);
  R cth = m_sqrt(fmax((R)0, (R)1 - (m.vz / vm) * (m.vz / vm)));  // cos(asin(dz/v))
  R ddx = tx - m.px, ddy = ty - m.py, ddz = tz - m.pz;
  R Rxy2 = ddx * ddx + ddy * ddy, Rxy = m_sqrt(Rxy2);
  R R2 = Rxy2 + ddz * ddz, Rxyz = m_sqrt(R2);
  R dbeta = ((tvy - m.vy) * ddx - (tvx - m.vx) * ddy) / Rxy2;
  R deps = ((tvz - m.vz) * Rxy2 - ddz * (ddx * (tvx - m.vx) + ddy * (tvy - m.vy))) / (R2 * Rxy);
  R K = fmax((R)P.K * (t_max - m.t) / t_max, (R)0);
  R ny = m_clamp(-nyz_max, K * vm / g * cth * dbeta, nyz_max);
  R nz = m_clamp(-nyz_max, K * vm / g * deps + cth, nyz_max);
  m.recede = (Rxyz > m.dprev) ? min(m.recede + 1, P.recede_max) : 0;   // (300 in a row decide; the count saturates there)
  m.dprev = Rxyz;
  if (Rxyz < (R)P.Rc && target_alive && m.status != MSL_MISS) {
    m.status = MSL_HIT;
  } else if (k >= P.k_timeout || vm < (R)P.v_min || m.recede >= P.recede_max || !target_alive) {
    m.status = MSL_MISS;
  } else {
    m.px += dt * m.vx; m.py += dt * m.vy; m.pz += dt * m.vz;
    R alt = missile_height(m.px, m.py, m.pz, c);
    R Isp = burning ? (R)P.Isp : (R)0;
    R Tt = g * Isp * (R)P.dm;
    R sd = m_sin(m.dth), sp = m_sin(m.dph);
    const R D0 = (R)P.Diameter, L0 = (R)P.Length;
    R S = (R)3.14159265358979323846 * (R)0.25 * D0 * D0 + m_sqrt(sd * sd + sp * sp) * D0 * L0;
    R rho = (R)1.225 * m_exp(-alt / (R)9300);
    R D = (R)0.5 * (R)P.cD * S * rho * vm * vm;
    R nx = (Tt - D) / (m.m * g);
    R st, ct; m_sincos(m.theta, &st, &ct);
    R dv = g * (nx - st);
    m.dph = g / vm * (ny / ct);
    m.dth = g / vm * (nz - ct);
    R v = vm + dt * dv;
    m.psi += dt * m.dph; m.theta += dt * m.dth;
    R s2, c2, s3, c3; m_sincos(m.theta, &s2, &c2); m_sincos(m.psi, &s3, &c3);
    m.vx = v * c2 * c3; m.vy = v * c2 * s3; m.vz = v * s2;
    if (burning) m.m -= dt * (R)P.dm;
  }
}

// fp64 munitions of the scenario tasks: the same update, still entirely in fp64, without the three fp64 sincos calls (~200
// instructions each). After its first update a missile's velocity IS (v cos(theta) cos(psi), v cos(theta) sin(psi), v sin(theta)),
// so sin / cos of the current angles are ratios of the velocity components, and the angles after the update follow by the
// angle-addition formulas with the per-tick increment (|d| <= 0.03 rad: degree-9 Taylor terms are below 1e-18). Only the first
// tick after launch, whose velocity was inherited from the aircraft (with its down-for-up z component) while theta / psi came
// from the aircraft attitude, evaluates the trigonometric functions.
__device__ __forceinline__ void small_sincos(double d, double* s, double* c) {
  const double d2 = d * d;
  *s = d * (1.0 + d2 * (-1.0 / 6.0 + d2 * (1.0 / 120.0 + d2 * (-1.0 / 5040.0 + d2 * (1.0 / 362880.0)))));
  *c = 1.0 + d2 * (-0.5 + d2 * (1.0 / 24.0 + d2 * (-1.0 / 720.0 + d2 * (1.0 / 40320.0))));
}
// fp32 sine of a turn rate (|x| up to a few pi): nearest multiple of pi off in two pieces, odd Taylor polynomial to y^11 on [-pi/2, pi/2]
// (truncation 6e-8; v_sin_f32 alone is 4e-7 absolute, which the reference-area term of the drag would see as 1e-5 relative)
__device__ __forceinline__ float sin_rate(float x) {
  const float n = rintf(x * 0.318309886f);
  float y = fmaf(-n, 3.14159274f, x);
  y = fmaf(-n, -8.74227766e-8f, y);
  const float t = y * y;
  float p = -2.50521084e-8f;               // -1/11!
  p = fmaf(p, t, 2.75573192e-6f);          //  1/9!
  p = fmaf(p, t, -1.98412698e-4f);         // -1/7!
  p = fmaf(p, t, 8.33333333e-3f);          //  1/5!
  p = fmaf(p, t, -1.66666667e-1f);         // -1/3!
  const float r = fmaf(y * t, p, y);
  return ((int)n & 1) ? -r : r;
}
// What stays in fp64 and what does not. A munition's STATE is integrated in fp64 -- position, speed, pitch / heading and the velocity
// vector rebuilt from them by the angle-addition formulas, the clock, the mass -- and so are the line of sight, the range that the 5 m
// fuse, the receding count and `dprev` see, and the proportional-navigation demands with the two turn rates they give (near a pass the
// demands hinge on centimetres: taking them in fp32 was measured, 2v2 "closing" parity case 1.05 -> 12.2 times the observation bound).
// The DRAG side of a tick -- geodetic height, air density, reference area from the two turn rates, thrust minus drag, dv/dt -- is fp32:
// it only moves the speed (1e-7 relative on an acceleration of a few g: millimetres over a flight; the parity cases use the same
// fraction of their bounds with it, tools/diag/missile_drift.py), and it held a third of the update's fp64 instructions, whose
// reciprocal / square root / exp / sin are 8-15 instruction sequences where fp32 has one each: scenario1 28.9 -> 26.4 us per step.
// Returns whether the entry MOVED (took the state-transition branch): a finished entry that stands still only advances its clock, its
// receding count and `dprev`, and the caller writes back just those.
__device__ __forceinline__ bool missile_run(MslD& m, const MslParam& P, double tx, double ty, double tz, double tvx, double tvy, double tvz,
                                            bool target_alive, const DevCfg& c) {
  const double dt = 1.0 / 60.0;
  const int k = (int)rint(m.t * 60.0) + 1;
  m.t = (double)k * dt;
  const bool burning = k < P.k_burnout;
  const double g = P.g, t_max = P.t_max, nyz_max = P.nyz_max;
  // (every length comes with its reciprocal from the one refined rsqrt seed: 1 / Rxy^2 = (1 / Rxy)^2, 1 / (R^2 Rxy) = (1 / R)^2 (1 / Rxy);
  //  cos(asin(vz / v)) IS hxy / v)
  const double hxy2 = m.vx * m.vx + m.vy * m.vy;
  double vm, ivm, hxy = 0.0, ih = 0.0;
  fx::sqrt_both(hxy2 + m.vz * m.vz, &vm, &ivm);
  if (hxy2 > 0.0) fx::sqrt_both(hxy2, &hxy, &ih);
  const double cth = hxy * ivm;
  const double ddx = tx - m.px, ddy = ty - m.py, ddz = tz - m.pz;
  const double Rxy2 = ddx * ddx + ddy * ddy, R2 = Rxy2 + ddz * ddz;
  double Rxy, iRxy, Rxyz, iR;
  fx::sqrt_both(Rxy2, &Rxy, &iRxy);
  fx::sqrt_both(R2, &Rxyz, &iR);
  const double dbeta = ((tvy - m.vy) * ddx - (tvx - m.vx) * ddy) * (iRxy * iRxy);
  const double deps = ((tvz - m.vz) * Rxy2 - ddz * (ddx * (tvx - m.vx) + ddy * (tvy - m.vy))) * (iR * iR * iRxy);
  const double K = fmax((double)P.K * (t_max - m.t) * (1.0 / t_max), 0.0);
  const double ny = m_clamp(-nyz_max, K * vm * (1.0 / g) * cth * dbeta, nyz_max);
  const double nz = m_clamp(-nyz_max, K * vm * (1.0 / g) * deps + cth, nyz_max);
  m.recede = (Rxyz > m.dprev) ? min(m.recede + 1, P.recede_max) : 0;   // (300 in a row decide; the count saturates there)
  m.dprev = Rxyz;
  if (Rxyz < (double)P.Rc && target_alive && m.status != MSL_MISS) {
    m.status = MSL_HIT;
  } else if (k >= P.k_timeout || vm < (double)P.v_min || m.recede >= P.recede_max || !target_alive) {
    m.status = MSL_MISS;
  } else {
    m.px += dt * m.vx; m.py += dt * m.vy; m.pz += dt * m.vz;
    // ---- drag and dv/dt (simulatior.py:578-590), fp32
    const float gf = P.g, fvm = (float)vm;
    const float alt = missile_height((float)m.px, (float)m.py, (float)m.pz, c);
    const float Tt = burning ? gf * P.Isp * P.dm : 0.0f;
    const float sd = sin_rate((float)m.dth), sp = sin_rate((float)m.dph);
    const float S = 3.14159265358979323846f * 0.25f * P.Diameter * P.Diameter + sqrtf(sd * sd + sp * sp) * P.Diameter * P.Length;
    const float rho = 1.225f * __expf(-alt * (1.0f / 9300.0f));
    const float D = 0.5f * P.cD * S * rho * fvm * fvm;
    const float nx = (Tt - D) / ((float)m.m * gf);
    double st, ct, sps, cps;   // of the CURRENT theta, psi
    if (k == 1) { fx::sincos(m.theta, &st, &ct); fx::sincos(m.psi, &sps, &cps); }
    else { st = m.vz * ivm; ct = cth; cps = m.vx * ih; sps = m.vy * ih; }
    const float dv = gf * (nx - (float)st);
    m.dph = g * ivm * (ny * fx::rcp(ct));
    m.dth = g * ivm * (nz - ct);
    const double v = vm + dt * (double)dv;
    const double dps = dt * m.dph, dts = dt * m.dth;
    m.psi += dps; m.theta += dts;
    double s2, c2, s3, c3;
    if (fabs(dps) > 0.05 || fabs(dts) > 0.05) { fx::sincos(m.theta, &s2, &c2); fx::sincos(m.psi, &s3, &c3); }   // (never in the guidance envelope)
    else {
      double sa, ca, sb, cb;
      small_sincos(dts, &sa, &ca); small_sincos(dps, &sb, &cb);
      s2 = st * ca + ct * sa; c2 = ct * ca - st * sa;
      s3 = sps * cb + cps * sb; c3 = cps * cb - sps * sb;
    }
    m.vx = v * c2 * c3; m.vy = v * c2 * s3; m.vz = v * s2;
    if (burning) m.m -= dt * (double)P.dm;
    return true;
  }
  return false;
}

// Stage the 7 KB table pack into LDS: every lane issues all of its 16-byte global loads before the first LDS store, so
// the workgroup pays one L2 round trip instead of one per loop iteration.
// The same copy in two halves, so that a kernel can put its state loads between them: the table loads are issued first, the
// state loads behind them, and the LDS writes wait (in-order vmcnt) for the table loads alone.
template <int THREADS>
struct TableCopy {
  static constexpr int NV = F16_PACK_LEN / 4;
  static constexpr int PER = (NV + THREADS - 1) / THREADS;
  float4 v[PER];
  __device__ __forceinline__ void issue(const float* __restrict__ g) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = threadIdx.x + k * THREADS;
      v[k] = (i < NV) ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void write(float* lds) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = threadIdx.x + k * THREADS;
      if (i < NV) reinterpret_cast<float4*>(lds)[i] = v[k];
    }
  }
  __device__ __forceinline__ void commit(float* lds) {
    write(lds);
    AC_CLKW(0, 120); AC_CLKW(1, 121); AC_CLKW(2, 122);     // (scratch builds: when each wave of workgroup 0 reaches the barrier)
    __syncthreads();
  }
};
template <int THREADS = 64>
__device__ __forceinline__ void stage_tables(float* lds, const float* __restrict__ g) {
  constexpr int NV = F16_PACK_LEN / 4;               // float4 count (the pack is padded to a multiple of 4)
  constexpr int PER = (NV + THREADS - 1) / THREADS;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = threadIdx.x + k * THREADS;
    v[k] = (i < NV) ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = threadIdx.x + k * THREADS;
    if (i < NV) reinterpret_cast<float4*>(lds)[i] = v[k];
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ step outputs
// The outputs of the 64 aircraft of a workgroup, written by ONE wave with every lane active. A lane's observation row is `ow`
// floats (60-260 B): storing it lane by lane gives 64 partial-line writes per field, which the L2 merges for HBM but which cross
// PCIe one by one when the destination is mapped host memory (ac_step_host) -- so the rows go through LDS and leave as contiguous
// 16-byte vectors (the block's 64 * ow floats are contiguous in [N][ow]); done flags leave as 16 dwords built from a ballot, the
// info row of an env as one 16-byte store. Output arrays are padded to whole workgroups, so the tail lanes of the last block
// (which shadow the last env) write their copy into the padding: no bounds logic here.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (the observation rows are already in LDS: lds[lane * ow + k], written by the lane itself)
__device__ __forceinline__ void emit_rows(const DevPtrs& P, float* lds /* [64 * ow], 16-byte aligned */, int ow, int lane,
                                          float reward, bool done, int A, int i0, int i1, int i2, int i3);
template <int NOB>
__device__ __forceinline__ void emit_outputs(const DevPtrs& P, float* lds /* [64 * ow], 16-byte aligned */, int ow, int lane, const float (&ob)[NOB],
                                             float reward, bool done, int A, int i0, int i1, int i2, int i3) {
  float* row = lds + lane * ow;
#pragma unroll
  for (int k = 0; k < NOB; ++k) if (k < ow) row[k] = ob[k];
  for (int k = NOB; k < ow; ++k) row[k] = 0.0f;             // reserved slots of the *_RWR variants
  emit_rows(P, lds, ow, lane, reward, done, A, i0, i1, i2, i3);
}
// The two halves of the output: the observation rows (the bulk, and ready first) ...
__device__ __forceinline__ void emit_obs_rows(const DevPtrs& P, float* lds, int ow, int lane) {
  wave_lds_fence();
  AC_CLK(56);
  const size_t blk = blockIdx.x;
  const int nvec = 16 * ow;                                 // float4 count of the block's rows
  const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
  for (int set = 1; set >= 0; --set) {                      // the copy that crosses PCIe first
    float* obs = set ? P.obs2 : P.obs;
    if (!obs) continue;
    float4* o4 = reinterpret_cast<float4*>(obs + blk * 64 * (size_t)ow);
    int i = lane;
    for (; i + 192 < nvec; i += 256) {     // four rows of the copy in flight at a time
      const float4 v0 = l4[i], v1 = l4[i + 64], v2 = l4[i + 128], v3 = l4[i + 192];
      o4[i] = v0; o4[i + 64] = v1; o4[i + 128] = v2; o4[i + 192] = v3;
    }
    for (; i < nvec; i += 64) o4[i] = l4[i];
  }
  AC_CLK(57);
  wave_lds_fence();
}
// ... and the per-aircraft reward and done flag with the env's info row
__device__ __forceinline__ void emit_scalars(const DevPtrs& P, int lane, float reward, bool done, int A, int i0, int i1, int i2, int i3) {
  const size_t blk = blockIdx.x;
  const unsigned long long dmask = __ballot(done);
  unsigned dword = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) dword |= (unsigned)((dmask >> (4 * (lane & 15) + b)) & 1ull) << (8 * b);
  const int4 inf = make_int4(i0, i1, i2, i3);
  // (current_step < 65536 is checked by ac_create; the turn count saturates at its field's 127 -- the reference's increment_size list has 15 stages)
  const int packed = (i0 & 0xFFFF) | ((i1 & 0xFF) << 16) | (min(i2, 127) << 24) | ((i3 & 1) << 31);
  const size_t n = blk * 64 + lane;
  if (!(fabsf(reward) < INFINITY)) {
    // One winner whatever the hardware's store order: a system-scope max on the page-locked word. An aircraft whose OWN probe fired
    // (poison_if's payload) outranks one whose reward only inherited the NaN; within a rank the lowest aircraft index wins.
    const int own = ((__float_as_int(reward) & 0x7fffffff) == 0x7fc00001) ? 1 : 0;
    __hip_atomic_fetch_max(P.err, (own << 30) | (0x3fffffff - (int)n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
#pragma unroll
  for (int set = 1; set >= 0; --set) {
    float* rew = set ? P.rew2 : P.rew; uint8_t* dn = set ? P.done2 : P.done; int* info = set ? P.info2 : P.info;
    if (!rew) continue;
    rew[n] = reward;
    if (lane < 16) reinterpret_cast<unsigned*>(dn + blk * 64)[lane] = dword;
    // the device copy keeps the four-word row; the host copy of ac_step_host is one packed word per env (AC_INFO_* in aircombat.h):
    // a quarter of the bytes across PCIe, and the lanes' words are adjacent
    if (set) {   // (env e of the block sits in lane e * A: fetched into lane e, so that the words leave as one run of adjacent lanes)
      const int w = __shfl(packed, (lane * A) & 63);
      if (lane < 64 / A) info[blk * (64 / A) + lane] = w;
    } else if (lane % A == 0) *reinterpret_cast<int4*>(info + (n / A) * 4) = inf;
  }
}
__device__ __forceinline__ void emit_rows(const DevPtrs& P, float* lds, int ow, int lane, float reward, bool done, int A, int i0, int i1, int i2, int i3) {
  emit_obs_rows(P, lds, ow, lane);
  emit_scalars(P, lane, reward, done, A, i0, i1, i2, i3);
}
// the four control indices of an aircraft's action row: one 16-byte load where the row width allows it (a row read from mapped
// host memory then crosses PCIe as one request per 4 lanes instead of four)
__device__ __forceinline__ float4 load_controls(const float* act, int act_dim) {
  if ((act_dim & 3) == 0) return *reinterpret_cast<const float4*>(act);
  return make_float4(act[0], act[1], act[2], act[3]);
}

// ------------------------------------------------------------------------------------------------ 1v1 observation
// singlecombat_task.py:88-139 (15 values, 2-D AO/TA, clipped to +-10) and
// singlecombat_with_missile_task.py:31-99 (21 values, 3-D AO/TA, unclipped, missile-warning block).
struct Enemy { float n, e, u, vn, ve, vd, ub, alt; };
__device__ __forceinline__ Enemy exchange_1v1(const Props& pr) {
  Enemy E;
  E.n = __shfl_xor(pr.n, 1); E.e = __shfl_xor(pr.e, 1); E.u = __shfl_xor(pr.u, 1);
  E.vn = __shfl_xor(pr.vn, 1); E.ve = __shfl_xor(pr.ve, 1); E.vd = __shfl_xor(pr.vd, 1);
  E.ub = __shfl_xor(pr.ub, 1); E.alt = __shfl_xor(pr.alt_m, 1);
  return E;
}
struct Incoming { bool any; float px, py, pz, vx, vy, vz; };
template <int TASK>
__device__ __forceinline__ void observe_1v1(const Props& pr, const Enemy& E, const Incoming& in, float* ob) {
  ob[0] = pr.alt_m / 5000.0f;
  ob[1] = pr.sphi; ob[2] = pr.cphi; ob[3] = pr.stht; ob[4] = pr.ctht;
  ob[5] = pr.ub / 340.0f; ob[6] = pr.vb / 340.0f; ob[7] = pr.wb / 340.0f; ob[8] = pr.vc / 340.0f;
  Geo g = (TASK == AC_TASK_SINGLECOMBAT)
              ? ao_ta_r<true>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd)
              : ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
  ob[9] = (E.ub - pr.ub) / 340.0f;
  ob[10] = (E.alt - pr.alt_m) / 1000.0f;
  ob[11] = g.AO; ob[12] = g.TA; ob[13] = g.R / 10000.0f; ob[14] = g.side;
  if (TASK == AC_TASK_SINGLECOMBAT) {
#pragma unroll
    for (int k = 0; k < 15; ++k) ob[k] = clampf(-10.0f, ob[k], 10.0f);
  } else {
#pragma unroll
    for (int k = 15; k < 21; ++k) ob[k] = 0.0f;
    if (in.any) {
      Geo gm = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, in.px, in.py, in.pz, in.vx, in.vy, in.vz);
      ob[15] = (sqrtf(in.vx * in.vx + in.vy * in.vy + in.vz * in.vz) - pr.ub) / 340.0f;
      ob[16] = (in.pz - pr.alt_m) / 1000.0f;
      ob[17] = gm.AO; ob[18] = gm.TA; ob[19] = gm.R / 10000.0f; ob[20] = gm.side;
    }
  }
}
// PostureReward / AltitudeReward raw values for this lane (posture_reward.py:26-49, altitude_reward.py:20-40)
__device__ __forceinline__ float posture_raw(const Props& pr, const Enemy& E) {
  Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
  return posture_fn(g.AO, g.TA, g.R * 0.001f);
}
__device__ __forceinline__ float altitude_raw(const Props& pr, const DevCfg& c) {
  return altitude_fn(pr.u * 0.001f, pr.vd / 340.0f, c);
}

template <int TASK>
struct TaskTraits {
  static constexpr bool HAS_MSL = (TASK == AC_TASK_SHOOT_MISSILE || TASK == AC_TASK_DODGE_MISSILE);
  static constexpr int MSLOTS = HAS_MSL ? AC_MAX_MISSILES_PER_AGENT : 1;
  static constexpr int OBS = (TASK == AC_TASK_SINGLECOMBAT) ? 15 : 21;
};

// ------------------------------------------------------------------------------------------------ the step kernel
// One lane per aircraft, the two aircraft of a 1v1 env in lanes (2k, 2k+1).
// Occupancy: WPE = waves per SIMD the register allocation is bounded for. WPE = 2 (256 VGPRs, 12 dwords of scratch) lets two
// waves share a SIMD and fill its 2-cycle issue rate: +38 % throughput once there are more waves than SIMDs. WPE = 1 keeps
// everything in registers (269 incl. accumulation VGPRs): 5 % less latency when each SIMD has at most one wave (E*A <= 65536).
#include "split_kernel.hpp"
#include "pair_kernel.hpp"

// FORM: 0 = one wave per 64 aircraft, 1 = the three-wave form (split_kernel.hpp) for the task without munitions at small grids,
// 2 = the pair form (pair_kernel.hpp: flight wave + environment wave) for the tasks with missiles.
template <int TASK, int WPE, int FORM = 0>
__global__ __launch_bounds__(FORM == 1 ? 192 : (FORM == 2 ? 128 : (FORM == 3 ? 256 : 64)), WPE) void step_kernel_1v1(DevPtrs P, DevCfg c) {
  // forms launch_step reaches: SingleCombat in the one-wave (0) and three-wave (1) forms; the two tasks with munitions in the pair (2) and quad (3) forms only
  static_assert(TASK == AC_TASK_SINGLECOMBAT ? FORM <= 1 : FORM >= 2, "no launch path for this (task, form)");
  using TT = TaskTraits<TASK>;
  constexpr bool SPLIT = FORM == 1, QUAD = FORM == 3, PAIR = FORM == 2 || QUAD;   // (the quad form's environment wave runs the pair form's code)
  constexpr bool HAS_MSL = TT::HAS_MSL;
  constexpr int MSLOTS = TT::MSLOTS;
  constexpr int OBS = TT::OBS;
  AC_CLK(0);
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  __shared__ __attribute__((aligned(16))) float lds_out[64 * OBS];
  __shared__ __attribute__((aligned(16))) char split_lds[SPLIT ? sizeof(SplitLds) : (QUAD ? sizeof(QuadLds) : (PAIR ? sizeof(PairLds) : 16))];
  SplitLds& L = *reinterpret_cast<SplitLds*>(split_lds);
  QuadLds& LQ = *reinterpret_cast<QuadLds*>(split_lds);
  PairLds& LP = QUAD ? LQ.P : *reinterpret_cast<PairLds*>(split_lds);
  const Tab T{lds_tab};
  const int N = c.N;
  const int l = threadIdx.x & 63;
  const int n = blockIdx.x * 64 + l;
  const bool live = n < N;           // N is even, so both lanes of a pair are live or not together
  const int nn = live ? n : (N - 2 + (n & 1));  // tail lanes shadow the last env and never store
  const int slot = nn & 1;

  State s; Task t; Derived d; Props pr;
  int pre_st[MSLOTS];
#pragma unroll
  for (int k = 0; k < MSLOTS; ++k) pre_st[k] = MSL_INACTIVE;
  // One HBM round trip for everything a wave can ask for up front: the table loads are issued first, the action row and the state
  // loads behind them, and the LDS copy of the tables waits (in-order vmcnt) for the table loads alone.
  const float* act = P.actions + (size_t)nn * c.act_dim;
  float4 a4;
  ActionFetch arow;                      // three-wave / quad form: the systems wave's action row (asked for late, waited for by hand)
  float shoot_raw = 0.0f;
  if (SPLIT) {
    TableCopy<192> tc;
    tc.issue(P.tab);
    AC_CLKW(0, 123);
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s = State{}; t = Task{};
    if (role == 0) { load_flight_role<0>(P.F, P.I, P.D, N, nn, s); AC_CLKW(0, 124); load_task(P.F, P.I, N, nn, t); AC_CLKW(0, 125); }
    else if (role == 1) load_flight_role<1>(P.F, P.I, P.D, N, nn, s);
    else { load_flight_role<2>(P.F, P.I, P.D, N, nn, s); t.status = state_word(P.F, SW_status, N, nn); }
    // The action row may live in mapped host memory (ac_step_host): a read across PCIe takes ~5 k cycles, and the CU's vector memory
    // path returns loads in the order they were issued ACROSS its waves -- a state load issued behind it, by any wave of the
    // workgroup, waits those 5 k cycles too (measured: the dynamics wave, which asks for the most, reached the table barrier at 8-10 k
    // cycles instead of 3 k; profiles/round4_cycle_stamps.txt). So: a bare barrier (no wait for memory) once every wave has ISSUED
    // its state loads; then the tables go to LDS (the compiler's waits are for its own loads, all older than the row), and only then
    // the one wave that decodes the commands -- the systems wave, after B1 of the first tick -- asks for the row, with a load whose
    // wait is placed by hand (ActionRow).
    __builtin_amdgcn_s_barrier();
    tc.write(lds_tab);
    a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // every load the compiler knows of has landed -- on every path, so that no compiler-placed `s_waitcnt vmcnt(0)` for a state field is
    // left anywhere behind this point (the counter counts all loads: it would wait for the row as well) -- before the one it does not
    // know of is issued
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wave's own state, one round trip like the tables
    if (role == 1) arow.issue(act, c.act_dim);
    AC_CLKW(0, 120); AC_CLKW(1, 121); AC_CLKW(2, 122);
    __syncthreads();                      // (LDS visibility; loads in flight are not waited for)
  } else if (QUAD) {
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 dynamics, 1 systems, 2 kinematics, 3 environment
    TableCopy<256> tc;
    tc.issue(P.tab);
    s = State{}; t = Task{};
    if (role == 0) { load_flight_role<0>(P.F, P.I, P.D, N, nn, s); t.status = state_word(P.F, SW_status, N, nn); }
    else if (role == 1) load_flight_role<1>(P.F, P.I, P.D, N, nn, s);
    else if (role == 2) { load_flight_role<2>(P.F, P.I, P.D, N, nn, s); t.status = state_word(P.F, SW_status, N, nn); }
    else {   // the environment wave owns the task bookkeeping; the status word of every missile slot and the tick count come with it
      load_task(P.F, P.I, N, nn, t);
      s.ticks = state_word(P.F, SW_ticks, N, nn);
#pragma unroll
      for (int k = 0; k < MSLOTS; ++k) pre_st[k] = P.MI[((size_t)k * NMI + MI_status) * (size_t)N + nn];
    }
    // The action row like in the three-wave form above (it may cross PCIe, and loads return in issue order across the CU's waves): a bare
    // barrier once every wave has issued its state loads, the tables, every known load landed; then the systems wave alone asks for the
    // row (control indices decoded after B1 of the first tick; the shoot bit goes to the environment wave with the systems wave's fields).
    __builtin_amdgcn_s_barrier();
    tc.write(lds_tab);
    a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (role == 1) arow.issue(act, c.act_dim);
    __syncthreads();
    if (role == 0) { quad_dynamics_wave(P, c, T, LQ, l, n, live, s, t); return; }
    if (role != 3 && split_helper_wave<true>(s, t, T, LQ.S, l, c.substeps, nullptr, nullptr, &arow)) return;
  } else if (PAIR) {
    const bool flight_role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 1;
    PairFlightIn fin;
    if (!flight_role) {   // the environment wave owns the task bookkeeping; the status word of every missile slot comes with it
      a4 = make_float4(0.f, 0.f, 0.f, 0.f);              // (the control indices are the flight wave's; of the row this wave needs the shoot bit, late)
      s = State{}; load_task(P.F, P.I, N, nn, t);
#pragma unroll
      for (int k = 0; k < MSLOTS; ++k) pre_st[k] = P.MI[((size_t)k * NMI + MI_status) * (size_t)N + nn];
    }
    // (the flight wave asks for its state behind the table copy: issuing it in front -- one round trip for both -- measured 0.3 us
    // slower per step here, the environment wave's longer prologue covers the second round trip anyway)
    stage_tables<128>(lds_tab, P.tab);
    if (flight_role) {
      pair_flight_load(P, c, nn, fin);
      pair_flight_wave<true>(P, c, T, LP, l, n, live, fin);
      return;
    }

  } else {
    TableCopy<64> tc;
    tc.issue(P.tab);
    a4 = load_controls(act, c.act_dim);
    if (TASK == AC_TASK_SHOOT_MISSILE) shoot_raw = act[4];
    load_state(P.F, P.I, P.D, N, nn, s, t);
    tc.commit(lds_tab);
  }
  AC_CLK(1);
  const Task was = t;   // the task record as loaded: groups 16-18 are written back only where they changed (store_task_changed)
  // Three-wave form: after the last tick the kinematics wave stays and builds the observation rows from the pose the dynamics wave
  // posts (two more barriers between those two waves), while the dynamics wave runs terminations, rewards and the state stores.
  constexpr bool OBS_BY_KIN = SPLIT && TASK == AC_TASK_SINGLECOMBAT;
  if (SPLIT) {   // helper waves: run their part of every substep (the systems wave decodes the commands it integrates)
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (split_helper_wave(s, t, T, L, l, c.substeps, nullptr, nullptr, &arow)) {
      if (OBS_BY_KIN && role == 1) {                 // the systems wave, idle after its last tick: the two geometry rewards of the final pose
        using namespace mail;
        wg_sync();                                   // the dynamics wave has posted the final pose
        Props q;
        q.alt_m = L.M[T_PR][l]; q.ub = L.M[T_PR + 5][l];
        q.n = L.M[T_PR + 9][l]; q.e = L.M[T_PR + 10][l]; q.u = L.M[T_PR + 11][l];
        q.vn = L.M[T_PR + 12][l]; q.ve = L.M[T_PR + 13][l]; q.vd = L.M[T_PR + 14][l];
        const Enemy Eq = exchange_1v1(q);
        L.M[T_RALT][l] = altitude_raw(q, c);
        L.M[T_RPOS][l] = posture_raw(q, Eq);
        wg_sync();                                   // read by the dynamics wave behind this barrier (its terminations ran meanwhile)
      }
      if (OBS_BY_KIN && role == 2) {
        using namespace mail;
        wg_sync();                                   // the dynamics wave has posted the final pose
        Props q;
        q.alt_m = L.M[T_PR][l]; q.sphi = L.M[T_PR + 1][l]; q.cphi = L.M[T_PR + 2][l]; q.stht = L.M[T_PR + 3][l]; q.ctht = L.M[T_PR + 4][l];
        q.ub = L.M[T_PR + 5][l]; q.vb = L.M[T_PR + 6][l]; q.wb = L.M[T_PR + 7][l]; q.vc = L.M[T_PR + 8][l];
        q.n = L.M[T_PR + 9][l]; q.e = L.M[T_PR + 10][l]; q.u = L.M[T_PR + 11][l];
        q.vn = L.M[T_PR + 12][l]; q.ve = L.M[T_PR + 13][l]; q.vd = L.M[T_PR + 14][l];
        const Enemy Eq = exchange_1v1(q);
        const Incoming none{false, 0, 0, 0, 0, 0, 0};
        float obq[OBS];
        observe_1v1<TASK>(q, Eq, none, obq);
        wg_sync();                                   // ... and whether the env ends its episode: then the template's observation goes out
        if (L.M[T_DONE][l] != 0.0f) {
          const float* tobs = (const float*)(P.tF + (size_t)NSW * 2) + slot * OBS;
#pragma unroll
          for (int k = 0; k < OBS; ++k) obq[k] = tobs[k];
        }
        float* row = lds_out + l * OBS;
#pragma unroll
        for (int k = 0; k < OBS; ++k) row[k] = obq[k];
        emit_obs_rows(P, lds_out, OBS, l);
      }
      return;
    }
  }
  Msl ms[MSLOTS];
  int nslots = 0, msl_was_active = 0;
  if (HAS_MSL) {
    nslots = min(c.num_missiles[slot], MSLOTS);
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) {
      // a slot that has not been launched since the last reset holds zeros and MSL_INACTIVE (reset_all_kernel, the reset branch
      // below): only its status is read, and it is written back only once it has been launched or reset
      const int st = (k < nslots) ? (PAIR ? pre_st[k] : P.MI[((size_t)k * NMI + MI_status) * (size_t)N + nn]) : MSL_INACTIVE;
      if (st != MSL_INACTIVE) { load_msl(P.MF, P.MI, N, nn, k, ms[k]); msl_was_active |= 1 << k; }
      else { ms[k] = Msl{}; ms[k].status = MSL_INACTIVE; }
    }
  }

  // ---- apply actions (normalize_action, singlecombat_task.py:141-153; property bounds catalog.py:189-197)
  t.cur_step += 1;
  if (!SPLIT) {
    s.da = clampf(-1.0f, a4.x / 20.0f - 1.0f, 1.0f);
    s.de = clampf(-1.0f, a4.y / 20.0f - 1.0f, 1.0f);
    s.dr = clampf(-1.0f, a4.z / 20.0f - 1.0f, 1.0f);
    s.thr = clampf(0.0f, a4.w / 58.0f + 0.4f, 0.9f);
  }
  if (TASK == AC_TASK_SHOOT_MISSILE && !PAIR) t.shoot_action = (shoot_raw != 0.0f) ? 1 : 0;  // singlecombat_with_missile_task.py:182-184 (pair / quad forms: below, when the bit is first needed)

  // ---- substeps (env_base.py:139-154): every aircraft, then every missile against this substep's aircraft poses
  const MslParam MP = aim9l();
  bool have_pose = false;
  int last_tick = -1;   // three-wave form: the last substep this aircraft flew
  // pair form: with no missile entry in the env there is nothing to fly between the ticks, and only the last substep's pose is needed
  bool env_has_missiles = false;
  if (PAIR) {
    bool mine = false;
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) mine = mine || ms[k].status != MSL_INACTIVE;
    const int other = __shfl_xor((int)mine, 1);   // (fetched first: a shuffle on the right of || is skipped by the lanes that short-circuit)
    env_has_missiles = mine || (bool)other;
  }
  int quad_nrun = 0;
  const int quad_ticks0 = s.ticks;   // (quad form: the environment wave loaded the tick count for the Earth angle of the poses)
  for (int sub = 0; sub < c.substeps; ++sub) {
    bool fly_munitions = HAS_MSL;
    if (QUAD) {
      quad_substep_begin<true>(t, LQ, l, sub, env_has_missiles, quad_ticks0, quad_nrun, pr, c);   // (B1 and B2 inside)
      fly_munitions = env_has_missiles;
    } else if (PAIR) {
      AC_CLK(2 + 8 * sub);
      pair_substep<true>(t, LP, l, sub, env_has_missiles, pr, c);
      AC_CLK(4 + 8 * sub);
      if (!env_has_missiles) continue;
    } else if (SPLIT) {        // the FDM tick over three waves; what follows in the substep (munitions) stays on this wave
      if (dynamics_wave_tick(s, t, d, T, L, l, sub)) { have_pose = true; last_tick = sub; }
    } else if (t.status == AC_ALIVE) {
      if (t.bloods <= 0.0f) t.status = AC_SHOTDOWN;  // simulatior.py:220-222: this tick still integrates
      f16::tick<false>(s, d, T);
      have_pose = true;
    }
    if (HAS_MSL && fly_munitions) {
      if (!PAIR) {
      f16::locate(s, d);                                       // fp64 geodetic reduction for the NEU pose of this substep
      if (!have_pose) { f16::body_frame(s, d); have_pose = true; }  // frozen pose of a dead aircraft
      make_props(s, d, c, pr);
      }
      float tx = __shfl_xor(pr.n, 1), ty = __shfl_xor(pr.e, 1), tz = __shfl_xor(pr.u, 1);
      float tvx = __shfl_xor(pr.vn, 1), tvy = __shfl_xor(pr.ve, 1), tvz = __shfl_xor(pr.vd, 1);
      bool talive = __shfl_xor(t.status, 1) == AC_ALIVE;
      bool hit_now = false;
#pragma unroll
      for (int k = 0; k < MSLOTS; ++k) {
        if (k < nslots && ms[k].status != MSL_INACTIVE) {  // run() is called on finished missiles too (env_base.py:142-143)
          missile_run(ms[k], MP, tx, ty, tz, tvx, tvy, tvz, talive, c);
          if (ms[k].status == MSL_HIT && talive) { hit_now = true; talive = false; }  // target.shotdown() (:527)
        }
      }
      if (__shfl_xor((int)hit_now, 1) && t.status == AC_ALIVE) t.status = AC_SHOTDOWN;
      AC_CLK(5 + 8 * sub);
    }
    if (QUAD) wg_sync();                                       // B3 of the tick
  }
  const bool split_located = SPLIT && dynamics_wave_finish(s, d, L, l, last_tick, c.substeps);   // (+ the helper waves' fields)
  // (three-wave form: the decoded commands of the stored state came with the systems wave's fields -- the dynamics wave itself never reads
  //  the action row)
  if (PAIR) {
    if (QUAD) wg_sync();                           // (quad form: the helper waves hand their fields to the dynamics wave here)
    wg_sync();                                     // the flight wave has posted its final values and stored the flight state
    pair_read_final(LP, l, s, d, pr);
  } else if (!HAS_MSL || c.substeps == 0) {
    if (!split_located) f16::locate(s, d);
    if (!have_pose) f16::body_frame(s, d);
    make_props(s, d, c, pr);
  }
  AC_CLK(52);
  if (OBS_BY_KIN) {
    using namespace mail;
    L.M[T_PR][l] = pr.alt_m; L.M[T_PR + 1][l] = pr.sphi; L.M[T_PR + 2][l] = pr.cphi; L.M[T_PR + 3][l] = pr.stht; L.M[T_PR + 4][l] = pr.ctht;
    L.M[T_PR + 5][l] = pr.ub; L.M[T_PR + 6][l] = pr.vb; L.M[T_PR + 7][l] = pr.wb; L.M[T_PR + 8][l] = pr.vc;
    L.M[T_PR + 9][l] = pr.n; L.M[T_PR + 10][l] = pr.e; L.M[T_PR + 11][l] = pr.u;
    L.M[T_PR + 12][l] = pr.vn; L.M[T_PR + 13][l] = pr.ve; L.M[T_PR + 14][l] = pr.vd;
    wg_sync();                                       // the kinematics wave builds the observation from here
  }
  Enemy E = exchange_1v1(pr);

  // ---- task.step
  if (c.use_artillery) {  // singlecombat_task.py:162-188: every shooter drains the blood of each ALIVE enemy
    Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
    bool ealive = __shfl_xor(t.status, 1) == AC_ALIVE;
    float of = (g.AO <= 0.5236f) ? 1.0f - g.AO / 0.5236f : 0.0f;
    float Rk = g.R * 0.001f;
    float df = (Rk <= 1.0f) ? 1.0f : ((Rk <= 3.0f) ? (3.0f - Rk) * 0.5f : 0.0f);
    float dmg = ealive ? of * df : 0.0f;
    t.bloods -= __shfl_xor(dmg, 1);
  }
  if (HAS_MSL) {
    bool launch;
    // singlecombat_with_missile_task.py:182-184 (pair / quad form: the wave that flies read the action row and posted the shoot bit with its final values)
    if (TASK == AC_TASK_SHOOT_MISSILE && PAIR) t.shoot_action = ((QUAD ? LQ.S.M[mail::F_BITS][l] : LP.FIN[pair::FIN_BITS][l]) != 0.0f) ? 1 : 0;
    if (TASK == AC_TASK_DODGE_MISSILE) {
      // singlecombat_with_missile_task.py:108-124: rule-based launch — the enemy within max_attack_angle of the velocity vector
      // for a full lock window (1 s of env steps), inside max_attack_distance, min_attack_interval steps after the last shot.
      // The window is updated by every aircraft, dead ones included (their cached pose keeps being read).
      const float dx = E.n - pr.n, dy = E.e - pr.e, dz = E.u - pr.u;
      const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
      const float sp = sqrtf(pr.vn * pr.vn + pr.ve * pr.ve + pr.vd * pr.vd);
      const float ang = 57.29577951f * acosf(clampf(-1.0f, (dx * pr.vn + dy * pr.ve + dz * pr.vd) / (dist * sp + 1e-8f), 1.0f));
      const int len = c.lock_len;                       // deque(maxlen = int(1 / time_interval))
      const unsigned bit = 1u << (unsigned)(t.lock_pos % len);
      t.lock_bits = (ang < c.max_attack_angle) ? (t.lock_bits | (int)bit) : (t.lock_bits & ~(int)bit);
      t.lock_pos += 1;
      const bool locked = __popc((unsigned)t.lock_bits & ((1u << len) - 1u)) >= len;   // np.sum(lock_duration) >= maxlen
      launch = t.status == AC_ALIVE && locked && dist <= c.max_attack_distance && t.remaining > 0 &&
               (t.cur_step - t.last_shoot_time) >= c.min_attack_interval;
      if (launch) t.last_shoot_time = t.cur_step;
    } else {  // :194-204: learned shoot bit, previous missile must be done
      bool prev_done = t.last_missile < 0;
#pragma unroll
      for (int k = 0; k < MSLOTS; ++k)
        if (k == t.last_missile) prev_done = (ms[k].status == MSL_HIT || ms[k].status == MSL_MISS);
      launch = t.status == AC_ALIVE && t.shoot_action && t.remaining > 0 && prev_done;
    }
    int k = nslots - t.remaining;  // slots are consumed in launch order
    if (launch && k >= 0 && k < nslots) {
      // MissileSimulator.launch (simulatior.py:497-514): parent's cached NEU position, (vN, vE, vDOWN), pitch, yaw
      float tht = asinf(pr.stht);
      float psi = atan2f(pr.m12, pr.m11);
      if (psi < 0.0f) psi += 2.0f * f16::kPi;
#pragma unroll
      for (int q = 0; q < MSLOTS; ++q)
        if (q == k) {
          ms[q].px = pr.n; ms[q].py = pr.e; ms[q].pz = pr.u; ms[q].vx = pr.vn; ms[q].vy = pr.ve; ms[q].vz = pr.vd;
          ms[q].theta = tht; ms[q].psi = psi; ms[q].t = 0.0f; ms[q].m = MP.m0; ms[q].dth = 0.0f; ms[q].dph = 0.0f;
          ms[q].dprev = INFINITY; ms[q].recede = 0; ms[q].status = MSL_LAUNCHED; ms[q].order = t.cur_step;
        }
      t.last_missile = k;
      t.remaining -= 1;
    }
  }

  // ---- my first alive incoming missile (check_missile_warning, simulatior.py:321-325) is the enemy's live one
  Incoming inc{false, 0, 0, 0, 0, 0, 0};
  int my_hits = 0;
  int inc_slot = -1;   // which of the enemy's missile slots that is
  if (HAS_MSL) {
    int best = 0x7fffffff, mine = -1;
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) {
      if (k < nslots && ms[k].status == MSL_LAUNCHED && ms[k].order < best) { best = ms[k].order; mine = k; }
      if (k < nslots && ms[k].status == MSL_HIT) my_hits += 1;
    }
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k)
      if (k == mine) { a0 = ms[k].px; a1 = ms[k].py; a2 = ms[k].pz; a3 = ms[k].vx; a4 = ms[k].vy; a5 = ms[k].vz; }
    inc.any = __shfl_xor((int)(mine >= 0), 1);
    inc_slot = __shfl_xor(mine, 1);
    inc.px = __shfl_xor(a0, 1); inc.py = __shfl_xor(a1, 1); inc.pz = __shfl_xor(a2, 1);
    inc.vx = __shfl_xor(a3, 1); inc.vy = __shfl_xor(a4, 1); inc.vz = __shfl_xor(a5, 1);
  }
  float ob[OBS];
  if (!OBS_BY_KIN) observe_1v1<TASK>(pr, E, inc, ob);
  AC_CLK(58);

  // ---- terminations (singlecombat_task.py:34-40; first condition that fires wins, task_base.py:88-112).
  // Agents are evaluated in order (env_base.py:159-166): agent 0's SafeReturn sees agent 1's status from before agent 1's
  // own checks, agent 1 sees agent 0's status after them.
  const int status_pre = t.status;
  int code = AC_DONE_NONE;
  bool done = false;
  bool nonfinite = false;
  {
    float np_max = fmaxf(fabsf(s.npx), fmaxf(fabsf(s.npy), fabsf(s.npz)));
    float pqr = sqrtf(d.p * d.p + d.q * d.q + d.r * d.r);
    nonfinite = nonfinite_probe(d.veci, pqr, d.h_sl_ft, np_max);
    bool extreme = nonfinite || (d.veci >= 1e10f) || (pqr >= 1000.0f) || (d.h_sl_ft >= 1e10f) || (np_max > 10.0f);  // catalog.py:386-416
    bool overload = (s.ticks >= kTickOverload) &&
                    (fabsf(s.npx) > c.acc_x || fabsf(s.npy) > c.acc_y || fabsf(s.npz + 1.0f) > c.acc_z);  // overload.py:38-46
    if (pr.alt_m <= c.altitude_limit) { t.status = AC_CRASH; code = AC_DONE_LOW_ALTITUDE; done = true; }
    else if (extreme) { t.status = AC_CRASH; code = AC_DONE_EXTREME_STATE; done = true; }
    else if (overload) { t.status = AC_CRASH; code = AC_DONE_OVERLOAD; done = true; }
  }
  {
    int other_pre = __shfl_xor(status_pre, 1), other_post = __shfl_xor(t.status, 1);
    int enemy_status = (slot == 0) ? other_pre : other_post;
    if (!done) {  // safe_return.py:15-50, timeout.py:14-32
      if (t.status == AC_SHOTDOWN) { code = AC_DONE_SHOTDOWN; done = true; }
      else if (t.status == AC_CRASH) { code = AC_DONE_CRASHED; done = true; }
      else if (enemy_status != AC_ALIVE && !inc.any) { code = AC_DONE_MISSION_COMPLETE; done = true; }
      else if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; }
    }
  }

  // ---- episode end: every agent done => the env is reset and its observation replaced (env_wrappers.py:191-204)
  const int other_done = __shfl_xor((int)done, 1);   // fetched outside the && (a short-circuited shuffle would read a masked-off lane)
  const bool all_done = done && (bool)other_done;
  const int other_code = __shfl_xor(code, 1);
  if (OBS_BY_KIN) {
    L.M[mail::T_DONE][l] = all_done ? 1.0f : 0.0f;
    wg_sync();                                       // the kinematics wave sends the rows (the template's if the episode ends)
  } else if (all_done) {
    const float* tobs = (const float*)(P.tF + (size_t)NSW * 2) + slot * OBS;  // template observation follows the template fields
#pragma unroll
    for (int k = 0; k < OBS; ++k) ob[k] = tobs[k];
  }

  // ---- rewards (after every termination ran, env_base.py:168-171; die-flag latch singlecombat_task.py:190-195)
  AC_CLK(59);
  float reward = 0.0f;
  const bool evaluates = !t.die_flag;
  if (evaluates) {
    t.die_flag = (t.status != AC_ALIVE) ? 1 : 0;
    // (three-wave SingleCombat: the systems wave evaluated the two geometry terms from the posted pose while this wave ran the terminations)
    float r_alt = potential(OBS_BY_KIN ? L.M[mail::T_RALT][l] : altitude_raw(pr, c), c.altitude_scale, c.altitude_pot, t.pre_altitude);
    float r_pos = potential(OBS_BY_KIN ? L.M[mail::T_RPOS][l] : posture_raw(pr, E), c.posture_scale, c.posture_pot, t.pre_posture);
    float ev = ((t.status != AC_ALIVE) ? -200.0f : 0.0f) + 200.0f * (float)my_hits;  // event_driven_reward.py:15-34
    float r_ev = potential(ev, c.event_scale, c.event_pot, t.pre_event);
    reward = r_alt + r_pos + r_ev;
    if (TASK == AC_TASK_SHOOT_MISSILE) {  // shoot_penalty_reward.py:13-32
      float sp = (t.remaining == t.pre_remaining - 1) ? -30.0f : 0.0f;
      t.pre_remaining = t.remaining;
      reward += potential(sp, c.shoot_penalty_scale, c.shoot_pot, t.pre_shoot);
    }
  }
  if (TASK == AC_TASK_DODGE_MISSILE) {
    // MissilePostureReward (missile_posture_reward.py:18-46): ONE `previous_missile_v` for the whole env, an alias of the live
    // velocity array of the first missile it saw, cleared whenever an evaluating agent has no incoming missile; agents are
    // walked in env order. The remembered missile id (1 + launcher * MSLOTS + slot, 0 = none) lives in `shoot_action`, which this
    // task does not otherwise use, identically in both lanes of the env.
    float spd[MSLOTS];
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) spd[k] = sqrtf(ms[k].vx * ms[k].vx + ms[k].vy * ms[k].vy + ms[k].vz * ms[k].vz);
    const int base = (threadIdx.x & 63) - slot;
    const int my_inc_id = inc.any ? 1 + (slot ^ 1) * MSLOTS + inc_slot : 0;
    int prev = t.shoot_action;
    float r_mp = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ev_i = __shfl((int)evaluates, base + i), id_i = __shfl(my_inc_id, base + i);
      if (!ev_i) continue;
      if (id_i) {
        if (!prev) prev = id_i;
        float v_prev = 0.0f, v_cur = 0.0f;
#pragma unroll
        for (int k = 0; k < MSLOTS; ++k) {
          float a = __shfl(spd[k], base + (prev - 1) / MSLOTS), b = __shfl(spd[k], base + (id_i - 1) / MSLOTS);
          if (k == (prev - 1) % MSLOTS) v_prev = a;
          if (k == (id_i - 1) % MSLOTS) v_cur = b;
        }
        if (slot == i) {
          const float v_dec = (v_prev - v_cur) / 340.0f * c.missile_posture_scale;
          const float va = sqrtf(pr.vn * pr.vn + pr.ve * pr.ve + pr.vd * pr.vd);
          const float ang = (inc.vx * pr.vn + inc.vy * pr.ve + inc.vz * pr.vd) / (v_cur * va);
          r_mp = (ang < 0.0f) ? ang / (fmaxf(v_dec, 0.0f) + 1.0f) : ang * fmaxf(v_dec, 0.0f);
        }
      } else prev = 0;
    }
    t.shoot_action = prev;
    reward += r_mp;
  }

  // ---- the reset itself
  int step_out = t.cur_step;
  if (all_done) {
    load_state(P.tF, P.tI, P.tD, 2, slot, s, t);
    zero_controller_state(P, N, n, live);
#pragma unroll
    for (int k = 0; k < MSLOTS; ++k) { ms[k] = Msl{}; ms[k].status = MSL_INACTIVE; }
  }
  AC_CLK(53);
  if (live) {
    if (!PAIR) { store_flight(P.F, P.I, P.D, N, n, s); store_task_changed(P.F, P.I, N, n, t, was); }
    else {   // the flight wave has stored the flown state (ordered before this by the barrier): an episode reset overwrites it
      if (all_done) store_flight(P.F, P.I, P.D, N, n, s);
      store_task_changed(P.F, P.I, N, n, t, was);
    }
    if (HAS_MSL) {
#pragma unroll
      for (int k = 0; k < MSLOTS; ++k)
        if (k < nslots && (ms[k].status != MSL_INACTIVE || ((msl_was_active >> k) & 1))) store_msl(P.MF, P.MI, N, n, k, ms[k]);
    }
  }
  AC_CLK(55);
  // info['done_condition'] keeps the last agent's message
  // (sending the observation rows ahead of the rewards and the state stores was tried for the host-boundary path, where their trip
  // across PCIe is what the step waits for in the end: 0.5 us slower, not faster)
  reward = poison_if(nonfinite, reward);
  if (OBS_BY_KIN) emit_scalars(P, l, reward, done, 2, step_out, other_code ? other_code : code, 0, all_done ? 1 : 0);
  else emit_outputs(P, lds_out, OBS, l, ob, reward, done, 2, step_out, other_code ? other_code : code, 0, all_done ? 1 : 0);
  AC_CLK(54);
}

// ------------------------------------------------------------------------------------------------ NvN (MultipleCombat)
// MultipleCombatEnv.step (envs/JSBSim/envs/multiplecombat_env.py:119-182) with MultipleCombatTask
// (envs/JSBSim/tasks/multiplecombat_task.py:15-151): the A aircraft of an env sit in A adjacent lanes (ego team first),
// every pairwise quantity is fetched from the owning lane with __shfl. Differences from the 1v1 family reproduced here:
// rewards are computed BEFORE terminations and only while alive, each agent receives its team's mean reward, the
// termination order is SafeReturn, ExtremeState, Overload, LowAltitude, Timeout, and observations list partners then enemies.
__device__ __forceinline__ Enemy gather_pose(const Props& pr, int lane) {
  Enemy E;
  E.n = __shfl(pr.n, lane); E.e = __shfl(pr.e, lane); E.u = __shfl(pr.u, lane);
  E.vn = __shfl(pr.vn, lane); E.ve = __shfl(pr.ve, lane); E.vd = __shfl(pr.vd, lane);
  E.ub = __shfl(pr.ub, lane); E.alt = __shfl(pr.alt_m, lane);
  return E;
}
// observation (9 + 6*(A-1), clipped) and the raw posture reward (sum over enemies) in one pass over the other aircraft. The row is
// this lane's row of the output staging buffer in LDS (null: the posture alone): a block's place in it is a run-time index -- an LDS
// address; on a register array the compiler turns the select chain into a dynamically indexed store and the array goes to scratch.
template <int A>
__device__ __forceinline__ float observe_nvn(const Props& pr, int slot, int base_lane, int n_ego, float* ob) {
  auto c10 = [](float v) { return clampf(-10.0f, v, 10.0f); };
  if (ob) {
    ob[0] = c10(pr.alt_m / 5000.0f);
    ob[1] = c10(pr.sphi); ob[2] = c10(pr.cphi); ob[3] = c10(pr.stht); ob[4] = c10(pr.ctht);
    ob[5] = c10(pr.ub / 340.0f); ob[6] = c10(pr.vb / 340.0f); ob[7] = c10(pr.wb / 340.0f); ob[8] = c10(pr.vc / 340.0f);
  }
  const int my_team = slot < n_ego ? 0 : 1;
  // position of aircraft j in my observation: partners (same team, in env order, skipping me) first, then enemies
  const int n_mine = my_team == 0 ? n_ego : A - n_ego;
  float posture = 0.0f;
#pragma unroll
  for (int j = 0; j < A; ++j) {
    Enemy E = gather_pose(pr, base_lane + j);   // every lane takes part in the shuffle
    const int team_j = j < n_ego ? 0 : 1;
    if (j == slot) continue;
    Geo g = ao_ta_r<false>(pr.n, pr.e, pr.u, pr.vn, pr.ve, pr.vd, E.n, E.e, E.u, E.vn, E.ve, E.vd);
    int idx;
    if (team_j == my_team) {
      int first = my_team == 0 ? 0 : n_ego;
      idx = (j - first) - (j > slot ? 1 : 0);
    } else {
      int first = my_team == 0 ? n_ego : 0;
      idx = (n_mine - 1) + (j - first);
      posture += posture_fn(g.AO, g.TA, g.R * 0.001f);
    }
    if (ob) {
      float* blk = ob + 9 + idx * 6;
      blk[0] = c10((E.ub - pr.ub) / 340.0f); blk[1] = c10((E.alt - pr.alt_m) / 1000.0f);
      blk[2] = c10(g.AO); blk[3] = c10(g.TA); blk[4] = c10(g.R / 10000.0f); blk[5] = c10(g.side);
    }
  }
  return posture;
}

// HierarchicalMultipleCombatShootTask (`hierarchical_multiplecombat_shoot`, multiplecombat_with_missile_task.py:206-238) observes
// like MultipleCombatDodgeMissileTask.get_obs (:33-117): 21 values, 3-D AO / TA, unclipped, against the enemy with the agent's own
// index in its team, and a missile block that stays zero (this task's step() never launches anything, :202-203).
template <int A>
__device__ __forceinline__ void legacy_obs_nvn(const Props& pr, int slot, int base_lane, int n_ego, float* row) {
  const int team = slot < n_ego ? 0 : 1;
  const int paired = (team == 0 ? n_ego : 0) + (slot - (team == 0 ? 0 : n_ego));
  const Enemy E = gather_pose(pr, base_lane + paired);
  const Incoming none{false, 0, 0, 0, 0, 0, 0};
  float o21[21];
  observe_1v1<AC_TASK_SHOOT_MISSILE>(pr, E, none, o21);
#pragma unroll
  for (int k = 0; k < 21; ++k) row[k] = o21[k];
}

template <int A, int WPE, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 192 : 64, WPE) void step_kernel_nvn(DevPtrs P, DevCfg c) {
  constexpr int OBS = 9 + 6 * (A - 1);
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  __shared__ __attribute__((aligned(16))) float lds_out[64 * OBS];
  __shared__ __attribute__((aligned(16))) char split_lds[SPLIT ? sizeof(SplitLds) : 16];
  SplitLds& L = *reinterpret_cast<SplitLds*>(split_lds);
  const Tab T{lds_tab};
  const int N = c.N;
  const int l = threadIdx.x & 63;
  const int n = blockIdx.x * 64 + l;
  const bool live = n < N;                      // N is a multiple of A, so an env is live or not as a whole
  const int slot = l % A;
  const int nn = live ? n : (N - A + slot);     // tail lanes shadow the last env and never store
  const int base_lane = l - slot;
  const int team = slot < c.n_ego ? 0 : 1;

  State s; Task t; Derived d; Props pr;
  TableCopy<SPLIT ? 192 : 64> tc;               // table loads, the action row and the state behind them: one HBM round trip
  tc.issue(P.tab);
  const float4 a4 = load_controls(P.actions + (size_t)nn * c.act_dim, c.act_dim);
  load_state(P.F, P.I, P.D, N, nn, s, t);
  const Task was = t;
  tc.commit(lds_tab);
  t.cur_step += 1;
  s.da = clampf(-1.0f, a4.x / 20.0f - 1.0f, 1.0f);   // multiplecombat_task.py:137-145
  s.de = clampf(-1.0f, a4.y / 20.0f - 1.0f, 1.0f);
  s.dr = clampf(-1.0f, a4.z / 20.0f - 1.0f, 1.0f);
  s.thr = clampf(0.0f, a4.w / 58.0f + 0.4f, 0.9f);
  if (SPLIT && split_helper_wave(s, t, T, L, l, c.substeps)) return;
  bool have_pose = false;
  int nrun_split = 0;
  const bool split_located = SPLIT && dynamics_wave_ticks(s, t, d, T, L, l, c.substeps, nrun_split);
  for (int sub = 0; sub < c.substeps && !SPLIT; ++sub) {
    if (t.status == AC_ALIVE) {
      if (t.bloods <= 0.0f) t.status = AC_SHOTDOWN;
      f16::tick<false>(s, d, T);
      have_pose = true;
    }
  }
  if (!split_located) {
    f16::locate(s, d);
    if (!have_pose) f16::body_frame(s, d);
  }
  make_props(s, d, c, pr);

  // (the template keeps the kernel's own OBS stride; hierarchical_multiplecombat_shoot puts out the first 21 values)
  const int ow = c.legacy_obs ? 21 : OBS;
  float* orow = lds_out + l * ow;                      // this lane's row of the output staging buffer
  float posture = observe_nvn<A>(pr, slot, base_lane, c.n_ego, c.legacy_obs ? nullptr : orow);
  if (c.legacy_obs) legacy_obs_nvn<A>(pr, slot, base_lane, c.n_ego, orow);

  // ---- rewards first (multiplecombat_env.py:166-175), only while alive (multiplecombat_task.py:147-151)
  float own = 0.0f;
  if (t.status == AC_ALIVE) {
    float r_alt = potential(altitude_raw(pr, c), c.altitude_scale, c.altitude_pot, t.pre_altitude);
    float r_pos = potential(posture, c.posture_scale, c.posture_pot, t.pre_posture);
    float r_ev = potential(0.0f, c.event_scale, c.event_pot, t.pre_event);  // alive and no missiles in this task: raw value 0
    own = r_alt + r_pos + r_ev;
  }
  float tsum = 0.0f;
#pragma unroll
  for (int j = 0; j < A; ++j) {
    float rj = __shfl(own, base_lane + j);
    if ((j < c.n_ego ? 0 : 1) == team) tsum += rj;
  }
  const float reward = tsum / (float)(team == 0 ? c.n_ego : A - c.n_ego);

  // ---- terminations, agent by agent in env order (multiplecombat_env.py:177-180): agent i sees the final status of agents
  // before it and the not-yet-evaluated status of agents after it
  bool done = false;
  int code = AC_DONE_NONE;
  int last_code = AC_DONE_NONE;
  float np_max = fmaxf(fabsf(s.npx), fmaxf(fabsf(s.npy), fabsf(s.npz)));
  float pqr = sqrtf(d.p * d.p + d.q * d.q + d.r * d.r);
  const bool nonfinite = nonfinite_probe(d.veci, pqr, d.h_sl_ft, np_max);
  const bool extreme = nonfinite || (d.veci >= 1e10f) || (pqr >= 1000.0f) || (d.h_sl_ft >= 1e10f) || (np_max > 10.0f);
  const bool overload = (s.ticks >= kTickOverload) &&
                        (fabsf(s.npx) > c.acc_x || fabsf(s.npy) > c.acc_y || fabsf(s.npz + 1.0f) > c.acc_z);
  const bool low = pr.alt_m <= c.altitude_limit;
  // (To the others a status only matters as "alive or not", and an agent's own checks change it in one way -- a crash condition makes
  // it CRASH. Two ballots give every lane the env's inputs as A-bit masks; every lane runs the A rounds on those bits in its own
  // registers and keeps what happened at its own turn; the messages are assigned afterwards: SafeReturn comes first, so only an
  // aircraft that is still flying and has no mission-complete reaches the crash checks. Same form as the scenario kernels' walk.)
  const unsigned long long env_mask = ((A == 64) ? ~0ull : ((1ull << A) - 1ull)) << base_lane;
  const int st0 = t.status;
  const bool crash_cond = extreme || overload || low;
  constexpr unsigned amask = (A >= 32) ? ~0u : ((1u << A) - 1u);
  const unsigned al0 = (unsigned)(__ballot(st0 == AC_ALIVE) >> base_lane) & amask;
  const unsigned ccb = (unsigned)(__ballot(crash_cond) >> base_lane) & amask;
  const unsigned team0 = (1u << c.n_ego) - 1u, team1 = amask & ~team0;
  unsigned alive = al0;
  bool enemies_dead = false, crash_mine = false;
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const bool ed = (alive & (i < c.n_ego ? team1 : team0)) == 0;
    const bool crash_now = ((al0 >> i) & 1u) && !ed && ((ccb >> i) & 1u);
    if (crash_now) alive &= ~(1u << i);
    if (i == slot) { enemies_dead = ed; crash_mine = crash_now; }
  }
  if (crash_mine) t.status = AC_CRASH;
  if (st0 == AC_SHOTDOWN) { code = AC_DONE_SHOTDOWN; done = true; }
  else if (st0 == AC_CRASH) { code = AC_DONE_CRASHED; done = true; }
  else if (enemies_dead) { code = AC_DONE_MISSION_COMPLETE; done = true; }   // no missiles in this task
  else if (extreme) { code = AC_DONE_EXTREME_STATE; done = true; }
  else if (overload) { code = AC_DONE_OVERLOAD; done = true; }
  else if (low) { code = AC_DONE_LOW_ALTITUDE; done = true; }
  else if (t.cur_step >= c.max_steps) { code = AC_DONE_TIMEOUT; done = true; }
  {   // info['done_condition'] keeps the message of the last agent (in env order) that has one
    const unsigned long long coded = __ballot(code != AC_DONE_NONE) & env_mask;
    const int last = coded ? 63 - __clzll((long long)coded) : l;
    last_code = __shfl(code, last);
  }
  const bool all_done = ((unsigned)(__ballot(done) >> base_lane) & amask) == amask;
  int step_out = t.cur_step;
  if (all_done) {
    load_state(P.tF, P.tI, P.tD, A, slot, s, t);
    zero_controller_state(P, N, n, live);
    const float* tobs = P.tF + (size_t)NSW * A + slot * OBS;
    for (int k = 0; k < ow; ++k) orow[k] = tobs[k];
  }
  if (live) {
    store_flight(P.F, P.I, P.D, N, n, s);
    store_task_changed(P.F, P.I, N, n, t, was);
  }
  emit_rows(P, lds_out, ow, l, poison_if(nonfinite, reward), done, A, step_out, last_code, 0, all_done ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------ initial conditions
// AircraftSimulator.reload (simulatior.py:152-190) for the scenario's A aircraft, run once per ac_create:
// FGInitialCondition setters -> FGPropagate::SetInitialState -> RunIC (two suspended executive passes,
// InitializeDerivatives) -> engine InitRunning -> GetSteadyState. Output: the reset template
// (state, task bookkeeping with the potential-reward seeds of reward_function_base.py:20-32, and the reset observation).
struct InitArgs { ac_init_state_t ic[AC_MAX_AGENTS]; };

__device__ double geod_alt_from_asl(double geod_lat, double alt) {  // FGInitialCondition.cpp:749-823 (setgeod branch)
  const double a = f16::kA, b = f16::kB, e2 = 1.0 - b * b / (a * a);
  double cg = cos(geod_lat), sg = sin(geod_lat), Nn = a / sqrt(1 - e2 * sg * sg);
  double n = e2, prev_n = 1.0;
  int iter = 0;
  if (cg > fabs(sg)) {
    double tg = sg / cg, x0 = Nn * e2 * cg, x = 0.0;
    while (fabs(n - prev_n) > 1E-15 && iter < 10) {
      double tl = (1 - n) * tg, c2 = 1. / (1. + tl * tl), slr = b / sqrt(1. - e2 * c2), R = slr + alt;
      x = R * sqrt(c2); prev_n = n; n = x0 / x; iter++;
    }
    return x / cg - Nn;
  }
  double ctg = cg / sg, z0 = Nn * e2 * sg, z = 0.0;
  while (fabs(n - prev_n) > 1E-15 && iter < 10) {
    double ctl = ctg / (1 - n), s2 = 1. / (1. + ctl * ctl), c2 = 1. - s2, slr = b / sqrt(1. - e2 * c2), R = slr + alt;
    z = R * (ctl >= 0 ? 1.0 : -1.0) * sqrt(s2); prev_n = n; n = z0 / (z0 + z); iter++;
  }
  return z / sg - Nn * (1 - e2);
}

__device__ void initial_state(const ac_init_state_t& ic, const Tab& T, State& s, Derived& d) {
  s = State{};
  const double D2R = 3.14159265358979323846 / 180.0;
  double lon = ic.lon_deg * D2R, lat = ic.lat_geod_deg * D2R, psi = ic.psi_deg * D2R;
  double hg = geod_alt_from_asl(lat, ic.h_sl_ft);
  const double ee = 1.0 - (f16::kB / f16::kA) * (f16::kB / f16::kA);
  double sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
  double RN = f16::kA / sqrt(1.0 - ee * sl * sl);
  s.rx = (RN + hg) * cl * co; s.ry = (RN + hg) * cl * so; s.rz = ((1 - ee) * RN + hg) * sl;  // epa = 0: ECI == ECEF
  // qAttitudeECI = Ti2l.GetQuaternion() * qAttitudeLocal (FGPropagate.cpp:166-167); Ti2l = Tec2l at epa = 0 and the
  // local attitude is a pure yaw (phi = theta = 0). Built the same way so the quaternion's sign matches as well.
  double Tl[9] = {-co * sl, -so * sl, cl, -so, co, 0.0, -co * cl, -so * cl, -sl};  // Tec2l rows N, E, D
  double tr[4] = {1.0 + Tl[0] + Tl[4] + Tl[8], 1.0 + Tl[0] - Tl[4] - Tl[8], 1.0 - Tl[0] + Tl[4] - Tl[8], 1.0 - Tl[0] - Tl[4] + Tl[8]};
  int idx = 0;
  for (int i = 1; i < 4; ++i) if (tr[i] > tr[idx]) idx = i;
  double a4[4];
  double m12 = Tl[1], m21 = Tl[3], m13 = Tl[2], m31 = Tl[6], m23 = Tl[5], m32 = Tl[7];
  if (idx == 0) { a4[0] = 0.5 * sqrt(tr[0]); a4[1] = 0.25 * (m23 - m32) / a4[0]; a4[2] = 0.25 * (m31 - m13) / a4[0]; a4[3] = 0.25 * (m12 - m21) / a4[0]; }
  else if (idx == 1) { a4[1] = 0.5 * sqrt(tr[1]); a4[0] = 0.25 * (m23 - m32) / a4[1]; a4[2] = 0.25 * (m12 + m21) / a4[1]; a4[3] = 0.25 * (m31 + m13) / a4[1]; }
  else if (idx == 2) { a4[2] = 0.5 * sqrt(tr[2]); a4[0] = 0.25 * (m31 - m13) / a4[2]; a4[1] = 0.25 * (m12 + m21) / a4[2]; a4[3] = 0.25 * (m23 + m32) / a4[2]; }
  else { a4[3] = 0.5 * sqrt(tr[3]); a4[0] = 0.25 * (m12 - m21) / a4[3]; a4[1] = 0.25 * (m13 + m31) / a4[3]; a4[2] = 0.25 * (m23 + m32) / a4[3]; }
  double b0 = cos(0.5 * psi), b3 = sin(0.5 * psi);  // yaw-only local quaternion (b1 = b2 = 0)
  double q[4] = {a4[0] * b0 - a4[3] * b3, a4[1] * b0 + a4[2] * b3, a4[2] * b0 - a4[1] * b3, a4[0] * b3 + a4[3] * b0};
  double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] *= qn;
  s.q0 = (float)q[0]; s.q1 = (float)q[1]; s.q2 = (float)q[2]; s.q3 = (float)q[3];
  double Tb[9] = {q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3], 2.0 * (q[1] * q[2] + q[0] * q[3]), 2.0 * (q[1] * q[3] - q[0] * q[2]),
                  2.0 * (q[1] * q[2] - q[0] * q[3]), q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3], 2.0 * (q[2] * q[3] + q[0] * q[1]),
                  2.0 * (q[1] * q[3] + q[0] * q[2]), 2.0 * (q[2] * q[3] - q[0] * q[1]), q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]};
  // inertial rates and velocity: PQRi = PQR + Ti2b*Omega; v_eci = Tb2i*UVW + Omega x r
  s.wp = (float)(ic.p_rad_sec + f16::kOmega * Tb[2]); s.wq = (float)(ic.q_rad_sec + f16::kOmega * Tb[5]); s.wr = (float)(ic.r_rad_sec + f16::kOmega * Tb[8]);
  double vix = Tb[0] * ic.u_fps + Tb[3] * ic.v_fps + Tb[6] * ic.w_fps;
  double viy = Tb[1] * ic.u_fps + Tb[4] * ic.v_fps + Tb[7] * ic.w_fps;
  double viz = Tb[2] * ic.u_fps + Tb[5] * ic.v_fps + Tb[8] * ic.w_fps;
  s.vx = (float)(vix - f16::kOmega * s.ry); s.vy = (float)(viy + f16::kOmega * s.rx); s.vz = (float)viz;
  s.tank0 = (float)F16_TANK0_CONTENTS; s.tank1 = (float)F16_TANK1_CONTENTS;
  s.eng = f16::PH_OFF | f16::ENG_CUTOFF;  // FGTurbine::ResetToIC
  // RunIC: Initialize() and RunIC() each run the executive once with dT = 0
  f16::tick<true, true>(s, d, T);
  f16::tick<true>(s, d, T);
  // InitializeDerivatives: every history slot holds the current derivative
  s.hv1x = s.hv2x = s.vx; s.hv1y = s.hv2y = s.vy; s.hv1z = s.hv2z = s.vz;
  s.ha1x = s.aix; s.ha1y = s.aiy; s.ha1z = s.aiz;
  // engine.init_running(): N1/N2 to idle at throttle 0, phase Run; get_steady_state(): fuel flow settles
  // (Seek at 500 pph per 0.5 s iteration for >= 121 iterations reaches its clamp) on max(thrust*tsfc, idle flow)
  s.n2 = (float)F16_ENG_IDLEN2; s.n1 = (float)F16_ENG_IDLEN1; s.n2norm = 0.0f;
  s.eng = f16::PH_RUN | f16::ENG_RUNNING;
  {
    f16::Atmos A = f16::atmosphere(d.h_sl_ft);
    float idle_f, mil_f, aug_f;
    f16::engine_factors(T, s.mach, d.h_sl_ft, idle_f, mil_f, aug_f);
    float idle = (float)F16_ENG_MILTHRUST * idle_f;
    float tsfc = (float)F16_ENG_TSFC * sqrtf(A.T * (1.0f / 389.7f)) * (0.84f + 1.0f);
    float target = idle * tsfc;            // thrust at N2norm = 0 is the idle thrust
    float ff = (target > 0.0f) ? target : 0.0f;  // Seek from 0 never goes below 0 here: a negative target is approached from above only
    s.ff = fmaxf(ff, 757.648518f);
  }
}

template <int TASK>
__global__ __launch_bounds__(64) void init_kernel_1v1(InitArgs ia, DevCfg c, const float* tab, float* tF, int* tI, double* tD) {
  using TT = TaskTraits<TASK>;
  constexpr int OBS = TT::OBS;
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  stage_tables(lds_tab, tab);
  const Tab T{lds_tab};
  const int slot = threadIdx.x & 1;
  State s; Derived d; Task t{}; Props pr;
  initial_state(ia.ic[slot], T, s, d);
  t.bloods = 100.0f; t.status = AC_ALIVE;
  t.remaining = c.num_missiles[slot]; t.pre_remaining = c.num_missiles[slot];
  t.last_missile = -1; t.last_shoot_time = -c.min_attack_interval;
  f16::locate(s, d);
  make_props(s, d, c, pr);
  Enemy E = exchange_1v1(pr);
  Incoming inc{false, 0, 0, 0, 0, 0, 0};
  float ob[OBS];
  observe_1v1<TASK>(pr, E, inc, ob);
  // potential-based terms are seeded with their value at reset (reward_function_base.py:28-31)
  if (c.altitude_pot) t.pre_altitude = altitude_raw(pr, c) * c.altitude_scale;
  if (c.posture_pot) t.pre_posture = posture_raw(pr, E) * c.posture_scale;
  if (threadIdx.x < 2) {
    store_state(tF, tI, tD, 2, slot, s, t);
    float* tobs = tF + (size_t)NSW * 2 + slot * OBS;
    for (int k = 0; k < OBS; ++k) tobs[k] = ob[k];
  }
}

template <int A>
__global__ __launch_bounds__(64) void init_kernel_nvn(InitArgs ia, DevCfg c, const float* tab, float* tF, int* tI, double* tD) {
  constexpr int OBS = 9 + 6 * (A - 1);
  __shared__ __attribute__((aligned(16))) float lds_tab[F16_PACK_LEN];
  stage_tables(lds_tab, tab);
  const Tab T{lds_tab};
  const int slot = threadIdx.x % A;
  const int base_lane = (threadIdx.x & 63) - slot;
  State s; Derived d; Task t{}; Props pr;
  initial_state(ia.ic[slot], T, s, d);
  t.bloods = 100.0f; t.status = AC_ALIVE;
  t.remaining = c.num_missiles[slot]; t.pre_remaining = c.num_missiles[slot];
  t.last_missile = -1; t.last_shoot_time = -c.min_attack_interval;
  f16::locate(s, d);
  make_props(s, d, c, pr);
  float ob[OBS];
  for (int k = 0; k < OBS; ++k) ob[k] = 0.0f;
  float posture = observe_nvn<A>(pr, slot, base_lane, c.n_ego, ob);
  if (c.legacy_obs) legacy_obs_nvn<A>(pr, slot, base_lane, c.n_ego, ob);
  if (c.altitude_pot) t.pre_altitude = altitude_raw(pr, c) * c.altitude_scale;
  if (c.posture_pot) t.pre_posture = posture * c.posture_scale;
  if (threadIdx.x < A) {
    store_state(tF, tI, tD, A, slot, s, t);
    float* tobs = tF + (size_t)NSW * A + slot * OBS;
    for (int k = 0; k < OBS; ++k) tobs[k] = ob[k];
  }
}

#include "scenario_kernel.hpp"
#include "controller_common.hpp"
#include "controller_pieces.hpp"
#include "controller8_kernel.hpp"
#include "heading_kernel.hpp"

// reset(): every env takes the template (SubprocVecEnv.reset -> env.reset(), env_base.py:98-113)
__global__ void reset_all_kernel(DevPtrs P, DevCfg c) {
  const int OBS = c.obs_dim;
  const int N = c.N;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int slot = n % c.A;
  State s; Task t;
  load_state(P.tF, P.tI, P.tD, c.A, slot, s, t);
  store_state(P.F, P.I, P.D, N, n, s, t);
  for (int k = 0; k < c.msl_slots; ++k) {
    if (P.MD) { MslD m{}; m.status = MSL_INACTIVE; store_msl(P.MD, P.MI, N, n, k, m); }
    else { Msl m{}; m.status = MSL_INACTIVE; store_msl(P.MF, P.MI, N, n, k, m); }
  }
  const int TOBS = c.tobs;   // the template holds the kernel family's own layout; RWR appends two zero slots, WVR uses the first 15
  const float* tobs = P.tF + (size_t)NSW * c.A + slot * TOBS;
  for (int k = 0; k < OBS; ++k) P.obs[(size_t)n * OBS + k] = (k < TOBS) ? tobs[k] : 0.0f;
  zero_controller_state(P, N, n, true);
  P.rew[n] = 0.0f; P.done[n] = 0;
  if (slot == 0) { int* inf = P.info + (size_t)(n / c.A) * 4; inf[0] = 0; inf[1] = 0; inf[2] = 0; inf[3] = 0; }
}

// ------------------------------------------------------------------------------------------------ host side (C ABI)
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

struct ac_env {
  ac_config_t cfg;
  DevCfg dc;
  DevPtrs dp;
  int E, A, N, obs_dim, act_dim, device;
  hipStream_t stream;
  float* d_actions;
  float* d_tab;
  float* d_tF; int* d_tI; double* d_tD;
  float* d_XF; int* d_XI;                // scenario-task extension state
  double* d_state_io;                    // ac_get_state / ac_set_state: one aircraft's record on its way through (state_io_kernel)
  float* d_low;                          // hierarchical tasks: low-level action buffer (the controller's output, the step kernel's input)
  float* d_ctlWs8;                       // controller weights as fp16 pieces in the kernel's tiling (controller8_kernel.hpp)
  int ctl_rows;                          // aircraft per controller workgroup pinned by AIRCOMBAT_CTL_ROWS=32/64 (0: chosen per grid)
  HeadingPtrs hp; HeadingCfg hc;         // HeadingTask: targets, check clock, numpy-PCG64 state per env
  int act_low;                           // width of the low-level action the step kernels decode
  hipEvent_t ev0, ev1;
  hipEvent_t ev_mid;                     // ac_step_timed_device: between the controller kernel and the step kernel
  bool mark_mid;
  hipEvent_t ev_order;                   // stream-ordering hand-shake with the caller's streams (ac_order_after / ac_order_before)
  // ac_step_host: two library-owned sets of pinned host buffers mapped into the device (actions in; obs, rewards, dones, info out)
  struct HostSet { float* act; float* obs; float* rew; uint8_t* done; int* info; } hs[AC_HOST_SETS];   // allocated on first use (ac_host_buffers)
  bool have_hs[AC_HOST_SETS];
  int* err_host;                         // P.err: one word of page-locked host memory the step kernels write on a non-finite state (sticky until ac_reset)
  int err_sticky;
  int* count_host;                       // ac_munitions_in_flight's counter: one page-locked, device-mapped word allocated with the handle
  bool timing;
  bool quad_waves;                       // the 1v1 tasks with munitions up to one workgroup per CU: three FDM waves + the environment wave (FORM 3 / FORM_QUAD)
  bool split_waves;                      // SingleCombat below one wave per SIMD: three waves per 64 aircraft (step_kernel_1v1<.., SPLIT>)
};

static void geodetic2ecef_m(double lat_deg, double lon_deg, double alt, double* x, double* y, double* z) {
  const double a = 6378137.0, b = 6356752.314245179, D2R = M_PI / 180.0;  // pymap3d WGS84
  double lat = lat_deg * D2R, lon = lon_deg * D2R;
  double Nn = a * a / hypot(a * cos(lat), b * sin(lat));
  *x = (Nn + alt) * cos(lat) * cos(lon);
  *y = (Nn + alt) * cos(lat) * sin(lon);
  *z = (Nn * (b / a) * (b / a) + alt) * sin(lat);
}

// The non-finite guard's host half: after a completed step, a set error word fails the call with the reference's message
// (RuntimeError("JSBSim failed."), simulatior.py:223-225) and the aircraft it was seen on; it stays set until ac_reset.
static int check_nonfinite(ac_env* h, const char* who) {
  if (h->err_host && *(volatile int*)h->err_host) h->err_sticky = *(volatile int*)h->err_host;
  if (!h->err_sticky) return 0;
  const int n = std::max(0, std::min(0x3fffffff - (h->err_sticky & 0x3fffffff), h->N - 1));   // emit_scalars: (own probe << 30) | (0x3fffffff - aircraft)
  char msg[192];
  snprintf(msg, sizeof msg, "%s: JSBSim failed. Non-finite state or reward in env %d, agent %d (ac_reset clears the condition)", who, n / h->A, n % h->A);
  return fail(msg);
}
static int launch_step(ac_env* h, const float* d_actions, int host_set = -1) {
  DevPtrs p = h->dp;
  p.actions = d_actions ? d_actions : h->d_actions;
  if (host_set >= 0) {   // actions read from, and a second copy of every output written to, mapped host memory
    const ac_env::HostSet& hs = h->hs[host_set];
    p.actions = hs.act; p.obs2 = hs.obs; p.rew2 = hs.rew; p.done2 = hs.done; p.info2 = hs.info;
  }
  dim3 block(64), grid((h->N + 63) / 64);
  if (h->cfg.hierarchical) {   // [3,5,3] (+ weapon bits) -> control indices, then the ordinary step on those
    if (!h->d_ctlWs8) return fail("hierarchical task: ac_load_controller has not been called");
    ctl::Args a{h->d_ctlWs8, p.actions, p.obs, p.H, h->d_low, h->N, h->obs_dim, h->act_dim, h->act_low,
                h->cfg.use_baseline, h->A, h->cfg.n_ego, h->cfg.use_artillery,
                (float)h->cfg.agent_interaction_steps / (float)h->cfg.sim_freq, p.man_step, p.man_h0, p, h->dc};
    // (the scripted opponents' inputs -- use_baseline -- are computed inside that instantiation of the kernel)
    {   // eight waves per tile (controller8_kernel.hpp): 32 aircraft per workgroup while that leaves no CU with two tiles to do one
        // after the other, 64 beyond (AIRCOMBAT_CTL_ROWS pins it for tests)
      const int rows = h->ctl_rows ? h->ctl_rows : ((h->N + 31) / 32 > 256 ? 64 : 32);
      const dim3 g8((h->N + rows - 1) / rows);
      if (rows == 64) {
        if (h->cfg.use_baseline) hipLaunchKernelGGL((controller8_kernel<true, 4>), g8, dim3(512), 0, h->stream, a);
        else hipLaunchKernelGGL((controller8_kernel<false, 4>), g8, dim3(512), 0, h->stream, a);
      } else if (h->cfg.use_baseline) hipLaunchKernelGGL((controller8_kernel<true, 2>), g8, dim3(512), 0, h->stream, a);
      else hipLaunchKernelGGL((controller8_kernel<false, 2>), g8, dim3(512), 0, h->stream, a);
    }
    HIP_OK(hipGetLastError());
    if (h->mark_mid) HIP_OK(hipEventRecord(h->ev_mid, h->stream));
    p.actions = h->d_low;
  }
  const bool one_wave_per_simd = grid.x <= 1024;  // 256 CUs x 4 SIMDs
  if (h->cfg.task == AC_TASK_HEADING) {
    if (h->split_waves) hipLaunchKernelGGL(step_kernel_heading<true>, grid, dim3(192), 0, h->stream, p, h->dc, h->hp, h->hc, 0);
    else if (one_wave_per_simd) hipLaunchKernelGGL((step_kernel_heading<false, 1>), grid, block, 0, h->stream, p, h->dc, h->hp, h->hc, 0);
    else hipLaunchKernelGGL((step_kernel_heading<false, 2>), grid, block, 0, h->stream, p, h->dc, h->hp, h->hc, 0);
    HIP_OK(hipGetLastError());
    return 0;
  }
  // pair form (every task with munitions): two waves per workgroup, so one wave per SIMD up to 512 workgroups
  const bool pair_wpe1 = grid.x <= 512;
  const bool gun_only = h->cfg.task == AC_TASK_WVR || h->cfg.task == AC_TASK_MANEUVER;
  if (gun_only) {   // the scenario kernel family without munitions: ticks only between the env steps
    if (h->split_waves) hipLaunchKernelGGL((step_kernel_scenario<2, 1, FORM_SPLIT>), grid, dim3(192), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);
    else if (pair_wpe1) hipLaunchKernelGGL((step_kernel_scenario<2, 1, FORM_PAIR>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);
    else hipLaunchKernelGGL((step_kernel_scenario<2, 2, FORM_PAIR>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);
  } else if (h->cfg.task == AC_TASK_DODGE_MISSILE && h->A > 2) {   // multiplecombat_dodge_missile: the scenario kernel family's NvN pair form, DODGE build
#define AC_LAUNCH_DODGE(AA)                                                                                                                                     \
  do {                                                                                                                                                          \
    if (pair_wpe1) hipLaunchKernelGGL((step_kernel_scenario<AA, 1, FORM_PAIR, true>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr); \
    else hipLaunchKernelGGL((step_kernel_scenario<AA, 2, FORM_PAIR, true>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);          \
  } while (0)
    if (h->A == 4) AC_LAUNCH_DODGE(4); else AC_LAUNCH_DODGE(8);
#undef AC_LAUNCH_DODGE
  } else if (h->cfg.task == AC_TASK_SCENARIO1 || h->cfg.task == AC_TASK_SCENARIO_NVN) {
#define AC_LAUNCH_PAIR(AA)                                                                                                                                \
  do {                                                                                                                                                    \
    if (pair_wpe1) hipLaunchKernelGGL((step_kernel_scenario<AA, 1, FORM_PAIR>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr); \
    else hipLaunchKernelGGL((step_kernel_scenario<AA, 2, FORM_PAIR>), grid, dim3(128), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);          \
  } while (0)
    // (the quad form for the 1v1 scenario only: for 2v2 / 4v4 it takes fewer cycles than the pair form and more time -- with four busy
    // waves per CU and the fp64 munitions of four or eight aircraft the part clocks lower, 1.8 against 2.05 GHz)
    if (h->A == 2 && h->quad_waves) hipLaunchKernelGGL((step_kernel_scenario<2, 1, FORM_QUAD>), grid, dim3(256), 0, h->stream, p, h->dc, h->d_XF, h->d_XI, nullptr, nullptr);
    else if (h->A == 2) AC_LAUNCH_PAIR(2); else if (h->A == 4) AC_LAUNCH_PAIR(4); else AC_LAUNCH_PAIR(8);
#undef AC_LAUNCH_PAIR
  } else if (h->cfg.task == AC_TASK_MULTICOMBAT) {
    if (h->split_waves) {
      if (h->A == 4) hipLaunchKernelGGL((step_kernel_nvn<4, 1, true>), grid, dim3(192), 0, h->stream, p, h->dc);
      else hipLaunchKernelGGL((step_kernel_nvn<8, 1, true>), grid, dim3(192), 0, h->stream, p, h->dc);
    } else if (h->A == 4) {
      if (one_wave_per_simd) hipLaunchKernelGGL((step_kernel_nvn<4, 1>), grid, block, 0, h->stream, p, h->dc);
      else hipLaunchKernelGGL((step_kernel_nvn<4, 2>), grid, block, 0, h->stream, p, h->dc);
    } else {
      if (one_wave_per_simd) hipLaunchKernelGGL((step_kernel_nvn<8, 1>), grid, block, 0, h->stream, p, h->dc);
      else hipLaunchKernelGGL((step_kernel_nvn<8, 2>), grid, block, 0, h->stream, p, h->dc);
    }
  } else if (h->cfg.task == AC_TASK_SINGLECOMBAT) {
    if (h->split_waves) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SINGLECOMBAT, 1, 1>), grid, dim3(192), 0, h->stream, p, h->dc);
    else if (one_wave_per_simd) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SINGLECOMBAT, 1, 0>), grid, block, 0, h->stream, p, h->dc);
    else hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SINGLECOMBAT, 2, 0>), grid, block, 0, h->stream, p, h->dc);
  } else if (h->cfg.task == AC_TASK_DODGE_MISSILE) {
    if (h->quad_waves) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_DODGE_MISSILE, 1, 3>), grid, dim3(256), 0, h->stream, p, h->dc);
    else if (pair_wpe1) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_DODGE_MISSILE, 1, 2>), grid, dim3(128), 0, h->stream, p, h->dc);
    else hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_DODGE_MISSILE, 2, 2>), grid, dim3(128), 0, h->stream, p, h->dc);
  } else {
    if (h->quad_waves) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SHOOT_MISSILE, 1, 3>), grid, dim3(256), 0, h->stream, p, h->dc);
    else if (pair_wpe1) hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SHOOT_MISSILE, 1, 2>), grid, dim3(128), 0, h->stream, p, h->dc);
    else hipLaunchKernelGGL((step_kernel_1v1<AC_TASK_SHOOT_MISSILE, 2, 2>), grid, dim3(128), 0, h->stream, p, h->dc);
  }
  HIP_OK(hipGetLastError());
  return 0;
}
static int launch_reset(ac_env* h) {
  dim3 block(64), grid((h->N + 63) / 64);
  if (h->cfg.task == AC_TASK_HEADING) {   // every env draws a new episode from its own generator
    hipLaunchKernelGGL((step_kernel_heading<false, 1>), grid, block, 0, h->stream, h->dp, h->dc, h->hp, h->hc, 1);
    HIP_OK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(reset_all_kernel, grid, block, 0, h->stream, h->dp, h->dc);
  HIP_OK(hipGetLastError());
  if (h->d_XF) {
    hipLaunchKernelGGL(reset_ext_kernel, grid, block, 0, h->stream, h->dc, h->d_XF, h->d_XI);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

extern "C" {

const char* ac_last_error(void) { return g_err.c_str(); }
const char* ac_version(void) { return "aircombat-hip 0.1 (gfx950)"; }
const char* ac_state_field_name(int i) { return (i >= 0 && i < AC_STATE_LEN && kStateNames[i]) ? kStateNames[i] : ""; }

int ac_create(const ac_config_t* cfg, int32_t n_envs, int32_t device_id, uint64_t seed, ac_env_t** out) {
  if (!cfg || !out) return fail("ac_create: null argument");
  // MultipleCombatDodgeMissileTask (multiplecombat_with_missile_task.py:13-145; `multiplecombat_dodge_missile`): AC_TASK_DODGE_MISSILE with more
  // than two aircraft. It runs on the scenario kernel family's NvN machinery (MultipleCombatEnv.step order, two munition uids per aircraft in fp64,
  // the 21-value paired-enemy observation) with the rule-based launch, the base-class missile and four reward terms.
  const bool nvn_dodge = cfg->task == AC_TASK_DODGE_MISSILE && cfg->n_agents > 2;
  const bool scenario = cfg->task == AC_TASK_SCENARIO1 || cfg->task == AC_TASK_SCENARIO_NVN || cfg->task == AC_TASK_WVR || cfg->task == AC_TASK_MANEUVER || nvn_dodge;
  const bool heading = cfg->task == AC_TASK_HEADING;
  if (cfg->task != AC_TASK_SINGLECOMBAT && cfg->task != AC_TASK_SHOOT_MISSILE && cfg->task != AC_TASK_DODGE_MISSILE &&
      cfg->task != AC_TASK_MULTICOMBAT && !scenario && !heading)
    return fail("ac_create: unknown task (supported: AC_TASK_HEADING, AC_TASK_SINGLECOMBAT, AC_TASK_DODGE_MISSILE, AC_TASK_SHOOT_MISSILE, AC_TASK_MULTICOMBAT, AC_TASK_SCENARIO1, AC_TASK_SCENARIO_NVN)");
  if (heading && (cfg->n_agents != 1 || cfg->n_ego != 1 || cfg->hierarchical))
    return fail("ac_create: AC_TASK_HEADING is a single-aircraft task with control-index actions (n_agents == n_ego == 1)");
  if (cfg->task == AC_TASK_DODGE_MISSILE && (cfg->sim_freq / cfg->agent_interaction_steps < 1 || cfg->sim_freq / cfg->agent_interaction_steps > 31))
    return fail("ac_create: AC_TASK_DODGE_MISSILE keeps its lock window in 31 bits (needs 1 <= sim_freq / agent_interaction_steps <= 31)");
  if (cfg->task == AC_TASK_SCENARIO_NVN || nvn_dodge) {
    if ((cfg->n_agents != 4 && cfg->n_agents != 8) || cfg->n_ego * 2 != cfg->n_agents)
      return fail("ac_create: AC_TASK_SCENARIO_NVN (and AC_TASK_DODGE_MISSILE with more than two aircraft) needs n_agents in {4, 8} split into two equal teams");
    if (nvn_dodge && cfg->rwr) return fail("ac_create: multiplecombat_dodge_missile has no rwr variant");
  }
  const bool gun_only = cfg->task == AC_TASK_WVR || cfg->task == AC_TASK_MANEUVER;
  if (scenario && !gun_only)
    for (int i = 0; i < cfg->n_agents; ++i)
      if (cfg->num_missiles[i] != 2) return fail("ac_create: the scenario tasks are built for 'missile: 2' (two munition uids per aircraft), as every shipped YAML has");
  if (cfg->task == AC_TASK_SCENARIO_NVN || nvn_dodge) {
  } else if (cfg->task == AC_TASK_MULTICOMBAT) {
    if ((cfg->n_agents != 4 && cfg->n_agents != 8) || cfg->n_ego <= 0 || cfg->n_ego >= cfg->n_agents)
      return fail("ac_create: AC_TASK_MULTICOMBAT needs n_agents in {4, 8} and 0 < n_ego < n_agents");
  } else if (!heading && (cfg->n_agents != 2 || cfg->n_ego != 1)) return fail("ac_create: 1v1 tasks need n_agents == 2 and n_ego == 1");
  if (cfg->use_baseline && (!cfg->hierarchical || cfg->use_baseline < 0 || cfg->use_baseline > 2 || cfg->n_ego * 2 != cfg->n_agents))
    return fail("ac_create: use_baseline (1 pursue, 2 maneuver) needs the hierarchical form and equal teams (enemy k is flown by scripted agent k)");
  if ((cfg->task == AC_TASK_WVR || cfg->task == AC_TASK_MANEUVER) && (cfg->n_agents != 2 || cfg->n_ego != 1 || cfg->rwr))
    return fail("ac_create: AC_TASK_WVR / AC_TASK_MANEUVER are 1v1 tasks");
  if (cfg->legacy_obs && ((cfg->task != AC_TASK_SCENARIO_NVN && cfg->task != AC_TASK_MULTICOMBAT && !nvn_dodge) || cfg->rwr))
    return fail("ac_create: legacy_obs is the observation of Scenario2 / Scenario3 (AC_TASK_SCENARIO_NVN without rwr) and of "
                "multiplecombat_shoot / hierarchical_multiplecombat_shoot (AC_TASK_MULTICOMBAT)");
  if (cfg->rwr && !scenario) return fail("ac_create: rwr is a variant of the scenario tasks (Scenario1_RWR, Scenario2_RWR, Scenario3_RWR)");
  if (n_envs <= 0) return fail("ac_create: n_envs must be positive");
  if (cfg->max_steps > 65535) return fail("ac_create: max_steps above 65535 (the host-boundary info word carries current_step in 16 bits)");
  if ((long long)n_envs * cfg->n_agents > (1 << 23))   // 32-bit byte offsets into the per-field arrays (AC_AT): 63 fields x 4 B x N < 4 GB
    return fail("ac_create: more than 2^23 aircraft in one handle; shard the envs over handles / GPUs");
  if (cfg->sim_freq != 60) return fail("ac_create: sim_freq must be 60 (the FDM tick is compiled for 1/60 s)");
  for (int i = 0; i < cfg->n_agents; ++i)
    if (cfg->num_missiles[i] < 0 || cfg->num_missiles[i] > AC_MAX_MISSILES_PER_AGENT) return fail("ac_create: num_missiles out of range");
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail("ac_create: no such HIP device");
  HIP_OK(hipSetDevice(device_id));
  ac_env* h = new ac_env();
  memset(h, 0, sizeof *h);
  h->cfg = *cfg; h->E = n_envs; h->A = cfg->n_agents; h->N = n_envs * cfg->n_agents; h->device = device_id;
  {  // Kernel form. Tasks whose substeps are the FDM tick alone: three waves per 64 aircraft (split_kernel.hpp) while the chip has
     // SIMDs to spare -- up to 512 workgroups (1536 waves on 256 CUs x 4 SIMDs; measured faster than one wave per 64 aircraft up to
     // there, slower from 768 on); AIRCOMBAT_SPLIT=0/1 overrides that choice. Tasks with munitions always run the pair form
     // (pair_kernel.hpp: a flight wave and an environment wave per 64 aircraft).
    const char* e = getenv("AIRCOMBAT_SPLIT");
    const int wgs = (h->N + 63) / 64;
    const bool ticks_only = cfg->task == AC_TASK_SINGLECOMBAT || cfg->task == AC_TASK_MULTICOMBAT || cfg->task == AC_TASK_HEADING ||
                            cfg->task == AC_TASK_WVR || cfg->task == AC_TASK_MANEUVER;
    h->split_waves = ticks_only && (e ? (e[0] == '1') : (wgs <= 512));
    const char* cr = getenv("AIRCOMBAT_CTL_ROWS");
    h->ctl_rows = cr ? (atoi(cr) == 64 ? 64 : 32) : 0;
    const char* qe = getenv("AIRCOMBAT_QUAD");   // 0 / 1 overrides the choice of the quad form
    const bool munitions_1v1 = cfg->task == AC_TASK_SHOOT_MISSILE || (cfg->task == AC_TASK_DODGE_MISSILE && !nvn_dodge) || cfg->task == AC_TASK_SCENARIO1;
    h->quad_waves = munitions_1v1 && (qe ? (qe[0] == '1') : (wgs <= 256));
  }
  h->obs_dim = heading ? 12 : (cfg->task == AC_TASK_SINGLECOMBAT) ? 15 : (cfg->task == AC_TASK_MULTICOMBAT ? 9 + 6 * (cfg->n_agents - 1) : 21);
  if (cfg->task == AC_TASK_SCENARIO_NVN || nvn_dodge) h->obs_dim = 9 + 6 * cfg->n_agents + 6;
  const int tmpl_obs = h->obs_dim;   // (the scenario kernel family's template keeps 21 slots for WVR too)
  if (gun_only) h->obs_dim = 15;
  if (((cfg->task == AC_TASK_SCENARIO_NVN || cfg->task == AC_TASK_MULTICOMBAT) && cfg->legacy_obs) || nvn_dodge) h->obs_dim = 21;   // multiplecombat_with_missile_task.py:30-31
  if (cfg->rwr) h->obs_dim += 2;   // scenario1_task.py:213-216, scenario2_task.py:403-413
  const bool weapon_bits = scenario && !gun_only && !nvn_dodge;
  h->act_low = weapon_bits ? 8 : ((cfg->task == AC_TASK_SHOOT_MISSILE) ? 5 : 4);
  // multiplecombat_shoot (MultipleCombatShootMissileTask, multiplecombat_with_missile_task.py:165-216): Tuple([41,41,41,30], Discrete(2)); like its
  // hierarchical child below, the shoot bit is stored (:204-206) and never used (its step() is MultipleCombatTask.step, :215-216)
  if (!cfg->hierarchical && cfg->task == AC_TASK_MULTICOMBAT && cfg->legacy_obs) h->act_low = 5;
  // hierarchical tasks (HierarchicalSingleCombatTask and everything built on it): [3,5,3] (+ the four weapon bits)
  // (HierarchicalSingleCombatShootTask: Tuple([3,5,3], Discrete(2)), singlecombat_with_missile_task.py:221-223; the Dodge variant: [3,5,3])
  h->act_dim = cfg->hierarchical ? (weapon_bits ? 7 : (cfg->task == AC_TASK_SHOOT_MISSILE ? 4 : 3)) : h->act_low;
  // hierarchical_multiplecombat_shoot: Tuple([3,5,3], Discrete(2)); the shoot bit is stored and never used (the task's step() launches nothing)
  if (cfg->hierarchical && cfg->task == AC_TASK_MULTICOMBAT && cfg->legacy_obs) h->act_dim = 4;
  DevCfg& c = h->dc;
  memset(&c, 0, sizeof c);
  c.task = cfg->task; c.A = h->A; c.n_ego = cfg->n_ego; c.substeps = cfg->agent_interaction_steps; c.max_steps = cfg->max_steps;
  c.obs_dim = h->obs_dim; c.act_dim = h->act_low; c.N = h->N;
  c.msl_slots = scenario ? 2 : ((cfg->task == AC_TASK_SHOOT_MISSILE || cfg->task == AC_TASK_DODGE_MISSILE) ? AC_MAX_MISSILES_PER_AGENT : 0);
  c.chaff_seed = seed;
  c.rwr = cfg->rwr ? 1 : 0;
  c.tobs = tmpl_obs;
  c.legacy_obs = (((cfg->task == AC_TASK_SCENARIO_NVN || cfg->task == AC_TASK_MULTICOMBAT) && cfg->legacy_obs) || nvn_dodge) ? 1 : 0;
  c.altitude_limit = (float)cfg->altitude_limit; c.acc_x = (float)cfg->acc_limit_x; c.acc_y = (float)cfg->acc_limit_y; c.acc_z = (float)cfg->acc_limit_z;
  c.posture_scale = (float)cfg->posture_scale; c.altitude_scale = (float)cfg->altitude_scale; c.event_scale = (float)cfg->event_scale;
  c.missile_posture_scale = (float)cfg->missile_posture_scale; c.shoot_penalty_scale = (float)cfg->shoot_penalty_scale;
  c.posture_pot = cfg->posture_potential; c.altitude_pot = cfg->altitude_potential; c.event_pot = cfg->event_potential; c.shoot_pot = cfg->shoot_penalty_potential;
  c.alt_safe = (float)cfg->alt_safe; c.alt_danger = (float)cfg->alt_danger; c.alt_kv = (float)cfg->alt_kv;
  c.max_attack_angle = (float)cfg->max_attack_angle; c.max_attack_distance = (float)cfg->max_attack_distance;
  c.min_attack_interval = cfg->min_attack_interval; c.use_artillery = cfg->use_artillery;
  c.lock_len = (int)(1.0 / ((double)cfg->agent_interaction_steps / cfg->sim_freq));
  for (int i = 0; i < AC_MAX_AGENTS; ++i) c.num_missiles[i] = cfg->num_missiles[i];
  geodetic2ecef_m(cfg->center_lat, cfg->center_lon, cfg->center_alt, &c.P0x, &c.P0y, &c.P0z);
  c.sLat0 = sin(cfg->center_lat * M_PI / 180.0); c.cLat0 = cos(cfg->center_lat * M_PI / 180.0);
  {
    const double a = 6378137.0, f = 1.0 / 298.257223563, e2 = f * (2.0 - f), w2 = 1.0 - e2 * c.sLat0 * c.sLat0;
    c.rn0 = (float)(a / sqrt(w2) + cfg->center_alt);
    c.rm0 = (float)(a * (1.0 - e2) / (w2 * sqrt(w2)) + cfg->center_alt);
    c.h0 = (float)cfg->center_alt;
  }
  c.sLon0 = sin(cfg->center_lon * M_PI / 180.0); c.cLon0 = cos(cfg->center_lon * M_PI / 180.0);

  HIP_OK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIP_OK(hipEventCreate(&h->ev0));
  HIP_OK(hipEventCreate(&h->ev1));
  HIP_OK(hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming));
  HIP_OK(hipEventCreate(&h->ev_mid));
  HIP_OK(hipHostMalloc((void**)&h->err_host, sizeof(int), hipHostMallocDefault));
  *h->err_host = 0; h->err_sticky = 0;
  h->dp.err = h->err_host;
  HIP_OK(hipHostMalloc((void**)&h->count_host, sizeof(int), hipHostMallocDefault));
  const size_t N = (size_t)h->N;
  DevPtrs& p = h->dp;
  HIP_OK(hipMalloc(&p.F, sizeof(float) * NSW * N));   // 19 groups of four words per aircraft, ints among them (see SW_*)
  p.I = reinterpret_cast<int*>(p.F);                  // (the same storage: kept as a name for the signatures that take the pair)
  HIP_OK(hipMalloc(&p.D, sizeof(double) * ND * N));
  const size_t ms = c.msl_slots > 0 ? (size_t)c.msl_slots : 1;
  HIP_OK(hipMalloc(&p.MF, sizeof(float) * ms * NMF * N));
  p.MD = nullptr;
  if (scenario) HIP_OK(hipMalloc(&p.MD, sizeof(double) * ms * NMF * N));
  HIP_OK(hipMalloc(&p.MI, sizeof(int) * ms * NMI * N));
  // outputs: rows padded to whole workgroups (emit_outputs stores whole blocks)
  const size_t Npad = (N + 63) / 64 * 64;
  HIP_OK(hipMalloc(&p.obs, sizeof(float) * Npad * h->obs_dim));
  HIP_OK(hipMalloc(&p.rew, sizeof(float) * Npad));
  HIP_OK(hipMalloc(&p.done, Npad));
  HIP_OK(hipMalloc(&p.info, sizeof(int) * 4 * (Npad / h->A)));
  p.obs2 = nullptr; p.rew2 = nullptr; p.done2 = nullptr; p.info2 = nullptr;
  HIP_OK(hipMalloc(&h->d_actions, sizeof(float) * N * h->act_dim));
  HIP_OK(hipMemset(h->d_actions, 0, sizeof(float) * N * h->act_dim));
  std::vector<float> tab(F16_PACK_LEN);
  for (int i = 0; i < F16_PACK_LEN; ++i) tab[i] = (float)F16_PACK[i];
  HIP_OK(hipMalloc(&h->d_tab, sizeof(float) * F16_PACK_LEN));
  HIP_OK(hipMemcpy(h->d_tab, tab.data(), sizeof(float) * F16_PACK_LEN, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&h->d_tF, sizeof(float) * ((size_t)NSW * h->A + (size_t)h->A * tmpl_obs)));
  h->d_tI = reinterpret_cast<int*>(h->d_tF);
  HIP_OK(hipMalloc(&h->d_tD, sizeof(double) * ND * h->A));
  HIP_OK(hipMalloc(&h->d_state_io, sizeof(double) * 256));
  if (scenario) {
    HIP_OK(hipMalloc(&h->d_XF, sizeof(float) * NXF * N));
    HIP_OK(hipMalloc(&h->d_XI, sizeof(int) * NXI * N));
  }
  if (heading) {
    HIP_OK(hipMalloc(&h->hp.HD, sizeof(double) * NHD * N));
    HIP_OK(hipMalloc(&h->hp.HF, sizeof(float) * NHF * N));
    HIP_OK(hipMalloc(&h->hp.HI, sizeof(int) * N));
    HIP_OK(hipMalloc(&h->hp.HR, sizeof(unsigned long long) * 4 * N));
    HIP_OK(hipMemset(h->hp.HD, 0, sizeof(double) * NHD * N));
    HIP_OK(hipMemset(h->hp.HF, 0, sizeof(float) * NHF * N));
    HIP_OK(hipMemset(h->hp.HI, 0, sizeof(int) * N));
    // stand-in seeding until ac_seed_envs supplies numpy's PCG64 states: splitmix64 of (seed, env), increment forced odd
    std::vector<unsigned long long> r(4 * N);
    unsigned long long z = seed;
    auto sm = [&z]() { z += 0x9E3779B97F4A7C15ULL; unsigned long long v = z; v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ULL; v = (v ^ (v >> 27)) * 0x94D049BB133111EBULL; return v ^ (v >> 31); };
    for (size_t e = 0; e < N; ++e) { r[0 * N + e] = sm(); r[1 * N + e] = sm(); r[2 * N + e] = sm(); r[3 * N + e] = sm() | 1ULL; }
    HIP_OK(hipMemcpy(h->hp.HR, r.data(), sizeof(unsigned long long) * 4 * N, hipMemcpyHostToDevice));
    h->hc.ic = cfg->init[0];
    h->hc.max_heading_increment = cfg->max_heading_increment; h->hc.max_altitude_increment = cfg->max_altitude_increment;
    h->hc.max_velocities_u_increment = cfg->max_velocities_u_increment; h->hc.check_interval = cfg->check_interval;
    h->hc.heading_scale = (float)cfg->heading_scale; h->hc.heading_pot = cfg->heading_potential;
    h->hc.approach = cfg->approach ? 1 : 0;
  }
  p.H = nullptr; p.man_step = nullptr; p.man_h0 = nullptr;
  if (cfg->hierarchical) {
    HIP_OK(hipMalloc(&p.H, sizeof(float) * 128 * N));
    HIP_OK(hipMemset(p.H, 0, sizeof(float) * 128 * N));
    HIP_OK(hipMalloc(&h->d_low, sizeof(float) * N * h->act_low));
    HIP_OK(hipMemset(h->d_low, 0, sizeof(float) * N * h->act_low));
    HIP_OK(hipMalloc(&p.man_step, sizeof(int) * N));
    HIP_OK(hipMemset(p.man_step, 0, sizeof(int) * N));
    HIP_OK(hipMalloc(&p.man_h0, sizeof(float) * N));
    HIP_OK(hipMemset(p.man_h0, 0, sizeof(float) * N));
  }
  p.tF = h->d_tF; p.tI = h->d_tI; p.tD = h->d_tD; p.tab = h->d_tab; p.actions = h->d_actions;
  InitArgs ia;
  for (int i = 0; i < AC_MAX_AGENTS; ++i) ia.ic[i] = cfg->init[i];
  if (heading) {
    // no reset template: every reset runs the initial-condition procedure on fresh draws inside the kernel
  } else if (scenario && h->A == 2)
    hipLaunchKernelGGL(init_kernel_scenario<2>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else if (scenario && h->A == 4)
    hipLaunchKernelGGL(init_kernel_scenario<4>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else if (scenario)
    hipLaunchKernelGGL(init_kernel_scenario<8>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else if (cfg->task == AC_TASK_MULTICOMBAT && h->A == 4)
    hipLaunchKernelGGL(init_kernel_nvn<4>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else if (cfg->task == AC_TASK_MULTICOMBAT)
    hipLaunchKernelGGL(init_kernel_nvn<8>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else if (cfg->task == AC_TASK_SINGLECOMBAT)
    hipLaunchKernelGGL(init_kernel_1v1<AC_TASK_SINGLECOMBAT>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  else
    hipLaunchKernelGGL(init_kernel_1v1<AC_TASK_SHOOT_MISSILE>, dim3(1), dim3(64), 0, h->stream, ia, h->dc, h->d_tab, h->d_tF, h->d_tI, h->d_tD);
  HIP_OK(hipGetLastError());
  if (launch_reset(h)) return -1;
  HIP_OK(hipStreamSynchronize(h->stream));
  *out = h;
  return 0;
}

int ac_destroy(ac_env_t* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  void* bufs[] = {h->dp.F, h->dp.D, h->dp.MF, h->dp.MD, h->dp.MI, h->dp.obs, h->dp.rew, h->dp.done, h->dp.info,
                  h->d_actions, h->d_tab, h->d_tF, h->d_tD, h->d_state_io, h->d_XF, h->d_XI, h->dp.H, h->dp.man_step, h->dp.man_h0, h->d_ctlWs8, h->d_low, h->hp.HD, h->hp.HF, h->hp.HI, h->hp.HR};
  for (void* b : bufs) (void)hipFree(b);
  for (int k = 0; k < AC_HOST_SETS; ++k)
    if (h->have_hs[k]) ac_host_set_free(h->hs[k].act, h->hs[k].obs, h->hs[k].rew, h->hs[k].done, h->hs[k].info);   // (a detached set is the caller's)
  if (h->err_host) (void)hipHostFree(h->err_host);
  if (h->count_host) (void)hipHostFree(h->count_host);
  (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1); (void)hipEventDestroy(h->ev_order); (void)hipEventDestroy(h->ev_mid);
  (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}
int ac_obs_dim(const ac_env_t* h) { return h ? h->obs_dim : -1; }
int ac_act_dim(const ac_env_t* h) { return h ? h->act_dim : -1; }
int ac_num_envs(const ac_env_t* h) { return h ? h->E : -1; }
int ac_num_agents(const ac_env_t* h) { return h ? h->A : -1; }

int ac_reset(ac_env_t* h, float* obs) {
  if (!h) return fail("ac_reset: null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));    // (a step still in flight may yet write the error word)
  *h->err_host = 0; h->err_sticky = 0;
  if (launch_reset(h)) return -1;
  if (obs) HIP_OK(hipMemcpyAsync(obs, h->dp.obs, sizeof(float) * (size_t)h->N * h->obs_dim, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

int ac_step(ac_env_t* h, const float* actions, float* obs, float* rewards, uint8_t* dones, int32_t* info) {
  if (!h || !actions) return fail("ac_step: null argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t N = (size_t)h->N;
  HIP_OK(hipMemcpyAsync(h->d_actions, actions, sizeof(float) * N * h->act_dim, hipMemcpyHostToDevice, h->stream));
  if (launch_step(h, nullptr)) return -1;
  if (obs) HIP_OK(hipMemcpyAsync(obs, h->dp.obs, sizeof(float) * N * h->obs_dim, hipMemcpyDeviceToHost, h->stream));
  if (rewards) HIP_OK(hipMemcpyAsync(rewards, h->dp.rew, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
  if (dones) HIP_OK(hipMemcpyAsync(dones, h->dp.done, N, hipMemcpyDeviceToHost, h->stream));
  if (info) HIP_OK(hipMemcpyAsync(info, h->dp.info, sizeof(int) * 4 * h->E, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return check_nonfinite(h, "ac_step");
}

// ---- zero-copy host boundary. The step kernel reads the actions straight from pinned host memory mapped into the device and
// writes a second copy of its outputs there (whole 16-byte vectors, emit_outputs): no H2D / D2H copy commands, no staging, one
// launch and one completion wait per step. Two buffer sets, so the arrays of step t stay untouched while step t+1 runs.
int ac_host_buffers(ac_env_t* h, int32_t set, float** actions, float** obs, float** rewards, uint8_t** dones, int32_t** info) {
  if (!h || set < 0 || set >= AC_HOST_SETS) return fail("ac_host_buffers: bad argument (sets 0 .. AC_HOST_SETS - 1)");
  HIP_OK(hipSetDevice(h->device));
  ac_env::HostSet& hs = h->hs[set];
  if (!h->have_hs[set]) {
    const size_t Npad = ((size_t)h->N + 63) / 64 * 64;
    // hipHostMallocDefault: page-locked, mapped into the device's address space, coherent (kernel stores are visible to the host
    // once the kernel has completed)
    HIP_OK(hipHostMalloc((void**)&hs.act, sizeof(float) * Npad * h->act_dim, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&hs.obs, sizeof(float) * Npad * h->obs_dim, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&hs.rew, sizeof(float) * Npad, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&hs.done, Npad, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&hs.info, sizeof(int) * (Npad / h->A), hipHostMallocDefault));
    memset(hs.act, 0, sizeof(float) * Npad * h->act_dim); memset(hs.obs, 0, sizeof(float) * Npad * h->obs_dim);
    memset(hs.rew, 0, sizeof(float) * Npad); memset(hs.done, 0, Npad); memset(hs.info, 0, sizeof(int) * (Npad / h->A));
    h->have_hs[set] = true;
  }
  if (actions) *actions = hs.act;
  if (obs) *obs = hs.obs;
  if (rewards) *rewards = hs.rew;
  if (dones) *dones = hs.done;
  if (info) *info = hs.info;
  return 0;
}
// A set whose arrays the caller still holds when the VecEnv closes (the reference's step() returns arrays the caller owns for good,
// env_wrappers.py:276-282): the handle gives the set up -- ac_destroy leaves it alone -- and the holder frees it when the last array is gone.
int ac_host_set_detach(ac_env_t* h, int32_t set) {
  if (!h || set < 0 || set >= AC_HOST_SETS || !h->have_hs[set]) return fail("ac_host_set_detach: no such set");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));     // no step may still be writing into it
  h->have_hs[set] = false;
  memset(&h->hs[set], 0, sizeof h->hs[set]);
  return 0;
}
void ac_host_set_free(void* actions, void* obs, void* rewards, void* dones, void* info) {
  (void)hipHostFree(actions); (void)hipHostFree(obs); (void)hipHostFree(rewards); (void)hipHostFree(dones); (void)hipHostFree(info);
}
int ac_step_host_async(ac_env_t* h, int32_t set) {
  if (!h || set < 0 || set >= AC_HOST_SETS) return fail("ac_step_host_async: bad argument");
  if (!h->have_hs[set]) return fail("ac_step_host_async: call ac_host_buffers for this set first");
  HIP_OK(hipSetDevice(h->device));
  return launch_step(h, nullptr, set);
}
int ac_step_host_wait(ac_env_t* h) {
  if (!h) return fail("ac_step_host_wait: null handle");
  // (hipStreamSynchronize spins on the completion signal itself; polling hipStreamQuery measured 5 us slower per step,
  //  tools/micro/host_io.hip)
  HIP_OK(hipStreamSynchronize(h->stream));
  return check_nonfinite(h, "ac_step_host_wait");
}
int ac_step_host(ac_env_t* h, int32_t set) {   // step_async + step_wait in one call (VecEnv.step, env_wrappers.py:30-42)
  if (ac_step_host_async(h, set)) return -1;
  return ac_step_host_wait(h);
}

// ---- stream ordering for the device-resident path: the handle's stream is non-blocking, so work on the caller's streams (a torch
// policy writing the actions, a buffer reading the observations) is not ordered against the step kernel unless asked for.
int ac_order_after(ac_env_t* h, void* producer_stream) {   // the next step waits for everything queued on producer_stream so far
  if (!h) return fail("ac_order_after: null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipEventRecord(h->ev_order, (hipStream_t)producer_stream));
  HIP_OK(hipStreamWaitEvent(h->stream, h->ev_order, 0));
  return 0;
}
int ac_order_before(ac_env_t* h, void* consumer_stream) {  // consumer_stream waits for every step queued so far
  if (!h) return fail("ac_order_before: null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipEventRecord(h->ev_order, h->stream));
  HIP_OK(hipStreamWaitEvent((hipStream_t)consumer_stream, h->ev_order, 0));
  return 0;
}

int ac_step_async_device(ac_env_t* h, const float* d_actions) {
  if (!h) return fail("ac_step_async_device: null handle");
  HIP_OK(hipSetDevice(h->device));
  return launch_step(h, d_actions);
}
int ac_device_buffers(ac_env_t* h, float** d_actions, float** d_obs, float** d_rewards, uint8_t** d_dones, int32_t** d_info) {
  if (!h) return fail("ac_device_buffers: null handle");
  if (d_actions) *d_actions = h->d_actions;
  if (d_obs) *d_obs = h->dp.obs;
  if (d_rewards) *d_rewards = h->dp.rew;
  if (d_dones) *d_dones = h->dp.done;
  if (d_info) *d_info = h->dp.info;
  return 0;
}
void* ac_stream(ac_env_t* h) { return h ? (void*)h->stream : nullptr; }
int ac_sync(ac_env_t* h) {
  if (!h) return fail("ac_sync: null handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  return check_nonfinite(h, "ac_sync");
}
int ac_timing_begin(ac_env_t* h) {
  if (!h) return fail("ac_timing_begin: null handle");
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  return 0;
}
int ac_timing_end(ac_env_t* h, float* total_ms) {
  if (!h || !total_ms) return fail("ac_timing_end: null argument");
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  HIP_OK(hipEventSynchronize(h->ev1));
  HIP_OK(hipEventElapsedTime(total_ms, h->ev0, h->ev1));
  return 0;
}

int ac_step_timed_device(ac_env_t* h, const float* d_actions, float* controller_ms, float* step_ms) {
  if (!h || !controller_ms || !step_ms) return fail("ac_step_timed_device: null argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipEventRecord(h->ev0, h->stream));
  h->mark_mid = h->cfg.hierarchical != 0;
  const int rc = launch_step(h, d_actions);
  h->mark_mid = false;
  if (rc) return rc;
  HIP_OK(hipEventRecord(h->ev1, h->stream));
  HIP_OK(hipEventSynchronize(h->ev1));
  *controller_ms = 0.0f;
  if (h->cfg.hierarchical) {
    HIP_OK(hipEventElapsedTime(controller_ms, h->ev0, h->ev_mid));
    HIP_OK(hipEventElapsedTime(step_ms, h->ev_mid, h->ev1));
  } else HIP_OK(hipEventElapsedTime(step_ms, h->ev0, h->ev1));
  return check_nonfinite(h, "ac_step_timed_device");
}

// state vector layout: rx ry rz | float fields | task floats | eng ticks | task ints   (names: ac_state_field_name)
// where a word of the external vector lives in the device's group storage (SW_*), and the fp64 position (a pair + rz)
static float* storage_word(float* S, int w, size_t N, size_t n) { return S + (((size_t)(w >> 2) * N + n) * 4 + (size_t)(w & 3)); }
static double* position_word(double* D, int f, size_t N, size_t n) { return f < 2 ? D + 2 * n + f : D + 2 * N + n; }
static int check_idx(ac_env_t* h, int env, int agent) {
  if (!h) return fail("null handle");
  if (env < 0 || env >= h->E || agent < 0 || agent >= h->A) return fail("env/agent index out of range");
  return 0;
}
// One aircraft's record in one launch and one copy (the tests' teacher-forcing reads and writes it for every aircraft of every step: as
// one small hipMemcpy per word it was most of the GPU suite's run time). io: [ND position doubles][NF floats][NI ints][NXI raw words][NXF floats].
struct SlotMap { int f[NF]; int i[NI]; };
constexpr int kStateIoWords = ND + NF + NI + NXI + NXF;
static_assert(kStateIoWords <= 256, "state_io_kernel: one thread per word of a 256-thread block");
__global__ void state_io_kernel(float* F, double* D, const float* XF, const int* XI, size_t N, size_t n, double* io, int write, SlotMap m) {
  const int t = threadIdx.x;
  if (t < ND) {
    double* w = t < 2 ? D + 2 * n + t : D + 2 * N + n;
    if (write) *w = io[t]; else io[t] = *w;
  } else if (t < ND + NF) {
    const int sw = m.f[t - ND];
    float* w = F + (((size_t)(sw >> 2) * N + n) * 4 + (size_t)(sw & 3));
    if (write) *w = (float)io[t]; else io[t] = (double)*w;
  } else if (t < ND + NF + NI) {
    const int sw = m.i[t - ND - NF];
    int* w = reinterpret_cast<int*>(F + (((size_t)(sw >> 2) * N + n) * 4 + (size_t)(sw & 3)));
    if (write) *w = (int)llround(io[t]); else io[t] = (double)*w;
  } else if (!write && t < kStateIoWords) {
    const int q = t - (ND + NF + NI);
    if (q < NXI) io[t] = XI ? (double)XI[(size_t)q * N + n] : 0.0;
    else io[t] = XF ? (double)XF[(size_t)(q - NXI) * N + n] : 0.0;
  }
}
static SlotMap slot_map() {
  SlotMap m;
  for (int f = 0; f < NF; ++f) m.f[f] = kSlotF[f];
  for (int f = 0; f < NI; ++f) m.i[f] = kSlotI[f];
  return m;
}
int ac_get_state(ac_env_t* h, int32_t env, int32_t agent, double* out) {
  if (check_idx(h, env, agent) || !out) return fail("ac_get_state: bad argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  double io[kStateIoWords];
  hipLaunchKernelGGL(state_io_kernel, dim3(1), dim3(256), 0, h->stream, h->dp.F, h->dp.D, h->d_XF, h->d_XI, N, n, h->d_state_io, 0, slot_map());
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(io, h->d_state_io, sizeof io, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  int k = 0;
  for (int f = 0; f < ND + NF + NI; ++f) out[k++] = io[f];
  if (h->d_XF) {  // read-only tail: scenario-task extension (weapon counters, chaff clouds, shared reward references)
    {   // the fourteen counters / flags live in two packed words (scenario_kernel.hpp: ext_pack0 / ext_pack1); reported one by one
      const double* w = io + ND + NF + NI;
      Ext x{};
      ext_unpack((int)w[XI_w0], (int)w[XI_w1], x);
      const int v[NXI_UNPACKED] = {x.rem_gun, x.rem_9m, x.rem_120b, x.rem_chaff, x.bits, x.last_chaff, x.orphan_hits, x.mp_prev, x.ref_set,
                                   x.ch_status[0], x.ch_mult[0], x.ch_status[1], x.ch_mult[1], x.n_ch};
      for (int f = 0; f < NXI_UNPACKED; ++f) out[k++] = v[f];
    }
    for (int f = 0; f < NXF; ++f) out[k++] = io[ND + NF + NI + NXI + f];
  }
  for (; k < AC_STATE_LEN; ++k) out[k] = 0.0;
  return 0;
}
int ac_set_state(ac_env_t* h, int32_t env, int32_t agent, const double* in) {
  if (check_idx(h, env, agent) || !in) return fail("ac_set_state: bad argument");
  HIP_OK(hipSetDevice(h->device));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  HIP_OK(hipMemcpyAsync(h->d_state_io, in, sizeof(double) * (ND + NF + NI), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(state_io_kernel, dim3(1), dim3(256), 0, h->stream, h->dp.F, h->dp.D, h->d_XF, h->d_XI, N, n, h->d_state_io, 1, slot_map());
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}
int ac_set_status(ac_env_t* h, int32_t env, int32_t agent, int32_t status) {
  if (check_idx(h, env, agent)) return -1;
  if (status < AC_ALIVE || status > AC_SHOTDOWN) return fail("ac_set_status: bad status");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  int v = status;
  HIP_OK(hipMemcpy(storage_word(h->dp.F, SW_status, N, n), &v, sizeof v, hipMemcpyHostToDevice));
  return 0;
}

__global__ void entity_kernel(DevPtrs P, DevCfg c, int n, double* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  State s; Task t; Derived d; Props pr;
  load_state(P.F, P.I, P.D, c.N, n, s, t);
  f16::locate(s, d); f16::body_frame(s, d);
  make_props(s, d, c, pr);
  const double R2D = 180.0 / 3.14159265358979323846;
  out[0] = atan2(d.sLon64, d.cLon64) * R2D; out[1] = atan2(d.sLat64, d.cLat64) * R2D; out[2] = pr.alt_m;
  out[3] = atan2f(pr.sphi, pr.cphi); out[4] = asinf(pr.stht);
  float psi = atan2f(pr.m12, pr.m11); if (psi < 0.0f) psi += 2.0f * f16::kPi;
  out[5] = psi; out[6] = pr.vn; out[7] = pr.ve; out[8] = pr.vd; out[9] = pr.n; out[10] = pr.e; out[11] = pr.u;
}
int ac_get_entity(ac_env_t* h, int32_t env, int32_t agent, double out[12]) {
  if (check_idx(h, env, agent) || !out) return fail("ac_get_entity: bad argument");
  HIP_OK(hipSetDevice(h->device));
  double* d_out = h->d_state_io;   // (the handle's scratch record: no allocation per call)
  hipLaunchKernelGGL(entity_kernel, dim3(1), dim3(64), 0, h->stream, h->dp, h->dc, env * h->A + agent, d_out);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(out, d_out, sizeof(double) * 12, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}
// Order-independent 64-bit digest of the live aircraft state: every lane reads its SoA columns exactly like the step kernel
// (4 B per lane, coalesced; 8 B for the ECI position), mixes each word with its field index and the lanes' digests are summed.
__global__ void state_checksum_kernel(DevPtrs P, DevCfg c, unsigned long long* out) {
  const int N = c.N;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  if (n < N) {
    auto mix = [](unsigned long long v, unsigned f) {
      unsigned long long z = v + 0x9E3779B97F4A7C15ULL * (unsigned long long)(f + 1);
      z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
      return z;
    };
    for (int g = 0; g < NSG; ++g) {   // the step kernels' own access pattern: one 16-byte group per lane and instruction
      const float4 v = reinterpret_cast<const float4*>(P.F)[(size_t)g * N + n];
      acc += mix((unsigned long long)__float_as_uint(v.x), 4 * g) + mix((unsigned long long)__float_as_uint(v.y), 4 * g + 1) +
             mix((unsigned long long)__float_as_uint(v.z), 4 * g + 2) + mix((unsigned long long)__float_as_uint(v.w), 4 * g + 3);
    }
    {
      const double2 xy = reinterpret_cast<const double2*>(P.D)[n];
      acc += mix((unsigned long long)__double_as_longlong(xy.x), NSW) + mix((unsigned long long)__double_as_longlong(xy.y), NSW + 1) +
             mix((unsigned long long)__double_as_longlong(P.D[2 * (size_t)N + n]), NSW + 2);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
int ac_state_checksum(ac_env_t* h, uint64_t* out) {
  if (!h || !out) return fail("ac_state_checksum: null argument");
  HIP_OK(hipSetDevice(h->device));
  unsigned long long* d_out;
  HIP_OK(hipMalloc(&d_out, sizeof(unsigned long long)));
  HIP_OK(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), h->stream));
  dim3 block(64), grid((h->N + 63) / 64);
  hipLaunchKernelGGL(state_checksum_kernel, grid, block, 0, h->stream, h->dp, h->dc, d_out);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  unsigned long long v = 0;
  HIP_OK(hipMemcpy(&v, d_out, sizeof v, hipMemcpyDeviceToHost));
  HIP_OK(hipFree(d_out));
  *out = (uint64_t)v;
  return 0;
}
// Munitions in flight (status LAUNCHED) over the whole handle: SURVEY 8(d) counts 192 algorithmic bytes per live missile-step, so the bench needs
// the number; one status word per slot, summed per wave.
__global__ void munitions_in_flight_kernel(DevPtrs P, DevCfg c, int* out) {
  const int N = c.N;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  int cnt = 0;
  if (n < N)
    for (int k = 0; k < c.msl_slots; ++k) cnt += P.MI[((size_t)k * NMI + MI_status) * (size_t)N + n] == MSL_LAUNCHED;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) __hip_atomic_fetch_add(out, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (`out` is page-locked host memory)
}
int ac_munitions_in_flight(ac_env_t* h, int32_t* count) {
  if (!h || !count) return fail("ac_munitions_in_flight: null argument");
  *count = 0;
  if (!h->dc.msl_slots || !h->dp.MI) return 0;
  HIP_OK(hipSetDevice(h->device));
  // the counter lives with the handle (no allocation, no device-wide synchronisation in a call): cleared and added to on the handle's
  // stream, read after one wait for that stream
  HIP_OK(hipStreamSynchronize(h->stream));
  *(volatile int*)h->count_host = 0;
  hipLaunchKernelGGL(munitions_in_flight_kernel, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->dp, h->dc, h->count_host);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(h->stream));
  *count = *(volatile int*)h->count_host;
  return 0;
}
int ac_seed_envs(ac_env_t* h, const uint64_t* states) {
  if (!h || !states) return fail("ac_seed_envs: null argument");
  if (h->cfg.task != AC_TASK_HEADING) return fail("ac_seed_envs: only AC_TASK_HEADING draws from env.np_random");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N;
  std::vector<unsigned long long> r(4 * N);
  for (size_t e = 0; e < N; ++e)
    for (int k = 0; k < 4; ++k) r[k * N + e] = states[e * 4 + k];
  HIP_OK(hipMemcpy(h->hp.HR, r.data(), sizeof(unsigned long long) * 4 * N, hipMemcpyHostToDevice));
  return 0;
}
int ac_get_heading_state(ac_env_t* h, int32_t env, double out[8]) {
  if (check_idx(h, env, 0) || !out) return fail("ac_get_heading_state: bad argument");
  if (h->cfg.task != AC_TASK_HEADING) return fail("ac_get_heading_state: not an AC_TASK_HEADING handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N, n = env;
  for (int f = 0; f < NHD; ++f) HIP_OK(hipMemcpy(&out[f], h->hp.HD + f * N + n, sizeof(double), hipMemcpyDeviceToHost));
  int tc; HIP_OK(hipMemcpy(&tc, h->hp.HI + n, sizeof tc, hipMemcpyDeviceToHost));
  out[5] = tc;
  float lp, lq;
  HIP_OK(hipMemcpy(&lp, h->hp.HF + HF_last_p * N + n, sizeof lp, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&lq, h->hp.HF + HF_last_q * N + n, sizeof lq, hipMemcpyDeviceToHost));
  out[6] = lp; out[7] = lq;
  return 0;
}
int ac_pin_host_buffer(ac_env_t* h, void* ptr, int64_t bytes) {
  if (!h || !ptr || bytes <= 0) return fail("ac_pin_host_buffer: bad argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
  return 0;
}
int ac_unpin_host_buffer(ac_env_t* h, void* ptr) {
  if (!h || !ptr) return fail("ac_unpin_host_buffer: bad argument");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipHostUnregister(ptr));
  return 0;
}
int ac_load_controller(ac_env_t* h, const float* weights, int64_t n) {
  using namespace ctl;
  if (!h || !weights) return fail("ac_load_controller: null argument");
  if (!h->cfg.hierarchical) return fail("ac_load_controller: the handle was not created with cfg.hierarchical");
  if (n != S_END) return fail("ac_load_controller: expected 137753 floats (layout of tools/export_baseline_actor.py)");
  HIP_OK(hipSetDevice(h->device));
  {   // the weights as fp16 pieces in the kernel's tiling: tile(c) = the 16 columns 16 c .. 16 c + 15 = K/32 k-steps x 2 pieces x 64 lanes x 8 values,
      // element (s, p, lane, i) = piece p of W[j = 16 c + lane % 16][k = 32 s + 8 (lane / 16) + i]
    using namespace ctl8;
    std::vector<float> e(C_END, 0.0f);
    unsigned short* e16 = reinterpret_cast<unsigned short*>(e.data());
    auto tiles_s = [&](int src, int dst, int J, int K, int Kpad, int ntiles) {
      for (int c = 0; c < ntiles; ++c)
        for (int st = 0; st < Kpad / 32; ++st)
          for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 8; ++i) {
              const int k = 32 * st + 8 * (lane / 16) + i, j = 16 * c + lane % 16;
              unsigned pc[NP];
              ctls::split2((j < J && k < K) ? weights[src + j * K + k] : 0.0f, pc[0], pc[1]);
              for (int pp = 0; pp < NP; ++pp)
                e16[(size_t)dst * 2 + ((((size_t)c * (Kpad / 32) + st) * NP + pp) * 64 + lane) * 8 + i] = (unsigned short)pc[pp];
            }
    };
    auto copy_s = [&](int src, int dst, int cnt) { for (int i = 0; i < cnt; ++i) e[dst + i] = weights[src + i]; };
    tiles_s(S_W1, C_W1, 128, 12, 32, 8); copy_s(S_B1, C_B1, 128); copy_s(S_G1, C_G1, 128); copy_s(S_BE1, C_BE1, 128);
    tiles_s(S_W2, C_W2, 128, 128, 128, 8); copy_s(S_B2, C_B2, 128); copy_s(S_G2, C_G2, 128); copy_s(S_BE2, C_BE2, 128);
    // (W_ih / W_hh rows are gate-major: r 0..127, z 128..255, n 256..383, so tile 8 g + u is columns 128 g + 16 u .. + 15 = tile index c of the stacked matrix)
    tiles_s(S_WIH, C_WIH, 384, 128, 128, 24); tiles_s(S_WHH, C_WHH, 384, 128, 128, 24); copy_s(S_BIH, C_BIH, 384); copy_s(S_BHH, C_BHH, 384);
    copy_s(S_G3, C_G3, 128); copy_s(S_BE3, C_BE3, 128);
    tiles_s(S_WA, C_WA, NH, 128, 128, 10); copy_s(S_BA, C_BA, NH);
    if (!h->d_ctlWs8) HIP_OK(hipMalloc(&h->d_ctlWs8, sizeof(float) * C_END));
    HIP_OK(hipMemcpy(h->d_ctlWs8, e.data(), sizeof(float) * C_END, hipMemcpyHostToDevice));
  }
  return 0;
}
int ac_split_f16x2(const float* x, int64_t n, float* hi, float* lo) {
  if (!x || !hi || !lo || n < 0) return fail("ac_split_f16x2: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    unsigned p[2];
    ctls::split2(x[i], p[0], p[1]);
    float* out[2] = {hi, lo};
    for (int k = 0; k < 2; ++k) { const unsigned short b = (unsigned short)p[k]; _Float16 v; memcpy(&v, &b, 2); out[k][i] = (float)v; }
  }
  return 0;
}
int ac_selftest_missile_walk(int32_t device_id, int32_t* mismatches) {
  if (!mismatches) return fail("ac_selftest_missile_walk: null argument");
  HIP_OK(hipSetDevice(device_id));
  int* d = nullptr;
  HIP_OK(hipMalloc(&d, 2 * sizeof(int)));
  HIP_OK(hipMemset(d, 0, 2 * sizeof(int)));
  const unsigned long long c4 = 2ull << (2 * 4), c8 = 2ull << (2 * 8);   // 4^A agent-state combinations x (carried-in missile or not)
  hipLaunchKernelGGL(mp_walk_selftest_kernel<4>, dim3((unsigned)((c4 + 15) / 16)), dim3(64), 0, 0, d, c4);
  hipLaunchKernelGGL(mp_walk_selftest_kernel<8>, dim3((unsigned)((c8 + 7) / 8)), dim3(64), 0, 0, d, c8);
  HIP_OK(hipGetLastError());
  int out[2] = {-1, 0};
  HIP_OK(hipMemcpy(out, d, sizeof out, hipMemcpyDeviceToHost));
  HIP_OK(hipFree(d));
  if (out[1] < 100000) return fail("ac_selftest_missile_walk: the sweep did not reach the cases it is there for");   // (an agent scoring against another agent's missile)
  *mismatches = out[0];
  return 0;
}
int ac_get_controller_state(ac_env_t* h, int32_t env, int32_t agent, float* hidden, float* low_action) {
  if (check_idx(h, env, agent)) return fail("ac_get_controller_state: bad argument");
  if (!h->cfg.hierarchical) return fail("ac_get_controller_state: not a hierarchical handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  if (hidden) HIP_OK(hipMemcpy2D(hidden, sizeof(float), h->dp.H + n, sizeof(float) * N, sizeof(float), 128, hipMemcpyDeviceToHost));
  if (low_action) HIP_OK(hipMemcpy(low_action, h->d_low + n * h->act_low, sizeof(float) * h->act_low, hipMemcpyDeviceToHost));
  return 0;
}
int ac_set_controller_state(ac_env_t* h, int32_t env, int32_t agent, const float* hidden) {
  if (check_idx(h, env, agent) || !hidden) return fail("ac_set_controller_state: bad argument");
  if (!h->cfg.hierarchical) return fail("ac_set_controller_state: not a hierarchical handle");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  HIP_OK(hipMemcpy2D(h->dp.H + n, sizeof(float) * N, hidden, sizeof(float), sizeof(float), 128, hipMemcpyHostToDevice));
  return 0;
}
int ac_get_missile(ac_env_t* h, int32_t env, int32_t agent, int32_t k, double out[12]) {
  if (check_idx(h, env, agent) || !out) return fail("ac_get_missile: bad argument");
  if (k < 0 || k >= h->dc.msl_slots) return fail("ac_get_missile: no such missile slot");
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipStreamSynchronize(h->stream));
  const size_t N = h->N, n = (size_t)env * h->A + agent;
  int st;
  HIP_OK(hipMemcpy(&st, h->dp.MI + ((size_t)k * NMI + MI_status) * N + n, sizeof st, hipMemcpyDeviceToHost));
  out[0] = st;
  static const int order[10] = {MF_px, MF_py, MF_pz, MF_vx, MF_vy, MF_vz, MF_theta, MF_psi, MF_t, MF_m};
  for (int i = 0; i < 10; ++i) {
    if (h->dp.MD) {
      HIP_OK(hipMemcpy(&out[1 + i], h->dp.MD + ((size_t)k * NMF + order[i]) * N + n, sizeof(double), hipMemcpyDeviceToHost));
    } else {
      float v;
      HIP_OK(hipMemcpy(&v, h->dp.MF + ((size_t)k * NMF + order[i]) * N + n, sizeof v, hipMemcpyDeviceToHost));
      out[1 + i] = v;
    }
  }
  {   // which munition the slot holds (ACMI name): 0 AIM-9L (the 1v1 missile tasks), 1 AIM-120B, 2 AIM-9M
    int rw;
    HIP_OK(hipMemcpy(&rw, h->dp.MI + ((size_t)k * NMI + MI_recede) * N + n, sizeof rw, hipMemcpyDeviceToHost));
    out[11] = (h->dp.MD && h->cfg.task != AC_TASK_DODGE_MISSILE) ? (double)(1 + ((rw >> 9) & 1)) : 0.0;
  }
  return 0;
}

}  // extern "C"

#include "rollout_buffer.hpp"
