// The three-wave form of the step kernels (included by aircombat.hip after the state layout and before the kernels).
//
// Below one wave per SIMD the step time is the length of ONE lane's instruction stream (a lone wave64 issues one instruction per
// ~4 cycles and the ticks are a dependent chain) while most SIMDs idle, so the FDM tick is cut by function over three waves that
// work on the same 64 aircraft: wave 0 "dynamics" (the critical path and the whole environment layer), wave 1 "systems" (flight
// control system, turbine and fuel, mass balance) and wave 2 "kinematics" (attitude / position / geodetic frame / gravity /
// atmosphere of the NEXT tick, which depend on this tick's rates and velocities only, plus a share of the table look-ups).
// f16_split.hpp holds the statements of tick() cut into those pieces (generated, so every kernel form runs the same arithmetic);
// this file holds the LDS mailbox and the per-wave drivers.
#pragma once

// SPLIT (tasks without munitions, small grids): a workgroup is THREE waves over the same 64 aircraft. Wave 0 ("dynamics") runs the rigid-body
// part of every tick and the whole environment layer; wave 1 ("systems") runs the flight control system and the turbine of the same
// aircraft on another SIMD, concurrently with wave 0's atmosphere / mass / auxiliary / table look-up work. The two exchange ~25 floats
// per aircraft and tick through LDS, with three workgroup barriers per tick (f16_split.hpp holds the statements of tick(), cut into
// those pieces). A lone wave issues one dependent instruction every ~4.3 cycles, so below one wave per SIMD this shortens the tick's
// critical path instead of competing for issue slots.
namespace mail {  // LDS mailbox rows (64 floats each)
enum { RUNF,                                                       // dynamics -> both helpers before B1: does this aircraft run this tick
       CTH, VB,                                                    // dynamics -> systems after part 1
       MACH, QBAR, RHO, TEMP, HSL, ALPHA, BETA, QC, VG, NPY, NPZ, AP, AQ, AR,   // dynamics -> systems after part 2
       S_AIL, S_FLAP, S_ELEV, S_RUD, S_LEF, S_SB,                   // systems -> dynamics after the FCS
       THRUST,                                                      // systems -> dynamics after the turbine
       LK_CLB, LK_CNB, LK_G7,                                       // kinematics -> dynamics: its share of the table look-ups
       MASS0 = LK_G7 + 4, MASS_N = 18,                              // systems -> dynamics: mass properties of the coming tick
       F_TEF = MASS0 + MASS_N, F_PINR, F_PINP, F_PINY, F_PIR, F_PIP, F_PIY, F_AIL, F_ELEV, F_SBDEG, F_N1, F_N2, F_N2NORM, F_FF, F_TANK0, F_TANK1,
       F_ENG, F_DA, F_DE, F_DR, F_THR,                              // final hand-over of the fields the systems wave owns (the decoded commands among them)
       F_BITS,                                                      // ... and, quad form, what else the action row holds: the shoot bit / the four weapon bits, packed (environment wave)
       K_W, K_V = K_W + 3,                                          // dynamics -> kinematics after part 1: body rates, ECI velocity
       K_OUT = K_V + 3, K_OUT_N = 27,                               // kinematics -> dynamics: f16::KinOut of the coming tick
       G_Q = K_OUT + K_OUT_N,                                       // kinematics -> dynamics: quaternion of tick k in rows G_Q + 4 (k & 1) ..
       G_H = G_Q + 8, G_NED = G_H + 1,                              // ... and, from the step's last substep, the env-layer frame
       QP_F = G_NED + 8, QP_F_N = 9,                                // quad form with reduced poses: altitude [m] and the local frame of tick k in rows QP_F + 9 (k & 1) ..
       T_PR = QP_F + 2 * QP_F_N, T_PR_N = 15,                       // three-wave SingleCombat: dynamics -> kinematics after the last tick, the pose
       T_DONE = T_PR + T_PR_N,                                      //   the observation is built from, and whether the env ends its episode
       T_RALT = T_DONE + 1, T_RPOS,                                 // systems -> dynamics before the second of those barriers: AltitudeReward / PostureReward of the final pose, unscaled
       ROWS = T_RPOS + 1 };
// fp64 rows: ECI position of tick k in rows GD_R + 3 (k & 1) .. (double-buffered: the kinematics wave is one tick ahead), ECEF
// position and geodetic cosines of the step's last substep
// (quad form with reduced poses: the NEU position of tick k in rows GD_QP + 3 (k & 1) ..)
enum { GD_R, GD_X = GD_R + 6, GD_LAT = GD_X + 3, GD_QP = GD_LAT + 4, DROWS = GD_QP + 6 };
}
__device__ __forceinline__ void post_kin(float (*M)[64], int l, const f16::KinOut& o) {
  const float v[mail::K_OUT_N] = {o.T[0], o.T[1], o.T[2], o.T[3], o.T[4], o.T[5], o.T[6], o.T[7], o.T[8], o.h_sl_ft, o.n_eci[0], o.n_eci[1], o.n_eci[2],
                                  o.e_eci[0], o.e_eci[1], o.d_eci[0], o.d_eci[1], o.d_eci[2], o.gx, o.gy, o.gz, o.rxf, o.ryf, o.A.T, o.A.P, o.A.rho, o.A.a};
#pragma unroll
  for (int i = 0; i < mail::K_OUT_N; ++i) M[mail::K_OUT + i][l] = v[i];
}
__device__ __forceinline__ void fetch_kin(float (*M)[64], int l, f16::KinOut& o) {
  float v[mail::K_OUT_N];
#pragma unroll
  for (int i = 0; i < mail::K_OUT_N; ++i) v[i] = M[mail::K_OUT + i][l];
#pragma unroll
  for (int i = 0; i < 9; ++i) o.T[i] = v[i];
  o.h_sl_ft = v[9]; o.n_eci[0] = v[10]; o.n_eci[1] = v[11]; o.n_eci[2] = v[12]; o.e_eci[0] = v[13]; o.e_eci[1] = v[14];
  o.d_eci[0] = v[15]; o.d_eci[1] = v[16]; o.d_eci[2] = v[17]; o.gx = v[18]; o.gy = v[19]; o.gz = v[20]; o.rxf = v[21]; o.ryf = v[22];
  o.A.T = v[23]; o.A.P = v[24]; o.A.rho = v[25]; o.A.a = v[26];
}
// mass, CG, inertia tensor, its cofactors / determinant, 1/mass: sys_mass() on the systems wave -> DynVars of the dynamics wave
__device__ __forceinline__ void post_mass(float (*M)[64], int l, const f16::DynVars& k) {
  const float v[mail::MASS_N] = {k.mass, k.cgx, k.cgy, k.cgz, k.Jxx, k.Jyy, k.Jzz, k.Jxy, k.Jxz, k.Jyz, k.c00, k.c01, k.c02, k.c11, k.c12, k.c22, k.idet, k.im_};
#pragma unroll
  for (int i = 0; i < mail::MASS_N; ++i) M[mail::MASS0 + i][l] = v[i];
}
__device__ __forceinline__ void fetch_mass(float (*M)[64], int l, f16::DynVars& k) {
  float v[mail::MASS_N];
#pragma unroll
  for (int i = 0; i < mail::MASS_N; ++i) v[i] = M[mail::MASS0 + i][l];
  k.mass = v[0]; k.cgx = v[1]; k.cgy = v[2]; k.cgz = v[3]; k.Jxx = v[4]; k.Jyy = v[5]; k.Jzz = v[6]; k.Jxy = v[7]; k.Jxz = v[8]; k.Jyz = v[9];
  k.c00 = v[10]; k.c01 = v[11]; k.c02 = v[12]; k.c11 = v[13]; k.c12 = v[14]; k.c22 = v[15]; k.idet = v[16]; k.im_ = v[17];
}
__device__ __forceinline__ void wg_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// The systems wave of a SPLIT workgroup: FCS and turbine of every substep, then the final values of the fields it owns.
// `raw`: the aircraft's action row still on its way (in a step through ac_step_host it crosses PCIe: about three times the latency of
// the state loads it was issued behind). The commands are this wave's alone to integrate and first needed after B1 of the first tick,
// so they are decoded there and the other two waves start the tick without waiting for them.
// `row`: the same row asked for by ActionRow::issue and possibly still in flight: then the wait for it is HERE, after B1 of the first tick.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct ActionRow {
  f32x4 v;
  // One 16-byte load the compiler does not know about: it cannot put a wait for it (or a copy of its registers) anywhere before take().
  // With a plain load the optimiser hoisted the loop-invariant decode, and the wait with it, in front of B1 -- and a row that comes
  // from mapped host memory (ac_step_host) takes ~5 k cycles across PCIe (profiles/round4_cycle_stamps.txt).
  __device__ __forceinline__ void issue(const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
  __device__ __forceinline__ void take() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); }
};
// the same for one word of the row (the shoot bit behind the four control indices)
struct ActionWord {
  float v;
  __device__ __forceinline__ void issue(const float* p) { asm volatile("global_load_dword %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
  __device__ __forceinline__ void take() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); }
};
// The whole action row as the wave that decodes it sees it: the four control indices, and behind them nothing (act_dim 4), one shoot bit
// (5) or the four weapon bits (8). Only THIS wave reads the row; what the environment wave needs of it comes through LDS (F_BITS).
struct ActionFetch {
  ActionRow idx, bits;
  ActionWord bit;
  int extra;
  __device__ __forceinline__ void issue(const float* act, int act_dim) {
    extra = act_dim == 8 ? 4 : (act_dim == 5 ? 1 : 0);
    idx.issue(act);
    if (extra == 4) bits.issue(act + 4);
    if (extra == 1) bit.issue(act + 4);
  }
  __device__ __forceinline__ void take() { idx.take(); if (extra == 4) bits.take(); if (extra == 1) bit.take(); }
  __device__ __forceinline__ float packed() const {       // bit k set = element 4 + k of the row is non-zero
    if (extra == 4) return (float)((bits.v.x != 0.0f ? 1 : 0) | (bits.v.y != 0.0f ? 2 : 0) | (bits.v.z != 0.0f ? 4 : 0) | (bits.v.w != 0.0f ? 8 : 0));
    if (extra == 1) return bit.v != 0.0f ? 1.0f : 0.0f;
    return 0.0f;
  }
};
__device__ __forceinline__ void systems_wave(f16::State& s, Task& t, const f16::Tab& T, float (*M)[64], int l, int substeps, const float4* raw = nullptr,
                                             ActionFetch* row = nullptr) {
  using namespace mail;
  f16::DynVars km{};
  f16::sys_mass(s, km);
  post_mass(M, l, km);                                     // read by the dynamics wave after B1 of the first tick
  (void)t;
  for (int sub = 0; sub < substeps; ++sub) {
    f16::Surf sf{};
    AC_CLKW(1, 64 + sub * 8);
    wg_sync();                                             // B1: this tick's attitude is known
    if (row && sub == 0) {                                 // normalize_action (singlecombat_task.py:141-153), property bounds catalog.py:189-197
      row->take();
      s.da = f16::clampf(-1.0f, row->idx.v.x / 20.0f - 1.0f, 1.0f);
      s.de = f16::clampf(-1.0f, row->idx.v.y / 20.0f - 1.0f, 1.0f);
      s.dr = f16::clampf(-1.0f, row->idx.v.z / 20.0f - 1.0f, 1.0f);
      s.thr = f16::clampf(0.0f, row->idx.v.w / 58.0f + 0.4f, 0.9f);
    } else if (raw && sub == 0) {
      s.da = f16::clampf(-1.0f, raw->x / 20.0f - 1.0f, 1.0f);
      s.de = f16::clampf(-1.0f, raw->y / 20.0f - 1.0f, 1.0f);
      s.dr = f16::clampf(-1.0f, raw->z / 20.0f - 1.0f, 1.0f);
      s.thr = f16::clampf(0.0f, raw->w / 58.0f + 0.4f, 0.9f);
    }
    const bool run = M[RUNF][l] != 0.0f;                   // the dynamics wave decides who flies (status can change between substeps)
    if (run) {
      f16::sys_fcs(s, M[CTH][l], M[VB][l], sf);
      M[S_AIL][l] = sf.aileron_rad; M[S_FLAP][l] = sf.flaperon_rad; M[S_ELEV][l] = sf.elevator_rad;
      M[S_RUD][l] = sf.rudder_rad;  M[S_LEF][l] = sf.lef_rad;       M[S_SB][l] = sf.sb_rad;
    }
    AC_CLKW(1, 65 + sub * 8);
    wg_sync();                                             // B2: this tick's air data are known
    if (run) {
      f16::Atmos A{};
      A.rho = M[RHO][l]; A.T = M[TEMP][l];
      const float mach = M[MACH][l], alpha = M[ALPHA][l], h_sl = M[HSL][l];
      float thrust;
      f16::sys_engine(s, T, mach, M[QBAR][l], A, h_sl, sf.throttle_pos, thrust);
      M[THRUST][l] = thrust;
    }
    AC_CLKW(1, 66 + sub * 8);
    wg_sync();                                             // B3: surfaces and thrust are known
    if (run) {                                             // while the dynamics wave assembles, integrates and propagates:
      // what FGAuxiliary published this tick is what the next tick's FCS reads
      s.alpha = M[ALPHA][l]; s.mach = M[MACH][l]; s.qc = M[QC][l]; s.vg = M[VG][l];
      s.npy = M[NPY][l]; s.npz = M[NPZ][l]; s.ap = M[AP][l]; s.aq = M[AQ][l]; s.ar = M[AR][l];
      f16::sys_mass(s, km);                                // the tanks after this tick's draw give the next tick's mass balance
      post_mass(M, l, km);
    }
  }
  if (substeps == 0 && (row || raw)) {                     // (no tick flown: the commands are still decoded for the stored state)
    f32x4 v;
    if (row) { row->take(); v = row->idx.v; } else { v.x = raw->x; v.y = raw->y; v.z = raw->z; v.w = raw->w; }
    s.da = f16::clampf(-1.0f, v.x / 20.0f - 1.0f, 1.0f); s.de = f16::clampf(-1.0f, v.y / 20.0f - 1.0f, 1.0f);
    s.dr = f16::clampf(-1.0f, v.z / 20.0f - 1.0f, 1.0f); s.thr = f16::clampf(0.0f, v.w / 58.0f + 0.4f, 0.9f);
  }
  M[F_TEF][l] = s.tef; M[F_PINR][l] = s.pin_r; M[F_PINP][l] = s.pin_p; M[F_PINY][l] = s.pin_y;
  M[F_PIR][l] = s.pi_r; M[F_PIP][l] = s.pi_p; M[F_PIY][l] = s.pi_y; M[F_AIL][l] = s.ail; M[F_ELEV][l] = s.elev; M[F_SBDEG][l] = s.sbdeg;
  M[F_N1][l] = s.n1; M[F_N2][l] = s.n2; M[F_N2NORM][l] = s.n2norm; M[F_FF][l] = s.ff; M[F_TANK0][l] = s.tank0; M[F_TANK1][l] = s.tank1;
  M[F_ENG][l] = __int_as_float(s.eng);
  M[F_DA][l] = s.da; M[F_DE][l] = s.de; M[F_DR][l] = s.dr; M[F_THR][l] = s.thr;   // (the dynamics wave stores the flight state and may never have seen the action row)
  M[F_BITS][l] = row ? row->packed() : 0.0f;
  wg_sync();
}
// The kinematics wave of a SPLIT workgroup. Attitude and position are integrated explicitly from the PREVIOUS tick's rates and
// velocity, so the quaternion, the fp64 position, the geodetic reduction, the direction cosine matrix and gravity of tick k+1 are
// computed here while the other two waves are still in tick k; on an aircraft's last tick of the step it does the fp64 geodetic
// reduction of the environment layer instead.
// QUAD: a fourth wave flies the munitions against each tick's pose (step_kernel_1v1, FORM 3). It reads the tick's fp64 position right
// after B1, so this wave posts every position as soon as it has it -- tick 0's before the first barrier, tick k+1's during tick k.
// POSE (the scenario kernels' quad form, whose environment wave has no time for it): the fp64 geodetic reduction and the NEU offset
// of every tick's position as well, a tick ahead, in the gap this wave has between B3 and the next B1.
__device__ __forceinline__ void post_quad_pose(float (*M)[64], double (*MD)[64], int l, int pb, const f16::State& s, int ticks, const DevCfg& c) {
  using namespace mail;
  f16::State sp{};
  sp.rx = s.rx; sp.ry = s.ry; sp.rz = s.rz; sp.ticks = ticks;
  f16::Derived dp;
  f16::locate(sp, dp);
  dp.vn = 0.0f; dp.ve = 0.0f; dp.vd = 0.0f;              // (the velocity part of make_pose is the environment wave's: it has the tick's velocity)
  Props pp;
  make_pose(dp, c, pp);
  MD[GD_QP + 3 * pb][l] = pp.n64; MD[GD_QP + 3 * pb + 1][l] = pp.e64; MD[GD_QP + 3 * pb + 2][l] = pp.u64;
  const int r = QP_F + QP_F_N * pb;
  M[r][l] = pp.alt_m;
  M[r + 1][l] = dp.n_eci[0]; M[r + 2][l] = dp.n_eci[1]; M[r + 3][l] = dp.n_eci[2];
  M[r + 4][l] = dp.e_eci[0]; M[r + 5][l] = dp.e_eci[1];
  M[r + 6][l] = dp.d_eci[0]; M[r + 7][l] = dp.d_eci[1]; M[r + 8][l] = dp.d_eci[2];
}
template <bool QUAD = false, bool POSE = false>
__device__ __forceinline__ void kinematics_wave(f16::State& s, Task& t, const f16::Tab& T, float (*M)[64], double (*MD)[64], int l, int substeps,
                                                const DevCfg* cfg = nullptr) {
  using namespace mail;
  f16::KinOut o;
  const int ticks0 = s.ticks;
  // tick 0 (the dynamics wave integrates its own): from the stored state, if the aircraft is alive at all
  const bool alive0 = t.status == AC_ALIVE && substeps > 0;
  if (alive0) { f16::kin_position(s, o); f16::kin_attitude(s, o); }
  if (QUAD) { MD[GD_R][l] = s.rx; MD[GD_R + 1][l] = s.ry; MD[GD_R + 2][l] = s.rz; }
  if (POSE) post_quad_pose(M, MD, l, 0, s, ticks0 + (alive0 ? 1 : 0), *cfg);
  for (int sub = 0; sub < substeps; ++sub) {
    AC_CLKW(2, 68 + sub * 8);
    wg_sync();                                             // B1: this tick's rates and velocity are known
    const bool run = M[RUNF][l] != 0.0f;                   // the dynamics wave decides who flies (status can change between substeps)
    const bool last = sub + 1 == substeps;
    if (run) {                                             // this wave's position / attitude ARE tick `sub`'s: hand them over
      const int pb = sub & 1;
      if (!QUAD) { MD[GD_R + 3 * pb][l] = s.rx; MD[GD_R + 3 * pb + 1][l] = s.ry; MD[GD_R + 3 * pb + 2][l] = s.rz; }   // (QUAD: posted a tick ago)
      M[G_Q + 4 * pb][l] = s.q0; M[G_Q + 4 * pb + 1][l] = s.q1; M[G_Q + 4 * pb + 2][l] = s.q2; M[G_Q + 4 * pb + 3][l] = s.q3;
    }
    if (run && !last) {                                    // one tick ahead (unused if the aircraft is shot down in between)
      s.wp = M[K_W][l]; s.wq = M[K_W + 1][l]; s.wr = M[K_W + 2][l];
      s.vx = M[K_V][l]; s.vy = M[K_V + 1][l]; s.vz = M[K_V + 2][l];
      f16::kin_position(s, o);
      if (QUAD) { const int nb = (sub + 1) & 1; MD[GD_R + 3 * nb][l] = s.rx; MD[GD_R + 3 * nb + 1][l] = s.ry; MD[GD_R + 3 * nb + 2][l] = s.rz; }
    } else if (run) {                                      // last substep of the step: the pose the env layer reads
      f16::Derived d;
      s.ticks += substeps;
      f16::locate(s, d);
      M[G_H][l] = d.h_sl_ft;
      M[G_NED][l] = d.n_eci[0]; M[G_NED + 1][l] = d.n_eci[1]; M[G_NED + 2][l] = d.n_eci[2];
      M[G_NED + 3][l] = d.e_eci[0]; M[G_NED + 4][l] = d.e_eci[1];
      M[G_NED + 5][l] = d.d_eci[0]; M[G_NED + 6][l] = d.d_eci[1]; M[G_NED + 7][l] = d.d_eci[2];
      MD[GD_X][l] = d.X; MD[GD_X + 1][l] = d.Y; MD[GD_X + 2][l] = d.Z;
      MD[GD_LAT][l] = d.sLat64; MD[GD_LAT + 1][l] = d.cLat64; MD[GD_LAT + 2][l] = d.sLon64; MD[GD_LAT + 3][l] = d.cLon64;
    }
    AC_CLKW(2, 69 + sub * 8);
    wg_sync();                                             // B2: this tick's air data are known
    if (run) {   // this wave's share of the aerodynamic tables: those on the sideslip axis (all else dyn_p3 computes is dead here)
      f16::DynVars kl{};
      f16::Derived dd{};
      dd.h_sl_ft = 1e6f;
      kl.alpha = M[ALPHA][l]; kl.beta = M[BETA][l]; kl.mach = M[MACH][l]; kl.vt = 1.0f;
      f16::dyn_p3(dd, T, kl, f16::Surf{});
      M[LK_CLB][l] = kl.clb; M[LK_CNB][l] = kl.cnb;
      M[LK_G7][l] = kl.g7.x; M[LK_G7 + 1][l] = kl.g7.y; M[LK_G7 + 2][l] = kl.g7.z; M[LK_G7 + 3][l] = kl.g7.w;
    }
    if (run && !last) {
      f16::kin_attitude(s, o);
      post_kin(M, l, o);
    }
    AC_CLKW(2, 70 + sub * 8);
    wg_sync();                                             // B3
    if (POSE && run && !last) post_quad_pose(M, MD, l, (sub + 1) & 1, s, ticks0 + sub + 2, *cfg);   // (position of tick sub + 1: integrated after B1)
  }
  wg_sync();
}
// LDS of a three-wave workgroup
struct SplitLds {
  float M[mail::ROWS][64];
  double MD[mail::DROWS][64];
};
// Helper waves of a three-wave workgroup: run their part of every substep and return true (the caller returns); the dynamics
// wave (wave 0) gets false. Commands (s.da .. s.thr) must be decoded before the call, or handed over as the raw action row.
template <bool QUAD = false, bool POSE = false>
__device__ __forceinline__ bool split_helper_wave(f16::State& s, Task& t, const f16::Tab& T, SplitLds& L, int l, int substeps, const float4* raw = nullptr,
                                                  const DevCfg* cfg = nullptr, ActionFetch* row = nullptr) {
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (role == 2) { kinematics_wave<QUAD, POSE>(s, t, T, L.M, L.MD, l, substeps, cfg); return true; }   // (a caller with work left for this wave tests the role itself)
  if (role == 1) { systems_wave(s, t, T, L.M, l, substeps, raw, row); return true; }
  return false;
}
// One substep of the dynamics wave (three workgroup barriers inside; every lane of the wave must call it). Returns whether this
// aircraft flew the tick. Afterwards s holds the tick's state except the quaternion (handed over by dynamics_wave_finish); the fp64
// ECI position IS current, so the caller may run f16::locate(s, d) after any substep (the missile tasks do).
// QUAD (step_kernel_1v1, FORM 3): who flies a tick is decided by the environment wave, which posts RUNF before B1 -- too late for this
// wave's first piece, which runs before that barrier. So the piece runs on the belief "still alive" (`t.status` here is this wave's
// private copy of that belief: exact for tick 0, cleared when the environment wave says otherwise) and the sixteen values it
// integrates are put back if the aircraft turns out to have been grounded in the previous substep (the kinematics wave already
// works one tick ahead like that). *d_stale is set then: d holds the frames of a tick that was not flown.
template <bool QUAD = false>
__device__ __forceinline__ bool dynamics_wave_tick(f16::State& s, Task& t, f16::Derived& d, const f16::Tab& T, SplitLds& L, int l, int sub, bool* d_stale = nullptr) {
  using namespace mail;
  float (*M)[64] = L.M;
  double (*MD)[64] = L.MD;
  bool run = t.status == AC_ALIVE;
  if (!QUAD && run && t.bloods <= 0.0f) t.status = AC_SHOTDOWN;        // simulatior.py:220-222: this tick still integrates
  f16::DynVars k;
  AC_CLK(2 + sub * 8);
  if (!QUAD) M[RUNF][l] = run ? 1.0f : 0.0f;
  const float b_wp = s.wp, b_wq = s.wq, b_wr = s.wr, b_vx = s.vx, b_vy = s.vy, b_vz = s.vz, b_ax = s.ha1x, b_ay = s.ha1y, b_az = s.ha1z;
  const float b_h1x = s.hv1x, b_h1y = s.hv1y, b_h1z = s.hv1z, b_h2x = s.hv2x, b_h2y = s.hv2y, b_h2z = s.hv2z;
  const int b_ticks = s.ticks;
  if (QUAD && !run) { M[K_V][l] = s.vx; M[K_V + 1][l] = s.vy; M[K_V + 2][l] = s.vz; }   // a grounded aircraft's (frozen) velocity, for the pose of substep 0
  if (run) {
    if (sub == 0) f16::dyn_p1(s, d, k);                        // (its quaternion goes stale from here on: the kinematics wave
    else {                                                     //  hands the final one over)
      f16::KinOut o;
      fetch_kin(M, l, o);
      f16::dyn_p1_lite(s, d, k, o);
    }
    M[K_W][l] = s.wp; M[K_W + 1][l] = s.wq; M[K_W + 2][l] = s.wr;
    M[K_V][l] = s.vx; M[K_V + 1][l] = s.vy; M[K_V + 2][l] = s.vz;
    M[CTH][l] = d.T[6] * d.d_eci[0] + d.T[7] * d.d_eci[1] + d.T[8] * d.d_eci[2];
    M[VB][l] = d.v;
  }
  AC_CLK(3 + sub * 8);
  wg_sync();                                                   // B1
  AC_CLK(4 + sub * 8);
  if (QUAD && run && M[RUNF][l] == 0.0f) {                     // grounded in the previous substep after all: undo the first piece
    s.wp = b_wp; s.wq = b_wq; s.wr = b_wr; s.vx = b_vx; s.vy = b_vy; s.vz = b_vz; s.ha1x = b_ax; s.ha1y = b_ay; s.ha1z = b_az; s.ticks = b_ticks;
    s.hv1x = b_h1x; s.hv1y = b_h1y; s.hv1z = b_h1z; s.hv2x = b_h2x; s.hv2y = b_h2y; s.hv2z = b_h2z;
    t.status = AC_SHOTDOWN;                                    // (only "not alive" matters to this wave)
    run = false;
    if (d_stale) *d_stale = true;
  }
  if (run) {
    fetch_mass(M, l, k);
    if (sub == 0) f16::dyn_p2<false>(s, d, k);
    else f16::dyn_p2<true>(s, d, k);                          // the atmosphere at this altitude came with the kinematics
    M[BETA][l] = k.beta;
    M[MACH][l] = k.mach; M[QBAR][l] = k.qbar; M[RHO][l] = k.A.rho; M[TEMP][l] = k.A.T; M[HSL][l] = d.h_sl_ft;
    M[ALPHA][l] = k.alpha; M[QC][l] = k.qc; M[VG][l] = k.vg; M[NPY][l] = k.npy; M[NPZ][l] = k.npz;
    M[AP][l] = d.p; M[AQ][l] = d.q; M[AR][l] = d.r;
  }
  AC_CLK(5 + sub * 8);
  wg_sync();                                                   // B2
  AC_CLK(6 + sub * 8);
  f16::Surf sf{};
  if (run) {
    sf = f16::Surf{M[S_AIL][l], M[S_FLAP][l], M[S_ELEV][l], M[S_RUD][l], M[S_LEF][l], M[S_SB][l], 0.0f};
    f16::dyn_p3(d, T, k, sf);
  }
  AC_CLK(7 + sub * 8);
  wg_sync();                                                   // B3
  AC_CLK(8 + sub * 8);
  if (run) {
    k.clb = M[LK_CLB][l]; k.cnb = M[LK_CNB][l];
    k.g7 = make_float4(M[LK_G7][l], M[LK_G7 + 1][l], M[LK_G7 + 2][l], M[LK_G7 + 3][l]);
    f16::dyn_p4(s, d, k, sf, M[THRUST][l]);
    if (sub > 0) {                                             // this tick's fp64 position, integrated by the kinematics wave a tick ago
      const int pb = sub & 1;
      s.rx = MD[GD_R + 3 * pb][l]; s.ry = MD[GD_R + 3 * pb + 1][l]; s.rz = MD[GD_R + 3 * pb + 2][l];
    }
  }
  return run;
}
// After the last substep: the fields the helper waves integrated. `last_tick` = the last substep this aircraft flew (-1: none).
// Returns true when d also holds the fp64 geodetic reduction of the final pose (f16::locate's outputs), which the kinematics wave
// prepares when the aircraft flew the step's last substep; otherwise the caller runs f16::locate(s, d) itself.
__device__ __forceinline__ bool dynamics_wave_finish(f16::State& s, f16::Derived& d, SplitLds& L, int l, int last_tick, int substeps) {
  using namespace mail;
  float (*M)[64] = L.M;
  double (*MD)[64] = L.MD;
  AC_CLK(50);
  wg_sync();
  AC_CLK(51);
  s.tef = M[F_TEF][l]; s.pin_r = M[F_PINR][l]; s.pin_p = M[F_PINP][l]; s.pin_y = M[F_PINY][l];
  s.pi_r = M[F_PIR][l]; s.pi_p = M[F_PIP][l]; s.pi_y = M[F_PIY][l]; s.ail = M[F_AIL][l]; s.elev = M[F_ELEV][l]; s.sbdeg = M[F_SBDEG][l];
  s.n1 = M[F_N1][l]; s.n2 = M[F_N2][l]; s.n2norm = M[F_N2NORM][l]; s.ff = M[F_FF][l]; s.tank0 = M[F_TANK0][l]; s.tank1 = M[F_TANK1][l];
  s.eng = __float_as_int(M[F_ENG][l]);
  s.da = M[F_DA][l]; s.de = M[F_DE][l]; s.dr = M[F_DR][l]; s.thr = M[F_THR][l];
  if (last_tick >= 1) {
    const int pb = last_tick & 1;
    s.q0 = M[G_Q + 4 * pb][l]; s.q1 = M[G_Q + 4 * pb + 1][l]; s.q2 = M[G_Q + 4 * pb + 2][l]; s.q3 = M[G_Q + 4 * pb + 3][l];
  }
  if (last_tick < 0 || last_tick != substeps - 1) return false;
  d.X = MD[GD_X][l]; d.Y = MD[GD_X + 1][l]; d.Z = MD[GD_X + 2][l];
  d.sLat64 = MD[GD_LAT][l]; d.cLat64 = MD[GD_LAT + 1][l]; d.sLon64 = MD[GD_LAT + 2][l]; d.cLon64 = MD[GD_LAT + 3][l];
  d.h_sl_ft = M[G_H][l];
  d.n_eci[0] = M[G_NED][l]; d.n_eci[1] = M[G_NED + 1][l]; d.n_eci[2] = M[G_NED + 2][l];
  d.e_eci[0] = M[G_NED + 3][l]; d.e_eci[1] = M[G_NED + 4][l]; d.e_eci[2] = 0.0f;
  d.d_eci[0] = M[G_NED + 5][l]; d.d_eci[1] = M[G_NED + 6][l]; d.d_eci[2] = M[G_NED + 7][l];
  return true;
}
// All substeps of a task without per-substep work: the same result as `for (sub) if (alive) { latch; tick<false>(s, d, T); }`
// followed by f16::locate(s, d) when any tick ran (returns that, and the number of ticks run); d holds the last tick's body-frame
// quantities.
__device__ __forceinline__ bool dynamics_wave_ticks(f16::State& s, Task& t, f16::Derived& d, const f16::Tab& T, SplitLds& L, int l, int substeps,
                                                    int& nrun) {
  int last_tick = -1;
  nrun = 0;
  for (int sub = 0; sub < substeps; ++sub)
    if (dynamics_wave_tick(s, t, d, T, L, l, sub)) { last_tick = sub; nrun += 1; }
  const bool located = dynamics_wave_finish(s, d, L, l, last_tick, substeps);
  if (last_tick >= 0 && !located) { f16::locate(s, d); return true; }
  return located;
}
