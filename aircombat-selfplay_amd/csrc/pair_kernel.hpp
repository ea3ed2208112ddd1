// The pair form of the step kernels for the tasks with munitions (included by aircombat.hip after split_kernel.hpp).
//
// A workgroup is TWO waves over the same 64 aircraft, each on a SIMD of its own:
//   wave 1, "flight": JSBSim's part -- the six FDM ticks of the step and the aircraft's pose in the battle-field frame (what
//            AircraftSimulator caches after every tick, simulatior.py:238-258). It loads and stores the flight state.
//   wave 0, "environment": the reference's Python part -- AircraftSimulator's status / blood bookkeeping, the munitions, chaff and decoy
//            test of every substep, the weapon rules, observations, rewards, terminations, the episode reset. It loads and stores the task
//            bookkeeping, the munition slots and the step's outputs.
// The integrators of FGPropagate are explicit (f16::propagate): an aircraft's position and velocity after tick k are fixed by what
// tick k-1 left in the state. So the flight wave posts the pose of tick k BEFORE the tick (the fp64 geodetic reduction and the NEU
// offset included), and the environment wave flies the substep's munitions against it -- proportional navigation, fuse, decoys --
// WHILE the flight wave works through the tick's aerodynamics. What comes back is one flag per aircraft: does it fly the next tick
// (a hit grounds the target from the next substep on, simulatior.py:220-229). An aircraft that turns out not to fly tick k keeps the
// pose the environment wave already holds. One workgroup barrier per substep; the mailboxes are double-buffered by substep parity.
//
// Against the one-wave form this takes the munitions (more than half of a substep there) off the critical path and splits the
// register file demand in two: neither wave spills (the one-wave scenario kernels kept 100-1160 values in scratch).
#pragma once

namespace pair {
enum { FIN_VN, FIN_VE, FIN_VD, FIN_ALT, FIN_UB, FIN_VB, FIN_WB, FIN_VC, FIN_SPHI, FIN_CPHI, FIN_STHT, FIN_CTHT, FIN_M11, FIN_M12,
       FIN_P, FIN_Q, FIN_R, FIN_VECI, FIN_HSL, FIN_NPX, FIN_NPY, FIN_NPZ, FIN_TICKS,
       FIN_BITS,   // what the action row holds behind the control indices (the shoot bit / the four weapon bits, packed): only the flight wave reads the row
       NFIN };
enum { RUN_FLY = 1, RUN_NEED_POSE = 2 };
}
struct PairLds {
  double P64[2][3][64];       // flight -> environment: NEU position after the substep's integration step, by substep parity
  float PV[2][4][64];         //                        NED velocity (clipped like the catalogue does) and altitude
  int RUNF[2][64];            // environment -> flight: pair::RUN_* bits for the coming tick, by substep parity
  double F64[3][64];          // flight -> environment after the last tick: the final pose ...
  float FIN[pair::NFIN][64];  // ... and everything else the observation / reward / termination code reads off the aircraft
};

// local NED velocity from the ECI velocity and the frame of f16::locate (the part of f16::body_frame the pose needs)
__device__ __forceinline__ void ned_velocity(const f16::State& s, f16::Derived& d) {
  const float om = (float)f16::kOmega;
  const float rvx = s.vx + om * (float)s.ry, rvy = s.vy - om * (float)s.rx, rvz = s.vz;
  d.vn = d.n_eci[0] * rvx + d.n_eci[1] * rvy + d.n_eci[2] * rvz;
  d.ve = d.e_eci[0] * rvx + d.e_eci[1] * rvy;
  d.vd = d.d_eci[0] * rvx + d.d_eci[1] * rvy + d.d_eci[2] * rvz;
}

// what the environment wave reads off the aircraft after the last tick (pair_read_final)
__device__ __forceinline__ void pair_post_final(PairLds& L, int l, const f16::State& s, const f16::Derived& d, const Props& pp, float bits = 0.0f) {
  using namespace pair;
  L.FIN[FIN_BITS][l] = bits;
  L.F64[0][l] = pp.n64; L.F64[1][l] = pp.e64; L.F64[2][l] = pp.u64;
  L.FIN[FIN_VN][l] = pp.vn; L.FIN[FIN_VE][l] = pp.ve; L.FIN[FIN_VD][l] = pp.vd; L.FIN[FIN_ALT][l] = pp.alt_m;
  L.FIN[FIN_UB][l] = pp.ub; L.FIN[FIN_VB][l] = pp.vb; L.FIN[FIN_WB][l] = pp.wb; L.FIN[FIN_VC][l] = pp.vc;
  L.FIN[FIN_SPHI][l] = pp.sphi; L.FIN[FIN_CPHI][l] = pp.cphi; L.FIN[FIN_STHT][l] = pp.stht; L.FIN[FIN_CTHT][l] = pp.ctht;
  L.FIN[FIN_M11][l] = pp.m11; L.FIN[FIN_M12][l] = pp.m12;
  L.FIN[FIN_P][l] = d.p; L.FIN[FIN_Q][l] = d.q; L.FIN[FIN_R][l] = d.r; L.FIN[FIN_VECI][l] = d.veci; L.FIN[FIN_HSL][l] = d.h_sl_ft;
  L.FIN[FIN_NPX][l] = s.npx; L.FIN[FIN_NPY][l] = s.npy; L.FIN[FIN_NPZ][l] = s.npz;
  L.FIN[FIN_TICKS][l] = __int_as_float(s.ticks);
}
// The flight wave. Returns when its part of the step is done (the caller returns).
// RAW_POSE: post the tick's ECI position / velocity as they are and leave the geodetic reduction to the environment wave -- for the 1v1
// missile tasks, whose fp32 AIM-9L update leaves that wave the slack (there the flight wave is the longer of the two).
// What the flight wave reads from HBM: the flight state, the aircraft's status (only to know whether tick 0 flies: the environment
// wave owns the field) and its action row. Separate from the wave's body so that a kernel can issue the loads behind its table loads.
struct PairFlightIn { f16::State s; int status0; float4 a4; float bits; };
__device__ __forceinline__ void pair_flight_load(const DevPtrs& P, const DevCfg& c, int nn, PairFlightIn& in) {
  load_flight(P.F, P.I, P.D, c.N, nn, in.s);
  in.status0 = state_word(P.F, SW_status, c.N, nn);
  const float* act = P.actions + (size_t)nn * c.act_dim;
  in.a4 = load_controls(act, c.act_dim);
  // the rest of the row, for the environment wave (which never reads the row itself: in a host-boundary step it lives in mapped host
  // memory, and a PCIe read asked for early holds every later load of the workgroup back)
  in.bits = 0.0f;
  if (c.act_dim == 8) {
    const float4 b = load_controls(act + 4, c.act_dim);
    in.bits = (float)((b.x != 0.0f ? 1 : 0) | (b.y != 0.0f ? 2 : 0) | (b.z != 0.0f ? 4 : 0) | (b.w != 0.0f ? 8 : 0));
  } else if (c.act_dim == 5) in.bits = act[4] != 0.0f ? 1.0f : 0.0f;
}
// `tail(pp)`: work for this wave after its final values are posted and the flight state is stored, while the environment wave runs the
// weapon rules, rewards and terminations (the NvN scenario kernels build the observation rows here). It receives this aircraft's final
// Props; whatever it synchronises with the environment wave is its own business.
struct PairNoTail { __device__ __forceinline__ void operator()(const Props&) const {} };
// LATE_STORE: the TAIL stores the flight state (store_flight of in.s), after the barrier that hands the final values over, so that
// the environment wave starts on the weapon rules while the stores drain -- for tails that meet the environment wave at a later
// barrier of their own (whose release half then orders the stores before that wave's episode reset).
template <bool RAW_POSE = false, typename Tail = PairNoTail, bool LATE_STORE = false>
__device__ __forceinline__ void pair_flight_wave(const DevPtrs& P, const DevCfg& c, const f16::Tab& T, PairLds& L, int l, int n, bool live, PairFlightIn& in,
                                                 Tail tail = Tail()) {
  using namespace pair;
  f16::State& s = in.s; f16::Derived d;
  const int status0 = in.status0;
  const float4 a4 = in.a4;
  s.da = f16::clampf(-1.0f, a4.x / 20.0f - 1.0f, 1.0f);   // normalize_action (singlecombat_task.py:141-153), property bounds catalog.py:189-197
  s.de = f16::clampf(-1.0f, a4.y / 20.0f - 1.0f, 1.0f);
  s.dr = f16::clampf(-1.0f, a4.z / 20.0f - 1.0f, 1.0f);
  s.thr = f16::clampf(0.0f, a4.w / 58.0f + 0.4f, 0.9f);
  // pose after the coming tick (or of the stored state when the aircraft does not fly it), posted under parity `par`
  f16::Derived dp; Props pp;
  auto post_pose = [&](int par, bool flies) {
    f16::State sp = s;
    if (flies) {   // the position / velocity part of f16::propagate, without committing it
      constexpr float dt = 1.0f / 60.0f, k = dt / 12.0f;
      sp.rx = s.rx + (double)(k * (23.0f * s.vx - 16.0f * s.hv1x + 5.0f * s.hv2x));
      sp.ry = s.ry + (double)(k * (23.0f * s.vy - 16.0f * s.hv1y + 5.0f * s.hv2y));
      sp.rz = s.rz + (double)(k * (23.0f * s.vz - 16.0f * s.hv1z + 5.0f * s.hv2z));
      sp.vx = s.vx + dt * (1.5f * s.aix - 0.5f * s.ha1x);
      sp.vy = s.vy + dt * (1.5f * s.aiy - 0.5f * s.ha1y);
      sp.vz = s.vz + dt * (1.5f * s.aiz - 0.5f * s.ha1z);
      sp.ticks = s.ticks + 1;
    }
    if (RAW_POSE) {
      L.P64[par][0][l] = sp.rx; L.P64[par][1][l] = sp.ry; L.P64[par][2][l] = sp.rz;
      L.PV[par][0][l] = sp.vx; L.PV[par][1][l] = sp.vy; L.PV[par][2][l] = sp.vz; L.PV[par][3][l] = __int_as_float(sp.ticks);
      return;
    }
    f16::locate(sp, dp);
    ned_velocity(sp, dp);
    make_pose(dp, c, pp);
    L.P64[par][0][l] = pp.n64; L.P64[par][1][l] = pp.e64; L.P64[par][2][l] = pp.u64;
    L.PV[par][0][l] = pp.vn; L.PV[par][1][l] = pp.ve; L.PV[par][2][l] = pp.vd; L.PV[par][3][l] = pp.alt_m;
  };
  bool flew = false, need = true, pose_is_final = false;
  AC_CLKW(1, 128);
  post_pose(0, status0 == AC_ALIVE);
  for (int sub = 0; sub < c.substeps; ++sub) {
    wg_sync();                                              // the environment wave has posted who flies this tick; tick `sub`'s pose is posted
    AC_CLKW(1, 129 + 4 * sub);
    const int rf = L.RUNF[sub & 1][l];
    const bool run = rf & RUN_FLY;
    need = rf & RUN_NEED_POSE;
    if (run) { f16::propagate(s); f16::tick_after_propagate<false>(s, d, T); flew = true; }
    pose_is_final = !RAW_POSE && sub + 1 == c.substeps && run;   // the pose posted for this (last) tick is the step's final pose
    AC_CLKW(1, 131 + 4 * sub);
    // the next tick's pose: needed when something flies between the ticks, and in any case for the step's last tick
    if (sub + 1 < c.substeps && (need || sub + 2 == c.substeps)) post_pose((sub + 1) & 1, true);
    AC_CLKW(1, 132 + 4 * sub);
  }
  // ---- final values: the pose of the stored state (that of the last tick flown, which is already in dp / pp if that was the step's
  // last tick) and what the wrapper reads off the FDM (catalog.py:292-338, 386-416)
  if (!pose_is_final) { f16::locate(s, dp); ned_velocity(s, dp); }
  if (!flew) { f16::locate_fast(s, d); f16::body_frame(s, d); }   // never flew this step: the body-frame quantities of the stored pose
  d.h_sl_ft = dp.h_sl_ft; d.vn = dp.vn; d.ve = dp.ve; d.vd = dp.vd;
  d.sLat64 = dp.sLat64; d.cLat64 = dp.cLat64; d.sLon64 = dp.sLon64; d.cLon64 = dp.cLon64;
#pragma unroll
  for (int i = 0; i < 3; ++i) { d.n_eci[i] = dp.n_eci[i]; d.e_eci[i] = dp.e_eci[i]; d.d_eci[i] = dp.d_eci[i]; }
  make_props(s, d, c, pp);
  pair_post_final(L, l, s, d, pp, in.bits);
  if (!LATE_STORE && live) store_flight(P.F, P.I, P.D, c.N, n, s);
  AC_CLKW(1, 160);
  wg_sync();   // final values posted, flight state stored (the release half of the barrier waits for the stores: an episode reset by
               // the environment wave overwrites them afterwards)
  tail(pp);
}

// Environment wave, one substep: post who flies (and whether poses are wanted between the ticks), meet the flight wave, take the
// pose of this substep. Returns whether this aircraft flew the tick.
template <bool RAW_POSE = false>
__device__ __forceinline__ bool pair_substep(Task& t, PairLds& L, int l, int sub, bool need_pose, Props& pr, const DevCfg& c) {
  using namespace pair;
  const bool run = t.status == AC_ALIVE;
  if (run && t.bloods <= 0.0f) t.status = AC_SHOTDOWN;      // simulatior.py:220-222: this tick still integrates
  L.RUNF[sub & 1][l] = (run ? RUN_FLY : 0) | (need_pose ? RUN_NEED_POSE : 0);
  wg_sync();
  if (need_pose && (run || sub == 0)) {                     // (a grounded aircraft keeps the pose it had)
    const int par = sub & 1;
    if (RAW_POSE) {                                         // the geodetic reduction of the posted ECI pose happens here
      f16::State sp{}; f16::Derived dp;
      sp.rx = L.P64[par][0][l]; sp.ry = L.P64[par][1][l]; sp.rz = L.P64[par][2][l];
      sp.vx = L.PV[par][0][l]; sp.vy = L.PV[par][1][l]; sp.vz = L.PV[par][2][l]; sp.ticks = __float_as_int(L.PV[par][3][l]);
      f16::locate(sp, dp);
      ned_velocity(sp, dp);
      make_pose(dp, c, pr);
      return run;
    }
    pr.n64 = L.P64[par][0][l]; pr.e64 = L.P64[par][1][l]; pr.u64 = L.P64[par][2][l];
    pr.n = (float)pr.n64; pr.e = (float)pr.e64; pr.u = (float)pr.u64;
    pr.vn = L.PV[par][0][l]; pr.ve = L.PV[par][1][l]; pr.vd = L.PV[par][2][l]; pr.alt_m = L.PV[par][3][l];
  }
  return run;
}
// after the last substep: wait for the flight wave's final values and read them
__device__ __forceinline__ void pair_read_final(const PairLds& L, int l, f16::State& s, f16::Derived& d, Props& pr) {
  using namespace pair;
  pr.n64 = L.F64[0][l]; pr.e64 = L.F64[1][l]; pr.u64 = L.F64[2][l];
  pr.n = (float)pr.n64; pr.e = (float)pr.e64; pr.u = (float)pr.u64;
  pr.vn = L.FIN[FIN_VN][l]; pr.ve = L.FIN[FIN_VE][l]; pr.vd = L.FIN[FIN_VD][l]; pr.alt_m = L.FIN[FIN_ALT][l];
  pr.ub = L.FIN[FIN_UB][l]; pr.vb = L.FIN[FIN_VB][l]; pr.wb = L.FIN[FIN_WB][l]; pr.vc = L.FIN[FIN_VC][l];
  pr.sphi = L.FIN[FIN_SPHI][l]; pr.cphi = L.FIN[FIN_CPHI][l]; pr.stht = L.FIN[FIN_STHT][l]; pr.ctht = L.FIN[FIN_CTHT][l];
  pr.m11 = L.FIN[FIN_M11][l]; pr.m12 = L.FIN[FIN_M12][l];
  d.p = L.FIN[FIN_P][l]; d.q = L.FIN[FIN_Q][l]; d.r = L.FIN[FIN_R][l]; d.veci = L.FIN[FIN_VECI][l]; d.h_sl_ft = L.FIN[FIN_HSL][l];
  s.npx = L.FIN[FIN_NPX][l]; s.npy = L.FIN[FIN_NPY][l]; s.npz = L.FIN[FIN_NPZ][l];
  s.ticks = __float_as_int(L.FIN[FIN_TICKS][l]);
}

// ------------------------------------------------------------------------------------------------ the quad form
// The 1v1 missile tasks at grids of up to one workgroup per CU: the three waves of the three-wave form (split_kernel.hpp) fly the FDM
// tick, a fourth -- the environment wave of the pair form -- flies the munitions against each tick's pose and owns the task
// bookkeeping. It joins the tick's three barriers: who flies (RUNF) is posted before B1, the tick's pose is read after B1 (fp64
// position from the kinematics wave, ECI velocity from the dynamics wave), the missiles fly between B2 and B3.
struct QuadLds {
  SplitLds S;
  PairLds P;    // (only F64 / FIN: the final values, dynamics wave -> environment wave)
};
// Environment wave, first part of a substep. `nrun` counts the ticks this aircraft has flown this step (the Earth angle of the pose).
// RAW_POSE (1v1 missile tasks, fp32 AIM-9L: this wave has the time): B1 and B2 inside, the geodetic reduction of the tick's raw ECI
// position between them; the caller flies the munitions and then calls wg_sync() for B3.
// !RAW_POSE (scenario kernels): B1 alone; the kinematics wave has posted the reduced position (kinematics_wave<.., POSE>), this wave adds
// the NED velocity; the caller spreads the munitions over the tick's three gaps and places B2 and B3 itself.
template <bool RAW_POSE>
__device__ __forceinline__ bool quad_substep_begin(Task& t, QuadLds& Q, int l, int sub, bool need_pose, int ticks0, int& nrun, Props& pr, const DevCfg& c) {
  using namespace mail;
  const bool run = t.status == AC_ALIVE;
  if (run && t.bloods <= 0.0f) t.status = AC_SHOTDOWN;      // simulatior.py:220-222: this tick still integrates
  Q.S.M[RUNF][l] = run ? 1.0f : 0.0f;
  wg_sync();                                                // B1
  nrun += run ? 1 : 0;
  if (need_pose && (run || sub == 0)) {                     // (a grounded aircraft keeps the pose it had)
    f16::State sp{}; f16::Derived dp;
    const int pb = sub & 1;
    sp.rx = Q.S.MD[GD_R + 3 * pb][l]; sp.ry = Q.S.MD[GD_R + 3 * pb + 1][l]; sp.rz = Q.S.MD[GD_R + 3 * pb + 2][l];
    sp.vx = Q.S.M[K_V][l]; sp.vy = Q.S.M[K_V + 1][l]; sp.vz = Q.S.M[K_V + 2][l];
    if (RAW_POSE) {
      sp.ticks = ticks0 + nrun;
      f16::locate(sp, dp);
      ned_velocity(sp, dp);
      make_pose(dp, c, pr);
    } else {
      const int r = QP_F + QP_F_N * pb;
      pr.n64 = Q.S.MD[GD_QP + 3 * pb][l]; pr.e64 = Q.S.MD[GD_QP + 3 * pb + 1][l]; pr.u64 = Q.S.MD[GD_QP + 3 * pb + 2][l];
      pr.n = (float)pr.n64; pr.e = (float)pr.e64; pr.u = (float)pr.u64;
      pr.alt_m = Q.S.M[r][l];
      dp.n_eci[0] = Q.S.M[r + 1][l]; dp.n_eci[1] = Q.S.M[r + 2][l]; dp.n_eci[2] = Q.S.M[r + 3][l];
      dp.e_eci[0] = Q.S.M[r + 4][l]; dp.e_eci[1] = Q.S.M[r + 5][l]; dp.e_eci[2] = 0.0f;
      dp.d_eci[0] = Q.S.M[r + 6][l]; dp.d_eci[1] = Q.S.M[r + 7][l]; dp.d_eci[2] = Q.S.M[r + 8][l];
      ned_velocity(sp, dp);
      pr.vn = mps(dp.vn); pr.ve = mps(dp.ve); pr.vd = mps(dp.vd);
    }
  }
  if (RAW_POSE) wg_sync();                                  // B2
  return run;
}
// The dynamics wave's whole step in the quad form (the caller returns afterwards).
__device__ __forceinline__ void quad_dynamics_wave(const DevPtrs& P, const DevCfg& c, const f16::Tab& T, QuadLds& Q, int l, int n, bool live,
                                                   f16::State& s, Task& t) {
  f16::Derived d;
  bool have_pose = false, d_stale = false;
  int last_tick = -1;
  for (int sub = 0; sub < c.substeps; ++sub)
    if (dynamics_wave_tick<true>(s, t, d, T, Q.S, l, sub, &d_stale)) { have_pose = true; last_tick = sub; }
  const bool located = dynamics_wave_finish(s, d, Q.S, l, last_tick, c.substeps);   // (+ the helper waves' fields)
  // (the decoded commands of the stored state came with the systems wave's fields)
  if (!located) f16::locate(s, d);
  if (!have_pose || d_stale) f16::body_frame(s, d);        // never flew, or the last piece it ran was undone: the frames of the stored pose
  Props pp;
  make_props(s, d, c, pp);
  pair_post_final(Q.P, l, s, d, pp);
  if (live) store_flight(P.F, P.I, P.D, c.N, n, s);
  wg_sync();   // final values posted, flight state stored (an episode reset by the environment wave overwrites it afterwards)
}

