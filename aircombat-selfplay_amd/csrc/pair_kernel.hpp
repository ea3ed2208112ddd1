// The pair form of the step kernels for the tasks with munitions (included by aircombat.hip after split_kernel.hpp).
//
// A workgroup is TWO waves over the same 64 aircraft, each on a SIMD of its own:
//   wave 1, "flight": JSBSim's part -- the six FDM ticks of the step, nothing else. It loads and stores the flight state.
//   wave 0, "environment": the reference's Python part -- AircraftSimulator's status / blood bookkeeping, the munitions, chaff and decoy
//            test of every substep, the weapon rules, observations, rewards, terminations, the episode reset. It loads and stores the task
//            bookkeeping, the munition slots and the step's outputs.
// The integrators of FGPropagate are explicit (f16::propagate): an aircraft's pose after tick k is fixed by what tick k-1 left in
// the state, so the flight wave posts it at the START of tick k and the environment wave flies the substep's munitions against it
// (fp64 geodetic reduction, proportional navigation, fuse, decoys) WHILE the flight wave works through the tick's aerodynamics. What
// comes back is one flag per aircraft -- does it fly the next tick (a hit grounds the target from the next substep on,
// simulatior.py:220-229). Two workgroup barriers per substep: "run flags posted" and "poses posted".
//
// Against the one-wave form this halves the length of the longest instruction stream of a munition task (the munitions took more
// than half of a substep) and splits the register file demand in two: neither wave spills (the one-wave scenario kernels kept 100-1160
// values in scratch).
#pragma once

namespace pair {
enum { FIN_T = 0, FIN_U = 9, FIN_V, FIN_W, FIN_P, FIN_Q, FIN_R, FIN_VECI, FIN_QC, FIN_NPX, FIN_NPY, FIN_NPZ, FIN_TICKS, FIN_FLEW, NFIN };
}
struct PairLds {
  double R[3][64];            // flight -> environment: ECI position after the substep's propagate step
  float V[3][64];             //                        ECI velocity
  float RUNF[64];             // environment -> flight: this aircraft flies the coming tick
  float FIN[pair::NFIN][64];  // flight -> environment after the last tick: what the observation / termination code reads off the FDM
};

// The flight wave. Returns when its part of the step is done (the caller returns).
__device__ __forceinline__ void pair_flight_wave(const DevPtrs& P, const DevCfg& c, const f16::Tab& T, PairLds& L, int l, int nn, int n, bool live) {
  using namespace pair;
  f16::State s; f16::Derived d;
  load_flight(P.F, P.I, P.D, c.N, nn, s);
  const float4 a4 = load_controls(P.actions + (size_t)nn * c.act_dim, c.act_dim);
  s.da = f16::clampf(-1.0f, a4.x / 20.0f - 1.0f, 1.0f);   // normalize_action (singlecombat_task.py:141-153), property bounds catalog.py:189-197
  s.de = f16::clampf(-1.0f, a4.y / 20.0f - 1.0f, 1.0f);
  s.dr = f16::clampf(-1.0f, a4.z / 20.0f - 1.0f, 1.0f);
  s.thr = f16::clampf(0.0f, a4.w / 58.0f + 0.4f, 0.9f);
  bool flew = false;
  AC_CLKW(1, 128);
  for (int sub = 0; sub < c.substeps; ++sub) {
    wg_sync();                                              // the environment wave has posted who flies this tick
    AC_CLKW(1, 129 + 4 * sub);
    const bool run = L.RUNF[l] != 0.0f;
    if (run) f16::propagate(s);
    L.R[0][l] = s.rx; L.R[1][l] = s.ry; L.R[2][l] = s.rz;  // (a grounded aircraft keeps posting its frozen pose)
    L.V[0][l] = s.vx; L.V[1][l] = s.vy; L.V[2][l] = s.vz;
    AC_CLKW(1, 130 + 4 * sub);
    wg_sync();                                              // poses posted
    AC_CLKW(1, 131 + 4 * sub);
    if (run) { f16::tick_after_propagate<false>(s, d, T); flew = true; }
    AC_CLKW(1, 132 + 4 * sub);
  }
  if (!flew) { f16::locate_fast(s, d); f16::body_frame(s, d); }   // never flew this step: the body-frame quantities of the stored pose
#pragma unroll
  for (int i = 0; i < 9; ++i) L.FIN[FIN_T + i][l] = d.T[i];
  L.FIN[FIN_U][l] = d.u; L.FIN[FIN_V][l] = d.v; L.FIN[FIN_W][l] = d.w;
  L.FIN[FIN_P][l] = d.p; L.FIN[FIN_Q][l] = d.q; L.FIN[FIN_R][l] = d.r; L.FIN[FIN_VECI][l] = d.veci;
  L.FIN[FIN_QC][l] = s.qc; L.FIN[FIN_NPX][l] = s.npx; L.FIN[FIN_NPY][l] = s.npy; L.FIN[FIN_NPZ][l] = s.npz;
  L.FIN[FIN_TICKS][l] = __int_as_float(s.ticks); L.FIN[FIN_FLEW][l] = flew ? 1.0f : 0.0f;
  if (live) store_flight(P.F, P.I, P.D, c.N, n, s);
  AC_CLKW(1, 160);
  wg_sync();   // final values posted, flight state stored (the release half of the barrier waits for the stores: an episode reset by
               // the environment wave overwrites them afterwards)
}

// Environment wave, one substep: post who flies, wait for the poses. Returns whether this aircraft flew the tick.
__device__ __forceinline__ bool pair_substep(Task& t, PairLds& L, int l) {
  const bool run = t.status == AC_ALIVE;
  if (run && t.bloods <= 0.0f) t.status = AC_SHOTDOWN;      // simulatior.py:220-222: this tick still integrates
  L.RUNF[l] = run ? 1.0f : 0.0f;
  wg_sync();
  wg_sync();
  return run;
}
// the posted pose as the fields of State the geodetic reduction and the NED velocity read
__device__ __forceinline__ void pair_read_pose(const PairLds& L, int l, int ticks, f16::State& s) {
  s.rx = L.R[0][l]; s.ry = L.R[1][l]; s.rz = L.R[2][l];
  s.vx = L.V[0][l]; s.vy = L.V[1][l]; s.vz = L.V[2][l];
  s.ticks = ticks;
}
// local NED velocity from the ECI velocity and the frame of f16::locate (the part of f16::body_frame the munitions need)
__device__ __forceinline__ void ned_velocity(const f16::State& s, f16::Derived& d) {
  const float om = (float)f16::kOmega;
  const float rvx = s.vx + om * (float)s.ry, rvy = s.vy - om * (float)s.rx, rvz = s.vz;
  d.vn = d.n_eci[0] * rvx + d.n_eci[1] * rvy + d.n_eci[2] * rvz;
  d.ve = d.e_eci[0] * rvx + d.e_eci[1] * rvy;
  d.vd = d.d_eci[0] * rvx + d.d_eci[1] * rvy + d.d_eci[2] * rvz;
}
// after the last substep: wait for the flight wave's final values and read them
__device__ __forceinline__ void pair_read_final(const PairLds& L, int l, f16::State& s, f16::Derived& d) {
  using namespace pair;
#pragma unroll
  for (int i = 0; i < 9; ++i) d.T[i] = L.FIN[FIN_T + i][l];
  d.u = L.FIN[FIN_U][l]; d.v = L.FIN[FIN_V][l]; d.w = L.FIN[FIN_W][l];
  d.p = L.FIN[FIN_P][l]; d.q = L.FIN[FIN_Q][l]; d.r = L.FIN[FIN_R][l]; d.veci = L.FIN[FIN_VECI][l];
  s.qc = L.FIN[FIN_QC][l]; s.npx = L.FIN[FIN_NPX][l]; s.npy = L.FIN[FIN_NPY][l]; s.npz = L.FIN[FIN_NPZ][l];
  s.ticks = __float_as_int(L.FIN[FIN_TICKS][l]);
}
