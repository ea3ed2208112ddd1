// Low-level controller of the hierarchical tasks on the device: the reference's BaselineActor
// (envs/JSBSim/model/baseline_actor.py:12-110: MLP 12->128->128 with ReLU+LayerNorm, GRU 128, LayerNorm, four argmax heads
// [41,41,41,30]) evaluated for every aircraft once per env step, driven exactly like
// HierarchicalSingleCombatTask.normalize_action (tasks/singlecombat_task.py:223-256): inputs = the [3,5,3] choice mapped to
// (delta altitude, delta heading, delta speed) + the first nine values of the aircraft's current observation; the four argmax
// indices become the control indices the step kernel decodes. Included by aircombat.hip.
//
// This file: what the controller kernel (controller8_kernel.hpp) shares with the host side -- network dimensions, the layout of the
// exported weight blob, the kernel's arguments -- and the inputs of the scripted opponents. One workgroup = 32 or 64 aircraft x 8 waves;
// each wave owns 16 columns of a layer's outputs (controller8_kernel.hpp).
// (History in controller_pieces.hpp. A first version with lane = aircraft and wave-uniform weights through the scalar cache took 257 us:
// every s_load missed.)
#pragma once

namespace ctl {
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int HID = 128, NH = 153, NHP = 160;   // hidden width, stacked head outputs, padded to 5 column tiles
constexpr int MT = 32;                          // aircraft per workgroup
constexpr int LS = 33;                          // LDS row stride of the [feature][aircraft] buffers (odd: column writes hit 32 banks)
// source blob of tools/export_baseline_actor.py ([out][in] like torch)
enum : int {
  S_W1 = 0, S_B1 = S_W1 + 128 * 12, S_G1 = S_B1 + 128, S_BE1 = S_G1 + 128,
  S_W2 = S_BE1 + 128, S_B2 = S_W2 + 128 * 128, S_G2 = S_B2 + 128, S_BE2 = S_G2 + 128,
  S_WIH = S_BE2 + 128, S_WHH = S_WIH + 384 * 128, S_BIH = S_WHH + 384 * 128, S_BHH = S_BIH + 384,
  S_G3 = S_BHH + 384, S_BE3 = S_G3 + 128, S_WA = S_BE3 + 128, S_BA = S_WA + 153 * 128, S_END = S_BA + 153
};

struct Args {
  const float* Ws8;        // the weights as bf16 pieces, tiled for controller8_kernel (16-column tiles, 32-k steps: controller8_kernel.hpp)
  const float* hi;         // [N][act_hi]: 3 high-level choices (+ weapon bits passed through)
  const float* obs;        // [N][obs_dim]: observation of the CURRENT state (last step's / the reset's output)
  float* H;                // [128][N] GRU state
  float* low;              // [N][act_low] out: 4 control indices (+ the weapon bits)
  int N, obs_dim, act_hi, act_low;
  // scripted opponents (use_baseline, model/baseline.py): the enemy team's inputs come from geometry instead of `hi`
  int use_baseline;        // 0 none, 1 PursueAgent, 2 ManeuverAgent('triangle')
  int A, n_ego, use_artillery;
  float time_interval;     // env.time_interval = agent_interaction_steps / sim_freq
  int* man_step;           // [N] ManeuverAgent.step
  float* man_h0;           // [N] ManeuverAgent.init_heading (latched when step == 0)
  DevPtrs P; DevCfg c;     // aircraft state (scripted inputs are computed from it)
};
__device__ __forceinline__ float in_range_rad_f(float a) {   // utils.py:114-119 with Python's % semantics
  a = fmodf(a, 6.283185307179586f);
  if (a < 0.0f) a += 6.283185307179586f;
  if (a > 3.14159265358979f) a -= 6.283185307179586f;
  return a;
}
__device__ __forceinline__ void aircraft_props(const DevPtrs& P, const DevCfg& c, int n, Props& pr, float& psi) {
  State s; Task t; Derived d;
  load_state(P.F, P.I, P.D, c.N, n, s, t);
  f16::locate(s, d); f16::body_frame(s, d);
  make_props(s, d, c, pr);
  psi = atan2f(pr.m12, pr.m11);
  if (psi < 0.0f) psi += 6.283185307179586f;
}

__device__ __forceinline__ floatx16 splat(float v) {
  floatx16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = v;
  return a;
}
// result layout of the 32x32 tile: acc[r] is (row = 8 (r / 4) + 4 (lane / 32) + r % 4, column = lane % 32)
__device__ __forceinline__ int c_row(int r, int lane) { return (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3); }

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }
}  // namespace ctl

// Inputs of a scripted opponent (`use_baseline`; aircraft n of the enemy team): BaselineAgent.get_observation (baseline.py:45-63) on the
// geometry of PursueAgent / ManeuverAgent. controller8_kernel<true, .> computes them while it stages its inputs.
namespace ctl {
__device__ __forceinline__ void scripted_inputs(const Args& a, int n, float (&x)[12]) {
  // (singlecombat_task.py:224-228, scenario1_task.py:41-49, scenario2_task.py:49-58)
  Props pr; float psi;
  aircraft_props(a.P, a.c, n, pr, psi);
  float dv0, dv1, dv2;
  if (a.use_baseline == 2) {   // ManeuverAgent('triangle').set_delta_value (baseline.py:137-155)
    int st = a.man_step[n];
    float h0 = (st == 0) ? psi : a.man_h0[n];
    int i = 0;
    for (i = 0; i < 300; ++i) if ((float)st <= (float)(i + 1) * 30.0f / a.time_interval) break;
    i = min(i, 299) % 3;
    dv1 = h0 + (i == 0 ? 1.0471975511965976f : (i == 1 ? 3.14159265358979f : -1.0471975511965976f)) - psi;
    dv0 = 6000.0f - pr.alt_m; dv2 = 243.0f - pr.ub;
    a.man_step[n] = st + 1; a.man_h0[n] = h0;
  } else {                      // PursueAgent.set_delta_value(env, task, k) (baseline.py:85-104): chase aircraft k of the list
    Props pt; float psit;
    aircraft_props(a.P, a.c, n - a.n_ego, pt, psit);
    dv0 = pt.u - pr.u;
    // get2d_AO_TA_R's angle-off (utils.py:86-103) is acos(dot / (R |v| + 1e-8)) with the sign of the 2-D cross product. A pursuer
    // drives exactly that angle to zero, where acos turns an fp32 rounding of its argument into sqrt(2 eps) = 3.5e-4 rad; the same angle
    // as atan2(cross, dot) is good to an ulp everywhere and differs from the float64 acos form by the 1e-8 in its denominator only
    const float dx = pt.n - pr.n, dy = pt.e - pr.e;
    const float cr = pr.vn * dy - pr.ve * dx;
    dv1 = (cr == 0.0f) ? 0.0f : atan2f(cr, dx * pr.vn + dy * pr.ve);
    dv2 = pt.ub - pr.ub;
  }
  // BaselineAgent.get_observation (baseline.py:45-63)
  x[0] = dv0 / 1000.0f; x[1] = in_range_rad_f(dv1); x[2] = dv2 / 340.0f; x[3] = pr.alt_m / 5000.0f;
  x[4] = pr.sphi; x[5] = pr.cphi; x[6] = pr.stht; x[7] = pr.ctht;
  x[8] = pr.ub / 340.0f; x[9] = pr.vb / 340.0f; x[10] = pr.wb / 340.0f; x[11] = pr.vc / 340.0f;
}
}  // namespace ctl
