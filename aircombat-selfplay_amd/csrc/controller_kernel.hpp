// Low-level controller of the hierarchical tasks on the device: the reference's BaselineActor
// (envs/JSBSim/model/baseline_actor.py:12-110: MLP 12->128->128 with ReLU+LayerNorm, GRU 128, LayerNorm, four argmax heads
// [41,41,41,30]) evaluated for every aircraft once per env step, driven exactly like
// HierarchicalSingleCombatTask.normalize_action (tasks/singlecombat_task.py:223-256): inputs = the [3,5,3] choice mapped to
// (delta altitude, delta heading, delta speed) + the first nine values of the aircraft's current observation; the four argmax
// indices become the control indices the step kernel decodes. Included by aircombat.hip.
//
// This is the one GEMM-shaped piece of the path, so it runs on the matrix cores in full fp32
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation in k order). One workgroup = 32 aircraft (the M of the tile)
// x 4 waves; each wave owns 32-column tiles of a layer's outputs (one tile of the 128-wide layers, the six gate tiles of its 32
// GRU units, one or two of the five head tiles). Activations live in LDS feature-major [k][32] — exactly the A-operand order
// (lane = (row, k parity)) — and weights are pre-tiled on the host so that one coalesced 16-byte load per lane feeds four
// MFMAs. LayerNorm / gate algebra / argmax are a few hundred VALU instructions around ~580 MFMAs per wave.
// (A first version kept lane = aircraft with wave-uniform weights through the scalar cache: every s_load missed and stalled
// its wave for ~600 cycles, 257 us per call.)
#pragma once

namespace ctl {
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int HID = 128, NH = 153, NHP = 160;   // hidden width, stacked head outputs, padded to 5 column tiles
constexpr int MT = 32;                          // aircraft per workgroup
constexpr int LS = 33;                          // LDS row stride of the [feature][aircraft] buffers (odd: column writes hit 32 banks)
// B-operand tiles: tile(c, K) = K/8 groups x 64 lanes x 4 floats; element (t4, lane, q) = W[j = 32c + lane%32][k = 2(4 t4 + q) + lane/32]
constexpr int tile_floats(int K) { return (K / 8) * 64 * 4; }
enum : int {
  D_W1 = 0,                                  // K = 16 (12 padded), 4 tiles
  D_W2 = D_W1 + 4 * tile_floats(16),         // K = 128, 4 tiles
  D_WIH = D_W2 + 4 * tile_floats(128),       // 12 tiles (r0..3, z0..3, n0..3)
  D_WHH = D_WIH + 12 * tile_floats(128),     // 12 tiles
  D_WA = D_WHH + 12 * tile_floats(128),      // 5 tiles (columns 153..159 zero)
  D_B1 = D_WA + 5 * tile_floats(128), D_G1 = D_B1 + 128, D_BE1 = D_G1 + 128,
  D_B2 = D_BE1 + 128, D_G2 = D_B2 + 128, D_BE2 = D_G2 + 128,
  D_BIH = D_BE2 + 128, D_BHH = D_BIH + 384, D_G3 = D_BHH + 384, D_BE3 = D_G3 + 128,
  D_BA = D_BE3 + 128,                        // [160]
  D_END = D_BA + NHP
};
// source blob of tools/export_baseline_actor.py ([out][in] like torch)
enum : int {
  S_W1 = 0, S_B1 = S_W1 + 128 * 12, S_G1 = S_B1 + 128, S_BE1 = S_G1 + 128,
  S_W2 = S_BE1 + 128, S_B2 = S_W2 + 128 * 128, S_G2 = S_B2 + 128, S_BE2 = S_G2 + 128,
  S_WIH = S_BE2 + 128, S_WHH = S_WIH + 384 * 128, S_BIH = S_WHH + 384 * 128, S_BHH = S_BIH + 384,
  S_G3 = S_BHH + 384, S_BE3 = S_G3 + 128, S_WA = S_BE3 + 128, S_BA = S_WA + 153 * 128, S_END = S_BA + 153
};

struct Args {
  const float* W;          // device blob (layout above)
  const float* Ws;         // the same weights as bf16 pieces, tiled for controller_split_kernel (controller_split_kernel.hpp)
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
  float* scripted;         // [N][12] controller inputs of the scripted aircraft (written by scripted_inputs_kernel)
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

// A operands of one layer for this lane: A[t] = act[k = 2t + lane/32][row = lane%32]
template <int K>
__device__ __forceinline__ void load_a(const float* act, int lane, float (&A)[K / 2]) {
#pragma unroll
  for (int t = 0; t < K / 2; ++t) A[t] = act[(2 * t + (lane >> 5)) * LS + (lane & 31)];
}
// acc += A(32 x K) * tile(K x 32); the 16-byte weight loads are interleaved with the MFMAs by the scheduler (an explicit
// double-buffered prefetch of whole tiles was tried: 400+ registers, 39 us instead of 31)
template <int K>
__device__ __forceinline__ void mma_tile(const float* __restrict__ tile, int lane, const float (&A)[K / 2], floatx16& acc) {
  const float4* t4 = reinterpret_cast<const float4*>(tile) + lane;
#pragma unroll
  for (int g = 0; g < K / 8; ++g) {
    const float4 b = t4[g * 64];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * g + 0], b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * g + 1], b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * g + 2], b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[4 * g + 3], b.w, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ floatx16 splat(float v) {
  floatx16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = v;
  return a;
}
// result layout of the 32x32 tile: acc[r] is (row = 8 (r / 4) + 4 (lane / 32) + r % 4, column = lane % 32)
__device__ __forceinline__ int c_row(int r, int lane) { return (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3); }

// torch.nn.LayerNorm(128) (eps 1e-5, biased variance) of buf[feature][row] in place; 256 threads: thread = (row, 16-feature part)
__device__ __forceinline__ void layer_norm(float* buf, float* red, const float* __restrict__ g, const float* __restrict__ b, int tid) {
  const int row = tid & 31, part = tid >> 5;
  float x[16], gg[16], bb[16];
  // scale / shift from L2 first: their latency then hides behind the two reductions instead of following them
  {
    const float4* g4 = reinterpret_cast<const float4*>(g + part * 16);
    const float4* b4 = reinterpret_cast<const float4*>(b + part * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 gv = g4[q], bv = b4[q];
      gg[4 * q] = gv.x; gg[4 * q + 1] = gv.y; gg[4 * q + 2] = gv.z; gg[4 * q + 3] = gv.w;
      bb[4 * q] = bv.x; bb[4 * q + 1] = bv.y; bb[4 * q + 2] = bv.z; bb[4 * q + 3] = bv.w;
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int f = 0; f < 16; ++f) { x[f] = buf[(part * 16 + f) * LS + row]; s += x[f]; }
  red[part * LS + row] = s;
  __syncthreads();
  float m = 0.0f;
#pragma unroll
  for (int p = 0; p < 8; ++p) m += red[p * LS + row];
  m *= (1.0f / HID);
  float v = 0.0f;
#pragma unroll
  for (int f = 0; f < 16; ++f) { const float d = x[f] - m; v = fmaf(d, d, v); }
  red[(8 + part) * LS + row] = v;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int p = 0; p < 8; ++p) var += red[(8 + p) * LS + row];
  const float is = rsqrtf(var * (1.0f / HID) + 1e-5f);
#pragma unroll
  for (int f = 0; f < 16; ++f) buf[(part * 16 + f) * LS + row] = fmaf((x[f] - m) * is, gg[f], bb[f]);
  __syncthreads();
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }
}  // namespace ctl

// Inputs of a scripted opponent (`use_baseline`; aircraft n of the enemy team): BaselineAgent.get_observation (baseline.py:45-63) on the
// geometry of PursueAgent / ManeuverAgent. controller_split_kernel computes them while it stages its inputs; the fp32 form of the
// controller keeps them in a kernel of their own (scripted_inputs_kernel) so that the state -> pose code (fp64 geodesy) does not
// inflate its register allocation.
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
__global__ __launch_bounds__(64) void scripted_inputs_kernel(ctl::Args a) {   // one lane per aircraft, enemy-team lanes only
  using namespace ctl;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.N) return;
  if (n % a.A < a.n_ego) return;
  float x[12];
  scripted_inputs(a, n, x);
  for (int k = 0; k < 12; ++k) a.scripted[(size_t)n * 12 + k] = x[k];
}

__global__ __launch_bounds__(256) void controller_kernel(ctl::Args a) {
  using namespace ctl;
  __shared__ float act0[HID * LS];   // activations, feature-major [k][row]
  __shared__ float act1[HID * LS];
  __shared__ float hbuf[HID * LS];   // GRU state of the 32 aircraft
  __shared__ float lg[NHP * LS];     // head logits
  __shared__ float red[16 * LS];     // LayerNorm partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * MT;
  const float* __restrict__ W = a.W;
  const int col = lane & 31;

  // ---- stage: 12 controller inputs (rows 12..15 of the K = 16 pad are zero) and the GRU state
  {
    const int row = tid & 31, part = tid >> 5;   // 8 parts
    const int n = min(i0 + row, a.N - 1);
    if (part == 0) {
      const float* hi = a.hi + (size_t)n * a.act_hi;
      const float* ob = a.obs + (size_t)n * a.obs_dim;
      const int slot = n % a.A;
      float x[12];
      if (a.use_baseline && slot >= a.n_ego) {
        // the enemy team is flown by BaselineAgent k: its 12 inputs were prepared by scripted_inputs_kernel
#pragma unroll
        for (int k = 0; k < 12; ++k) x[k] = a.scripted[(size_t)n * 12 + k];
      } else {
        const int c0 = (int)hi[0], c1 = (int)hi[1], c2 = (int)hi[2];
        // singlecombat_task.py:217-219, 235-241: below 3500 m the altitude choice is overridden by "climb"
        x[0] = (ob[0] * 5000.0f < 3500.0f) ? 0.1f : (c0 == 0 ? 0.1f : (c0 == 1 ? 0.0f : -0.1f));
        x[1] = (float)(c1 - 2) * 0.26179938779914943f;   // {-pi/6, -pi/12, 0, pi/12, pi/6}
        x[2] = c2 == 0 ? 0.05f : (c2 == 1 ? 0.0f : -0.05f);
#pragma unroll
        for (int k = 0; k < 9; ++k) x[3 + k] = ob[k];
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) act0[k * LS + row] = x[k];
    } else if (part == 1) {
#pragma unroll
      for (int k = 12; k < 16; ++k) act0[k * LS + row] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) hbuf[(part * 16 + f) * LS + row] = a.H[(size_t)(part * 16 + f) * a.N + n];
  }
  __syncthreads();

  // ---- MLP layer 1: Linear(12, 128) + ReLU + LayerNorm; wave w owns output columns 32 w .. 32 w + 31
  {
    float A[8];
    load_a<16>(act0, lane, A);
    floatx16 acc = splat(W[D_B1 + w * 32 + col]);
    mma_tile<16>(W + D_W1 + w * tile_floats(16), lane, A, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) act1[(w * 32 + col) * LS + c_row(r, lane)] = fmaxf(acc[r], 0.0f);
  }
  __syncthreads();
  layer_norm(act1, red, W + D_G1, W + D_BE1, tid);
  // ---- MLP layer 2
  {
    float A[64];
    load_a<HID>(act1, lane, A);
    floatx16 acc = splat(W[D_B2 + w * 32 + col]);
    mma_tile<HID>(W + D_W2 + w * tile_floats(HID), lane, A, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) act0[(w * 32 + col) * LS + c_row(r, lane)] = fmaxf(acc[r], 0.0f);   // act0's inputs were consumed before the last barriers
  }
  __syncthreads();
  layer_norm(act0, red, W + D_G2, W + D_BE2, tid);
  // ---- GRU cell (torch gate order r, z, n): wave w owns hidden units 32 w .. 32 w + 31, i.e. gate tiles w, 4 + w, 8 + w
  {
    floatx16 ir = splat(W[D_BIH + 0 * 128 + w * 32 + col]), iz = splat(W[D_BIH + 1 * 128 + w * 32 + col]), in_ = splat(W[D_BIH + 2 * 128 + w * 32 + col]);
    floatx16 hr = splat(W[D_BHH + 0 * 128 + w * 32 + col]), hz = splat(W[D_BHH + 1 * 128 + w * 32 + col]), hn = splat(W[D_BHH + 2 * 128 + w * 32 + col]);
    {
      float A[64];
      load_a<HID>(act0, lane, A);
      mma_tile<HID>(W + D_WIH + (0 + w) * tile_floats(HID), lane, A, ir);
      mma_tile<HID>(W + D_WIH + (4 + w) * tile_floats(HID), lane, A, iz);
      mma_tile<HID>(W + D_WIH + (8 + w) * tile_floats(HID), lane, A, in_);
    }
    {
      float A[64];
      load_a<HID>(hbuf, lane, A);
      mma_tile<HID>(W + D_WHH + (0 + w) * tile_floats(HID), lane, A, hr);
      mma_tile<HID>(W + D_WHH + (4 + w) * tile_floats(HID), lane, A, hz);
      mma_tile<HID>(W + D_WHH + (8 + w) * tile_floats(HID), lane, A, hn);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = c_row(r, lane), unit = w * 32 + col;
      const float rg = sigmoid_f(ir[r] + hr[r]);
      const float zg = sigmoid_f(iz[r] + hz[r]);
      const float ng = tanh_f(in_[r] + rg * hn[r]);
      const float hnew = (1.0f - zg) * ng + zg * hbuf[unit * LS + row];
      act1[unit * LS + row] = hnew;
    }
  }
  __syncthreads();
  {   // the new hidden state goes out row-contiguous (128-byte runs per feature) from LDS; the tile registers above would scatter it
      // one float per cache line. Thread = (row, 16-feature part): the very elements this thread normalises next.
    const int row = tid & 31, part = tid >> 5, n = i0 + row;
    if (n < a.N) {
#pragma unroll
      for (int f = 0; f < 16; ++f) a.H[(size_t)(part * 16 + f) * a.N + n] = act1[(part * 16 + f) * LS + row];
    }
  }
  layer_norm(act1, red, W + D_G3, W + D_BE3, tid);
  // ---- heads: 153 logits = five column tiles; wave w takes tile w and a quarter of the fifth tile's K range
  {
    float A[64];
    load_a<HID>(act1, lane, A);
    floatx16 acc = splat(W[D_BA + w * 32 + col]);
    mma_tile<HID>(W + D_WA + w * tile_floats(HID), lane, A, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) lg[(w * 32 + col) * LS + c_row(r, lane)] = acc[r];
    // fifth tile (logits 128 .. 152): its K range is split over the four waves (16 MFMAs each instead of 64 on one wave); the
    // partial sums go to act0 (free by now) and are added in a fixed order below
    {
      floatx16 part = splat(0.0f);
      const float4* t4 = reinterpret_cast<const float4*>(W + D_WA + 4 * tile_floats(HID)) + lane + (size_t)(4 * w) * 64;
      float Aw[16];   // this wave's K slice (k = 32 w .. 32 w + 31), read again from LDS: indexing A[] by w would put it in scratch
#pragma unroll
      for (int t = 0; t < 16; ++t) Aw[t] = act1[(2 * (16 * w + t) + (lane >> 5)) * LS + (lane & 31)];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = t4[g * 64];
        part = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[4 * g + 0], b.x, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[4 * g + 1], b.y, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[4 * g + 2], b.z, part, 0, 0, 0);
        part = __builtin_amdgcn_mfma_f32_32x32x2f32(Aw[4 * g + 3], b.w, part, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) act0[(w * 32 + col) * LS + c_row(r, lane)] = part[r];
    }
  }
  __syncthreads();
  // logits 128 .. 152 = bias + the four K-partials, summed in a fixed order (25 columns x 32 aircraft over 256 threads)
  for (int e = tid; e < 25 * 32; e += 256) {
    const int q = e >> 5, row = e & 31;
    lg[(128 + q) * LS + row] = (((W[D_BA + 128 + q] + act0[q * LS + row]) + act0[(32 + q) * LS + row]) + act0[(64 + q) * LS + row]) + act0[(96 + q) * LS + row];
  }
  __syncthreads();
  if (tid < 128) {   // thread = (head, aircraft): first maximum, like torch argmax
    const int head = tid >> 5, row = tid & 31;
    const int off = head * 41, cnt = (head == 3) ? 30 : 41;
    float best = lg[off * LS + row];
    int bi = 0;
    for (int j = 1; j < cnt; ++j) {
      const float v = lg[(off + j) * LS + row];
      if (v > best) { best = v; bi = j; }
    }
    if (i0 + row < a.N) a.low[(size_t)(i0 + row) * a.act_low + head] = (float)bi;
  } else if (tid < 160) {   // weapon bits ride along unchanged
    const int row = tid & 31;
    if (i0 + row < a.N) {
      const int nn = i0 + row;
      const bool scripted = a.use_baseline && (nn % a.A) >= a.n_ego;   // scenario1_task.py:42-48: bits [0,0,0,0], or all ones with artillery
      for (int k = 4; k < a.act_low; ++k)
        a.low[(size_t)nn * a.act_low + k] = scripted ? (a.use_artillery ? 1.0f : 0.0f) : a.hi[(size_t)nn * a.act_hi + (k - 1)];
    }
  }
}
