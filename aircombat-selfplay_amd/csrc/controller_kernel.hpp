// Low-level controller of the hierarchical tasks on the device: the reference's BaselineActor
// (envs/JSBSim/model/baseline_actor.py:12-110: MLP 12->128->128 with ReLU+LayerNorm, GRU 128, LayerNorm, four argmax heads
// [41,41,41,30]) evaluated for every aircraft once per env step, driven exactly like
// HierarchicalSingleCombatTask.normalize_action (tasks/singlecombat_task.py:223-256): inputs = the [3,5,3] choice mapped to
// (delta altitude, delta heading, delta speed) + the first nine values of the aircraft's current observation; the four argmax
// indices become the control indices the step kernel decodes. Included by aircombat.hip.
//
// Mapping: one workgroup = 64 aircraft x 8 waves. In every wave lane l is aircraft l, so LayerNorm, the GRU gate algebra and
// argmax are lane-local; the waves split the OUTPUT neurons of each layer (16 of 128, 3 x 16 of 384, 20 of 153). A wave's
// weights are wave-uniform: they arrive through the scalar cache (s_load_dwordx16 of a transposed [k][j] matrix) and feed
// v_fmac_f32 as SGPR operands; the 128 inputs of a layer sit in VGPRs, activations cross waves through LDS as [feature][lane]
// (bank-conflict free). fp32 FMA chains in k order: same arithmetic as an fp32 GEMV, deterministic.
#pragma once

namespace ctl {
constexpr int HID = 128, NH = 153, NHP = 160;   // hidden width, stacked head outputs, padded to 8 waves x 20
// device blob (floats): every matrix transposed to [k][j] so that 16 consecutive outputs of one input are one 64-byte scalar load
enum : int {
  D_W1T = 0,                        // [12][128]
  D_B1 = D_W1T + 12 * 128, D_G1 = D_B1 + 128, D_BE1 = D_G1 + 128,
  D_W2T = D_BE1 + 128,              // [128][128]
  D_B2 = D_W2T + 128 * 128, D_G2 = D_B2 + 128, D_BE2 = D_G2 + 128,
  D_WIHT = D_BE2 + 128,             // [128][384]
  D_WHHT = D_WIHT + 128 * 384,      // [128][384]
  D_BIH = D_WHHT + 128 * 384, D_BHH = D_BIH + 384,
  D_G3 = D_BHH + 384, D_BE3 = D_G3 + 128,
  D_WAT = D_BE3 + 128,              // [128][160] (columns 153..159 zero)
  D_BA = D_WAT + 128 * NHP,         // [160]
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
  const float* W;          // device blob
  const float* hi;         // [N][act_hi]: 3 high-level choices (+ weapon bits passed through)
  const float* obs;        // [N][obs_dim]: observation of the CURRENT state (last step's / the reset's output)
  float* H;                // [128][N] GRU state
  float* low;              // [N][act_low] out: 4 control indices (+ the weapon bits)
  int N, obs_dim, act_hi, act_low;
};

struct W16 { float v[16]; };
struct W4 { float v[4]; };

// acc[0..NJ) += sum_k WT[k][j0 + j] * x[k]; WT row length J. Wave-uniform addresses: scalar loads.
template <int K, int NJ>
__device__ __forceinline__ void gemv_slice(const float* __restrict__ WT, int J, int j0, const float (&x)[K], float (&acc)[NJ]) {
  static_assert(NJ % 4 == 0, "slices are multiples of 4 outputs");
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float* row = WT + k * J + j0;
#pragma unroll
    for (int b = 0; b < NJ; b += 16) {
      if (b + 16 <= NJ) {
        const W16 w = *reinterpret_cast<const W16*>(row + b);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[b + j] = fmaf(w.v[j], x[k], acc[b + j]);
      } else {
#pragma unroll
        for (int q = b; q < NJ; q += 4) {
          const W4 w = *reinterpret_cast<const W4*>(row + q);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q + j] = fmaf(w.v[j], x[k], acc[q + j]);
        }
      }
    }
  }
}

// torch.nn.LayerNorm(128), eps 1e-5, biased variance; g / b wave-uniform
__device__ __forceinline__ void layer_norm(float (&x)[HID], const float* __restrict__ g, const float* __restrict__ b) {
  float m = 0.0f;
#pragma unroll
  for (int i = 0; i < HID; ++i) m += x[i];
  m *= (1.0f / HID);
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < HID; ++i) { const float d = x[i] - m; v = fmaf(d, d, v); }
  const float is = rsqrtf(v * (1.0f / HID) + 1e-5f);
#pragma unroll
  for (int i = 0; i < HID; ++i) x[i] = fmaf((x[i] - m) * is, g[i], b[i]);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

__device__ __forceinline__ void lds_get(const float* buf, int lane, float (&x)[HID]) {
#pragma unroll
  for (int k = 0; k < HID; ++k) x[k] = buf[k * 64 + lane];
}
}  // namespace ctl

__global__ __launch_bounds__(512) void controller_kernel(ctl::Args a) {
  using namespace ctl;
  __shared__ float bufA[NHP * 64];   // activations / head logits, [feature][lane]
  __shared__ float bufB[HID * 64];
  __shared__ float bufH[HID * 64];   // GRU state of the 64 aircraft
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index, kept scalar so that weight addresses are uniform
  const int n = blockIdx.x * 64 + lane;
  const bool live = n < a.N;
  const int nn = live ? n : a.N - 1;
  const float* __restrict__ W = a.W;

  // ---- inputs: wave 0 builds the 12 controller inputs, every wave loads 16 features of the GRU state
  if (w == 0) {
    const float* hi = a.hi + (size_t)nn * a.act_hi;
    const float* ob = a.obs + (size_t)nn * a.obs_dim;
    const int c0 = (int)hi[0], c1 = (int)hi[1], c2 = (int)hi[2];
    // singlecombat_task.py:217-219, 235-241: below 3500 m the altitude choice is overridden by "climb"
    const float d_alt = (ob[0] * 5000.0f < 3500.0f) ? 0.1f : (c0 == 0 ? 0.1f : (c0 == 1 ? 0.0f : -0.1f));
    const float d_hdg = (float)(c1 - 2) * 0.26179938779914943f;   // {-pi/6, -pi/12, 0, pi/12, pi/6}
    const float d_vel = c2 == 0 ? 0.05f : (c2 == 1 ? 0.0f : -0.05f);
    bufA[0 * 64 + lane] = d_alt; bufA[1 * 64 + lane] = d_hdg; bufA[2 * 64 + lane] = d_vel;
#pragma unroll
    for (int k = 0; k < 9; ++k) bufA[(3 + k) * 64 + lane] = ob[k];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) bufH[(w * 16 + k) * 64 + lane] = a.H[(size_t)(w * 16 + k) * a.N + nn];
  __syncthreads();

  float x[HID];
  // ---- MLP layer 1: Linear(12, 128) + ReLU, then LayerNorm
  {
    float x12[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) x12[k] = bufA[k * 64 + lane];
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = W[D_B1 + w * 16 + j];
    gemv_slice<12, 16>(W + D_W1T, 128, w * 16, x12, acc);
#pragma unroll
    for (int j = 0; j < 16; ++j) bufB[(w * 16 + j) * 64 + lane] = fmaxf(acc[j], 0.0f);
  }
  __syncthreads();
  lds_get(bufB, lane, x);
  layer_norm(x, W + D_G1, W + D_BE1);
  // ---- MLP layer 2
  {
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = W[D_B2 + w * 16 + j];
    gemv_slice<HID, 16>(W + D_W2T, 128, w * 16, x, acc);
#pragma unroll
    for (int j = 0; j < 16; ++j) bufA[(w * 16 + j) * 64 + lane] = fmaxf(acc[j], 0.0f);
  }
  __syncthreads();
  lds_get(bufA, lane, x);
  layer_norm(x, W + D_G2, W + D_BE2);
  // ---- GRU cell (torch gate order r, z, n): this wave owns hidden units w*16 .. w*16+15
  float hnew[16];
  {
    float gi[48], gh[48];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int j = 0; j < 16; ++j) { gi[g * 16 + j] = W[D_BIH + g * 128 + w * 16 + j]; gh[g * 16 + j] = W[D_BHH + g * 128 + w * 16 + j]; }
    // three 16-wide slices of the 384 gate rows: columns g*128 + w*16 + j
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float (&acc)[16] = *reinterpret_cast<float (*)[16]>(&gi[g * 16]);
      gemv_slice<HID, 16>(W + D_WIHT, 384, g * 128 + w * 16, x, acc);
    }
    float h[HID];
    lds_get(bufH, lane, h);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float (&acc)[16] = *reinterpret_cast<float (*)[16]>(&gh[g * 16]);
      gemv_slice<HID, 16>(W + D_WHHT, 384, g * 128 + w * 16, h, acc);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float r = sigmoid_f(gi[j] + gh[j]);
      const float z = sigmoid_f(gi[16 + j] + gh[16 + j]);
      const float nn_ = tanh_f(gi[32 + j] + r * gh[32 + j]);
      const float hold = bufH[(w * 16 + j) * 64 + lane];
      hnew[j] = (1.0f - z) * nn_ + z * hold;
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    bufB[(w * 16 + j) * 64 + lane] = hnew[j];
    if (live) a.H[(size_t)(w * 16 + j) * a.N + n] = hnew[j];
  }
  __syncthreads();
  lds_get(bufB, lane, x);
  layer_norm(x, W + D_G3, W + D_BE3);
  // ---- heads: 153 logits split 8 x 20 (padded columns are zero weights)
  {
    float acc[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) acc[j] = W[D_BA + w * 20 + j];
    gemv_slice<HID, 20>(W + D_WAT, NHP, w * 20, x, acc);
#pragma unroll
    for (int j = 0; j < 20; ++j) bufA[(w * 20 + j) * 64 + lane] = acc[j];
  }
  __syncthreads();
  if (w < 4) {   // wave hd scans head hd of its 64 aircraft: first maximum, like torch argmax
    const int off = w * 41, cnt = (w == 3) ? 30 : 41;
    float best = bufA[off * 64 + lane];
    int bi = 0;
    for (int j = 1; j < cnt; ++j) {
      const float v = bufA[(off + j) * 64 + lane];
      if (v > best) { best = v; bi = j; }
    }
    if (live) a.low[(size_t)n * a.act_low + w] = (float)bi;
  } else if (w == 4 && live) {   // weapon bits ride along unchanged
    for (int k = 4; k < a.act_low; ++k) a.low[(size_t)n * a.act_low + k] = a.hi[(size_t)n * a.act_hi + (k - 1)];
  }
}
