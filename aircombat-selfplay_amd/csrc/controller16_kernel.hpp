// The low-level controller on 16-aircraft workgroups (included by aircombat.hip after controller_kernel.hpp, whose Args, weight
// blob source layout and helpers it shares).
//
// controller_kernel works on 32 aircraft per workgroup with v_mfma_f32_32x32x2_f32 tiles: at the BASELINE batch (8192 aircraft) that
// is 256 workgroups x 4 waves = ONE wave per SIMD, and everything that is not an MFMA -- staging the GRU state, three LayerNorms
// with their barriers, the gate algebra, the argmax -- leaves the matrix pipe idle (round 1: 27.9 us, MFMA busy 50 %). Here a
// workgroup is 16 aircraft on v_mfma_f32_16x16x4_f32 tiles (the same 64 flop per cycle and SIMD): twice the workgroups, two of them
// resident per CU, two waves per SIMD -- one workgroup's MFMA streams run under the other's LayerNorm / staging / argmax phases.
// fp32 products and accumulation in k order as before (the argmax indices must match the reference's float32 GEMV).
//
// Operand layouts of v_mfma_f32_16x16x4_f32 (lane i of 64): A[row = i % 16][k = i / 16], B[k = i / 16][col = i % 16],
// C/D 4 registers: [row = 4 (i / 16) + r][col = i % 16].
// Activations live in LDS as [k % 4][row][k / 4] (row stride 36 floats): lane (row, kq) reads the A operands of four consecutive
// MFMAs -- k = 4 t + kq, t .. t + 3 -- with ONE ds_read_b128, conflict-free over the 16-lane phases.
// Weights are pre-tiled on the host: column tile c (16 outputs), K group g (4 MFMAs): element (lane, q) = W[j = 16 c + lane % 16]
// [k = 16 g + 4 q + lane / 16], so one coalesced 16-byte load per lane feeds four MFMAs.
#pragma once

namespace ctl16 {
using ctl::HID; using ctl::NH; using ctl::NHP;
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int MT = 16;                       // aircraft per workgroup
constexpr int TS = 36;                       // floats per (kq, row) run of an activation buffer: 32 t + 4 pad (16-byte aligned, bank-staggered)
constexpr int ACT = 4 * MT * TS;             // floats of one [128 features][16 rows] activation buffer
constexpr int tile_floats(int K) { return (K / 16) * 64 * 4; }
enum : int {
  E_W1 = 0,                                  // K = 16 (12 padded), 8 column tiles
  E_W2 = E_W1 + 8 * tile_floats(16),         // K = 128, 8 tiles
  E_WIH = E_W2 + 8 * tile_floats(128),       // 24 tiles (r0..7, z0..7, n0..7)
  E_WHH = E_WIH + 24 * tile_floats(128),     // 24 tiles
  E_WA = E_WHH + 24 * tile_floats(128),      // 10 tiles (columns 153..159 zero)
  E_B1 = E_WA + 10 * tile_floats(128), E_G1 = E_B1 + 128, E_BE1 = E_G1 + 128,
  E_B2 = E_BE1 + 128, E_G2 = E_B2 + 128, E_BE2 = E_G2 + 128,
  E_BIH = E_BE2 + 128, E_BHH = E_BIH + 384, E_G3 = E_BHH + 384, E_BE3 = E_G3 + 128,
  E_BA = E_BE3 + 128,                        // [160]
  E_END = E_BA + NHP
};
__device__ __forceinline__ int act_index(int k, int row) { return ((k & 3) * MT + row) * TS + (k >> 2); }

// A operands of a K-wide layer for this lane: K / 16 float4s, a[g] = act[k = 4 (4 g + q) + kq][row], q = 0 .. 3
template <int K>
__device__ __forceinline__ void load_a(const float* act, int lane, float4 (&A)[K / 16]) {
  const float4* p = reinterpret_cast<const float4*>(act + ((lane >> 4) * MT + (lane & 15)) * TS);
#pragma unroll
  for (int g = 0; g < K / 16; ++g) A[g] = p[g];
}
template <int K>
__device__ __forceinline__ void mma_tile(const float* __restrict__ tile, int lane, const float4 (&A)[K / 16], floatx4& acc) {
  const float4* t4 = reinterpret_cast<const float4*>(tile) + lane;
#pragma unroll
  for (int g = 0; g < K / 16; ++g) {
    const float4 b = t4[g * 64];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, b.w, acc, 0, 0, 0);
  }
}
// One column tile's weights in registers (K = 128: eight 16-byte vectors per lane) so that the NEXT tile's loads are in flight
// while this tile's 32 MFMAs run: a 16 x 16 tile is only ~1000 cycles of matrix work, about one L2 round trip
struct BTile { float4 b[8]; };
__device__ __forceinline__ void load_b(const float* __restrict__ tile, int lane, BTile& B) {
  const float4* t4 = reinterpret_cast<const float4*>(tile) + lane;
#pragma unroll
  for (int g = 0; g < 8; ++g) B.b[g] = t4[g * 64];
}
__device__ __forceinline__ void mma_b(const BTile& B, const float4 (&A)[8], floatx4& acc) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B.b[g].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B.b[g].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, B.b[g].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, B.b[g].w, acc, 0, 0, 0);
  }
}
// two independent tiles interleaved (their accumulators do not depend on each other: the matrix pipe never waits for its own result)
__device__ __forceinline__ void mma_b2(const BTile& B0, const BTile& B1, const float4 (&A)[8], floatx4& acc0, floatx4& acc1) {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B0.b[g].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].x, B1.b[g].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B0.b[g].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].y, B1.b[g].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, B0.b[g].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].z, B1.b[g].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, B0.b[g].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[g].w, B1.b[g].w, acc1, 0, 0, 0);
  }
}
__device__ __forceinline__ floatx4 splat(float v) { floatx4 a; a[0] = v; a[1] = v; a[2] = v; a[3] = v; return a; }
__device__ __forceinline__ int c_row(int r, int lane) { return (lane >> 4) * 4 + r; }

// torch.nn.LayerNorm(128) (eps 1e-5, biased variance) of buf[feature][row] in place; 256 threads: thread = (row, 8-feature part)
__device__ __forceinline__ void layer_norm(float* buf, float* red, const float* __restrict__ g, const float* __restrict__ b, int tid) {
  const int row = tid & 15, part = tid >> 4;
  float x[8], gg[8], bb[8];
  {
    const float4* g4 = reinterpret_cast<const float4*>(g + part * 8);
    const float4* b4 = reinterpret_cast<const float4*>(b + part * 8);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 gv = g4[q], bv = b4[q];
      gg[4 * q] = gv.x; gg[4 * q + 1] = gv.y; gg[4 * q + 2] = gv.z; gg[4 * q + 3] = gv.w;
      bb[4 * q] = bv.x; bb[4 * q + 1] = bv.y; bb[4 * q + 2] = bv.z; bb[4 * q + 3] = bv.w;
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int f = 0; f < 8; ++f) { x[f] = buf[act_index(part * 8 + f, row)]; s += x[f]; }
  red[part * 17 + row] = s;
  __syncthreads();
  float m = 0.0f;
#pragma unroll
  for (int p = 0; p < 16; ++p) m += red[p * 17 + row];
  m *= (1.0f / HID);
  float v = 0.0f;
#pragma unroll
  for (int f = 0; f < 8; ++f) { const float d = x[f] - m; v = fmaf(d, d, v); }
  red[(16 + part) * 17 + row] = v;
  __syncthreads();
  float var = 0.0f;
#pragma unroll
  for (int p = 0; p < 16; ++p) var += red[(16 + p) * 17 + row];
  const float is = rsqrtf(var * (1.0f / HID) + 1e-5f);
#pragma unroll
  for (int f = 0; f < 8; ++f) buf[act_index(part * 8 + f, row)] = fmaf((x[f] - m) * is, gg[f], bb[f]);
  __syncthreads();
}
}  // namespace ctl16

__global__ __launch_bounds__(256, 2) void controller16_kernel(ctl::Args a) {
  using namespace ctl16;
  using ctl::sigmoid_f; using ctl::tanh_f;
  __shared__ __attribute__((aligned(16))) float act0[ACT];
  __shared__ __attribute__((aligned(16))) float act1[ACT];
  __shared__ __attribute__((aligned(16))) float hbuf[ACT];   // GRU state of the 16 aircraft
  __shared__ float lg[NHP * 17];                              // head logits [logit][row]
  __shared__ float red[32 * 17];                              // LayerNorm partial sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * MT;
  const float* __restrict__ W = a.W16;
  const int col = lane & 15;

  // ---- stage: the GRU state (HBM, feature-major [128][N]: 64-byte runs per feature) and the 12 controller inputs
  {
    const int row = tid & 15, part = tid >> 4;   // 16 parts x 8 features
    const int n = min(i0 + row, a.N - 1);
    float hv[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) hv[f] = a.H[(size_t)(part * 8 + f) * a.N + n];
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
      for (int k = 0; k < 12; ++k) act0[act_index(k, row)] = x[k];
#pragma unroll
      for (int k = 12; k < 16; ++k) act0[act_index(k, row)] = 0.0f;   // K = 16 pad
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) hbuf[act_index(part * 8 + f, row)] = hv[f];
  }
  __syncthreads();

  // ---- MLP layer 1: Linear(12, 128) + ReLU + LayerNorm; wave w owns output columns 32 w .. 32 w + 31 = column tiles 2 w, 2 w + 1
  {
    float4 A[1];
    load_a<16>(act0, lane, A);
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
      const int c = 2 * w + tc;
      floatx4 acc = splat(W[E_B1 + c * 16 + col]);
      mma_tile<16>(W + E_W1 + c * tile_floats(16), lane, A, acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) act1[act_index(c * 16 + col, c_row(r, lane))] = fmaxf(acc[r], 0.0f);
    }
  }
  __syncthreads();
  layer_norm(act1, red, W + E_G1, W + E_BE1, tid);
  // ---- MLP layer 2
  {
    BTile B0, B1;
    load_b(W + E_W2 + (2 * w) * tile_floats(HID), lane, B0);
    load_b(W + E_W2 + (2 * w + 1) * tile_floats(HID), lane, B1);
    float4 A[8];
    load_a<HID>(act1, lane, A);
    floatx4 acc0 = splat(W[E_B2 + (2 * w) * 16 + col]), acc1 = splat(W[E_B2 + (2 * w + 1) * 16 + col]);
    mma_b2(B0, B1, A, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // act0's inputs were consumed before the last barriers
      act0[act_index((2 * w) * 16 + col, c_row(r, lane))] = fmaxf(acc0[r], 0.0f);
      act0[act_index((2 * w + 1) * 16 + col, c_row(r, lane))] = fmaxf(acc1[r], 0.0f);
    }
  }
  __syncthreads();
  layer_norm(act0, red, W + E_G2, W + E_BE2, tid);
  // ---- GRU cell (torch gate order r, z, n): wave w owns hidden units 32 w .. 32 w + 31 = unit tiles 2 w, 2 w + 1 of each gate
  {
    floatx4 ir[2], iz[2], in_[2], hr[2], hz[2], hn[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) {
      const int u = (2 * w + tc) * 16 + col;
      ir[tc] = splat(W[E_BIH + 0 * 128 + u]); iz[tc] = splat(W[E_BIH + 1 * 128 + u]); in_[tc] = splat(W[E_BIH + 2 * 128 + u]);
      hr[tc] = splat(W[E_BHH + 0 * 128 + u]); hz[tc] = splat(W[E_BHH + 1 * 128 + u]); hn[tc] = splat(W[E_BHH + 2 * 128 + u]);
    }
    {
      // twelve tiles in six interleaved pairs (the two unit tiles of a gate), each pair's weights requested while the pair before runs
      const int c0 = 2 * w, c1 = 2 * w + 1, TF = tile_floats(HID);
      BTile P0, P1, Q0, Q1;
      float4 Ai[8], Ah[8];
      load_b(W + E_WIH + (0 + c0) * TF, lane, P0); load_b(W + E_WIH + (0 + c1) * TF, lane, P1);
      load_a<HID>(act0, lane, Ai);
      load_b(W + E_WIH + (8 + c0) * TF, lane, Q0); load_b(W + E_WIH + (8 + c1) * TF, lane, Q1);
      mma_b2(P0, P1, Ai, ir[0], ir[1]);
      load_b(W + E_WIH + (16 + c0) * TF, lane, P0); load_b(W + E_WIH + (16 + c1) * TF, lane, P1);
      mma_b2(Q0, Q1, Ai, iz[0], iz[1]);
      load_b(W + E_WHH + (0 + c0) * TF, lane, Q0); load_b(W + E_WHH + (0 + c1) * TF, lane, Q1);
      load_a<HID>(hbuf, lane, Ah);
      mma_b2(P0, P1, Ai, in_[0], in_[1]);
      load_b(W + E_WHH + (8 + c0) * TF, lane, P0); load_b(W + E_WHH + (8 + c1) * TF, lane, P1);
      mma_b2(Q0, Q1, Ah, hr[0], hr[1]);
      load_b(W + E_WHH + (16 + c0) * TF, lane, Q0); load_b(W + E_WHH + (16 + c1) * TF, lane, Q1);
      mma_b2(P0, P1, Ah, hz[0], hz[1]);
      mma_b2(Q0, Q1, Ah, hn[0], hn[1]);
    }
#pragma unroll
    for (int tc = 0; tc < 2; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = c_row(r, lane), unit = (2 * w + tc) * 16 + col;
        const float rg = sigmoid_f(ir[tc][r] + hr[tc][r]);
        const float zg = sigmoid_f(iz[tc][r] + hz[tc][r]);
        const float ng = tanh_f(in_[tc][r] + rg * hn[tc][r]);
        act1[act_index(unit, row)] = (1.0f - zg) * ng + zg * hbuf[act_index(unit, row)];
      }
  }
  __syncthreads();
  {   // the new hidden state goes out in 64-byte runs per feature from LDS. Thread = (row, 8-feature part): the elements it normalises next
    const int row = tid & 15, part = tid >> 4, n = i0 + row;
    if (n < a.N) {
#pragma unroll
      for (int f = 0; f < 8; ++f) a.H[(size_t)(part * 8 + f) * a.N + n] = act1[act_index(part * 8 + f, row)];
    }
  }
  layer_norm(act1, red, W + E_G3, W + E_BE3, tid);
  // ---- heads: 153 logits = ten column tiles; wave w takes tiles w and 4 + w, and waves 0 / 1 tiles 8 / 9
  {
    const int TF = tile_floats(HID);
    BTile B0, B1, B2;
    load_b(W + E_WA + w * TF, lane, B0);
    load_b(W + E_WA + (4 + w) * TF, lane, B1);
    if (w < 2) load_b(W + E_WA + (8 + w) * TF, lane, B2);
    float4 A[8];
    load_a<HID>(act1, lane, A);
    floatx4 acc0 = splat(W[E_BA + w * 16 + col]), acc1 = splat(W[E_BA + (4 + w) * 16 + col]);
    mma_b2(B0, B1, A, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lg[(w * 16 + col) * 17 + c_row(r, lane)] = acc0[r];
      lg[((4 + w) * 16 + col) * 17 + c_row(r, lane)] = acc1[r];
    }
    if (w < 2) {
      floatx4 acc2 = splat(W[E_BA + (8 + w) * 16 + col]);
      mma_b(B2, A, acc2);
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[((8 + w) * 16 + col) * 17 + c_row(r, lane)] = acc2[r];
    }
  }
  __syncthreads();
  if (tid < 64) {   // thread = (head, aircraft): first maximum, like torch argmax
    const int head = tid >> 4, row = tid & 15;
    const int off = head * 41, cnt = (head == 3) ? 30 : 41;
    float best = lg[off * 17 + row];
    int bi = 0;
    for (int j = 1; j < cnt; ++j) {
      const float v = lg[(off + j) * 17 + row];
      if (v > best) { best = v; bi = j; }
    }
    if (i0 + row < a.N) a.low[(size_t)(i0 + row) * a.act_low + head] = (float)bi;
  } else if (tid < 80) {   // weapon bits ride along unchanged
    const int row = tid & 15;
    if (i0 + row < a.N) {
      const int nn = i0 + row;
      const bool scripted = a.use_baseline && (nn % a.A) >= a.n_ego;   // scenario1_task.py:42-48: bits [0,0,0,0], or all ones with artillery
      for (int k = 4; k < a.act_low; ++k)
        a.low[(size_t)nn * a.act_low + k] = scripted ? (a.use_artillery ? 1.0f : 0.0f) : a.hi[(size_t)nn * a.act_hi + (k - 1)];
    }
  }
}
