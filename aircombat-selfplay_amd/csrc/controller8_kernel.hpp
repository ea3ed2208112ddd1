// The low-level controller kernel, eight-wave form (network, arguments: controller_common.hpp; the piece arithmetic -- every fp32
// product as three exact fp16 x fp16 terms accumulated in fp32 -- and its helpers: controller_pieces.hpp; the four-wave kernel of rounds
// 2-3 that this one replaced on every grid is in the history).
//
// Why eight waves. The four-wave kernel put ONE wave on each SIMD of a CU: 378 registers of weight prefetch per wave, and a wave issues
// in order -- so every weight load, every LDS read and every LayerNorm / gate / argmax instruction was time the SIMD's matrix pipe stood
// idle (round 3's counters: pipe busy 12.9 k of a wave's 38.8 k cycles; sharing a weight stream between two tiles, two workgroups per CU
// in lockstep and a software pipeline of two tiles all measured within 3 % of it). Here a workgroup is still one 32-aircraft tile, but
// EIGHT waves, two per SIMD, each owning 16 of a layer's output columns (v_mfma_f32_16x16x32_f16: M = 16 aircraft x N = 16 columns x
// K = 32 per instruction, two M-tiles per wave): the two waves of a SIMD run the same phase on different columns, so one's load issue and
// LDS waits sit under the other's matrix instructions, and the vector phases (LayerNorm, gate algebra, argmax, staging) are spread over
// twice the lanes. A operands are read from the LDS planes per k-step (two M-tiles x two pieces = four ds_read_b128 per 18 matrix
// instructions of a GRU k-step) instead of living in registers, weight pieces stream through a three-stage ring of one k-step each:
// under 200 registers, no scratch.
//
// Weight tiles for this form (ac_load_controller): tile(c, K) = the 16 output columns 16 c .. 16 c + 15 of a layer = K/32 k-steps x
// 2 pieces x 64 lanes x 8 fp16; element (s, p, lane, i) = piece p of W[j = 16 c + lane % 16][k = 32 s + 8 (lane / 16) + i] (the B
// operand map of the 16x16x32 instruction: lane l holds B[k = 8 (l >> 4) + i][col = l & 15]).
#pragma once

namespace ctl8 {
using ctl::HID; using ctl::NH; using ctl::NHP; using ctl::MT; using ctl::LS;
using ctls::KS; using ctls::RS;
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int NP = 2;                    // pieces per value
constexpr int tile_floats(int K) { return (K / 32) * NP * 64 * 4; }   // in floats (a uint4 = 8 fp16 = 4 floats)
enum : int {
  C_W1 = 0,                                  // K = 32 (12 padded), 8 tiles
  C_W2 = C_W1 + 8 * tile_floats(32),         // K = 128, 8 tiles
  C_WIH = C_W2 + 8 * tile_floats(128),       // 24 tiles: gate g (r, z, n), unit tile u -> tile 8 g + u
  C_WHH = C_WIH + 24 * tile_floats(128),     // 24 tiles
  C_WA = C_WHH + 24 * tile_floats(128),      // 10 tiles (columns 153..159 zero)
  C_B1 = C_WA + 10 * tile_floats(128), C_G1 = C_B1 + 128, C_BE1 = C_G1 + 128,
  C_B2 = C_BE1 + 128, C_G2 = C_B2 + 128, C_BE2 = C_G2 + 128,
  C_BIH = C_BE2 + 128, C_BHH = C_BIH + 384, C_G3 = C_BHH + 384, C_BE3 = C_G3 + 128,
  C_BA = C_BE3 + 128,                        // [160]
  C_END = C_BA + NHP
};
__device__ __forceinline__ floatx4 splat4(float v) { floatx4 a = {v, v, v, v}; return a; }
__device__ __forceinline__ floatx4 mf(const uint4& a, const uint4& b, floatx4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ctls::f16x8, a), __builtin_bit_cast(ctls::f16x8, b), acc, 0, 0, 0);
}
// result layout of a 16x16 tile: acc[i] is (row = 4 (lane / 16) + i, column = lane % 16)
__device__ __forceinline__ int c_row(int mt, int i, int lane) { return 16 * mt + 4 * (lane >> 4) + i; }

// A operands of one k-step for this lane: [M-tile][piece] = planes[piece][row = 16 mt + lane % 16][k = 32 s + 8 (lane / 16) .. + 7].
// MTL = M-tiles per wave: 2 (32 aircraft per workgroup) or 4 (64: the grids with more tiles than CUs, where a workgroup's fixed costs --
// first round trip, barriers, the latency-bound LayerNorm / gate / argmax phases -- are shared by twice the aircraft and every weight
// piece loaded feeds twice the matrix instructions).
template <int MTL>
struct Geo8 {
  static constexpr int R = 16 * MTL;               // aircraft per workgroup
  static constexpr int PLN = R * KS;               // fp16 per plane
  static constexpr int LSR = R + 1;                // row stride of the [feature][aircraft] buffers (odd: column writes spread over the banks)
  static constexpr int TPR = 512 / R;              // threads per aircraft in the row-wise phases (16 or 8)
  static constexpr int FPT = HID / TPR;            // features per thread there (8 or 16)
};
template <int MTL>
struct AF { uint4 a[MTL][NP]; };
template <int MTL>
__device__ __forceinline__ void load_af(const unsigned short* planes, int lane, int s, AF<MTL>& A) {
  const unsigned short* base = planes + (lane & 15) * KS + 8 * (lane >> 4) + 32 * s;
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
    for (int p = 0; p < NP; ++p) A.a[mt][p] = *reinterpret_cast<const uint4*>(base + p * Geo8<MTL>::PLN + 16 * mt * KS);
}
struct BS { uint4 b[NP]; };   // one k-step of one 16-column tile: the two pieces
__device__ __forceinline__ void load_bs(const uint4* __restrict__ t4 /* tile + lane */, int s, BS& B) {
#pragma unroll
  for (int p = 0; p < NP; ++p) B.b[p] = t4[(s * NP + p) * 64];
}
template <int K>
struct BT { BS s[K / 32]; };
template <int K>
__device__ __forceinline__ void prefetch_bt(const float* __restrict__ tile, int lane, BT<K>& B) {
  const uint4* t4 = reinterpret_cast<const uint4*>(tile) + lane;
#pragma unroll
  for (int s = 0; s < K / 32; ++s) load_bs(t4, s, B.s[s]);
}
// one k-step of one tile on two accumulation chains per M-tile: the two cross terms (2^-11 of the product) and the leading term
template <int MTL>
__device__ __forceinline__ void step2(floatx4 (&lo)[MTL], floatx4 (&acc)[MTL], const AF<MTL>& A, const BS& B) {
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt) lo[mt] = mf(A.a[mt][1], B.b[0], lo[mt]);
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt) acc[mt] = mf(A.a[mt][0], B.b[0], acc[mt]);
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt) lo[mt] = mf(A.a[mt][0], B.b[1], lo[mt]);
}
// a whole K = 128 layer for this wave's 16 columns: the weight tile is in registers (asked for a phase earlier), the A operands come
// from the planes one k-step ahead of their use (MTL = 2) or as they are needed (MTL = 4: registers)
template <int MTL>
__device__ __forceinline__ void layer128(const BT<HID>& B, const unsigned short* planes, int lane, floatx4 (&acc)[MTL]) {
  floatx4 lo[MTL];
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt) lo[mt] = splat4(0.0f);
  constexpr int NB = MTL == 2 ? 2 : 1;
  AF<MTL> A[NB];
  load_af<MTL>(planes, lane, 0, A[0]);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (NB == 2 && s + 1 < 4) load_af<MTL>(planes, lane, s + 1, A[(s + 1) % NB]);
    __builtin_amdgcn_sched_barrier(0);
    step2<MTL>(lo, acc, A[s % NB], B.s[s]);
    __builtin_amdgcn_sched_barrier(0);
    if (NB == 1 && s + 1 < 4) load_af<MTL>(planes, lane, s + 1, A[0]);
  }
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[mt][i] += lo[mt][i];
}
// the GRU's k-steps: three gate tiles of this wave's 16 hidden units; one term of the three, all gates and M-tiles
template <int MTL>
__device__ __forceinline__ void gru_term(floatx4 (&a0)[MTL], floatx4 (&a1)[MTL], floatx4 (&a2)[MTL], const AF<MTL>& A, int pa, const BS (&B)[3], int pb) {
#pragma unroll
  for (int mt = 0; mt < MTL; ++mt) {
    a0[mt] = mf(A.a[mt][pa], B[0].b[pb], a0[mt]); a1[mt] = mf(A.a[mt][pa], B[1].b[pb], a1[mt]); a2[mt] = mf(A.a[mt][pa], B[2].b[pb], a2[mt]);
  }
}
// the three terms smallest first into one accumulator per (gate, M-tile)
template <int MTL>
__device__ __forceinline__ void gru_step(floatx4 (&a0)[MTL], floatx4 (&a1)[MTL], floatx4 (&a2)[MTL], const AF<MTL>& A, const BS (&B)[3]) {
  gru_term<MTL>(a0, a1, a2, A, 1, B, 0); gru_term<MTL>(a0, a1, a2, A, 0, B, 1); gru_term<MTL>(a0, a1, a2, A, 0, B, 0);
}
__device__ __forceinline__ void ring_load(const float* __restrict__ W, int w, int lane, int st, BS (&dst)[3]) {
  const int TF = tile_floats(HID);
  const float* base = W + (st < 4 ? C_WIH : C_WHH);
#pragma unroll
  for (int g = 0; g < 3; ++g) load_bs(reinterpret_cast<const uint4*>(base + (8 * g + w) * TF) + lane, st & 3, dst[g]);
}
// eight consecutive features of one aircraft -> the two planes (one 16-byte LDS store per plane)
template <int MTL>
__device__ __forceinline__ void write_planes8(unsigned short* planes, int row, int k0, const float* v) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ctls::split2_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
  *reinterpret_cast<uint4*>(planes + 0 * Geo8<MTL>::PLN + row * KS + k0) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(planes + 1 * Geo8<MTL>::PLN + row * KS + k0) = make_uint4(l[0], l[1], l[2], l[3]);
}
// Sum over the adjacent lanes of an aircraft (16 or 8), the same value in all of them, with data-parallel-primitive moves (a few cycles
// each; __shfl_xor compiles to ds_bpermute_b32, an LDS round trip of ~100 cycles, eight of them in a dependent chain per LayerNorm): pairs
// and quads by quad_perm, the two quads of a half row by row_half_mirror (lane i <-> 7 - i), the two halves by row_mirror (i <-> 15 - i).
// Every lane adds its own and its partner's partial sum, which are the same two numbers on both sides: all lanes end bit-identical.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_f<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  if (LANES == 16) v += dpp_f<0x140>(v);   // row_mirror
  return v;
}
// torch.nn.LayerNorm(128) (eps 1e-5, biased variance) of buf[row][k] (fp32, row stride RS) into the two fp16 planes the next layer's
// A operands are read from. Thread = (aircraft = tid / TPR, part = tid % TPR) owns FPT consecutive features; the parts of an aircraft sit
// in adjacent lanes: mean and variance are a few DPP steps each, no partial sums through LDS. Scale / shift come from LDS (staged).
template <int MTL>
__device__ __forceinline__ void layer_norm_planes(const float* buf, unsigned short* planes, const float* g, const float* b, int tid) {
  constexpr int TPR = Geo8<MTL>::TPR, FPT = Geo8<MTL>::FPT;
  const int row = tid / TPR, part = tid % TPR;
  float x[FPT], gg[FPT], bb[FPT];
#pragma unroll
  for (int q = 0; q < FPT / 4; ++q) {
    const float4 xv = *reinterpret_cast<const float4*>(buf + row * RS + FPT * part + 4 * q);
    const float4 gv = *reinterpret_cast<const float4*>(g + FPT * part + 4 * q), bv = *reinterpret_cast<const float4*>(b + FPT * part + 4 * q);
    x[4 * q] = xv.x; x[4 * q + 1] = xv.y; x[4 * q + 2] = xv.z; x[4 * q + 3] = xv.w;
    gg[4 * q] = gv.x; gg[4 * q + 1] = gv.y; gg[4 * q + 2] = gv.z; gg[4 * q + 3] = gv.w;
    bb[4 * q] = bv.x; bb[4 * q + 1] = bv.y; bb[4 * q + 2] = bv.z; bb[4 * q + 3] = bv.w;
  }
  float sum = 0.0f;
#pragma unroll
  for (int q = 0; q < FPT / 8; ++q) sum += ((x[8 * q] + x[8 * q + 1]) + (x[8 * q + 2] + x[8 * q + 3])) + ((x[8 * q + 4] + x[8 * q + 5]) + (x[8 * q + 6] + x[8 * q + 7]));
  const float m = group_sum<TPR>(sum) * (1.0f / HID);
  float v = 0.0f;
#pragma unroll
  for (int q = 0; q < FPT; ++q) { x[q] -= m; v = fmaf(x[q], x[q], v); }
  const float is = rsqrtf(group_sum<TPR>(v) * (1.0f / HID) + 1e-5f);
  float y[FPT];
#pragma unroll
  for (int q = 0; q < FPT; ++q) y[q] = fmaf(x[q] * is, gg[q], bb[q]);
#pragma unroll
  for (int q = 0; q < FPT / 8; ++q) write_planes8<MTL>(planes, row, FPT * part + 8 * q, y + 8 * q);
  __syncthreads();
}
// the fp32 GRU state of (aircraft row, unit): two fp16 pieces do not add up to it exactly, so an fp32 copy [aircraft][k] (row stride RS)
// sits behind the two planes of the state buffer for the gate algebra
template <int MTL>
__device__ __forceinline__ float state_value(const unsigned short* planes, int row, int unit) {
  return reinterpret_cast<const float*>(planes + 2 * Geo8<MTL>::PLN)[row * RS + unit];
}
}  // namespace ctl8

// SCRIPTED: the handle has scripted opponents (`use_baseline`); their state -> pose code is compiled into that instantiation only.
// MTL: 16-row M-tiles per wave = aircraft per workgroup / 16 (2 or 4).
template <bool SCRIPTED, int MTL>
__global__ __launch_bounds__(512) void controller8_kernel(ctl::Args a) {
  using namespace ctl8;
  using ctl::sigmoid_f; using ctl::tanh_f;
  using G = Geo8<MTL>;
  constexpr int R = G::R, PLN = G::PLN, LSR = G::LSR, TPR = G::TPR, FPT = G::FPT;
  constexpr int PHN = 2 * PLN + 2 * R * RS;                            // two planes + the fp32 copy [aircraft][k] (state_value), in 16-bit units
  __shared__ __attribute__((aligned(16))) unsigned short PA[NP * PLN];  // activations as piece planes [piece][aircraft][k]
  __shared__ __attribute__((aligned(16))) unsigned short PH[PHN];       // the GRU state likewise; the head logits (fp32 [160][LSR]) later
  __shared__ __attribute__((aligned(16))) float stg[R * RS];            // a layer's fp32 outputs [aircraft][k] (row stride RS) on their way to LayerNorm
  static_assert(4 * 32 * LSR <= R * RS, "the head partials of tiles 8 and 9 fit the staging buffer");
  static_assert(sizeof(unsigned short) * PHN >= sizeof(float) * NHP * LSR && (2 * PLN) % 8 == 0, "the logits reuse the GRU-state planes");
  float* lg = reinterpret_cast<float*>(PH);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave 0..7: output columns 16 w .. 16 w + 15 of every 128-wide layer
  const int i0 = blockIdx.x * R;
  const float* __restrict__ W = a.Ws8;
  const int col = lane & 15;
  // every bias and LayerNorm scale / shift (1952 floats behind the weight tiles) goes to LDS with the first loads
  __shared__ __attribute__((aligned(16))) float prm[C_END - C_B1];
  static_assert((C_END - C_B1) % 4 == 0 && (C_END - C_B1) / 4 <= 512 && C_B1 % 4 == 0, "one float4 per thread");
#define CTL8_PRM(i) prm[(i) - C_B1]            /* W[i] for the vectors, from LDS */
  float4 prm4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < (C_END - C_B1) / 4) prm4 = reinterpret_cast<const float4*>(W + C_B1)[tid];

  // ---- stage. Loads return in the order they were asked for: the 12 controller inputs first (layer 1 waits for nothing else), then
  // layer 1's weights, the GRU state (first needed by the GRU) and layer 2's weights.
  AC_CLK(200);
  BT<32> b1;
  BT<HID> b2;
  const int srow = tid % R, spart = tid / R;     // staging: thread = (aircraft, FPT-feature part)
  const int sn = min(i0 + srow, a.N - 1);
  float x[16];
  if (spart == 0) {
    const float* hi = a.hi + (size_t)sn * a.act_hi;
    const float* ob = a.obs + (size_t)sn * a.obs_dim;
    const int slot = sn % a.A;
    if (SCRIPTED && a.use_baseline && slot >= a.n_ego) {
      // the enemy team is flown by BaselineAgent k: its 12 inputs come from the geometry (no action row is read for it)
      float xs[12];
      ctl::scripted_inputs(a, sn, xs);
#pragma unroll
      for (int k = 0; k < 12; ++k) x[k] = xs[k];
    } else {
      const int c0 = (int)hi[0], c1 = (int)hi[1], c2 = (int)hi[2];
      // singlecombat_task.py:217-219, 235-241: below 3500 m the altitude choice is overridden by "climb"
      x[0] = (ob[0] * 5000.0f < 3500.0f) ? 0.1f : (c0 == 0 ? 0.1f : (c0 == 1 ? 0.0f : -0.1f));
      x[1] = (float)(c1 - 2) * 0.26179938779914943f;   // {-pi/6, -pi/12, 0, pi/12, pi/6}
      x[2] = c2 == 0 ? 0.05f : (c2 == 1 ? 0.0f : -0.05f);
#pragma unroll
      for (int k = 0; k < 9; ++k) x[3 + k] = ob[k];
    }
    x[12] = 0.0f; x[13] = 0.0f; x[14] = 0.0f; x[15] = 0.0f;   // (k 12..31 of the one 32-k step are zero)
  }
  __builtin_amdgcn_sched_barrier(0);
  prefetch_bt<32>(W + C_W1 + w * tile_floats(32), lane, b1);
  float hv[FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) hv[f] = a.H[(size_t)(spart * FPT + f) * a.N + sn];
  prefetch_bt<HID>(W + C_W2 + w * tile_floats(HID), lane, b2);
  __builtin_amdgcn_sched_barrier(0);
  if (tid < (C_END - C_B1) / 4) reinterpret_cast<float4*>(prm)[tid] = prm4;
  if (spart == 0) {
    const float hi8[8] = {x[8], x[9], x[10], x[11], 0.0f, 0.0f, 0.0f, 0.0f};
    write_planes8<MTL>(PA, srow, 0, x); write_planes8<MTL>(PA, srow, 8, hi8);
  } else if (spart <= 2) {   // zero k 16..31 of the planes
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(PA + p * PLN + srow * KS + 8 * (spart + 1)) = z;
  }
  __syncthreads();

  AC_CLK(201);
  // ---- MLP layer 1: Linear(12, 128) + ReLU + LayerNorm; wave w owns output columns 16 w .. 16 w + 15
  {
    AF<MTL> A;
    load_af<MTL>(PA, lane, 0, A);
    const float bias = CTL8_PRM(C_B1 + w * 16 + col);
    floatx4 acc[MTL], lo[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) { acc[mt] = splat4(bias); lo[mt] = splat4(0.0f); }
    step2<MTL>(lo, acc, A, b1.s[0]);
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) stg[c_row(mt, i, lane) * RS + w * 16 + col] = fmaxf(acc[mt][i] + lo[mt][i], 0.0f);
  }
  {   // the GRU state has arrived behind layer 1: as the two fp16 planes the products read, and in fp32 for the gate algebra
#pragma unroll
    for (int q = 0; q < FPT / 8; ++q) write_planes8<MTL>(PH, srow, spart * FPT + 8 * q, hv + 8 * q);
    float* hf = reinterpret_cast<float*>(PH + 2 * PLN);
#pragma unroll
    for (int q = 0; q < FPT / 4; ++q)
      *reinterpret_cast<float4*>(hf + srow * RS + spart * FPT + 4 * q) = make_float4(hv[4 * q], hv[4 * q + 1], hv[4 * q + 2], hv[4 * q + 3]);
  }
  // The GRU's weight ring (one k-step per stage: two ahead, or one). Its first stages are asked for HERE, behind layer 1: LayerNorm 1 and
  // layer 2 (whose own weights came with the first loads) leave the L1 idle for ~3.5 k cycles. (Asked for behind layer 2, where round 4 first
  // had them, layer 2's phase ended with 96 KB per CU queueing at the L1: 1-2.5 % slower at every size. Asked for with the kernel's first
  // loads they queue in front of what layer 1 waits for: 2-4 % slower. With two fp16 pieces the registers would allow deeper rings --
  // a fourth stage at 32 rows +1 to +4 %, a third stage and A operands one k-step ahead at 64 rows +1 %: depth is not what the loop waits for.)
  constexpr int RING = MTL == 2 ? 3 : 2;
  BS ring[RING][3];   // [stage][gate]
#pragma unroll
  for (int st = 0; st < RING - 1; ++st) ring_load(W, w, lane, st, ring[st]);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  AC_CLK(202);
  layer_norm_planes<MTL>(stg, PA, prm + (C_G1 - C_B1), prm + (C_BE1 - C_B1), tid);
  AC_CLK(203);
  // ---- MLP layer 2
  {
    const float bias = CTL8_PRM(C_B2 + w * 16 + col);
    floatx4 acc[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) acc[mt] = splat4(bias);
    layer128<MTL>(b2, PA, lane, acc);
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) stg[c_row(mt, i, lane) * RS + w * 16 + col] = fmaxf(acc[mt][i], 0.0f);
  }
  __syncthreads();
  AC_CLK(204);
  layer_norm_planes<MTL>(stg, PA, prm + (C_G2 - C_B1), prm + (C_BE2 - C_B1), tid);
  AC_CLK(205);
  // ---- GRU cell (torch gate order r, z, n): wave w owns hidden units 16 w .. 16 w + 15, i.e. gate tiles w, 8 + w, 16 + w.
  // r and z only ever need W_ih x + W_hh h summed, so each has ONE accumulator for both products; the n gate keeps them apart (r * (W_hn h + b_hn)).
  BT<HID> bh;
  BS b5;
  {
    const float br = CTL8_PRM(C_BIH + 0 * 128 + w * 16 + col) + CTL8_PRM(C_BHH + 0 * 128 + w * 16 + col);
    const float bz = CTL8_PRM(C_BIH + 1 * 128 + w * 16 + col) + CTL8_PRM(C_BHH + 1 * 128 + w * 16 + col);
    const float bin = CTL8_PRM(C_BIH + 2 * 128 + w * 16 + col), bhn = CTL8_PRM(C_BHH + 2 * 128 + w * 16 + col);
    floatx4 gr[MTL], gz[MTL], in_[MTL], hn[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) { gr[mt] = splat4(br); gz[mt] = splat4(bz); in_[mt] = splat4(bin); hn[mt] = splat4(bhn); }
    {
      constexpr int NB = MTL == 2 ? 2 : 1;
      AF<MTL> A[NB];
      load_af<MTL>(PA, lane, 0, A[0]);
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        // (the scheduling fences keep the loads where they are written: left alone, the machine scheduler sinks every weight load
        // to just in front of its first use to save registers, which serialises an L2 round trip with every k-step)
        if (st + RING - 1 < 8) ring_load(W, w, lane, st + RING - 1, ring[(st + RING - 1) % RING]);
        if (NB == 2 && st + 1 < 8) load_af<MTL>(st + 1 < 4 ? PA : PH, lane, (st + 1) & 3, A[(st + 1) % NB]);
        // (Measured and left out: a bare barrier per k-step that keeps the two waves of a SIMD within a k-step of each other. Left alone the
        // older wave finishes all its products first and the younger runs on; in lockstep the pair was slower -- with three pieces 15.3 k
        // cycles instead of 13.6 k at 32 rows, 27.6 k against 24.0 k at 64. The matrix pipe is not what the pair waits for: the GRU's 384 KB
        // of weight pieces per workgroup are 6.1 k cycles of the 64 B / clk a CU's L1 fills at, its matrix instructions 2.3 k per wave.)
        __builtin_amdgcn_sched_barrier(0);
        if (st < 4) gru_step<MTL>(gr, gz, in_, A[st % NB], ring[st % RING]);
        else gru_step<MTL>(gr, gz, hn, A[st % NB], ring[st % RING]);
        __builtin_amdgcn_sched_barrier(0);
        if (NB == 1 && st + 1 < 8) load_af<MTL>(st + 1 < 4 ? PA : PH, lane, (st + 1) & 3, A[0]);
      }
    }
    // the heads' weights (this wave's tile and its k-step of the ninth / tenth), behind the gate algebra and LayerNorm 3
    prefetch_bt<HID>(W + C_WA + w * tile_floats(HID), lane, bh);
    load_bs(reinterpret_cast<const uint4*>(W + C_WA + (8 + (w & 1)) * tile_floats(HID)) + lane, w >> 1, b5);
    __builtin_amdgcn_sched_barrier(0);
    AC_CLK(206);
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = c_row(mt, i, lane), unit = w * 16 + col;
        const float rg = sigmoid_f(gr[mt][i]);
        const float zg = sigmoid_f(gz[mt][i]);
        // (explicit fused multiply-adds: which products the compiler fuses on its own depends on the code around them, and two builds of
        // this kernel would differ by an ulp)
        const float ng = tanh_f(fmaf(rg, hn[mt][i], in_[mt][i]));
        const float hnew = fmaf(zg, state_value<MTL>(PH, row, unit), (1.0f - zg) * ng);
        stg[row * RS + unit] = hnew;
      }
  }
  __syncthreads();
  AC_CLK(207);
  {   // the new hidden state goes out row-contiguous (runs of R floats per feature) from LDS; thread = (row, FPT-feature part)
    const int row = tid % R, part = tid / R, n = i0 + row;
    if (n < a.N) {
#pragma unroll
      for (int q = 0; q < FPT / 4; ++q) {
        const float4 h4 = *reinterpret_cast<const float4*>(stg + row * RS + part * FPT + 4 * q);
        a.H[(size_t)(part * FPT + 4 * q + 0) * a.N + n] = h4.x; a.H[(size_t)(part * FPT + 4 * q + 1) * a.N + n] = h4.y;
        a.H[(size_t)(part * FPT + 4 * q + 2) * a.N + n] = h4.z; a.H[(size_t)(part * FPT + 4 * q + 3) * a.N + n] = h4.w;
      }
    }
  }
  AC_CLK(208);
  layer_norm_planes<MTL>(stg, PA, prm + (C_G3 - C_B1), prm + (C_BE3 - C_B1), tid);
  AC_CLK(209);
  // ---- heads: 153 logits = ten 16-column tiles; wave w takes tile w, and one k-step of tile 8 + (w & 1) (logits 128 .. 159)
  {
    const float bias = CTL8_PRM(C_BA + w * 16 + col);
    floatx4 acc[MTL];
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt) acc[mt] = splat4(bias);
    layer128<MTL>(bh, PA, lane, acc);
#pragma unroll
    for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) lg[(w * 16 + col) * LSR + c_row(mt, i, lane)] = acc[mt][i];   // (the GRU-state planes under lg were last read before two barriers)
    // tiles 8 and 9: their K range is split over four waves each (k-step w >> 1); the partial sums go to stg (free by now) and are added
    // in a fixed order below
    {
      AF<MTL> A;
      load_af<MTL>(PA, lane, w >> 1, A);
      floatx4 part[MTL], lo[MTL];
#pragma unroll
      for (int mt = 0; mt < MTL; ++mt) { part[mt] = splat4(0.0f); lo[mt] = splat4(0.0f); }
      step2<MTL>(lo, part, A, b5);
#pragma unroll
      for (int mt = 0; mt < MTL; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) stg[((w >> 1) * 32 + (w & 1) * 16 + col) * LSR + c_row(mt, i, lane)] = part[mt][i] + lo[mt][i];
    }
  }
  __syncthreads();
  AC_CLK(210);
  // logits 128 .. 152 = bias + the four K-partials, summed in a fixed order (25 columns x R aircraft over 512 threads)
  for (int e = tid; e < 25 * R; e += 512) {
    const int q = e / R, row = e % R;
    lg[(128 + q) * LSR + row] = (((CTL8_PRM(C_BA + 128 + q) + stg[q * LSR + row]) + stg[(32 + q) * LSR + row]) + stg[(64 + q) * LSR + row]) + stg[(96 + q) * LSR + row];
  }
  __syncthreads();
  AC_CLK(211);
  {   // argmax: wave = (head, half of the rows), lane = (part of the head's logits, row): first maximum, like torch argmax
    constexpr int RW = R / 2, SPLIT = 64 / RW, PER = (41 + SPLIT - 1) / SPLIT;   // rows per wave, parts per head (4 or 2), logits per part (11 or 21)
    const int head = w >> 1, row = RW * (w & 1) + (lane % RW), part = lane / RW;
    const int off = head * 41, cnt = (head == 3) ? 30 : 41;
    const int j0 = PER * part;
    float lv[PER];
#pragma unroll
    for (int jj = 0; jj < PER; ++jj) lv[jj] = (j0 + jj < cnt) ? lg[(off + j0 + jj) * LSR + row] : -INFINITY;   // independent LDS reads
    // (part 0 starts from logit 0 like the sequential scan does; the others from -inf, so that a NaN logit is skipped, not adopted)
    float best = part == 0 ? lv[0] : -INFINITY;
    int bi = part == 0 ? 0 : cnt;
#pragma unroll
    for (int jj = 0; jj < PER; ++jj)
      if (!(part == 0 && jj == 0) && lv[jj] > best) { best = lv[jj]; bi = j0 + jj; }
    // the later part only wins with a strictly larger value (its indices are all higher)
#pragma unroll
    for (int d = RW; d <= 32; d <<= 1) {
      const float v2 = __shfl_down(best, d);
      const int i2 = __shfl_down(bi, d);
      if (v2 > best) { best = v2; bi = i2; }
    }
    const int nn = i0 + row;
    if (part == 0 && nn < a.N) a.low[(size_t)nn * a.act_low + head] = (float)bi;
    if (part == 1 && head == 0 && nn < a.N) {   // weapon bits ride along unchanged
      const bool scripted = a.use_baseline && (nn % a.A) >= a.n_ego;   // scenario1_task.py:42-48: bits [0,0,0,0], or all ones with artillery
      for (int k = 4; k < a.act_low; ++k)
        a.low[(size_t)nn * a.act_low + k] = scripted ? (a.use_artillery ? 1.0f : 0.0f) : a.hi[(size_t)nn * a.act_hi + (k - 1)];
    }
  }
  AC_CLK(212);
}
#undef CTL8_PRM
