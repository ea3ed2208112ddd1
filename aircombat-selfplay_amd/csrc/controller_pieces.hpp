// The piece arithmetic of the low-level controller kernel (network, arguments and weight blob: controller_common.hpp; the kernel:
// controller8_kernel.hpp), so that its GEMMs run on the 16-bit matrix path of gfx950 instead of the fp32 one (a sixteenth of its rate).
//
// An fp32 value x is split into two fp16 pieces: hi = x rounded to fp16 (11 significand bits), lo = (x - hi) rounded to fp16; the
// subtraction is exact, the second rounding leaves |x - hi - lo| <= 2^-22 |x| (2^-25 absolute where lo falls among fp16's subnormals,
// |x| < 2^-3). A product of two such sums has four terms; three are kept -- hi*hi, hi*lo, lo*hi -- and lo*lo (<= 2^-22 of the product) is
// left out: every product is good to about 2^-21 of itself (22 bits per value, not fp32's 24: on the golden sequences some five times
// the error of a plain fp32 matmul, the reference's own arithmetic -- torch on the CPU, baseline_actor.py -- numbers below), and each
// kept term is an exact fp16 x fp16 product accumulated in fp32 by the matrix core. Three v_mfma_f32_16x16x32_f16 (16 cycles
// each) per 32 k of a 16 x 16 tile are 48 cycles where the fp32 instructions take 256.
//
// What that costs in parity, measured (tests/test_gpu_parity.py::test_hierarchical_tasks_lowlevel_controller and the as-shipped NvN
// cases, teacher-forced against the float64 CPU restatement): 0 of 130 000 argmax indices differ; worst |d GRU state| 4.0 - 7.6e-6 per case
// against 2.8 - 6.1e-6 with three bf16 pieces (24 bits per value, six terms per product: rounds 2 - 4) -- the difference the fp32 flight
// model's observations feed INTO the network is what both numbers measure, the products' own error (numpy emulation on the golden
// sequences: 3.1e-6 of the state after 48 steps, 1.3e-5 of a logit; a plain fp32 matmul 6.4e-7 / 2.5e-6; three bf16 pieces 1.1e-7 /
// 5.5e-7; no argmax flips in any) is inside the tests' 5e-5 bound by an order of magnitude. In exchange: half the matrix instructions and
// two thirds of the weight bytes (the GRU phase is bound by the 64 B / clk a CU's L1 fills at): 18.0 -> 13.x us at 8192 aircraft.
//
// Weights are split once on the host (ac_load_controller); activations are split where they are produced -- the LayerNorm epilogue writes
// two fp16 planes [aircraft][k] to LDS, so an A operand (8 consecutive k of one aircraft) is one ds_read_b128 per piece.
// (History: round 1 ran the GEMMs on v_mfma_f32_32x32x2_f32 (27 us per call at 8192 aircraft); rounds 2-3 on a four-wave kernel of
// v_mfma_f32_32x32x16_bf16 with three bf16 pieces per value and six kept terms, one wave per SIMD with 378 registers of weight prefetch
// (20-22 us; 39 / 75 us at 16 384 / 32 768); round 4's eight-wave kernel replaced it on every grid (18.0 / 27.6 / 53.6 us), then went
// from three bf16 pieces to two fp16 ones.)
#pragma once

namespace ctls {
using ctl::HID;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
constexpr int KS = HID + 8;                      // 16-bit values per plane row (272 bytes: 16 aircraft's 16-byte reads fall on distinct banks)
constexpr int RS = HID + 4;                      // floats per row of the fp32 staging buffer [aircraft][k] (LayerNorm reads 16-byte vectors)
// x ~ hi + lo (bit patterns of the two fp16), round-to-nearest-even each
__host__ __device__ __forceinline__ void split2(float x, unsigned& hi, unsigned& lo) {
  const _Float16 h = (_Float16)x;
  const _Float16 l = (_Float16)(x - (float)h);
  unsigned short hb, lb;
  memcpy(&hb, &h, 2); memcpy(&lb, &l, 2);
  hi = hb; lo = lb;
}
// two values at once on the device (v_cvt_pk_f16_f32 rounds a pair to nearest-even and packs it)
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hi, unsigned& lo) {
  const floatx2 f = {a, b};
  const f16x2 h = __builtin_convertvector(f, f16x2);
  const floatx2 r = f - __builtin_convertvector(h, floatx2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
}  // namespace ctls
