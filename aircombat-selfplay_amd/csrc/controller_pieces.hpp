// The bf16-piece arithmetic of the low-level controller kernel (network, arguments and weight blob: controller_common.hpp; the kernel:
// controller8_kernel.hpp), so that its GEMMs run on the bf16 matrix path of gfx950 instead of the fp32 one (v_mfma_f32_32x32x2_f32: 2 k per
// 64 cycles).
//
// An fp32 value x is exactly hi + mid + lo, three bf16 numbers: hi = x rounded to bf16, mid = (x - hi) rounded to bf16, lo = x - hi - mid
// (8 significand bits each, 24 together; both subtractions are exact and the last remainder fits). A product of two such sums has
// nine terms; the six that can reach 2^-16 of the product are kept -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- and the three
// left out (mid*lo, lo*mid, lo*lo) are at most 2^-23 of it (one fp32 ulp, at the worst case of both roundings; typically 2^-27). Every kept term is an exact bf16 x bf16
// product accumulated in fp32 by the matrix core, like the fp32 instruction accumulates its own: six bf16 instructions per 16 k are
// 12 cycles per k against 32.
//
// Weights are split once on the host (ac_load_controller); activations are split where they are produced -- the LayerNorm epilogue writes
// three bf16 planes [aircraft][k] to LDS, so an A operand (8 consecutive k of one aircraft) is one ds_read_b128 per piece.
// (History: round 1 ran the GEMMs on v_mfma_f32_32x32x2_f32 (27 us per call at 8192 aircraft); rounds 2-3 on a four-wave kernel of
// v_mfma_f32_32x32x16_bf16 pieces, one wave per SIMD with 378 registers of weight prefetch (20-22 us; 39 / 75 us at 16 384 / 32 768);
// round 4's eight-wave kernel replaced it on every grid and it was removed.)
#pragma once

namespace ctls {
using ctl::HID;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KS = HID + 8;                      // bf16 per plane row (272 bytes: 16 aircraft's 16-byte reads fall on distinct banks)
constexpr int RS = HID + 4;                      // floats per row of the fp32 staging buffer [aircraft][k] (LayerNorm reads 16-byte vectors)
// x = hi + mid + lo (bit patterns of the three bf16, i.e. the high halves of three floats). Each piece is the round-to-nearest-even
// bf16 of what is left: |x - hi| <= 2^-8 |x|, |x - hi - mid| <= 2^-16 |x|, and the last remainder has at most 8 significant bits, so
// lo takes it exactly.
__host__ __device__ __forceinline__ unsigned bf16_rne_bits(unsigned b) {   // float bit pattern -> the same with the low half rounded away
  const unsigned r = b + 0x7FFFu + ((b >> 16) & 1u);
  return (((r & 0x7F800000u) == 0x7F800000u) ? b : r) & 0xFFFF0000u;      // (never round a finite value up to infinity)
}
__host__ __device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef __HIP_DEVICE_COMPILE__
  const unsigned hb = bf16_rne_bits(__float_as_uint(x));
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = bf16_rne_bits(__float_as_uint(r1));
  const float r2 = r1 - __uint_as_float(mb);
  hi = hb >> 16; mid = mb >> 16; lo = __float_as_uint(r2) >> 16;
#else
  unsigned xb; memcpy(&xb, &x, 4);
  const unsigned hb = bf16_rne_bits(xb); float h; memcpy(&h, &hb, 4);
  const float r1 = x - h; unsigned r1b; memcpy(&r1b, &r1, 4);
  const unsigned mb = bf16_rne_bits(r1b); float m; memcpy(&m, &mb, 4);
  const float r2 = r1 - m; unsigned r2b; memcpy(&r2b, &r2, 4);
  hi = hb >> 16; mid = mb >> 16; lo = r2b >> 16;
#endif
}
// Two values at once on the device: v_cvt_pk_bf16_f32 rounds a pair to nearest-even and packs it (the same rounding as bf16_rne_bits
// for every value the network produces).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const floatx2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pack_bf16x2(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = pack_bf16x2(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xFFFF0000u);
  lo = pack_bf16x2(sa, sb);   // (exact: at most 8 significant bits are left)
}
}  // namespace ctls
