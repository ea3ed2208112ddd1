// fp64 primitives for the munition path and the geodetic reductions: the reference flies its missiles in Python floats, and a 5 m
// fuse against ~25 m of relative travel per tick makes hit-or-miss hinge on centimetres of a 300-tick integration, so those
// updates stay in fp64 on the device -- but the library's division / sqrt / exp / sin are correctly-rounded, special-case-proof
// sequences of 25-200 instructions each, and one munition update held ~20 of them. These forms keep a relative error of a few
// 1e-16 (one or two ulp: the hardware seed refined by two Newton steps, Cody-Waite reductions, Taylor polynomials on small
// ranges) for the finite, ordinary arguments these call sites produce, at 5-15 instructions each.
#pragma once

namespace fx {

__device__ __forceinline__ double rcp(double x) {          // v_rcp_f64 (>= 26 bits) + two Newton-Raphson steps
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double div(double a, double b) {
  const double r = rcp(b), q = a * r;
  return fma(fma(-b, q, a), r, q);                          // one residual correction of the quotient
}
__device__ __forceinline__ double rsqrt(double x) {         // v_rsq_f64 + two Newton-Raphson steps (x > 0)
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = r * fma(-h * r, r, 1.5);
  r = r * fma(-h * r, r, 1.5);
  return r;
}
__device__ __forceinline__ double sqrt(double x) {          // x >= 0
  const double r = rsqrt(x), s = x * r;
  const double y = fma(fma(-s, s, x), 0.5 * r, s);
  return x > 0.0 ? y : 0.0;
}
// sqrt(x) and 1 / sqrt(x) from the one refined seed (x > 0): the callers that divide by a length they also need
__device__ __forceinline__ void sqrt_both(double x, double* s_out, double* r_out) {
  const double r = rsqrt(x), s = x * r;
  *s_out = fma(fma(-s, s, x), 0.5 * r, s);
  *r_out = r;
}
// sin for |x| up to a few pi (the per-second turn rates of a missile: |x| <= 3.4): reduction by the nearest multiple of pi in two
// pieces, Taylor series to x^21 on [-pi/2, pi/2] (truncation 3e-16)
__device__ __forceinline__ double sin(double x) {
  const double n = rint(x * 0.31830988618379067154);
  double y = fma(-n, 3.141592653589793116, x);
  y = fma(-n, 1.2246467991473532e-16, y);
  const double t = y * y;
  double p = -1.9572941063391263e-20;                       // -1/21!
  p = fma(p, t, 8.2206352466243295e-18);                    //  1/19!
  p = fma(p, t, -2.8114572543455206e-15);                   // -1/17!
  p = fma(p, t, 7.6471637318198164e-13);                    //  1/15!
  p = fma(p, t, -1.6059043836821613e-10);                   // -1/13!
  p = fma(p, t, 2.5052108385441720e-08);                    //  1/11!
  p = fma(p, t, -2.7557319223985893e-06);                   // -1/9!
  p = fma(p, t, 1.9841269841269841e-04);                    //  1/7!
  p = fma(p, t, -8.3333333333333332e-03);                   // -1/5!
  p = fma(p, t, 1.6666666666666666e-01);                    //  1/3!
  const double r = fma(-y * t, p, y);
  return ((long long)n & 1) ? -r : r;
}
// sin and cos together for |x| up to a few turns (a missile's pitch and heading): reduction by the nearest multiple of pi / 2 in two
// pieces, Taylor series on [-pi/4, pi/4] (truncation < 1e-17), quadrant swap
__device__ __forceinline__ void sincos(double x, double* sn, double* cs) {
  const double n = rint(x * 0.63661977236758134308);
  double y = fma(-n, 1.5707963267948965580, x);
  y = fma(-n, 6.1232339957367660e-17, y);
  const double t = y * y;
  double ps = 2.8114572543455206e-15;                       //  1/17!
  ps = fma(ps, t, -7.6471637318198164e-13);                 // -1/15!
  ps = fma(ps, t, 1.6059043836821613e-10);                  //  1/13!
  ps = fma(ps, t, -2.5052108385441720e-08);                 // -1/11!
  ps = fma(ps, t, 2.7557319223985893e-06);                  //  1/9!
  ps = fma(ps, t, -1.9841269841269841e-04);                 // -1/7!
  ps = fma(ps, t, 8.3333333333333332e-03);                  //  1/5!
  ps = fma(ps, t, -1.6666666666666666e-01);                 // -1/3!
  const double s0 = fma(y * t, ps, y);
  double pc = -1.1470745597729725e-11;                      // -1/14!  ... (the 1/16! term is below 1e-17 on this range)
  pc = fma(pc, t, 2.0876756987868100e-09);                  //  1/12!
  pc = fma(pc, t, -2.7557319223985888e-07);                 // -1/10!
  pc = fma(pc, t, 2.4801587301587302e-05);                  //  1/8!
  pc = fma(pc, t, -1.3888888888888889e-03);                 // -1/6!
  pc = fma(pc, t, 4.1666666666666664e-02);                  //  1/4!
  pc = fma(pc, t, -0.5);
  const double c0 = fma(t, pc, 1.0);
  const long long q = (long long)n & 3;
  const double s1 = (q & 1) ? c0 : s0, c1 = (q & 1) ? s0 : c0;
  *sn = (q == 2 || q == 3) ? -s1 : s1;
  *cs = (q == 1 || q == 2) ? -c1 : c1;
}
// exp for moderate arguments (the air-density law exp(-h / 9300): x in [-3, 0.1]): 2^k * exp(r), |r| <= ln2 / 2, Taylor to r^13
__device__ __forceinline__ double exp(double x) {
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                        // 1/13!
  p = fma(p, r, 2.0876756987868100e-09);                    // 1/12!
  p = fma(p, r, 2.5052108385441720e-08);
  p = fma(p, r, 2.7557319223985888e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.4801587301587302e-05);
  p = fma(p, r, 1.9841269841269841e-04);
  p = fma(p, r, 1.3888888888888889e-03);
  p = fma(p, r, 8.3333333333333332e-03);
  p = fma(p, r, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

}  // namespace fx
