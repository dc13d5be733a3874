// samsim_pow.h -- x**3.10 of the permeability law (mo_grav_drain.f90:105, mo_flush.f90:119,128, mo_flood.f90:73) for x >= 0.
//
// x**3.1 = x*x*x * exp(0.1*log(x)).  The cube is exact to 1.5 ulp; the remaining factor has a small exponent, so a plain
// (not double-double) logarithm is enough: an absolute error e in log(x) shows up as 0.1*e relative in the result.  log(x) =
// E*ln2 + 2*atanh(s), s = (m-1)/(m+1), m in [sqrt(1/2), sqrt(2)); exp by n*ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial.
// Within ~3 ulp of the correctly rounded power (tests/test_host_logic.py checks it on the CPU against math.pow with
// this header compiled for the host), about 60 vector instructions instead of the 139 of exp(3.1*log(x)) and the 214 of pow().
// On the GPU the power goes through a hardware-seeded tenth root (sp_pow_3p1 below; tools/div_probe measures both forms against
// pow() there).  Included by samsim_kernels.hip, tools/div_probe.hip and tools/pow_host.c.
#ifndef SAMSIM_POW_H
#define SAMSIM_POW_H

#if defined(__HIPCC__)
#define SP_FN __device__ __forceinline__
#define SP_FMA(a, b, c) __builtin_fma(a, b, c)
#define SP_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#define SP_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define SP_LDEXP(x, n) __builtin_amdgcn_ldexp(x, n)
#define SP_RINT(x) __builtin_rint(x)
// a*b + C and a*C + b with the constant C read from a scalar register pair: left to itself the compiler materialises every
// polynomial coefficient in a vector register pair (two v_mov_b32 per coefficient, as many vector instructions as the fma they
// feed -- on this machine every vector instruction, 32- or 64-bit, occupies the SIMD for four cycles); two s_mov_b32 issue on the
// scalar unit beside the vector work.  Same operation, same rounding.
__device__ __forceinline__ double sp_fma_vvs(double a, double b, double c_const) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_const));
  return r;
}
__device__ __forceinline__ double sp_fma_vsv(double a, double c_const, double b) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c_const), "v"(b));
  return r;
}
#define SP_FMA_C(a, b, C) sp_fma_vvs(a, b, C)
#define SP_FMA_MC(a, C, b) sp_fma_vsv(a, C, b)
#else
#include <math.h>
#define SP_FN static inline
#define SP_FMA(a, b, c) fma(a, b, c)
static inline double sp_frexp_mant(double x) { int e; return frexp(x, &e); }
static inline int sp_frexp_exp(double x) { int e; (void)frexp(x, &e); return e; }
#define SP_FREXP_MANT(x) sp_frexp_mant(x)
#define SP_FREXP_EXP(x) sp_frexp_exp(x)
#define SP_LDEXP(x, n) ldexp(x, n)
#define SP_RINT(x) rint(x)
#define SP_FMA_C(a, b, C) fma(a, b, C)
#define SP_FMA_MC(a, C, b) fma(a, C, b)
#endif
#ifndef SP_QUOT
#define SP_QUOT(a, b) ((a) / (b))   // the kernel passes its own quotient (samsim_div.h)
#endif

// natural logarithm of a positive finite x (subnormals included), absolute error ~1e-16 * max(1, |log x|)
SP_FN double sp_log(double x) {
  double m = SP_FREXP_MANT(x);  // [0.5, 1)
  int e = SP_FREXP_EXP(x);
  if (m < 0.70710678118654752440) { m = m + m; e = e - 1; }
  const double s = SP_QUOT(m - 1.0, m + 1.0);
  const double z = s * s;
  double p = 1.0 / 21.0;
  p = SP_FMA_C(p, z, 1.0 / 19.0);
  p = SP_FMA_C(p, z, 1.0 / 17.0);
  p = SP_FMA_C(p, z, 1.0 / 15.0);
  p = SP_FMA_C(p, z, 1.0 / 13.0);
  p = SP_FMA_C(p, z, 1.0 / 11.0);
  p = SP_FMA_C(p, z, 1.0 / 9.0);
  p = SP_FMA_C(p, z, 1.0 / 7.0);
  p = SP_FMA_C(p, z, 1.0 / 5.0);
  p = SP_FMA_C(p, z, 1.0 / 3.0);
  const double s2 = s + s;
  const double lm = SP_FMA(s2 * z, p, s2);  // 2*atanh(s)
  const double ed = (double)e;
  // E*ln2 in two parts (ln2_hi has 32 trailing zero bits: E*ln2_hi is exact for |E| < 2^20)
  return SP_FMA_MC(ed, 0.693147180369123816490, SP_FMA_MC(ed, 1.90821492927058770002e-10, lm));
}

// exp(y) for |y| < 700
SP_FN double sp_exp(double y) {
  const double n = SP_RINT(y * 1.44269504088896340736);
  double r = SP_FMA_MC(n, -0.693147180369123816490, y);
  r = SP_FMA_MC(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;           // 1/13!
  p = SP_FMA_C(p, r, 1.0 / 479001600.0);
  p = SP_FMA_C(p, r, 1.0 / 39916800.0);
  p = SP_FMA_C(p, r, 1.0 / 3628800.0);
  p = SP_FMA_C(p, r, 1.0 / 362880.0);
  p = SP_FMA_C(p, r, 1.0 / 40320.0);
  p = SP_FMA_C(p, r, 1.0 / 5040.0);
  p = SP_FMA_C(p, r, 1.0 / 720.0);
  p = SP_FMA_C(p, r, 1.0 / 120.0);
  p = SP_FMA_C(p, r, 1.0 / 24.0);
  p = SP_FMA_C(p, r, 1.0 / 6.0);
  p = SP_FMA(p, r, 0.5);
  p = SP_FMA(p, r, 1.0);
  p = SP_FMA(p, r, 1.0);
  return SP_LDEXP(p, (int)n);
}

// x**3.1, x >= 0 (0 -> 0; a cube that underflows gives 0 like the product 1e-17 * x**3.1 it feeds): the plain form
SP_FN double sp_pow_3p1_plain(double x) {
  const double x3 = x * x * x;
  if (!(x3 > 0.0)) return x3;
  return x3 * sp_exp(0.1 * sp_log(x));
}

// The same through a tenth root: y = x**0.1 seeded by the hardware's single-precision log2 / exp2 (v_log_f32, v_exp_f32: relative
// error of the seed e0 ~ 2e-7 for 2^-100 <= x <= 2^100) and ONE Halley step on y**10 = x,
//     y <- y * (9 y**10 + 11 x) / (11 y**10 + 9 x),      error 8.25 * e0**3 ~ 1e-19,
// so the root is as good as its six roundings (2-3 ulp), and x**3.1 = x*x*x*y within ~4 ulp of the correctly rounded power like the
// plain form -- in 26 vector instructions instead of 63.  Outside that range of x (a liquid fraction below 1e-33, zero, NaN) the
// plain form is taken, per lane: a column's bits never depend on its wave-mates.
#if defined(__HIPCC__)
SP_FN double sp_pow_3p1(double x) {
  if (!(x >= 0x1p-100 && x <= 0x1p100)) return sp_pow_3p1_plain(x);
  const double x3 = x * x * x;
  const double y = (double)__builtin_amdgcn_exp2f(0.1f * __builtin_amdgcn_logf((float)x));
  const double y2 = y * y, y4 = y2 * y2, y8 = y4 * y4, t = y8 * y2;
  const double num = SP_FMA(9.0, t, 11.0 * x), den = SP_FMA(11.0, t, 9.0 * x);
  return x3 * (y * SP_QUOT(num, den));
}
#else
SP_FN double sp_pow_3p1(double x) { return sp_pow_3p1_plain(x); }
#endif
// x**1.5 and x**4 of the per-step scalar code (freezing point mo_functions.f90:239-250, sea-water density :51-62, the two radiative
// iterations mo_heat_fluxes.f90:115-148): the reference's compiler calls pow() for both.  x*sqrt(x) (the square root is correctly
// rounded, on the host and on the GPU) and (x*x)*(x*x) are within 1.5 ulp of the exact power at a fifteenth of pow()'s instructions;
// tests/test_host_logic.py measures them against powl with this header compiled for the host.
#if defined(__HIPCC__)
SP_FN double sp_pow_1p5(double x) { return x * __builtin_sqrt(x); }
#else
SP_FN double sp_pow_1p5(double x) { return x * sqrt(x); }
#endif
SP_FN double sp_pow_4(double x) { const double t = x * x; return t * t; }
#endif
