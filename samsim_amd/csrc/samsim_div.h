// samsim_div.h -- the FP64 reciprocal and quotient of the sweeps (included by samsim_kernels.hip and by tools/div_probe.hip, which
// measures on the GPU how far they are from 1.0/x and a/b).
//
// The compiler's own a/b is v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal, a*r, one residual correction,
// v_div_fmas, v_div_fixup: eleven vector instructions on the critical path of every layer.  The sweeps divide by normal-range
// numbers (m, thick, S_br, sums of resistances), for which the scaling and the special-case fix-up do nothing, and the parity bar
// is 1e-6 relative, so they use the arithmetic core of that sequence alone, with ONE Newton step: v_rcp_f64 is good to about
// 2^-23, one step squares that (2^-46), and the residual correction q + r*(a - b*q) leaves a relative error of 2^-92 before the
// final rounding -- a quotient within one ulp of the correctly rounded one in six instructions; 1/x with two steps likewise in
// five.  (Round 2 kept one more step in each, which made them the bits of 1.0/x and a/b; tools/div_probe counts how many of 2^26
// operand pairs differ from those now, and by how many ulp at most: one.)
#ifndef SAMSIM_DIV_H
#define SAMSIM_DIV_H
__device__ __forceinline__ double recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double quot(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
#endif
