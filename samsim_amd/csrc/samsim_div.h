// samsim_div.h -- the FP64 reciprocal and quotient of the fused sweeps (included by samsim_kernels.hip and by
// tools/div_probe.hip, which checks on the GPU that they return the bits of 1.0/x and a/b).
#ifndef SAMSIM_DIV_H
#define SAMSIM_DIV_H
// SAMSIM_FAST_DIV 2: 1/x as the compiler's own division sequence forms it -- v_rcp_f64 and three Newton steps -- without the
// operand scaling and the special-case fix-up around it (v_div_scale x2, v_div_fmas, v_div_fixup): the divisors are normal-range
// numbers (m, thick, S_br, ...), for which both give the same bits (tools/div_probe.hip checks 2^26 operands on the GPU)
__device__ __forceinline__ double recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
// SAMSIM_FAST_DIV 3: the other quotients of the two fused sweeps the same way (two Newton steps, a*r, one residual correction: the
// arithmetic of the compiler's sequence, 8 instructions instead of 11)
__device__ __forceinline__ double quot(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
#endif
