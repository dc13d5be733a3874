// div_probe.hip -- how far are recip(x) / quot(a, b) of samsim_amd/csrc/samsim_div.h (v_rcp_f64 + Newton steps: the arithmetic core
// of the compiler's own FP64 division without operand scaling and special-case fix-up, one Newton step short of it) from 1.0/x and
// a/b, and sp_pow_3p1 of samsim_pow.h (x**3.1 through a hardware-seeded tenth root) from pow(x, 3.1)?
// 2^26 pseudo-random operand pairs over the magnitudes the sweeps divide by (2^-40 .. 2^40, both signs), and 2^26 arguments of the
// permeability law (1000 * liquid fraction: 1e-9 .. 3000, plus a share below 2^-100 where the plain form takes over), on the GPU.
//   make -C samsim_amd/csrc div_probe && tools/div_probe   (tests/test_gpu_parity.py runs it and asserts the bounds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

#include "../samsim_amd/csrc/samsim_div.h"
#define SP_QUOT(a, b) quot(a, b)
#include "../samsim_amd/csrc/samsim_pow.h"

__device__ __forceinline__ uint64_t mix(uint64_t z) {  // splitmix64
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double operand(uint64_t h, int emax) {
  // mantissa from the hash, exponent uniform in [-emax, emax], random sign
  const double m = 1.0 + (double)(h >> 12) * (1.0 / 4503599627370496.0);
  const int e = (int)((h >> 3) % (unsigned)(2 * emax + 1)) - emax;
  const double v = ldexp(m, e);
  return (h & 1) ? -v : v;
}
__device__ __forceinline__ unsigned long long ulps(double a, double b) {
  const long long d = __double_as_longlong(a) - __double_as_longlong(b);
  return (unsigned long long)(d < 0 ? -d : d);
}

// out: [0] recip != 1/x, [1] max ulp, [2] quot != a/b, [3] max ulp, [4] max rel err * 2^80 of sp_pow_3p1 over 1e-9..3000, [5] of the
//      plain form there, [6] arguments below 2^-100 (plain form either way), [7] max rel err * 2^80 over those,
//      [8] max relative difference * 2^80 between the two forms over 1e-9..3000
__global__ void probe(unsigned long long *out, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = operand(mix(2 * i), 40), b = operand(mix(2 * i + 1), 40);
  const double r0 = 1.0 / b, r1 = recip(b), q0 = a / b, q1 = quot(a, b);
  if (r0 != r1) { atomicAdd(out + 0, 1ull); atomicMax(out + 1, ulps(r0, r1)); }
  if (q0 != q1) { atomicAdd(out + 2, 1ull); atomicMax(out + 3, ulps(q0, q1)); }
  // permeability arguments: 15 of 16 log-uniform in [1e-9, 3000], the rest log-uniform in [2^-300, 2^-90]
  const uint64_t h = mix(3 * i + 7);
  const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
  const double x = ((h & 15) != 0) ? exp(log(1e-9) + u * (log(3000.0) - log(1e-9))) : ldexp(1.0 + u, -300 + (int)((h >> 4) % 210));
  const double ref = pow(x, 3.1);
  const double pf = sp_pow_3p1(x), pp = sp_pow_3p1_plain(x);
  const double e1 = fabs(pf - ref) / ref, e2 = fabs(pp - ref) / ref;
  if (x >= 0x1p-100 && x <= 0x1p100) {
    atomicMax(out + 4, (unsigned long long)(e1 * 0x1p80));
    atomicMax(out + 5, (unsigned long long)(e2 * 0x1p80));
    atomicMax(out + 8, (unsigned long long)(fabs(pf - pp) / pp * 0x1p80));
  } else {
    atomicAdd(out + 6, 1ull);
    atomicMax(out + 7, (unsigned long long)(e1 * 0x1p80));
  }
}

int main() {
  unsigned long long *d, h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long n = 1ull << 26;
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, n);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("operands %llu  recip != 1.0/x: %llu (max %llu ulp)  quot != a/b: %llu (max %llu ulp)\n", n, h[0], h[1], h[2], h[3]);
  // (the plain form is within 7e-16 of the exact x**(3 + 0.1) on the CPU, tests/test_host_logic.py; the device's pow() is the looser
  // of the two references here)
  printf("pow_3p1 vs the device's pow(x, 3.1) over 1e-9..3000: max rel err %.3e (plain form %.3e); tenth-root form vs plain form: %.3e; "
         "%llu arguments below 2^-100 (plain form): %.3e\n",
         (double)h[4] * 0x1p-80, (double)h[5] * 0x1p-80, (double)h[8] * 0x1p-80, h[6], (double)h[7] * 0x1p-80);
  (void)hipFree(d);
  return 0;
}
