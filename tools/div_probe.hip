// div_probe.hip -- are recip(x) / quot(a, b) of samsim_amd/csrc/samsim_div.h (SAMSIM_FAST_DIV 2, 3: v_rcp_f64 + Newton steps, the
// arithmetic of the compiler's own FP64 division sequence without operand scaling and special-case fix-up) the same bits as 1.0/x and a/b?
// 2^26 pseudo-random operand pairs over the magnitudes the sweeps divide by (1e-12 .. 1e12, both signs), counted on the GPU.
//   make -C samsim_amd/csrc div_probe && tools/div_probe   (tests/test_gpu_parity.py runs it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "../samsim_amd/csrc/samsim_div.h"

__device__ __forceinline__ uint64_t mix(uint64_t z) {  // splitmix64
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double operand(uint64_t h) {
  // mantissa from the hash, exponent uniform in [-40, 40], random sign
  const double m = 1.0 + (double)(h >> 12) * (1.0 / 4503599627370496.0);
  const int e = (int)((h >> 3) % 81) - 40;
  const double v = ldexp(m, e);
  return (h & 1) ? -v : v;
}

__global__ void probe(unsigned long long *bad_recip, unsigned long long *bad_quot, unsigned long long *ulp_recip, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = operand(mix(2 * i)), b = operand(mix(2 * i + 1));
  const double r0 = 1.0 / b, r1 = recip(b), q0 = a / b, q1 = quot(a, b);
  if (r0 != r1) {
    atomicAdd(bad_recip, 1ull);
    const long long d = __double_as_longlong(r0) - __double_as_longlong(r1);
    atomicMax(ulp_recip, (unsigned long long)(d < 0 ? -d : d));
  }
  if (q0 != q1) atomicAdd(bad_quot, 1ull);
}

int main() {
  unsigned long long *d, h[3] = {0, 0, 0};
  const unsigned long long n = 1ull << 26;
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, d + 1, d + 2, n);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("operands %llu  recip != 1.0/x: %llu (max %llu ulp)  quot != a/b: %llu\n", n, h[0], h[2], h[1]);
  (void)hipFree(d);
  return 0;
}
