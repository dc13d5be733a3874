// layout_probe.hip -- does the memory layout of the layer block matter for this path's access pattern?
// One lane walks its column twice per "step" (down: 7 loads + 4 stores per layer, up: 6 loads + 4 stores per layer) with a
// dependent FP64 chain per layer and 4 waves/SIMD, like samsim_step_kernel.  Layout A = [array][layer][column] (8 MB between
// two rows at 1 M columns), layout B = [column block of 64][layer][array][64 lanes] (a wave's whole column block is one
// contiguous 0.6 MB piece).
//   hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/layout_probe && tools/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NA = 12, NL = 80, WORK = 24;

template <bool BLOCKED>
__device__ __forceinline__ size_t idx(size_t ncol, size_t col, int a, int k) {
  if (BLOCKED) return ((col >> 6) * (size_t)(NL * NA) + (size_t)(k * NA + a)) * 64 + (col & 63);
  return ((size_t)a * NL + k) * ncol + col;
}

template <bool BLOCKED>
__global__ void __launch_bounds__(64, 4) walk(double *__restrict__ lay, size_t ncol, int steps) {
  extern __shared__ double pad[];          // sized so that 16 blocks fit a CU: 4 waves / SIMD
  const size_t col = (size_t)blockIdx.x * 64 + threadIdx.x;
  double carry = 0.0;
  for (int s = 0; s < steps; ++s) {
    for (int k = 0; k < NL; ++k) {                       // down sweep
      double v = carry;
      for (int a = 0; a < 7; ++a) v += lay[idx<BLOCKED>(ncol, col, a, k)];
      for (int i = 0; i < WORK; ++i) v = v * 1.0000001 + 1e-9;
      for (int a = 7; a < 11; ++a) lay[idx<BLOCKED>(ncol, col, a, k)] = v + a;
      carry = v * 1e-3;
    }
    for (int k = NL - 1; k >= 0; --k) {                  // up sweep
      double v = carry;
      for (int a = 5; a < 11; ++a) v += lay[idx<BLOCKED>(ncol, col, a, k)];
      for (int i = 0; i < 2 * WORK; ++i) v = v * 1.0000001 + 1e-9;
      for (int a = 0; a < 4; ++a) lay[idx<BLOCKED>(ncol, col, a, k)] = v * 1e-6 + a;
      carry = v * 1e-3;
    }
  }
  if (carry == 123.456) pad[threadIdx.x] = carry;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t ncol = 1 << 20;
  double *lay;
  CHK(hipMalloc(&lay, ncol * NA * NL * sizeof(double)));
  CHK(hipMemset(lay, 0, ncol * NA * NL * sizeof(double)));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int steps = 10;
  const double bytes = (double)ncol * NL * (7 + 4 + 6 + 4) * 8.0 * steps;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(walk<false>, dim3(ncol / 64), dim3(64), 10000, 0, lay, ncol, steps);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("layout A [array][layer][column]       : %.1f ms  %.0f GB/s logical\n", ms, bytes / 1e6 / ms);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(walk<true>, dim3(ncol / 64), dim3(64), 10000, 0, lay, ncol, steps);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("layout B [block64][layer][array][lane]: %.1f ms  %.0f GB/s logical\n", ms, bytes / 1e6 / ms);
  }
  return 0;
}
