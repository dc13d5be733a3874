// hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS path's access pattern
// (8 bytes per lane, 512 B per wave instruction, [layer][column] walk), as MI355X_MICROARCH.md section HBM asks:
// "calibrate on a known byte count in your own access pattern before trusting an absolute".
// Kernel read8 fetches exactly nlayer*ncol*8 bytes, kernel write8 stores exactly that many.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
//   rocprofv3 --pmc FETCH_SIZE -- tools/hbm_calib     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double *__restrict__ a, double *__restrict__ out, size_t ncol, int nlayer) {
  size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  double s = 0.0;
  for (int k = 0; k < nlayer; ++k) s += a[(size_t)k * ncol + col];
  if (s == 12345.678) out[col] = s;  // never true for the fill value: keeps the loads alive without a store stream
}

__global__ void write8(double *__restrict__ a, size_t ncol, int nlayer, double v) {
  size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncol) return;
  for (int k = 0; k < nlayer; ++k) a[(size_t)k * ncol + col] = v + k;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t ncol = 1 << 20;
  const int nlayer = 400;                     // 3.36 GB, far beyond the 256 MiB Infinity Cache
  double *a, *out;
  CHK(hipMalloc(&a, ncol * nlayer * sizeof(double)));
  CHK(hipMalloc(&out, ncol * sizeof(double)));
  CHK(hipMemset(a, 0, ncol * nlayer * sizeof(double)));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(write8, dim3(ncol / 64), dim3(64), 0, 0, a, ncol, nlayer, 1.0);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("write8: %.3f GB in %.3f ms = %.1f GB/s\n", ncol * nlayer * 8.0 / 1e9, ms, ncol * nlayer * 8.0 / 1e6 / ms);
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(read8, dim3(ncol / 64), dim3(64), 0, 0, a, out, ncol, nlayer);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("read8 : %.3f GB in %.3f ms = %.1f GB/s\n", ncol * nlayer * 8.0 / 1e9, ms, ncol * nlayer * 8.0 / 1e6 / ms);
  }
  printf("bytes_per_kernel %zu\n", ncol * nlayer * sizeof(double));
  return 0;
}
