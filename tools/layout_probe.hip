// layout_probe.hip -- what does the memory system give this path's access pattern, and does the position of the hot arrays matter?
// One lane walks its column twice per "step" exactly as the fused sweeps of samsim_step_kernel do: down (top -> bottom) loads T, S_abs,
// m, H_abs of a layer and stores m, S_abs, H_abs; up (bottom -> top) loads H_abs, m, S_abs and stores T: 7 loads + 4 stores = 88 bytes
// per layer-cell, with a dependent FP64 chain per layer (WORK fused multiply-adds down, 2 WORK up), 4 waves/SIMD, rows requested where
// they are used.  Layouts of the layer block (NA = 16 arrays of NL layers):
//   0  [block64][layer][array][lane], hot arrays H_abs, S_abs, m at 0..2 and T at 4 (round 2 / the product's order: thick sits between)
//   1  the same with T at 3: the four hot arrays of a layer row are one contiguous 2 KB piece
//   2  [block64][array][layer][lane]: consecutive layers of one array are contiguous (512 B apart), arrays NL * 512 B apart
//   3  [array][layer][column] (round 1: 8 MB between two rows at 1 M columns)
//   4  [block64][layer pair][array][lane][2]: a lane's two values of a layer pair are adjacent, one 16-byte access moves both
//      (dwordx4: 1 KB per wave instruction, half as many requests)
//   hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/layout_probe && tools/layout_probe [WORK]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NA = 16, NL = 80;

template <int LAYOUT>
__device__ __forceinline__ size_t idx(size_t ncol, size_t col, int a, int k) {
  if (LAYOUT == 0 || LAYOUT == 1) return ((col >> 6) * (size_t)(NL * NA) + (size_t)(k * NA + a)) * 64 + (col & 63);
  if (LAYOUT == 2) return ((col >> 6) * (size_t)(NL * NA) + (size_t)(a * NL + k)) * 64 + (col & 63);
  return ((size_t)a * NL + k) * ncol + col;
}

// NT: the loads carry the non-temporal hint, as the product's sweeps do
template <bool NT>
__device__ __forceinline__ double ld(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int LAYOUT, bool NT = false>
__global__ void __launch_bounds__(64, 4) walk(double *__restrict__ lay, size_t ncol, int steps, int work) {
  extern __shared__ double pad[];          // sized so that 16 blocks fit a CU: 4 waves / SIMD
  const size_t col = (size_t)blockIdx.x * 64 + threadIdx.x;
  constexpr int H = 0, S = 1, M = 2, T = (LAYOUT == 0) ? 4 : 3;
  double carry = 0.0;
  for (int s = 0; s < steps; ++s) {
    for (int k = 0; k < NL; ++k) {                       // down sweep
      double v = carry + ld<NT>(lay + idx<LAYOUT>(ncol, col, T, k)) + ld<NT>(lay + idx<LAYOUT>(ncol, col, S, k)) + ld<NT>(lay + idx<LAYOUT>(ncol, col, M, k)) +
                 ld<NT>(lay + idx<LAYOUT>(ncol, col, H, k));
      for (int i = 0; i < work; ++i) v = v * 1.0000001 + 1e-9;
      lay[idx<LAYOUT>(ncol, col, M, k)] = v + 1.0;
      lay[idx<LAYOUT>(ncol, col, S, k)] = v + 2.0;
      lay[idx<LAYOUT>(ncol, col, H, k)] = v + 3.0;
      carry = v * 1e-3;
    }
    for (int k = NL - 1; k >= 0; --k) {                  // up sweep
      double v = carry + ld<NT>(lay + idx<LAYOUT>(ncol, col, H, k)) + ld<NT>(lay + idx<LAYOUT>(ncol, col, M, k)) + ld<NT>(lay + idx<LAYOUT>(ncol, col, S, k));
      for (int i = 0; i < 2 * work; ++i) v = v * 1.0000001 + 1e-9;
      lay[idx<LAYOUT>(ncol, col, T, k)] = v * 1e-6;
      carry = v * 1e-3;
    }
  }
  if (carry == 123.456) pad[threadIdx.x] = carry;
}

// layout 4: two layers per access
__global__ void __launch_bounds__(64, 4) walk_pairs(double2 *__restrict__ lay, size_t ncol, int steps, int work) {
  extern __shared__ double pad[];
  const size_t col = (size_t)blockIdx.x * 64 + threadIdx.x;
  constexpr int H = 0, S = 1, M = 2, T = 3;
  auto at = [&](int a, int kp) -> double2 & { return lay[((col >> 6) * (size_t)((NL / 2) * NA) + (size_t)(kp * NA + a)) * 64 + (col & 63)]; };
  double carry = 0.0;
  for (int s = 0; s < steps; ++s) {
    for (int kp = 0; kp < NL / 2; ++kp) {                // down sweep, a layer pair per trip
      const double2 t = at(T, kp), sa = at(S, kp), m = at(M, kp), h = at(H, kp);
      double v = carry + t.x + sa.x + m.x + h.x;
      for (int i = 0; i < work; ++i) v = v * 1.0000001 + 1e-9;
      double w = v * 1e-3 + t.y + sa.y + m.y + h.y;
      for (int i = 0; i < work; ++i) w = w * 1.0000001 + 1e-9;
      at(M, kp) = make_double2(v + 1.0, w + 1.0);
      at(S, kp) = make_double2(v + 2.0, w + 2.0);
      at(H, kp) = make_double2(v + 3.0, w + 3.0);
      carry = w * 1e-3;
    }
    for (int kp = NL / 2 - 1; kp >= 0; --kp) {           // up sweep
      const double2 h = at(H, kp), m = at(M, kp), sa = at(S, kp);
      double v = carry + h.y + m.y + sa.y;
      for (int i = 0; i < 2 * work; ++i) v = v * 1.0000001 + 1e-9;
      double w = v * 1e-3 + h.x + m.x + sa.x;
      for (int i = 0; i < 2 * work; ++i) w = w * 1.0000001 + 1e-9;
      at(T, kp) = make_double2(w * 1e-6, v * 1e-6);
      carry = w * 1e-3;
    }
  }
  if (carry == 123.456) pad[threadIdx.x] = carry;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LAYOUT, bool NT = false>
int run(double *lay, size_t ncol, int steps, int work, const char *name) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const double bytes = (double)ncol * NL * (7 + 4) * 8.0 * steps;
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((walk<LAYOUT, NT>), dim3(ncol / 64), dim3(64), 10000, 0, lay, ncol, steps, work);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("work %3d  %-58s: %7.1f ms  %5.0f GB/s of the 88 B per layer-cell\n", work, name, best, bytes / 1e6 / best);
  return 0;
}

int main(int argc, char **argv) {
  const size_t ncol = 1 << 20;
  double *lay;
  CHK(hipMalloc(&lay, ncol * NA * NL * sizeof(double)));
  CHK(hipMemset(lay, 0, ncol * NA * NL * sizeof(double)));
  const int steps = 10;
  for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
    const int work = argc > 1 ? atoi(argv[a]) : 24;
    if (run<0>(lay, ncol, steps, work, "[block][layer][array][lane], T behind thick (the product)")) return 1;
    if (run<0, true>(lay, ncol, steps, work, "the same with non-temporal loads (the product's sweeps)")) return 1;
    if (run<1>(lay, ncol, steps, work, "[block][layer][array][lane], H_abs S_abs m T contiguous")) return 1;
    if (run<2>(lay, ncol, steps, work, "[block][array][layer][lane]")) return 1;
    if (run<3>(lay, ncol, steps, work, "[array][layer][column]")) return 1;
    {
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      const double bytes = (double)ncol * NL * (7 + 4) * 8.0 * steps;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(walk_pairs, dim3(ncol / 64), dim3(64), 10000, 0, (double2 *)lay, ncol, steps, work);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("work %3d  %-58s: %7.1f ms  %5.0f GB/s of the 88 B per layer-cell\n", work, "[block][layer pair][array][lane][2], 16-byte accesses", best, bytes / 1e6 / best);
    }
  }
  return 0;
}
