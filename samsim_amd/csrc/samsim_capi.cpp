// samsim_capi.cpp -- host side of the C-ABI declared in include/samsim.h.
//
// Owns the HBM allocations of a handle (layout: samsim_device.h), mirrors the uniform clock of the ensemble on
// the host (time, step, output counter, forcing-table cursor are the same for every column, so nothing has to
// be read back to know them) and launches the step kernel on the handle's HIP stream.  There is no CPU
// implementation of the physics here or anywhere else in the product: without a HIP device every entry point
// that would compute returns SAMSIM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "samsim_device.h"

extern "C" hipError_t samsim_launch_step(const DevParams *d_params, const DevParams *hp, long long grid, hipStream_t stream);

namespace {

constexpr int kRing = 16;

thread_local char g_last_hip_error[256] = "";

bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  std::snprintf(g_last_hip_error, sizeof(g_last_hip_error), "%s: %s", what, hipGetErrorString(e));
  return false;
}
#define HIPCHK(call)                                  \
  do {                                                \
    if (!hip_ok((call), #call)) return SAMSIM_ERR_HIP; \
  } while (0)

}  // namespace

struct samsim_handle {
  samsim_config cfg{};
  long long ncol = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  // A step of a large ensemble is two launches: the first half of the 64-column blocks on `stream`, the rest on `stream2`.  Columns
  // never meet, so each part only follows its own previous launch; the workgroups of a launch finish raggedly (the last of the
  // four rounds of a 16 384-block launch leaves the chip partly idle for 15 % of a workgroup's run time), and with two launches
  // in flight per step, and the next step's enqueued behind them, a draining launch is topped up by the others.  Every other entry point waits for both.
  hipStream_t stream2 = nullptr;
  hipEvent_t fork = nullptr;      // recorded on `stream` when other work was enqueued there since the last launch: stream2 waits for it
  bool other_work = true;
  int split_eighths = 4;          // share of the first part in eighths (samsim_set_launch_split; 4, 5, 6, 7 eighths measured 962, 973,
                                  // 968, 992 ms per 500-step step of 1 048 576 columns, one launch 1 009 ms; three and four parts 1 015, 969-986)
  long long split_blocks = 8192;  // a launch of at least this many 64-column blocks is split (samsim_set_launch_split; 0 = never)
  hipEvent_t ev0b = nullptr, ev1b = nullptr;
  // device memory
  double *lay = nullptr, *scal = nullptr;
  int32_t *n_active = nullptr, *status = nullptr, *err_layer = nullptr;
  long long *err_step = nullptr, *work = nullptr;
  double *spec = nullptr;      // hand-over block of the up sweep, [DEV_NSPEC][ncol]
  int32_t *flags = nullptr;    // COLF_* per column
  void *d_stat = nullptr;      // block partials of samsim_get_ensemble_stats
  double *stage = nullptr;     // staging buffer of samsim_set_state / samsim_get_state (boundary layout), grown on demand and kept:
  size_t stage_n = 0;          // no hipMalloc / hipFree -- both wait for the whole device -- per call
  // passive tracers (bgc_flag 2)
  double *bgc = nullptr, *bgc_bot = nullptr, *bfl = nullptr, *out_bgc = nullptr, *out_bgc_bot = nullptr;
  int32_t n_bgc = 0;
  double bgc_total0 = 0.0;
  double *f_sw = nullptr, *f_lw = nullptr, *f_T2m = nullptr, *f_precip = nullptr;
  int32_t flen = 0, nsites = 1;
  int32_t *site = nullptr;     // [ncol] forcing set of each column (nsites > 1)
  double *ocean_dflq = nullptr, *ocean_sbu = nullptr;   // samsim_set_ocean: per-column oceanic heat-flux offset / salinity below
  double *out_lay = nullptr, *out_scal = nullptr;
  int32_t *out_n_active = nullptr;
  long long out_col0 = 0, out_ncols = 0;
  // parameter ring (pinned host + device), one event per slot
  DevParams *h_params = nullptr, *d_params = nullptr;
  hipEvent_t slot_done[kRing]{};
  int slot = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // uniform clock, mirrored on the host
  samsim_clock clk{};
  bool stepped = false;        // the kernel has run since the last samsim_set_state: S_bu is refreshed from S_abs / m on get_state
  bool snap_valid = false;
  double snap_time = 0.0;
  long long snap_step = 0;
  double p17 = 0, p14 = 0, tf_c3 = 0;
};

namespace {

// the handle's staging buffer with room for n doubles
hipError_t stage_for(samsim_handle *h, size_t n) {
  if (n <= h->stage_n) return hipSuccess;
  (void)hipFree(h->stage);
  h->stage = nullptr; h->stage_n = 0;
  hipError_t e = hipMalloc((void **)&h->stage, n * sizeof(double));
  if (e == hipSuccess) h->stage_n = n;
  return e;
}

int validate(const samsim_config &c) {
  if (c.struct_size != (int32_t)sizeof(samsim_config)) return SAMSIM_ERR_ABI;
  if (c.nlayer < 3 || c.nlayer > SAMSIM_MAX_NLAYER) return SAMSIM_ERR_ARG;
  if (c.n_top < 3 || c.n_bottom < 1 || c.n_middle < 1 || c.n_top + c.n_middle + c.n_bottom != c.nlayer) return SAMSIM_ERR_ARG;
  if (!(c.dt > 0.0) || !(c.thick_0 > 0.0) || c.i_time_out < 0) return SAMSIM_ERR_ARG;
  auto in = [](int v, std::initializer_list<int> ok) { for (int o : ok) if (v == o) return true; return false; };
  // testcases whose per-step specifics (mo_grotz.f90:505-565) read the lab's forcing tables or re-impose a snow cover: refused
  // rather than run without them
  if (in(c.testcase, {8, 44, 45, 99, 101, 102, 103, 104, 105, 111})) return SAMSIM_ERR_UNSUPPORTED;
  if (!in(c.boundflux_flag, {1, 2, 3}) || (c.boundflux_flag == 3 && c.lab_snow_flag != 0)) return SAMSIM_ERR_UNSUPPORTED;
  if (!in(c.atmoflux_flag, {1, 2, 3})) return SAMSIM_ERR_UNSUPPORTED;
  if (c.tank_flag == 2 && !(c.m_total > 0.0)) return SAMSIM_ERR_ARG;
  if (!in(c.grav_flag, {1, 2, 3}) || !in(c.prescribe_flag, {1, 2}) || !in(c.grav_heat_flag, {1, 2}) || !in(c.flush_heat_flag, {1, 2}))
    return SAMSIM_ERR_UNSUPPORTED;
  if (!in(c.turb_flag, {1, 2}) || !in(c.salt_flag, {1, 2}) || !in(c.flush_flag, {1, 4, 5, 6}) || !in(c.flood_flag, {1, 2, 3}))
    return SAMSIM_ERR_UNSUPPORTED;
  if (!in(c.bottom_flag, {1, 2}) || !in(c.precip_flag, {0, 1}) || !in(c.harmonic_flag, {1, 2}) || !in(c.tank_flag, {1, 2}))
    return SAMSIM_ERR_UNSUPPORTED;
  if (!in(c.albedo_flag, {1, 2}) || !in(c.freeboard_snow_flag, {0, 1}) || !in(c.snow_flush_flag, {0, 1}) || !in(c.bgc_flag, {1, 2}))
    return SAMSIM_ERR_UNSUPPORTED;
  return SAMSIM_OK;
}

template <typename T>
hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }

bool launch_needs_forcing(const samsim_config &c) { return c.atmoflux_flag == 2; }

// ---- ensemble statistics: two passes (mean / min / max, then the squared deviations) over one [ncol] row, columns with a
// STOP code skipped; fixed grid + tree reduction, so the result does not depend on scheduling
constexpr int kStatBlock = 256, kStatGrid = 512;
struct StatPartial { double sum, mn, mx, ssq; long long n; };

template <bool PASS2>
__global__ void __launch_bounds__(kStatBlock) stat_kernel(const double *__restrict__ row, const int32_t *__restrict__ irow,
                                                          const int32_t *__restrict__ status, long long n, double mean,
                                                          StatPartial *__restrict__ part) {
  __shared__ double s_sum[kStatBlock], s_mn[kStatBlock], s_mx[kStatBlock];
  __shared__ long long s_n[kStatBlock];
  double sum = 0.0, mn = 1.0e300, mx = -1.0e300;
  long long cnt = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (status[i]) continue;
    const double v = row ? row[i] : (double)irow[i];
    if (PASS2) {
      sum += (v - mean) * (v - mean);
    } else {
      sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++cnt;
    }
  }
  const int t = threadIdx.x;
  s_sum[t] = sum; s_mn[t] = mn; s_mx[t] = mx; s_n[t] = cnt;
  __syncthreads();
  for (int w = kStatBlock / 2; w > 0; w >>= 1) {
    if (t < w) {
      s_sum[t] += s_sum[t + w];
      s_mn[t] = s_mn[t + w] < s_mn[t] ? s_mn[t + w] : s_mn[t];
      s_mx[t] = s_mx[t + w] > s_mx[t] ? s_mx[t + w] : s_mx[t];
      s_n[t] += s_n[t + w];
    }
    __syncthreads();
  }
  if (t == 0) {
    if (PASS2) part[blockIdx.x].ssq = s_sum[0];
    else { part[blockIdx.x].sum = s_sum[0]; part[blockIdx.x].mn = s_mn[0]; part[blockIdx.x].mx = s_mx[0]; part[blockIdx.x].n = s_n[0]; }
  }
}

// fill a [rows][ncol] device block with one value per row-set
__global__ void fill_rows(double *dst, size_t n, double v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}
__global__ void fill_i32(int32_t *dst, size_t n, int32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

// S_bu(k) = S_abs(k)/m(k) for the active layers: the kernel keeps the bulk salinity in registers only (every reader
// derives it from S_abs and m), so the array is brought up to date when the host asks for the state
__global__ void refresh_s_bu(double *lay, const int32_t *n_active, const int32_t *status, size_t ncol, int N, size_t col0, size_t w) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w) return;
  const size_t c = col0 + i;
  if (status[c]) return;   // a column that stopped keeps what it held (its masses may be zero)
  const int na = n_active[c];
  for (int k = 0; k < na; ++k) {
    const double m = lay[DEV_LAY_INDEX(SAMSIM_A_M, k, c, N, ncol)];
    if (m != 0.0) lay[DEV_LAY_INDEX(SAMSIM_A_S_BU, k, c, N, ncol)] = lay[DEV_LAY_INDEX(SAMSIM_A_S_ABS, k, c, N, ncol)] / m;
  }
}

// The boundary keeps [array][layer][column] (samsim_state_soa); the device layout is DEV_LAY_INDEX.  A window of columns passes
// through a staging buffer in the boundary's layout: scatter = staging -> device layout, gather = the reverse.
template <bool GATHER>
__global__ void lay_window(double *lay, double *stage, int narr, int N, size_t ncol, size_t col0, size_t w) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)narr * N * w) return;
  const size_t cw = i % w, ak = i / w;
  const int k = (int)(ak % (size_t)N), a = (int)(ak / (size_t)N);
  const size_t j = DEV_LAY_INDEX(a, k, col0 + cw, N, ncol);
  if (GATHER) stage[i] = lay[j]; else lay[j] = stage[i];
}
// one value into every layer of one array
__global__ void lay_fill_array(double *lay, int a, int N, size_t ncol, double v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * ncol) return;
  lay[DEV_LAY_INDEX(a, (int)(i / ncol), i % ncol, N, ncol)] = v;
}

hipError_t fill(double *dst, size_t n, double v, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, n, v);
  return hipGetLastError();
}

void advance_clock(samsim_handle *h, long long nsteps) {
  samsim_clock &k = h->clk;
  const samsim_config &c = h->cfg;
  for (long long s = 0; s < nsteps; ++s) {
    if (c.atmoflux_flag == 2) {
      const double ti = ((double)(float)k.time_counter - 1.0) * 3600.0 * 3.0;
      if (k.time > ti) k.time_counter += 1;
      if (k.time_counter > h->flen) k.time_counter = h->flen;
    }
    const bool out = (k.n_time_out == c.i_time_out) || (k.step + 1 == 1);
    if (out) {
      k.n_time_out = 0;
      k.n_outputs += 1;
      h->snap_valid = h->out_ncols > 0;
      h->snap_time = k.time;
      h->snap_step = k.step + 1;
    } else {
      k.n_time_out += 1;
    }
    k.time = k.time + c.dt;
    k.step += 1;
  }
}

int launch(samsim_handle *h, long long nsteps) {
  if (nsteps <= 0) return SAMSIM_OK;
  // the forcing tables are read whenever atmoflux_flag is 2 (T2m / precipitation in every step, the radiative fluxes with
  // boundflux_flag 2): no launch without them, whatever the other flags say
  if (launch_needs_forcing(h->cfg) && (!h->f_sw || !h->f_lw || !h->f_T2m || !h->f_precip || h->flen < 2)) return SAMSIM_ERR_ARG;
  if (h->cfg.bgc_flag == 2 && h->n_bgc < 1) return SAMSIM_ERR_ARG;   // samsim_set_tracers first
  const long long nblk = (h->ncol + 63) / 64;
  // two parts from 8 192 blocks up (two rounds of the chip's 4 096 wave slots): below that a launch has no rounds to speak of
  const bool split = h->stream2 && h->split_blocks > 0 && nblk >= h->split_blocks && nblk >= 2;
  const int nparts = split ? 2 : 1;
  auto stream_of = [&](int part) { return part == 0 ? h->stream : h->stream2; };
  // part boundaries: the first part takes split_eighths/8 of the blocks
  auto bound = [&](int part) -> long long {
    if (part <= 0) return 0;
    if (part >= nparts) return nblk;
    return (nblk * h->split_eighths + 7) / 8;
  };
  if (split && h->other_work) {
    HIPCHK(hipEventRecord(h->fork, h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->fork, 0));
  }
  h->other_work = false;
  for (int part = 0; part < nparts; ++part) {
    const long long b0 = bound(part), nb = bound(part + 1) - b0;
    if (nb <= 0) continue;
    hipStream_t st = stream_of(part);
    const int s = h->slot;
    h->slot = (h->slot + 1) % kRing;
    HIPCHK(hipEventSynchronize(h->slot_done[s]));
    DevParams &p = h->h_params[s];
    p.cfg = h->cfg;
    p.lay = h->lay; p.scal = h->scal; p.n_active = h->n_active; p.status = h->status; p.err_layer = h->err_layer;
    p.err_step = h->err_step; p.work = h->work;
    p.spec = h->spec; p.flags = h->flags;
    p.f_sw = h->f_sw; p.f_lw = h->f_lw; p.f_T2m = h->f_T2m; p.f_precip = h->f_precip; p.flen = h->flen;
    p.nsites = h->nsites; p.site = h->site;
    p.ocean_dflq = h->ocean_dflq; p.ocean_sbu = h->ocean_sbu;
    p.ncol = h->ncol;
    p.block0 = b0;
    p.time0 = h->clk.time; p.step0 = h->clk.step; p.n_time_out0 = h->clk.n_time_out; p.time_counter0 = h->clk.time_counter;
    p.nsteps = nsteps;
    p.out_lay = h->out_lay; p.out_scal = h->out_scal; p.out_n_active = h->out_n_active;
    p.out_col0 = h->out_col0; p.out_ncols = h->out_ncols;
    p.p17 = h->p17; p.p14 = h->p14; p.tf_c3 = h->tf_c3;
    p.bgc = h->bgc; p.bgc_bot = h->bgc_bot; p.bfl = h->bfl; p.out_bgc = h->out_bgc; p.out_bgc_bot = h->out_bgc_bot;
    p.n_bgc = h->n_bgc; p.bgc_total0 = h->bgc_total0;
    HIPCHK(hipMemcpyAsync(&h->d_params[s], &p, sizeof(DevParams), hipMemcpyHostToDevice, st));
    HIPCHK(samsim_launch_step(&h->d_params[s], &p, nb, st));
    HIPCHK(hipEventRecord(h->slot_done[s], st));
  }
  h->stepped = true;
  advance_clock(h, nsteps);
  return SAMSIM_OK;
}

// every entry point but the stepping ones waits for the second stream and makes the next launch's second part wait for whatever
// it leaves on the first
int use(samsim_handle *h, bool stepping = false) {
  if (!h) return SAMSIM_ERR_ARG;
  HIPCHK(hipSetDevice(h->device));
  if (!stepping) {
    if (h->stream2) HIPCHK(hipStreamSynchronize(h->stream2));
    h->other_work = true;
  }
  return SAMSIM_OK;
}

}  // namespace

extern "C" {

int samsim_abi_version(void) { return SAMSIM_ABI_VERSION; }

int samsim_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *samsim_strerror(int code) {
  switch (code) {
    case SAMSIM_OK: return "ok";
    case SAMSIM_ERR_ARG: return "bad argument";
    case SAMSIM_ERR_UNSUPPORTED: return "flag value not supported by the HIP path";
    case SAMSIM_ERR_HIP: return g_last_hip_error[0] ? g_last_hip_error : "HIP error";
    case SAMSIM_ERR_NO_DEVICE: return "no HIP device";
    case SAMSIM_ERR_NO_OUTPUT: return "no output snapshot has been taken";
    case SAMSIM_ERR_ABI: return "samsim_config.struct_size does not match this library";
    case SAMSIM_ERR_NOMEM: return "out of memory";
  }
  return "unknown error";
}

int samsim_create(const samsim_config *cfg, int64_t ncol, int32_t device, samsim_handle **out) {
  if (!cfg || !out || ncol <= 0) return SAMSIM_ERR_ARG;
  int rc = validate(*cfg);
  if (rc) return rc;
  // The kernel reaches the [slot][ncol] blocks (per-column scalars, hand-over block of the up sweep) with a scalar base and a 32-bit
  // byte offset slot * ncol * 8 + col * 8 (GSI / SPEC in samsim_kernels.hip): the widest of them must stay below 4 GiB.  The layer
  // block has no such bound: it is stored per 64-column block, whose base is 64-bit arithmetic and whose rows (nlayer * 8 KiB) lie
  // within a 32-bit offset for any nlayer samsim_config admits.
  constexpr unsigned long long kRows = (int)SAMSIM_NSCAL > (int)DEV_NSPEC ? (int)SAMSIM_NSCAL : (int)DEV_NSPEC;
  if (kRows * (unsigned long long)ncol * 8ull >= (1ull << 32)) return SAMSIM_ERR_ARG;
  if (device < 0 || device >= samsim_device_count()) return SAMSIM_ERR_NO_DEVICE;
  HIPCHK(hipSetDevice(device));
  samsim_handle *h = new (std::nothrow) samsim_handle();
  if (!h) return SAMSIM_ERR_NOMEM;
  h->cfg = *cfg; h->ncol = ncol; h->device = device;
  h->clk = samsim_clock{0.0, 0, 0, 1, 0};
  h->p17 = std::pow(10.0, -17.0);
  h->p14 = std::pow(10.0, -14.0);
  h->tf_c3 = (double)(5.33f * std::pow(10.0f, -7.0f));
  const size_t N = (size_t)cfg->nlayer, nc = (size_t)ncol;
  bool ok = hip_ok(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate");
  ok = ok && hip_ok(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking), "hipStreamCreate");
  ok = ok && hip_ok(hipEventCreateWithFlags(&h->fork, hipEventDisableTiming), "hipEventCreate");
  ok = ok && hip_ok(hipEventCreate(&h->ev0b), "hipEventCreate") && hip_ok(hipEventCreate(&h->ev1b), "hipEventCreate");
  ok = ok && hip_ok(dalloc(&h->lay, DEV_LAY_DOUBLES(N, nc)), "hipMalloc lay");
  ok = ok && hip_ok(dalloc(&h->scal, (size_t)SAMSIM_NSCAL * nc), "hipMalloc scal");
  ok = ok && hip_ok(dalloc(&h->n_active, nc), "hipMalloc n_active");
  ok = ok && hip_ok(dalloc(&h->status, nc), "hipMalloc status");
  ok = ok && hip_ok(dalloc(&h->err_layer, nc), "hipMalloc err_layer");
  ok = ok && hip_ok(dalloc(&h->err_step, nc), "hipMalloc err_step");
  ok = ok && hip_ok(dalloc(&h->work, nc), "hipMalloc work");
  ok = ok && hip_ok(dalloc(&h->spec, (size_t)DEV_NSPEC * nc), "hipMalloc spec");
  ok = ok && hip_ok(dalloc(&h->flags, nc), "hipMalloc flags");
  ok = ok && hip_ok(dalloc(&h->d_params, (size_t)kRing), "hipMalloc params");
  ok = ok && hip_ok(hipHostMalloc((void **)&h->h_params, sizeof(DevParams) * kRing, hipHostMallocDefault), "hipHostMalloc params");
  for (int i = 0; ok && i < kRing; ++i) ok = hip_ok(hipEventCreateWithFlags(&h->slot_done[i], hipEventDisableTiming), "hipEventCreate");
  ok = ok && hip_ok(hipEventCreate(&h->ev0), "hipEventCreate") && hip_ok(hipEventCreate(&h->ev1), "hipEventCreate");
  if (ok) {
    // sub_allocate zeros (mo_init.f90:2081-2087) and the defaults of mo_init.f90:1982-1990
    ok = hip_ok(hipMemsetAsync(h->lay, 0, sizeof(double) * DEV_LAY_DOUBLES(N, nc), h->stream), "memset lay");
    ok = ok && hip_ok(hipMemsetAsync(h->scal, 0, sizeof(double) * SAMSIM_NSCAL * nc, h->stream), "memset scal");
    ok = ok && hip_ok(hipMemsetAsync(h->status, 0, sizeof(int32_t) * nc, h->stream), "memset");
    ok = ok && hip_ok(hipMemsetAsync(h->err_layer, 0, sizeof(int32_t) * nc, h->stream), "memset");
    ok = ok && hip_ok(hipMemsetAsync(h->err_step, 0, sizeof(long long) * nc, h->stream), "memset");
    ok = ok && hip_ok(hipMemsetAsync(h->work, 0, sizeof(long long) * nc, h->stream), "memset");
    ok = ok && hip_ok(hipMemsetAsync(h->spec, 0, sizeof(double) * DEV_NSPEC * nc, h->stream), "memset");
    if (ok) {
      const unsigned fg = (unsigned)((N * nc + 255) / 256);
      hipLaunchKernelGGL(lay_fill_array, dim3(fg), dim3(256), 0, h->stream, h->lay, (int)SAMSIM_A_T, (int)N, nc, cfg->T_bottom);
      hipLaunchKernelGGL(lay_fill_array, dim3(fg), dim3(256), 0, h->stream, h->lay, (int)SAMSIM_A_S_BU, (int)N, nc, cfg->S_bu_bottom);
      hipLaunchKernelGGL(lay_fill_array, dim3(fg), dim3(256), 0, h->stream, h->lay, (int)SAMSIM_A_PSI_L, (int)N, nc, 1.0);
      ok = hip_ok(hipGetLastError(), "fill layer arrays");
    }
    ok = ok && hip_ok(fill(h->scal + (size_t)SAMSIM_S_PRECIP_SCALE * nc, nc, 1.0, h->stream), "fill precip_scale");
    ok = ok && hip_ok(fill(h->scal + (size_t)SAMSIM_S_S_BU_BOTTOM * nc, nc, cfg->S_bu_bottom, h->stream), "fill S_bu_bottom");
    if (ok) {
      hipLaunchKernelGGL(fill_i32, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream, h->n_active, nc, 1);
      hipLaunchKernelGGL(fill_i32, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream, h->flags, nc,
                         COLF_DIRTY | COLF_RESTART);
      ok = hip_ok(hipGetLastError(), "fill n_active/flags");
    }
    ok = ok && hip_ok(hipStreamSynchronize(h->stream), "sync");
  }
  if (!ok) { samsim_destroy(h); return SAMSIM_ERR_HIP; }
  rc = samsim_set_output_window(h, 0, 1);
  if (rc) { samsim_destroy(h); return rc; }
  *out = h;
  return SAMSIM_OK;
}

void samsim_destroy(samsim_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)hipFree(h->lay); (void)hipFree(h->scal); (void)hipFree(h->n_active); (void)hipFree(h->status);
  (void)hipFree(h->err_layer); (void)hipFree(h->err_step); (void)hipFree(h->work);
  (void)hipFree(h->spec); (void)hipFree(h->flags); (void)hipFree(h->d_stat); (void)hipFree(h->stage);
  (void)hipFree(h->bgc); (void)hipFree(h->bgc_bot); (void)hipFree(h->bfl); (void)hipFree(h->out_bgc); (void)hipFree(h->out_bgc_bot);
  (void)hipFree(h->f_sw); (void)hipFree(h->f_lw); (void)hipFree(h->f_T2m); (void)hipFree(h->f_precip); (void)hipFree(h->site);
  (void)hipFree(h->ocean_dflq); (void)hipFree(h->ocean_sbu);
  (void)hipFree(h->out_lay); (void)hipFree(h->out_scal); (void)hipFree(h->out_n_active);
  (void)hipFree(h->d_params);
  if (h->h_params) (void)hipHostFree(h->h_params);
  for (int i = 0; i < kRing; ++i) if (h->slot_done[i]) (void)hipEventDestroy(h->slot_done[i]);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->fork) (void)hipEventDestroy(h->fork);
  if (h->ev0b) (void)hipEventDestroy(h->ev0b);
  if (h->ev1b) (void)hipEventDestroy(h->ev1b);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int samsim_set_forcing(samsim_handle *h, int32_t len, const double *fl_sw, const double *fl_lw, const double *T2m,
                       const double *precip, const double *dT2m_col, const double *precip_scale_col) {
  return samsim_set_forcing_sites(h, 1, len, fl_sw, fl_lw, T2m, precip, nullptr, dT2m_col, precip_scale_col);
}

int samsim_set_forcing_sites(samsim_handle *h, int32_t nsites, int32_t len, const double *fl_sw, const double *fl_lw,
                             const double *T2m, const double *precip, const int32_t *site_of_column,
                             const double *dT2m_col, const double *precip_scale_col) {
  int rc = use(h);
  if (rc) return rc;
  if (len < 2 || nsites < 1 || !fl_sw || !fl_lw || !T2m || !precip || (nsites > 1 && !site_of_column)) return SAMSIM_ERR_ARG;
  for (long long c = 0; nsites > 1 && c < h->ncol; ++c)
    if (site_of_column[c] < 0 || site_of_column[c] >= nsites) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(h->f_sw); (void)hipFree(h->f_lw); (void)hipFree(h->f_T2m); (void)hipFree(h->f_precip); (void)hipFree(h->site);
  h->f_sw = h->f_lw = h->f_T2m = h->f_precip = nullptr; h->site = nullptr;
  const size_t bytes = sizeof(double) * (size_t)len * (size_t)nsites;
  HIPCHK(dalloc(&h->f_sw, (size_t)len * nsites)); HIPCHK(dalloc(&h->f_lw, (size_t)len * nsites));
  HIPCHK(dalloc(&h->f_T2m, (size_t)len * nsites)); HIPCHK(dalloc(&h->f_precip, (size_t)len * nsites));
  h->nsites = nsites;
  if (nsites > 1) {
    HIPCHK(dalloc(&h->site, (size_t)h->ncol));
    HIPCHK(hipMemcpy(h->site, site_of_column, sizeof(int32_t) * (size_t)h->ncol, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpy(h->f_sw, fl_sw, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->f_lw, fl_lw, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->f_T2m, T2m, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->f_precip, precip, bytes, hipMemcpyHostToDevice));
  h->flen = len;
  const size_t nc = (size_t)h->ncol;
  if (dT2m_col) HIPCHK(hipMemcpy(h->scal + (size_t)SAMSIM_S_DT2M * nc, dT2m_col, sizeof(double) * nc, hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(h->scal + (size_t)SAMSIM_S_DT2M * nc, 0, sizeof(double) * nc));
  if (precip_scale_col) {
    HIPCHK(hipMemcpy(h->scal + (size_t)SAMSIM_S_PRECIP_SCALE * nc, precip_scale_col, sizeof(double) * nc, hipMemcpyHostToDevice));
  } else {
    HIPCHK(fill(h->scal + (size_t)SAMSIM_S_PRECIP_SCALE * nc, nc, 1.0, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return SAMSIM_OK;
}

int samsim_set_ocean(samsim_handle *h, const double *dfl_q_bottom_col, const double *S_bu_bottom_col) {
  int rc = use(h);
  if (rc) return rc;
  // the offset rides on the flux sub_test4 sets every step; the salinity replaces cfg.S_bu_bottom, which the tank budget owns with tank_flag 2
  if (dfl_q_bottom_col && h->cfg.testcase != 4 && h->cfg.testcase != 7) return SAMSIM_ERR_UNSUPPORTED;
  if (S_bu_bottom_col && h->cfg.tank_flag == 2) return SAMSIM_ERR_UNSUPPORTED;
  const size_t nc = (size_t)h->ncol;
  for (size_t i = 0; S_bu_bottom_col && i < nc; ++i) if (!(S_bu_bottom_col[i] >= 0.0)) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  // the new arrays are complete on the device before the handle sees them: a call that fails leaves the handle as it was
  double *dq = nullptr, *sb = nullptr;
  bool ok = true;
  if (dfl_q_bottom_col)
    ok = hip_ok(dalloc(&dq, nc), "hipMalloc ocean_dflq") &&
         hip_ok(hipMemcpy(dq, dfl_q_bottom_col, sizeof(double) * nc, hipMemcpyHostToDevice), "hipMemcpy ocean_dflq");
  if (ok && S_bu_bottom_col)
    ok = hip_ok(dalloc(&sb, nc), "hipMalloc ocean_sbu") &&
         hip_ok(hipMemcpy(sb, S_bu_bottom_col, sizeof(double) * nc, hipMemcpyHostToDevice), "hipMemcpy ocean_sbu");
  if (!ok) { (void)hipFree(dq); (void)hipFree(sb); return SAMSIM_ERR_HIP; }
  (void)hipFree(h->ocean_dflq); (void)hipFree(h->ocean_sbu);
  h->ocean_dflq = dq; h->ocean_sbu = sb;
  return SAMSIM_OK;
}

static int check_soa(samsim_handle *h, const samsim_state_soa *s, int64_t col0) {
  if (!s || !s->lay || !s->scal || !s->n_active) return SAMSIM_ERR_ARG;
  if (s->nlayer != h->cfg.nlayer || s->ncol <= 0 || col0 < 0 || col0 + s->ncol > h->ncol) return SAMSIM_ERR_ARG;
  if (s->narr != SAMSIM_NPROG && s->narr != SAMSIM_NARR) return SAMSIM_ERR_ARG;
  return SAMSIM_OK;
}

int samsim_set_state(samsim_handle *h, const samsim_state_soa *s, int64_t col0) {
  int rc = use(h);
  if (rc) return rc;
  rc = check_soa(h, s, col0);
  if (rc) return rc;
  const size_t N = (size_t)s->nlayer, nc = (size_t)h->ncol, w = (size_t)s->ncol;
  for (size_t i = 0; i < w; ++i) if (s->n_active[i] < 1 || s->n_active[i] > (int)N) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  {
    const size_t n = (size_t)s->narr * N * w;
    HIPCHK(stage_for(h, n));
    HIPCHK(hipMemcpyAsync(h->stage, s->lay, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lay_window<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->lay, h->stage, (int)s->narr, (int)N, nc,
                       (size_t)col0, w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  // the perturbation slots (>= SAMSIM_S_DT2M) belong to the forcing: set_state leaves them alone
  HIPCHK(hipMemcpy2D(h->scal + col0, nc * sizeof(double), s->scal, w * sizeof(double), w * sizeof(double), (size_t)SAMSIM_S_DT2M,
                     hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->n_active + col0, s->n_active, w * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(h->status + col0, 0, w * sizeof(int32_t)));
  // the next step of these columns runs the full first sweep and treats RAY as the previous step's Rayleigh numbers
  hipLaunchKernelGGL(fill_i32, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, h->stream, h->flags + col0, w,
                     COLF_DIRTY | COLF_RESTART);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return SAMSIM_OK;
}

int samsim_get_state(samsim_handle *h, samsim_state_soa *s, int64_t col0) {
  int rc = use(h);
  if (rc) return rc;
  rc = check_soa(h, s, col0);
  if (rc) return rc;
  const size_t N = (size_t)s->nlayer, nc = (size_t)h->ncol, w = (size_t)s->ncol;
  if (s->narr == SAMSIM_NARR && h->stepped) {
    hipLaunchKernelGGL(refresh_s_bu, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, h->stream, h->lay, h->n_active, h->status,
                       nc, (int)N, (size_t)col0, w);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  {
    const size_t n = (size_t)s->narr * N * w;
    HIPCHK(stage_for(h, n));
    hipLaunchKernelGGL(lay_window<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->lay, h->stage, (int)s->narr, (int)N, nc,
                       (size_t)col0, w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(s->lay, h->stage, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  HIPCHK(hipMemcpy2D(s->scal, w * sizeof(double), h->scal + col0, nc * sizeof(double), w * sizeof(double), (size_t)SAMSIM_NSCAL,
                     hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(s->n_active, h->n_active + col0, w * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SAMSIM_OK;
}

int samsim_set_clock(samsim_handle *h, const samsim_clock *c) {
  if (!h || !c || c->time_counter < 1 || c->step < 0) return SAMSIM_ERR_ARG;
  h->clk = *c;
  return SAMSIM_OK;
}

int samsim_get_clock(samsim_handle *h, samsim_clock *c) {
  if (!h || !c) return SAMSIM_ERR_ARG;
  *c = h->clk;
  return SAMSIM_OK;
}

int samsim_step(samsim_handle *h, int64_t nsteps) {
  int rc = use(h, true);
  if (rc) return rc;
  if (nsteps < 0) return SAMSIM_ERR_ARG;
  return launch(h, nsteps);
}

int samsim_steps_timed(samsim_handle *h, int64_t nsteps, int32_t nlaunches, double *device_ms) {
  int rc = use(h);   // (waits for the second stream: the timed region starts with both streams idle or ordered behind ev0)
  if (rc) return rc;
  if (nsteps < 0 || nlaunches < 1 || !device_ms) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int32_t i = 0; i < nlaunches; ++i) {
    rc = launch(h, nsteps);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventRecord(h->ev1b, h->stream2));
  HIPCHK(hipEventSynchronize(h->ev1));
  HIPCHK(hipEventSynchronize(h->ev1b));
  float ms = 0.f, msb = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  HIPCHK(hipEventElapsedTime(&msb, h->ev0, h->ev1b));   // (the other streams' first launches are ordered behind ev0 by the fork event)
  if (msb > ms) ms = msb;
  *device_ms = (double)ms;
  return SAMSIM_OK;
}

int samsim_step_timed(samsim_handle *h, int64_t nsteps, double *kernel_ms) { return samsim_steps_timed(h, nsteps, 1, kernel_ms); }

int samsim_get_device(samsim_handle *h, int32_t *device, char *pci_bus_id, int32_t len) {
  if (!h) return SAMSIM_ERR_ARG;
  if (device) *device = h->device;
  if (pci_bus_id && len > 0) HIPCHK(hipDeviceGetPCIBusId(pci_bus_id, len, h->device));
  return SAMSIM_OK;
}

int samsim_set_launch_split(samsim_handle *h, int64_t min_blocks, int32_t first_part_eighths) {
  int rc = use(h);
  if (rc) return rc;
  if (min_blocks < 0 || first_part_eighths < 1 || first_part_eighths > 7) return SAMSIM_ERR_ARG;
  h->split_blocks = min_blocks;
  h->split_eighths = first_part_eighths;
  return SAMSIM_OK;
}

int samsim_synchronize(samsim_handle *h) {
  int rc = use(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return SAMSIM_OK;
}

int64_t samsim_steps_to_output(samsim_handle *h) {
  if (!h) return 0;
  if (h->clk.step == 0) return 1;
  return (int64_t)(h->cfg.i_time_out - h->clk.n_time_out) + 1;
}

int samsim_set_output_window(samsim_handle *h, int64_t col0, int64_t ncols) {
  int rc = use(h);
  if (rc) return rc;
  if (col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  (void)hipFree(h->out_lay); (void)hipFree(h->out_scal); (void)hipFree(h->out_n_active);
  h->out_lay = h->out_scal = nullptr; h->out_n_active = nullptr;
  h->out_col0 = col0; h->out_ncols = ncols; h->snap_valid = false;
  if (ncols > 0) {
    const size_t N = (size_t)h->cfg.nlayer, w = (size_t)ncols;
    HIPCHK(dalloc(&h->out_lay, (size_t)SAMSIM_NARR * N * w));
    HIPCHK(dalloc(&h->out_scal, (size_t)SAMSIM_NSCAL * w));
    HIPCHK(dalloc(&h->out_n_active, w));
    HIPCHK(hipMemset(h->out_lay, 0, sizeof(double) * SAMSIM_NARR * N * w));
    HIPCHK(hipMemset(h->out_scal, 0, sizeof(double) * SAMSIM_NSCAL * w));
    HIPCHK(hipMemset(h->out_n_active, 0, sizeof(int32_t) * w));
  }
  if (h->n_bgc > 0) {  // tracer snapshot follows the window
    (void)hipFree(h->out_bgc); (void)hipFree(h->out_bgc_bot);
    h->out_bgc = h->out_bgc_bot = nullptr;
    const size_t N = (size_t)h->cfg.nlayer, w = (size_t)(ncols > 0 ? ncols : 1);
    HIPCHK(dalloc(&h->out_bgc, (size_t)h->n_bgc * N * w));
    HIPCHK(dalloc(&h->out_bgc_bot, (size_t)h->n_bgc * w));
    HIPCHK(hipMemset(h->out_bgc, 0, sizeof(double) * (size_t)h->n_bgc * N * w));
    HIPCHK(hipMemset(h->out_bgc_bot, 0, sizeof(double) * (size_t)h->n_bgc * w));
  }
  return SAMSIM_OK;
}

int samsim_get_output(samsim_handle *h, samsim_output_soa *o) {
  int rc = use(h);
  if (rc) return rc;
  if (!o || !o->lay || !o->scal || !o->n_active || o->ncols != h->out_ncols || o->nlayer != h->cfg.nlayer) return SAMSIM_ERR_ARG;
  if (!h->snap_valid || h->out_ncols == 0) return SAMSIM_ERR_NO_OUTPUT;
  const size_t N = (size_t)o->nlayer, w = (size_t)o->ncols;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(o->lay, h->out_lay, sizeof(double) * SAMSIM_NARR * N * w, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o->scal, h->out_scal, sizeof(double) * SAMSIM_NSCAL * w, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(o->n_active, h->out_n_active, sizeof(int32_t) * w, hipMemcpyDeviceToHost));
  o->time = h->snap_time;
  o->step = h->snap_step;
  return SAMSIM_OK;
}

int samsim_get_status(samsim_handle *h, int32_t *status, int64_t *step, int32_t *layer) {
  int rc = use(h);
  if (rc) return rc;
  const size_t nc = (size_t)h->ncol;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (status) HIPCHK(hipMemcpy(status, h->status, sizeof(int32_t) * nc, hipMemcpyDeviceToHost));
  if (step) HIPCHK(hipMemcpy(step, h->err_step, sizeof(long long) * nc, hipMemcpyDeviceToHost));
  if (layer) HIPCHK(hipMemcpy(layer, h->err_layer, sizeof(int32_t) * nc, hipMemcpyDeviceToHost));
  return SAMSIM_OK;
}

int samsim_set_status(samsim_handle *h, const int32_t *status, const int64_t *step, const int32_t *layer, int64_t col0, int64_t ncols) {
  int rc = use(h);
  if (rc) return rc;
  if (!status || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t w = (size_t)ncols;
  HIPCHK(hipMemcpy(h->status + col0, status, sizeof(int32_t) * w, hipMemcpyHostToDevice));
  if (step) HIPCHK(hipMemcpy(h->err_step + col0, step, sizeof(long long) * w, hipMemcpyHostToDevice));
  if (layer) HIPCHK(hipMemcpy(h->err_layer + col0, layer, sizeof(int32_t) * w, hipMemcpyHostToDevice));
  return SAMSIM_OK;
}

int samsim_get_work(samsim_handle *h, int64_t *layer_cell_updates, int64_t *column_steps) {
  int rc = use(h);
  if (rc) return rc;
  const size_t nc = (size_t)h->ncol;
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<long long> w;
  try { w.resize(nc); } catch (const std::bad_alloc &) { return SAMSIM_ERR_NOMEM; }
  HIPCHK(hipMemcpy(w.data(), h->work, sizeof(long long) * nc, hipMemcpyDeviceToHost));
  long long tot = 0;
  for (size_t i = 0; i < nc; ++i) tot += w[i];
  if (layer_cell_updates) *layer_cell_updates = tot;
  if (column_steps) *column_steps = (int64_t)h->clk.step * (int64_t)h->ncol;
  return SAMSIM_OK;
}

// ---- passive tracers
int samsim_set_tracers(samsim_handle *h, int32_t n_bgc, const double *bgc_bottom, const double *bgc_total) {
  int rc = use(h);
  if (rc) return rc;
  if (h->cfg.bgc_flag != 2 || n_bgc < 1 || n_bgc > SAMSIM_MAX_NBGC || !bgc_bottom) return SAMSIM_ERR_ARG;
  if (h->cfg.tank_flag == 2 && !bgc_total) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t N = (size_t)h->cfg.nlayer, nc = (size_t)h->ncol, w = (size_t)(h->out_ncols > 0 ? h->out_ncols : 1);
  if (n_bgc != h->n_bgc) {
    (void)hipFree(h->bgc); (void)hipFree(h->bgc_bot); (void)hipFree(h->bfl); (void)hipFree(h->out_bgc); (void)hipFree(h->out_bgc_bot);
    h->bgc = h->bgc_bot = h->bfl = h->out_bgc = h->out_bgc_bot = nullptr;
    HIPCHK(dalloc(&h->bgc, (size_t)n_bgc * N * nc));
    HIPCHK(dalloc(&h->bgc_bot, (size_t)n_bgc * nc));
    HIPCHK(dalloc(&h->bfl, (size_t)BFL_NROW * N * nc));
    HIPCHK(dalloc(&h->out_bgc, (size_t)n_bgc * N * w));
    HIPCHK(dalloc(&h->out_bgc_bot, (size_t)n_bgc * w));
    HIPCHK(hipMemsetAsync(h->bgc, 0, sizeof(double) * (size_t)n_bgc * N * nc, h->stream));
    HIPCHK(hipMemsetAsync(h->out_bgc, 0, sizeof(double) * (size_t)n_bgc * N * w, h->stream));
    HIPCHK(hipMemsetAsync(h->out_bgc_bot, 0, sizeof(double) * (size_t)n_bgc * w, h->stream));
    h->n_bgc = n_bgc;
  }
  HIPCHK(hipMemsetAsync(h->bfl, 0, sizeof(double) * (size_t)BFL_NROW * N * nc, h->stream));
  for (int t = 0; t < n_bgc; ++t) HIPCHK(fill(h->bgc_bot + (size_t)t * nc, nc, bgc_bottom[t], h->stream));
  h->bgc_total0 = bgc_total ? bgc_total[0] : 0.0;
  HIPCHK(hipStreamSynchronize(h->stream));
  return SAMSIM_OK;
}

int samsim_set_tracer_state(samsim_handle *h, const double *bgc_abs, int64_t col0, int64_t ncols) {
  int rc = use(h);
  if (rc) return rc;
  if (!bgc_abs || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t N = (size_t)h->cfg.nlayer, nc = (size_t)h->ncol, w = (size_t)ncols;
  HIPCHK(hipMemcpy2D(h->bgc + col0, nc * sizeof(double), bgc_abs, w * sizeof(double), w * sizeof(double), (size_t)h->n_bgc * N,
                     hipMemcpyHostToDevice));
  return SAMSIM_OK;
}

int samsim_set_tracer_bottom(samsim_handle *h, const double *bgc_bottom, int64_t col0, int64_t ncols) {
  int rc = use(h);
  if (rc) return rc;
  if (!bgc_bottom || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t nc = (size_t)h->ncol, w = (size_t)ncols;
  HIPCHK(hipMemcpy2D(h->bgc_bot + col0, nc * sizeof(double), bgc_bottom, w * sizeof(double), w * sizeof(double), (size_t)h->n_bgc,
                     hipMemcpyHostToDevice));
  return SAMSIM_OK;
}

int samsim_get_tracer_state(samsim_handle *h, double *bgc_abs, double *bgc_bottom, int64_t col0, int64_t ncols) {
  int rc = use(h);
  if (rc) return rc;
  if (!bgc_abs || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t N = (size_t)h->cfg.nlayer, nc = (size_t)h->ncol, w = (size_t)ncols;
  HIPCHK(hipMemcpy2D(bgc_abs, w * sizeof(double), h->bgc + col0, nc * sizeof(double), w * sizeof(double), (size_t)h->n_bgc * N,
                     hipMemcpyDeviceToHost));
  if (bgc_bottom)
    HIPCHK(hipMemcpy2D(bgc_bottom, w * sizeof(double), h->bgc_bot + col0, nc * sizeof(double), w * sizeof(double), (size_t)h->n_bgc,
                       hipMemcpyDeviceToHost));
  return SAMSIM_OK;
}

int samsim_get_tracer_output(samsim_handle *h, double *bgc_abs, double *bgc_bottom) {
  int rc = use(h);
  if (rc) return rc;
  if (!bgc_abs || h->n_bgc < 1) return SAMSIM_ERR_ARG;
  if (!h->snap_valid) return SAMSIM_ERR_NO_OUTPUT;
  HIPCHK(hipStreamSynchronize(h->stream));
  const size_t N = (size_t)h->cfg.nlayer, w = (size_t)h->out_ncols;
  HIPCHK(hipMemcpy(bgc_abs, h->out_bgc, sizeof(double) * (size_t)h->n_bgc * N * w, hipMemcpyDeviceToHost));
  if (bgc_bottom) HIPCHK(hipMemcpy(bgc_bottom, h->out_bgc_bot, sizeof(double) * (size_t)h->n_bgc * w, hipMemcpyDeviceToHost));
  return SAMSIM_OK;
}

int samsim_get_ensemble_stats(samsim_handle *h, int32_t nslots, const int32_t *slots, samsim_stat *out) {
  int rc = use(h);
  if (rc) return rc;
  if (nslots < 0 || (nslots > 0 && (!slots || !out))) return SAMSIM_ERR_ARG;
  for (int i = 0; i < nslots; ++i)
    if (slots[i] != SAMSIM_STAT_N_ACTIVE && (slots[i] < 0 || slots[i] >= SAMSIM_NSCAL)) return SAMSIM_ERR_ARG;
  if (!h->d_stat) HIPCHK(hipMalloc(&h->d_stat, sizeof(StatPartial) * kStatGrid));
  StatPartial *d_part = (StatPartial *)h->d_stat;
  std::vector<StatPartial> part(kStatGrid);
  const long long nc = h->ncol;
  for (int i = 0; i < nslots; ++i) {
    const double *row = (slots[i] == SAMSIM_STAT_N_ACTIVE) ? nullptr : h->scal + (size_t)slots[i] * (size_t)nc;
    hipLaunchKernelGGL(stat_kernel<false>, dim3(kStatGrid), dim3(kStatBlock), 0, h->stream, row, h->n_active, h->status, nc, 0.0,
                       d_part);
    HIPCHK(hipMemcpyAsync(part.data(), d_part, sizeof(StatPartial) * kStatGrid, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    samsim_stat st{0, 0.0, 1.0e300, -1.0e300, 0.0};
    double sum = 0.0;
    for (const StatPartial &q : part) {
      st.count += q.n; sum += q.sum;
      st.min = q.mn < st.min ? q.mn : st.min;
      st.max = q.mx > st.max ? q.mx : st.max;
    }
    if (st.count > 0) {
      st.mean = sum / (double)st.count;
      hipLaunchKernelGGL(stat_kernel<true>, dim3(kStatGrid), dim3(kStatBlock), 0, h->stream, row, h->n_active, h->status, nc,
                         st.mean, d_part);
      HIPCHK(hipMemcpyAsync(part.data(), d_part, sizeof(StatPartial) * kStatGrid, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      double ssq = 0.0;
      for (const StatPartial &q : part) ssq += q.ssq;
      st.std = sqrt(ssq / (double)st.count);
    } else {
      st.min = st.max = 0.0;
    }
    out[i] = st;
  }
  return SAMSIM_OK;
}

}  // extern "C"
