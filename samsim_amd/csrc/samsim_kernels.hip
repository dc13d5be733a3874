// samsim_kernels.hip -- hand-written CDNA4 (gfx950) kernel for the per-timestep 1-D sea-ice column update.
//
// What it computes: the body of the reference time loop, pgriewank/SAMSIM mo_grotz.f90:182-835, for `ncol`
// independent columns, `nsteps` steps per launch.  One lane owns one column for the whole launch (columns
// never communicate), state lives in HBM per 64-column block as [block][layer][array][lane] float64 so that every
// per-layer access of a wave is one contiguous 512-byte line and a layer row's arrays sit within one row address's
// immediate range, and a column's layers are walked sequentially inside the lane with the neighbour values
// (k-1, k, k+1) carried in registers.  No MFMA (there is no contraction in this path); LDS holds the per-column
// scalars that must survive the two layer loops of a step (19 slots per lane, own words only: no barrier); no
// cross-lane traffic except wave-uniform votes.
//
// The sequential structure inside a column is dictated by the reference: getT's Newton iteration is seeded
// with the temperature of the layer below (mo_grotz.f90:298-303) and stops at |f| <= 1 J/kg, so the result
// depends on the guess (hard part 1) and the bottom->top chain has to be reproduced.
//
// Sweeps per step (direction, what is fused; reference lines in the functions below):
//   S1  up    S_bu,H -> getT chain -> S_br -> Expulsion; permeability + suffix scans -> Rayleigh number
//   P2  down  expulsion_flux recurrence + mass_transfer + S_bu refresh
//   P3  down  gravity-drainage fluxes + return-flow mass_transfer + Beer-law transmittance
//   P4  up    conductive stencil + explicit enthalpy update + second getT chain
//   rare: freeboard (2 down), flush3 (1 up + 1 down), flood, snow, layer_dynamics (regrid)
// The reference's O(N^2) loops (harmonic-mean permeability, freeboard search) are O(N) scans here; sums are
// therefore associated differently (1e-16 relative), everything else follows the reference's operation order.
#include <hip/hip_runtime.h>
#include <math.h>

#include <type_traits>

#include "samsim_device.h"

// SAMSIM_STAMPS (profiling builds only, never the product library): 1 = s_memtime stamps around the regions of a time step,
// summed per wave in LDS and added to g_stamps at the end of the launch; 2 = event counters (Newton evaluations, loop trips).
// tools/stamps.py reads g_stamps through samsim_debug_stamps.
#ifndef SAMSIM_STAMPS
#define SAMSIM_STAMPS 0
#endif
// SAMSIM_ISA_MARKS: comment lines in the assembly listing at the boundaries of the hot loops (tools/isa_loops.py --marks)
#ifdef SAMSIM_ISA_MARKS
#define ISA_MARK(name) asm volatile("; ISA_MARK " name)
#else
#define ISA_MARK(name) ((void)0)
#endif
#if SAMSIM_STAMPS
__device__ unsigned long long g_stamps[48];
extern "C" int samsim_debug_stamps(unsigned long long *out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
  if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

namespace {

#if SAMSIM_STAMPS
enum { ST_PRO = 0, ST_DFUSED, ST_DUNFUSED, ST_SURF, ST_UP, ST_POST, ST_HEAD, ST_TAIL,
       CT_WAVESTEPS = 8, CT_FUSED, CT_UNFUSED, CT_UP_TRIPS, CT_NEWTON_WAVE, CT_NEWTON_LANE, CT_LANES, CT_DOWN_TRIPS, CT_DRAIN_WAVE,
       CT_DRAIN_LANE, CT_DIRTY, CT_L_COUPLING, ST_U_HEAD = 20, ST_U_GETT, ST_U_TAIL, ST_D_A, ST_D_B,
       CT_L_FLOODP = 25, CT_L_IRREG, CT_L_DIRTY, CT_L_UNFUSED, CT_L_FLUSH3, CT_L_REGRID, CT_L_FREEBOARD,
       CT_REFILL = 32, CT_ROWS, CT_ROWS_STILL, CT_ODD_LANES, CT_ODD_WAVES, CT_ODD_EVALS_WAVE, CT_LITE, ST_NSLOT = 48 };
struct Stamps {
  unsigned long long *acc;   // [48] in LDS, one block = one wave
  unsigned long long t0;
};
__device__ __forceinline__ bool st_leader() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }
__device__ __forceinline__ void st_mark(Stamps &st, int region) {
#if SAMSIM_STAMPS == 1
  const unsigned long long t = __builtin_amdgcn_s_memtime();
  if (st_leader()) st.acc[region] += t - st.t0;
  st.t0 = t;
#endif
}
__device__ __forceinline__ void st_count(Stamps &st, int counter, unsigned long long n = 1) {
#if SAMSIM_STAMPS == 2
  if (st_leader()) st.acc[counter] += n;
#endif
}
#define ST_MARK(r) st_mark(x.st, r)
#define ST_COUNT(cn, n) st_count(x.st, cn, n)
#else
#define ST_MARK(r) ((void)0)
#define ST_COUNT(cn, n) ((void)0)
#endif

// ---------------------------------------------------------------- constants, mo_parameters.f90:38-112
// `pi` and `grav` are default REAL (float32) in the reference (mo_parameters.f90:38-39)
constexpr double pi_f = (double)3.1415f;
constexpr double grav_f = (double)9.8061f;
constexpr double k_s = 2.2, k_l = 0.523;
constexpr double c_s = 2020.0, c_s_beta = 7.6973, c_l = 3400.0;
constexpr double rho_s = 920.0, rho_l = 1028.0, latent_heat = 333500.0, zeroK = 273.15;
// `0.8_wp*1e-3`: float32 literal factor (mo_parameters.f90:56,57,59)
constexpr double bbeta = 0.8 * (double)1e-3f;
constexpr double mu = 2.55 * (double)1e-3f;
constexpr double kappa_l = k_l / rho_l / c_l;
constexpr double sigma = 5.6704 * (double)1e-8f;
constexpr double psi_s_min = 0.05, neg_free = -0.05;
constexpr double x_grav = 0.000584, ray_crit = 4.89;
constexpr double para_flush_horiz = 1.0, para_flush_gamma = 0.9;
constexpr double psi_s_top_min = 0.40, ratio_flood = 1.50, ref_salinity = 34.0;
constexpr double rho_snow = 330.0, gas_snow_ice2 = 0.20;
constexpr double emissivity_ice = 0.95, emissivity_snow = 1.00, penetr = 0.30, extinc = 2.00;
constexpr double Turb_A = 0.1 * 0.05 * rho_l / 86400.0;
constexpr double Turb_B = 0.05;

// Every device function that takes the column struct or the context by reference is force-inlined: if one of them stayed
// out of line the struct would escape, its fields would live in scratch memory, the data pointers in it would lose their
// address space and the (uniform) config reads would become vector loads.  Measured: out-of-line rare paths by reference
// 101 ms, by value (struct copied in and out) 209 ms, everything inline 90 ms per launch of the default bench.
#define RARE __forceinline__

// ---------------------------------------------------------------- kernel instantiations
// The step kernel is instantiated per flag set K.  KGeneric reads every flag of samsim_config at run time and contains
// all supported parametrisations.  A fixed set (KSheba = testcase 4 as shipped = BASELINE cfg3 / cfg5, KPlate = testcase 1 =
// cfg1 / cfg2) turns the flags into compile-time constants: the branches of the other parametrisations, their registers
// and the flag loads disappear from the hot sweeps.  samsim_launch_step picks the instantiation whose flags equal the
// handle's configuration, KGeneric otherwise; the code paths taken are the same either way.
#define SAMSIM_FLAG_LIST(X)                                                                                               \
  X(atmoflux_flag) X(grav_flag) X(prescribe_flag) X(grav_heat_flag) X(flush_heat_flag) X(turb_flag) X(salt_flag)         \
  X(boundflux_flag) X(flush_flag) X(flood_flag) X(bottom_flag) X(precip_flag) X(harmonic_flag) X(tank_flag) X(albedo_flag) \
  X(lab_snow_flag) X(freeboard_snow_flag) X(snow_flush_flag) X(snow_precip_flag) X(testcase)
struct KGeneric {
  static constexpr bool fixed = false, general = true, sites = true, bgc = true;
#define X(f) [[maybe_unused]] static constexpr int f = 0;
  SAMSIM_FLAG_LIST(X)
#undef X
};
struct KSheba {  // init(4), mo_init.f90:1127-1207 on the defaults of :83-109
  static constexpr bool fixed = true, general = false, sites = false, bgc = false;
  static constexpr int atmoflux_flag = 2, grav_flag = 2, prescribe_flag = 1, grav_heat_flag = 1, flush_heat_flag = 2, turb_flag = 2,
                       salt_flag = 1, boundflux_flag = 2, flush_flag = 5, flood_flag = 2, bottom_flag = 1, precip_flag = 1,
                       harmonic_flag = 2, tank_flag = 1, albedo_flag = 2, lab_snow_flag = 0, freeboard_snow_flag = 0,
                       snow_flush_flag = 1, snow_precip_flag = 1, testcase = 4;
};
struct KShebaSites : KSheba {  // the same on several forcing sets (samsim_set_forcing_sites): a grid of columns
  static constexpr bool sites = true;
};
struct KPlate {  // init(1), mo_init.f90:865-945 (bgc off)
  static constexpr bool fixed = true, general = false, sites = false, bgc = false;
  static constexpr int atmoflux_flag = 1, grav_flag = 2, prescribe_flag = 1, grav_heat_flag = 1, flush_heat_flag = 1, turb_flag = 1,
                       salt_flag = 2, boundflux_flag = 1, flush_flag = 1, flood_flag = 2, bottom_flag = 1, precip_flag = 0,
                       harmonic_flag = 2, tank_flag = 1, albedo_flag = 2, lab_snow_flag = 0, freeboard_snow_flag = 0,
                       snow_flush_flag = 1, snow_precip_flag = 1, testcase = 1;
};
// the same flag sets carrying passive tracers (bgc_flag 2: testcase 1 as init ships it; a SHEBA ensemble with tracers)
struct KPlateBgc : KPlate {
  static constexpr bool bgc = true;
};
struct KShebaBgc : KSheba {
  static constexpr bool bgc = true;
};
template <class K>
bool flags_match(const samsim_config &g) {
#define X(f) if (g.f != K::f) return false;
  SAMSIM_FLAG_LIST(X)
#undef X
  return true;
}
// flag read inside a function template over K with `g` = the run-time configuration in scope
#define CFG(f) (K::fixed ? K::f : g.f)

// Device data pointers carry the global address space in their type: an access through them is a global_load / global_store
// even where the pointer itself has been through memory (a struct passed to a non-inlined function), where the compiler
// would otherwise have to assume a generic (flat) address.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double gcdouble;
typedef __attribute__((address_space(1))) int32_t gint32;
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(3))) unsigned long long lu64;
// LDS-resident per-column scalars: slot s of lane l is word s*SAMSIM_BLOCK + l of the block's array
enum lds_slot {
  LD_grav_drain = 0, LD_grav_salt, LD_grav_temp,
  LD_albedo, LD_fl_sw, LD_fl_lw, LD_T2m, LD_liquid_precip, LD_solid_precip,
  LD_T_top, LD_fl_Q_snow, LD_melt_thick,   // state that only the code between the sweeps touches
  LD_fl_Q1,                                // fl_Q(1) of this step (surface balance -> top-layer block, melt film): not a slot of the scalar block
  // the snow cover: read and written before, between and after the two sweeps of every step, never inside them
  LD_m_snow, LD_H_abs_snow, LD_thick_snow, LD_T_snow, LD_psi_s_snow, LD_melt_thick_snow,
  LD_NSLOT
};
#define CL(f) c.ld[LD_##f * SAMSIM_BLOCK]
// Per-column scalars that the common step does not touch (melt-water accumulators, freeboard, T_freeze, the snow's salt and the
// volume fractions only snow_thermo itself reads) are read and written IN PLACE in the scalar block: GS(FREEBOARD) = slot
// SAMSIM_S_FREEBOARD of this lane's column.  19 LDS slots are what 16 one-wave workgroups per CU leave room for.
// (scalar base + 32-bit byte offset, like LAY: slot * bytes-per-row + this lane's column; 38 slots of at most 4 GiB / nlayer)
#define GSI(idx) (*(gdouble *)((gchar *)x.scal + (size_t)(unsigned)((unsigned)(idx) * c.rstride + c.coff)))
#define GS(IDX) GSI(SAMSIM_S_##IDX)

struct Salt {  // liquidus polynomial (func_S_br) and its derivative (func_ddT_S_br), mo_thermo_functions.f90:308-414
  double c2, c3, c4, d2, d3, d4;
};

struct Col {
  gdouble *lay;  // UNIFORM: 4096 bytes into the wave's 64-column block of the layer arrays 
  unsigned col; // this lane's column
  unsigned coff;     // col * 8: byte offset of the column inside a row of the scalar / hand-over blocks
  unsigned lcoff;    // lane * 8: byte offset of the column inside a row of its 64-column block
  unsigned rstride;  // UNIFORM ncol * 8: bytes per row
  size_t astride;    // UNIFORM nlayer * ncol * 8: bytes per layer array
  size_t ncol;
  int N;
  int Na;       // N_active
  int flags;          // COLF_*
  gdouble *spec;       // UNIFORM base of the [DEV_NSPEC][ncol] hand-over block
  int status;      // 0 or the reference's STOP code; where and when it stopped goes straight to the err_layer / err_step arrays
  long long step;  // completed steps; i = step + 1
  // per-column scalars (enum samsim_scalar)
  double fl_q_bottom;
  // The other per-column scalars live in LDS for the whole launch (CL(name), one 8-byte word per lane and slot, no bank
  // conflicts): accumulators (grav_*, melt_out*, melt_err), values that are set under conditions and otherwise carried
  // (freeboard, T_freeze), the forcing of the step (T2m, precipitation, albedo, short- and long-wave flux) and the ensemble
  // perturbation.  Kept in registers they would be live across both layer loops of every step, where the allocator has no
  // room for them: they were spilled to scratch memory, i.e. to HBM, around every sweep.
  ldouble *ld;
  double energy_stored, freshwater, total_resist, thickness, bulk_salin;  // vital signs: live at output points only
  // per-step temporaries that cross sweeps
  double frad;       // fl_rad(N_active)
  double flq2;       // fl_Q(2), handed from the down sweep (which applies the conductive update of layers >= 2) to the top-layer block
  double esum;       // SUM(H_abs before - H_abs after the conductive update) over layers >= 2 (energy assert, mo_heat_fluxes.f90:265-310)
  bool neg_psi;      // MINVAL(psi_s(1:N_active)) of this step's Expulsion is negative (health check at the end of the step)
  double buoy_s;     // SUM(psi_s*thick) over the active layers (from S1)
  double buoy_g;     // SUM(psi_g*thick) after expulsion_flux (from P2)
  double psi_l_top;  // psi_l(1) of this step's Expulsion (the albedo reads it before the down sweep stores the psi arrays)
  double bgc_flood;  // flood_brine of this step (fl_brine_bgc(N_active,1), mo_flood.f90:140-143)
  bool bgc_grav;     // fl_grav_drain ran this step (its fl_brine_bgc assignment, mo_grav_drain.f90:179)
  bool psi_full;     // this step's down sweep stored psi_s / psi_l / psi_g for every layer (not only for layer 1)
  bool ray_all;      // this step's first sweep was the full one (sweep_thermo_expulsion): every Rayleigh number of this column is in the array
};

// Row (a, k) of the layer block starts at a wave-uniform address whenever k is uniform (all top-down loops, and the
// bottom-up loops that run from the wave maximum of N_active); the lane only adds its 32-bit column offset, which lets
// the compiler use scalar-base addressing (global_load ... v_off, s[base]) instead of a 64-bit VGPR address per array.
//
// Arithmetic choices of the fused sweeps (each an ulp-level deviation from the reference's operation order; the parity bar is 1e-6
// relative, observed against the reference's own records <= 1e-11 on one-day windows, tests/test_gpu_reference_windows.py):
//  * quotients that share a divisor go through one reciprocal (Expulsion: /thick three times and the two density constants; getT:
//    /S_br and /S_br**2; S_abs/m and H_abs/m; H/c_l; the constant kappa_l*mu); recip() / quot() of samsim_div.h are the compiler's
//    own Newton sequence without the operand scaling and special-case fix-up around it (the divisors are normal-range numbers);
//  * the liquidus polynomial in Horner form (5 operations instead of 9 per evaluation);
//  * the thicknesses of a regular column from the grid rule: the semi-adaptive grid (mo_layer_dynamics.f90) keeps every layer but
//    the first at thick_0, except the N_middle elastic layers, which all share one value (they receive the same increments in the
//    same order).  Where a column follows that rule (COLF_REGULAR, checked by the full first sweep after samsim_set_state and after
//    every regrid) the sweeps form thick(k) from thick(1), thick(N_top+1) and thick_0 instead of streaming the array; a column that
//    does not follow it (a hand-made state) loads the array and takes the unfused order.
// RARE_CHUNK: the sweeps of the melt season (flushing, freeboard, the unfused order of a step with thin snow or possible flooding)
// walk a column with a per-lane trip count and little arithmetic per layer; with a row requested where it is used every
// iteration waits a full memory latency (2 us under load against 0.1-0.5 us of work).  They request RARE_CHUNK rows at a time
// -- unconditionally, from a clamped row beyond the column's last layer -- and then work through them in order.
#ifndef RARE_CHUNK
#define RARE_CHUNK 8
#endif
// SAMSIM_PATH_MODE 2 (the product): one order of the step per wave, see column_step; 1 = always the unfused order (the checker
// build of tools/path_equiv.py, which shows on the GPU that the two orders give a column the same bits)
#ifndef SAMSIM_PATH_MODE
#define SAMSIM_PATH_MODE 2
#endif

static_assert(SAMSIM_BLOCK == 64, "the blocked layer layout, launch() and DEV_LAY_INDEX are written for one 64-lane wave per column block");
// Address of element (a, k): one 32-bit offset register per row serves all arrays of the row (a 64-bit per-lane address for every
// array costs two registers each and 64-bit vector arithmetic per access).
// Blocked layout (samsim_device.h): c.lay points 4096 bytes into the wave's own column block, so that array a of layer row k is at
// c.lay + (k-1)*DEV_ROWB + (a*512 - 4096) + lane*8: sixteen arrays within the signed 13-bit immediate of one row address.
// LAY takes any k (one 32-bit offset register per row, the lane's part included); LAYU is for a wave-uniform k: the row address is
// scalar arithmetic and the vector offset is the lane's constant c.lcoff.
#define LAY(a, k) (*(gdouble *)((gchar *)c.lay + (size_t)(unsigned)(((unsigned)(k) - 1u) * (unsigned)DEV_ROWB + c.lcoff) + (ptrdiff_t)((int)(a) * 512 - 4096)))
#define LAYU(a, k) (*(gdouble *)((gchar *)c.lay + (size_t)(((unsigned)(k) - 1u) * (unsigned)DEV_ROWB) + (size_t)c.lcoff + (ptrdiff_t)((int)(a) * 512 - 4096)))
// The row loads of the two fused sweeps are streaming accesses: a row is read once per sweep and not again before gigabytes of
// other rows have passed.  With the non-temporal hint (`global_load ... nt`) they do not displace what IS read again soon -- the
// per-column words of a step, the wave's scratch lines, the rows the down sweep has just written near the column's bottom -- from
// the L2: 760 -> 736 ms per 500 steps.  (The same hint on the sweeps' stores costs half of that again: 749 ms.)
#define LAYU_LD(a, k) __builtin_nontemporal_load(&LAYU(a, k))
// hand-over block [DEV_NSPEC][ncol]: scalar base + 32-bit byte offset, like GSI (samsim_create bounds ncol for both)
#define SPEC(i) (*(gdouble *)((gchar *)c.spec + (size_t)(unsigned)((unsigned)(i) * c.rstride + c.coff)))
#define STOPC(code, layer)            \
  do {                                \
    if (!c.status) {                  \
      c.status = (code);              \
      x.err_step[c.col] = c.step + 1; \
      x.err_layer[c.col] = (layer);   \
    }                                 \
    return;                           \
  } while (0)

// Wave-uniform maximum of a per-lane integer over the lanes that are EXECUTING the call (the sweeps are called under
// divergent conditions -- fused / unfused path, frozen columns -- so a shuffle butterfly would read stale registers of
// inactive lanes).  Layer loops run k over 1..wave_max (or wave_max..1) with the body predicated on k <= N_active: k then
// lives in an SGPR and every row address (array, k) is scalar arithmetic; a lane with fewer layers idles exactly as long
// as it would have waited for its wave.
__device__ __forceinline__ int wave_max(int v) {
  unsigned long long mask = __ballot(1);
  int m = 0;
  while (mask) {
    const int lane = __ffsll((long long)mask) - 1;
    const int val = __builtin_amdgcn_readlane(v, lane);
    m = val > m ? val : m;
    mask &= mask - 1;
  }
  return m;
}

// first executing lane of the wave (divergent callers included)
// does any active lane of the wave hold the predicate?  (the ballot of a comparison result IS its lane mask: one scalar compare,
// where __ballot() first turns the predicate into an integer per lane and compares that again)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool wave_leader() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }

#include "samsim_div.h"
// MAX / MIN of the reference as one v_max_f64 / v_min_f64 each.  `a > b ? a : b` compiles to a compare and two 32-bit selects
// (the C semantics for NaN and signed zeros differ from the instruction's), and every vector instruction costs the same four
// cycles: the sweeps clamp some twenty times per layer-cell.  For ordered operands the value is the same (max(-0, +0) may come out
// as +0 instead of -0: equal numbers); a NaN operand loses against a number in both forms where the number is the constant.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double dmax(double a, double b) {
  double r;
  if (__builtin_constant_p(b) && b == 0.0) asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a));
  else asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double dmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
#endif
// a*b + C and max(a, C) with the constant C read from a scalar register pair.  Left to itself the compiler picks the accumulating
// form (v_fmac) for a*b + constant and first copies the constant into the accumulator -- two v_mov_b32 per fused multiply-add, and a
// vector move costs the SIMD the same four cycles as the arithmetic it feeds.  (One scalar operand per instruction is what the
// encoding allows, so a step with two constants is a multiply and an add.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_c(double a, double b, double c_const) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_const));
  return r;
}
__device__ __forceinline__ double max_c(double a, double c_const) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(c_const));
  return r;
}
#else
__device__ __forceinline__ double fma_c(double a, double b, double c_const) { return __builtin_fma(a, b, c_const); }
__device__ __forceinline__ double max_c(double a, double c_const) { return a > c_const ? a : c_const; }
#endif
// S_bu = S_abs/m and H = H_abs/m of one layer, mo_grotz.f90:298-299, 593-594
__device__ __forceinline__ void per_mass(double S_abs, double H_abs, double m, double &S_bu, double &H) {
  const double rm = recip(m);
  S_bu = S_abs * rm;
  H = H_abs * rm;
}
__device__ __forceinline__ double per_mass(double S_abs, double m) { return S_abs / m; }

// func_S_br without / with the S_bu clamp, mo_thermo_functions.f90:308-360.  flang lowers T**2._wp and T**3._wp
// to multiplications (verified bit for bit against the flang build), so do we.
__device__ __forceinline__ double S_br_poly(const Salt &s, double T) {
  return T * (s.c2 + T * (s.c3 + T * s.c4));
}
__device__ __forceinline__ double S_br_clamped(const Salt &s, double T, double S_bu) {
  double v = S_br_poly(s, T);
  return dmax(v, S_bu);   // one v_max_f64 for the compare and two 32-bit selects of `v < S_bu ? S_bu : v`: the same number for numbers
}
// func_ddT_S_br, mo_thermo_functions.f90:380-414 (derivative-only clamp below -20 C)
__device__ __forceinline__ double ddT_S_br(const Salt &s, double T) {
  const double T_crit = -20.0;
  double d = s.d2 + 2.0 * s.d3 * T + 3.0 * s.d4 * (T * T);
  if (T < T_crit) d = s.d2 + 2.0 * s.d3 * T_crit + 3.0 * s.d4 * (T_crit * T_crit);
  return d;
}

// residual f(T_0) and its derivative of the enthalpy relation, mo_thermo_functions.f90:95-96 / :109-110 (the first evaluation
// clamps S_br at 1e-9, the ones in the loop at 1e-10, as in the reference)
__device__ __forceinline__ void newton_terms(const Salt &s, double H, double S_bu, double T_0, double sb, double sb_floor,
                                             double &f, double &ddT_f) {
  if (sb > 0.0001) {  // neither clamp is active: one reciprocal serves both quotients
    const double inv = recip(sb);
    f = -latent_heat - H + latent_heat * S_bu * inv + c_s * T_0 + c_s_beta * T_0 * T_0 / 2.0;
    ddT_f = c_s + c_s_beta * T_0 - latent_heat * S_bu * ddT_S_br(s, T_0) * (inv * inv);
    return;
  }
  f = -latent_heat - H + latent_heat * S_bu / dmax(sb, sb_floor) + c_s * T_0 + c_s_beta * T_0 * T_0 / 2.0;
  ddT_f = c_s + c_s_beta * T_0 - latent_heat * S_bu * ddT_S_br(s, T_0) / dmax(sb * sb, 0.0000000001);
}

// One division per Newton step of getT instead of two: with f = N/sb**2 and f' = D/sb**2 (N = A*sb**2 + L*S_bu*sb, D = B*sb**2 -
// L*S_bu*S_br'(T), A and B the polynomial parts) the step is T_0 - N/D and the stopping rule |f| > 1 reads |N| > sb**2: the same
// iteration in exact arithmetic.  getT runs 3.6 evaluations per layer-cell on the bench ensemble, all of them on the critical path
// of the up sweep.
// One evaluation of the Newton step of getT from T_0 in the one-division form, with fused multiply-adds (one rounding per a*b+c
// instead of two; each iterate within an ulp or two of the reference's, like the shared reciprocals): T_new = T_0 - N/D, more =
// |N| > sb**2 (the reference's |f| > 1), ok = the liquidus salinity at T_0 is above 1e-4, i.e. the reference's clamps of S_br
// (1e-9 / 1e-10) are inactive and this form is the step.  A0 = -latent_heat - H and LS = latent_heat * S_bu are the caller's
// (the same for every evaluation of a layer).  Straight-line: no branch, 23 vector instructions.
// Three of its fused multiply-adds have a constant multiplier AND a constant addend (c3, c_s twice, 2*d3); the instruction takes one
// operand from a scalar register, so the compiler copies the other into a vector register pair first -- two v_mov_b32 per constant
// and evaluation, re-done inside getT's loop (no hoisting: Makefile).  NewtonConsts holds those three as vector values the caller
// forms once per layer.
struct NewtonConsts { double c3, cs, d3x2; };
__device__ __forceinline__ NewtonConsts newton_consts(const Salt &s) {
  NewtonConsts n = {s.c3, c_s, 2.0 * s.d3};
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(n.c3), "+v"(n.cs), "+v"(n.d3x2));   // (vector registers from here on: not rematerialised per use)
#endif
  return n;
}
__device__ __forceinline__ void newton_eval(const Salt &s, const NewtonConsts &n, double A0, double LS, double T_0, double &T_new, bool &more, bool &ok) {
  const double sbf = T_0 * fma_c(T_0, __builtin_fma(T_0, s.c4, n.c3), s.c2);      // (fma_c: the addend from a scalar register pair)
  const double sb2 = sbf * sbf;
  const double A = __builtin_fma(T_0, __builtin_fma(T_0, 0.5 * c_s_beta, n.cs), A0);
  const double B = __builtin_fma(c_s_beta, T_0, n.cs);
  const double num = __builtin_fma(A, sb2, LS * sbf);
  const double Tc = max_c(T_0, -20.0);                       // derivative-only clamp below -20 C, mo_thermo_functions.f90:408-412
  const double dd = fma_c(Tc, __builtin_fma(Tc, 3.0 * s.d4, n.d3x2), s.d2);
  const double den = __builtin_fma(B, sb2, -(LS * dd));
  T_new = T_0 - quot(num, den);
  more = fabs(num) > sb2;
  ok = sbf > 0.0001;
}

// one Newton step from T_0: returns the new iterate and whether |f(T_0)| > 1 (general routine: any S_br)
__device__ __forceinline__ bool newton_step(const Salt &s, double H, double S_bu, double T_0, double sb_floor, double &T_new) {
  {
    bool more, ok;
    double Tn;
    newton_eval(s, newton_consts(s), -latent_heat - H, latent_heat * S_bu, T_0, Tn, more, ok);
    if (ok) { T_new = Tn; return more; }
  }
  const double sb = S_br_poly(s, T_0);
  double f, ddT_f;
  newton_terms(s, H, S_bu, T_0, sb, sb_floor, f, ddT_f);
  T_new = T_0 - quot(f, ddT_f);
  return fabs(f) > 1.0;
}

// H/c_l: the temperature of pure brine of enthalpy H (first line of getT, mo_thermo_functions.f90:84)
__device__ __forceinline__ double T_liquid(double H) {
  return H * (1.0 / c_l);
}

// the temperature at which brine of salinity S_bu starts to freeze, mo_thermo_functions.f90:85-92 (Newton from -1 C)
__device__ __forceinline__ double T_freeze_of(const Salt &s, double S_bu) {
  double T_fr = -1.0;
  while (fabs(S_br_poly(s, T_fr) / S_bu - 1.0) > (double)0.0001f) {  // tolerance is a float32 literal (:87)
    const double t0 = T_fr;
    T_fr = t0 - (S_br_poly(s, t0) - S_bu) / ddT_S_br(s, t0);
  }
  return T_fr;
}

// getT, mo_thermo_functions.f90:62-143: guarded Newton iteration for T and the solid mass fraction phi.
// Returns 99 (the reference's STOP code) when 260 iterations do not converge.
__device__ __forceinline__ int getT(const Salt &s, double H, double S_bu, double T_in, double &T_out, double &phi_out, int *evals = nullptr) {
  double T = T_liquid(H), phi = phi_out;
  int rc = 0;
  if (S_br_clamped(s, T, S_bu) > S_bu && S_bu > 0.001) {
    double T_fr = 0.0, T_0;
    bool have_T_fr = false;
    T_0 = T_in;
    bool more = newton_step(s, H, S_bu, T_0, 0.000000001, T);
    int i = 0;
    while (more) {
      T_0 = T;
      if (T_0 > 0.0 || T_0 < -200.0) {
        // The reference computes the freezing temperature T_fr up front (mo_thermo_functions.f90:85-92) and only reads it
        // here.  It has no other effect, so it is evaluated on first use: same value, no Newton loop in the common case.
        if (!have_T_fr) {
          T_fr = T_freeze_of(s, S_bu);
          have_T_fr = true;
        }
        T_0 = T_fr;
      }
      more = newton_step(s, H, S_bu, T_0, 0.0000000001, T);
#if SAMSIM_STAMPS == 2
      if (evals) *evals += 1;
#endif
      if (++i == 260) { rc = 99; break; }
    }
    phi = 1.0 - quot(S_bu, S_br_clamped(s, T, S_bu));
  } else if (S_bu < 0.001) {
    if (H > 0.0) { phi = 0.0; T = H / c_l; }
    else if (H <= -latent_heat) { phi = 1.0; T = (H + latent_heat) / c_s; }
    else if (H <= 0.0 && -latent_heat < H) { T = 0.0; phi = -H / latent_heat; }
  } else {
    phi = 0.0;
  }
  T_out = T;
  phi_out = phi;
  return rc;
}

// getT for the layers of a sweep (the up sweeps, the full first sweep): the same iteration, arranged for a wave.  Winter columns
// are mushy layers whose iterates stay inside (-200, 0) and whose liquidus salinity stays above 1e-4: for them getT is a first
// evaluation and a loop of further ones, and every lane of the wave runs that loop together -- `while (some lane wants more)`, the
// update selected per lane -- so that the loop is straight-line vector code under ONE scalar branch, without the exec-mask
// bookkeeping of a per-lane `while` (a wave runs as many trips as its slowest lane either way: 3.57 against a lane mean of 3.33
// in winter, 7.1 against 4.4 in the melt season).  A lane that is anything else -- fresh ice, pure brine, an iterate that leaves
// the interval and needs T_fr, S_br under 1e-4, no convergence -- is redone by the general routine above, on its own: what a lane
// gets depends on its own column only, and the arithmetic (newton_eval) is the general routine's.
// WARM (the sweeps of a melt season: the full first sweep, the up sweep of a flushing wave): an iterate that leaves (-200, 0) is
// replaced by the freezing temperature inside the loop, exactly where the general routine does it, instead of sending the lane
// through the general routine afterwards -- near 0 C a third of the layers of a wave hold such a lane, and each cost the wave a
// second, slower iteration from the start.  The winter sweeps keep the loop without it (three registers less in their layer loop).
template <bool WARM = false>
__device__ __forceinline__ int getT_chain(const Salt &s, double H, double S_bu, double T_in, double &T_out, double &phi_out, int *evals = nullptr) {
  const double Tl = T_liquid(H);
  const bool mushy = S_br_clamped(s, Tl, S_bu) > S_bu && S_bu > 0.001;
  const double A0 = -latent_heat - H, LS = latent_heat * S_bu;
  const NewtonConsts nc = newton_consts(s);
  double T;
  bool more0, ok;
  newton_eval(s, nc, A0, LS, T_in, T, more0, ok);
  // `more` and `odd` travel through the loop as 0 / 1 words in vector registers, not as lane masks: the loop test is then one compare
  // whose result is the branch condition and the select mask of the update at once
  int odd_i = (!mushy || !ok) ? 1 : 0;
  int more_i = (more0 && odd_i == 0) ? 1 : 0;
  int i = 0;
  double T_fr = 0.0;
  bool have_T_fr = false;
  ISA_MARK("NEWTON_LOOP");
  for (;;) {
    const bool more = more_i != 0;
    if (__builtin_amdgcn_ballot_w64(more) == 0ull) break;
    if (WARM) {
      const bool out = more && (T > 0.0 || T < -200.0);
      if (wave_any(out)) {
        if (out) {
          if (!have_T_fr) { T_fr = T_freeze_of(s, S_bu); have_T_fr = true; }
          T = T_fr;
        }
      }
    }
    double Tn;
    bool m2, ok2;
    newton_eval(s, nc, A0, LS, T, Tn, m2, ok2);
    const bool left = WARM ? !ok2 : (T > 0.0 || T < -200.0 || !ok2);   // (the test is on the iterate the evaluation started from)
#if SAMSIM_STAMPS == 2
    if (evals && more) *evals += 1;
#endif
    T = more ? Tn : T;
    odd_i = left ? (odd_i | more_i) : odd_i;
    more_i = (left || !m2) ? 0 : more_i;
    if (++i == 260) { odd_i |= more_i; break; }       // no convergence in 260 evaluations: the general routine reports it (STOP 99)
  }
  const bool odd = odd_i != 0;
  ISA_MARK("NEWTON_LOOP_END");
  double phi = 1.0 - quot(S_bu, S_br_clamped(s, T, S_bu));
  int rc = 0;
  if (odd) {
    phi = phi_out;
#if SAMSIM_STAMPS == 2
    int ev0 = evals ? *evals : 0;
#endif
    rc = getT(s, H, S_bu, T_in, T, phi, evals);
#if SAMSIM_STAMPS == 2
    if (evals) *evals += ((*evals - ev0) << 16) | (1 << 30);   // (decoded by the caller: redone by the general routine, its evaluations)
#endif
  }
  T_out = T;
  phi_out = phi;
  return rc;
}

// The solid fraction getT returned for a layer, recomputed from the temperature it returned and the values it was called
// with (mo_thermo_functions.f90:84,129,131-143): same operands, same operations, so the same phi bit for bit.  The down sweeps
// use it instead of loading phi (one array less to hand over).
__device__ __forceinline__ double phi_from_T(const Salt &s, double H, double S_bu, double S_br_T) {
  // S_bu > 0.001 is a mushy layer or pure brine.  getT gives pure brine phi = 0 and T = H/c_l, whose clamped liquidus salinity
  // S_br_T is S_bu itself -- and quot(x, x) is exactly 1 (samsim_div.h: the residual correction removes what the rounded
  // product x*r is off by) -- so the mushy layer's formula serves both and the liquidus need not be evaluated at H/c_l again.
  if (S_bu > 0.001) return 1.0 - quot(S_bu, S_br_T);
  if (S_bu < 0.001) {
    if (H > 0.0) return 0.0;
    if (H <= -latent_heat) return 1.0;
    return -H / latent_heat;
  }
  return 0.0;
}

// x**3.10 of the permeability law (mo_grav_drain.f90:105, mo_flush.f90:119,128, mo_flood.f90:73) as exp(3.1*log(x)):
// within ~4e-15 relative of the correctly rounded pow() the reference links (|3.1*log x| <= 22 for x <= 1000), at a
// third of its instructions and without the double-double constant tables that push the layer loops into spills.
}  // namespace
#define SP_QUOT(a, b) quot(a, b)
#include "samsim_pow.h"
namespace {
// x*x*x * exp(0.1*log(x)) with a plain logarithm: within ~4 ulp of the correctly rounded power (samsim_pow.h)
__device__ __forceinline__ double pow_3p1(double x) { return sp_pow_3p1(x); }

__device__ __forceinline__ double pow_1p5(double x) { return sp_pow_1p5(x); }   // samsim_pow.h
__device__ __forceinline__ double pow_4(double x) { return sp_pow_4(x); }

// func_density, mo_functions.f90:51-62
__device__ double func_density(double T, double S) {
  double density_0 = 999.842594 + 6.8 / 100.0 * T;
  return density_0 + 0.825 * S + (-5.7 / 1000.0) * pow_1p5(dmax(S, 0.0));
}

// func_T_freeze, mo_functions.f90:239-250 (float32 products of default-REAL literals)
__device__ double func_T_freeze(double S_bu, int salt_flag, double tf_c3) {
  if (salt_flag == 2) {
    return -0.0592 * S_bu - (double)9.37f * (S_bu * S_bu) - tf_c3 * (S_bu * S_bu * S_bu);
  } else {
    const float a = 1.710523f * 1e-3f, b = 2.154996f * 1e-4f;
    return -0.0575 * S_bu + (double)a * pow_1p5(S_bu) - (double)b * (S_bu * S_bu);
  }
}

// func_albedo, mo_functions.f90:157-208 (float32 literals)
__device__ double func_albedo(double thick_snow, double T_snow, double psi_l, double thick_min, int albedo_flag) {
  const double ice_dry = (double)0.75f, ice_wet = (double)0.6f, snow_dry = (double)0.85f, snow_wet = (double)0.75f,
               water = (double)0.2f;
  double albedo;
  if (thick_snow > thick_min) {
    albedo = (T_snow < (double)(-0.01f)) ? snow_dry : snow_wet;
    albedo = ice_dry + (albedo - ice_dry) * dmin(1.0, quot(thick_snow, 0.3));
  } else {
    if (psi_l > 0.9) albedo = water;
    else if (psi_l > 0.6) albedo = ice_wet + (water - ice_wet) * ((psi_l - 0.6) / 0.3);
    else if (psi_l > 0.2) albedo = ice_wet;
    else albedo = ice_dry;
  }
  if (albedo_flag == 1) {
    if (thick_snow > thick_min) albedo = (T_snow < (double)(-0.01f)) ? snow_dry : snow_wet;
    else albedo = (psi_l < (double)0.8f) ? ice_dry : water;
  }
  return albedo;
}

// func_k_snow, mo_snow.f90:560-573
__device__ double func_k_snow(double m_snow, double thick_snow) {
  const double c0 = 0.138, c1 = -1.01 / 1000.0, c2 = 3.233 / 1000000.0;
  double r = quot(m_snow, thick_snow);
  double k_snow = c0 + quot(c1 * m_snow, thick_snow) + c2 * (r * r);
  return k_snow + (double)0.15f;
}

// 3-hourly table time axis, mo_functions.f90:323-325
__device__ __forceinline__ double time_input(int k) { return ((double)(float)k - 1.0) * 3600.0 * 3.0; }

struct Ctx {
  const DevParams *p;
  // The data pointers are taken from DIRECT kernel arguments, not from the parameter block: only then does the compiler
  // know they are global-memory pointers (global_load/global_store with scalar base) instead of generic flat ones.
  gcdouble *f_sw, *f_lw, *f_T2m, *f_precip;
  gdouble *out_lay, *out_scal;
  gdouble *scal;  // [SAMSIM_NSCAL][ncol] scalar block: slots that are not carried in registers (fl_rest) are read / written in place
  gint32 *out_n_active;
  gint32 *err_layer;                                   // [ncol] layer and step of a column's STOP (written once, when it stops)
  __attribute__((address_space(1))) long long *err_step;
  long long out_col0, out_ncols;
  Salt salt;
  double p17, p14, tf_c3;
  // salinity of the water below the ice: cfg.S_bu_bottom (uniform), or the column's tank budget with tank_flag 2 (mo_grotz.f90:573)
  double S_bu_bottom;
  double rho_bottom;   // func_density(T_bottom, S_bu_bottom) of sub_turb_flux, evaluated once per launch where the water below is uniform
  // passive tracers (bgc_flag 2, KGeneric only): amounts [n_bgc][N][ncol], concentration below the ice [n_bgc][ncol], this
  // step's brine fluxes [BFL_NROW][N][ncol], snapshot of the output window
  int soff;   // start of this column's forcing set in the tables (0 unless samsim_set_forcing_sites gave several)
  // the water below a grid of columns (samsim_set_ocean, K::sites instantiations): offset added to the oceanic heat flux the
  // testcase sets every step (sub_test4), and whether S_bu_bottom above is this column's own value
  double dflq;
  bool ocean_sbu;
  gdouble *bgc, *bgc_bot, *bfl, *out_bgc, *out_bgc_bot;
  int n_bgc;
  double bgc_total0;
  // Which rows of the Rayleigh-number array the last up sweep wrote (bit k-1 of word (k-1)/64 = row k), per wave, in LDS.
  // Gravity drainage only reads ray(k) where it exceeds ray_crit (mo_grav_drain.f90:144), which in winter holds in two or three
  // of 80 layers: the up sweep stores a row only when some column of the wave is above the threshold in that layer (or when the
  // whole array is wanted: output, end of a launch), the down sweeps load only those rows and take 0 elsewhere.
  // The words pass data between the lanes of the wave (the wave's first lane ORs a bit in, every lane reads it in the next step's
  // down sweep): volatile, so that every access is an LDS instruction in program order -- one wave issues its LDS instructions
  // in order and the LDS serves them in order -- and a wave barrier where the phases change (zeroing -> setting -> reading).
  volatile lu64 *rflag;
  bool ray_rows_all;   // this up sweep stores every row

#if SAMSIM_STAMPS
  mutable Stamps st;
#endif
};
#define BGC(t, k) (x.bgc + ((size_t)(t) * (size_t)c.N + (size_t)((k) - 1)) * c.ncol)[c.col]
#define BGC_BOT(t) (x.bgc_bot + (size_t)(t) * c.ncol)[c.col]
#define BFL(r, k) (x.bfl + ((size_t)(r) * (size_t)c.N + (size_t)((k) - 1)) * c.ncol)[c.col]
// tracers exist only in the run-time-flag instantiation; in the fixed ones the test folds to false
#define HAS_BGC (K::bgc && x.n_bgc > 0)

// density of the water below the ice (sub_turb_flux, mo_functions.f90:355): the same number in every step of every column unless
// the tank budget (tank_flag 2) moves S_bu_bottom
template <class K>
__device__ __forceinline__ double ocean_density(const Ctx &x) {
  if ((K::fixed ? K::tank_flag : x.p->cfg.tank_flag) == 2 || (K::sites && x.ocean_sbu)) return func_density(x.p->cfg.T_bottom, x.S_bu_bottom);
  return x.rho_bottom;
}

// thick(k), k >= 2, of a column that follows the grid rule; th_mid = thick(N_top+1)
__device__ __forceinline__ double thick_by_rule(int k, int n_top, int n_middle, double th_mid, double thick_0) {
  return (k > n_top && k <= n_top + n_middle) ? th_mid : thick_0;
}

// The thickness of layer kk for the sweeps of the melt season: from the grid rule where the column follows it, else from the array
struct ThickRule { bool reg; int n_top, n_middle; double th_mid, thick_0; };
#define THICK_RULE_INIT(tr)                                                                                   \
  ThickRule tr;                                                                                               \
  tr.reg = (c.flags & COLF_REGULAR) != 0; tr.n_top = x.p->cfg.n_top; tr.n_middle = x.p->cfg.n_middle; \
  tr.thick_0 = x.p->cfg.thick_0; tr.th_mid = LAY(SAMSIM_A_THICK, tr.n_top + 1)
#define THICK_AT(tr, kk) ((tr.reg && (kk) >= 2) ? thick_by_rule(kk, tr.n_top, tr.n_middle, tr.th_mid, tr.thick_0) : LAY(SAMSIM_A_THICK, kk))

// Does row k of the Rayleigh-number array hold this column's current value?  Row 1 is written by the first sweep of every step
// (prologue_top_layer / sweep_thermo_expulsion), the other rows by the last up sweep where flagged (Ctx::rflag), and all of them by
// this step's full first sweep.  A row that was not written held no value above ray_crit in any column of the wave.
__device__ __forceinline__ bool ray_row_valid(const Col &c, const Ctx &x, int k) {
  return k == 1 || c.ray_all || ((x.rflag[(k - 1) >> 6] >> ((k - 1) & 63)) & 1ull) != 0ull;
}


// ---------------------------------------------------------------- func_freeboard, mo_functions.f90:79-130
// O(N): one pass for the column totals, one pass for the waterline search with prefix sums (the reference
// recomputes the suffix sums for every candidate layer).
template <class K>
__device__ RARE double func_freeboard(Col &c, const Ctx &x) {
  const int Na = c.Na;
  double snowmass = ((K::fixed ? K::freeboard_snow_flag : x.p->cfg.freeboard_snow_flag) == 0) ? CL(m_snow) : 0.0;
  THICK_RULE_INIT(tr);
  // The column totals SUM(psi_s*thick) and SUM(psi_g*thick): the sweep that stored the volume-fraction rows (sweep_down_fused,
  // sweep_expulsion_transfer, refill_psi_rows) summed them over layers 2..N_active as it went, top -> bottom like the reference's
  // SUM, and left the two sums in the hand-over block; layer 1 -- whose thickness snow slush, the melt film and melt water may have
  // changed since -- is added here with what it holds now.  (Round 2 walked the whole column for them: two rows per layer-cell in
  // every step of a melt season.)
  const double th1 = LAY(SAMSIM_A_THICK, 1);
  const double A = LAY(SAMSIM_A_PSI_S, 1) * th1 + ((Na >= 2) ? SPEC(SP_FB_A2) : 0.0);
  const double G = LAY(SAMSIM_A_PSI_G, 1) * th1 + ((Na >= 2) ? SPEC(SP_FB_G2) : 0.0);
  double buoy = A * (rho_l - rho_s) + G * rho_l;
  double freeboard;
  if (snowmass > buoy) {
    freeboard = (buoy - snowmass) / rho_l;
  } else {
    double Ap = 0.0, Gp = 0.0, Mp = 0.0, Tp = 0.0;  // prefix sums over 1..k-1
    double test2 = 0.0, mk = 0.0, thk = 1.0;
    bool done = false;
    for (int k0 = 1; !done; k0 += RARE_CHUNK) {
      double m_[RARE_CHUNK], th_[RARE_CHUNK], ps_[RARE_CHUNK], pg_[RARE_CHUNK];
#pragma unroll
      for (int i = 0; i < RARE_CHUNK; ++i) {
        const int kk = (k0 + i <= c.N) ? k0 + i : c.N;
        m_[i] = LAY(SAMSIM_A_M, kk); th_[i] = THICK_AT(tr, kk);
        ps_[i] = LAY(SAMSIM_A_PSI_S, kk); pg_[i] = LAY(SAMSIM_A_PSI_G, kk);
      }
#pragma unroll
      for (int i = 0; i < RARE_CHUNK; ++i) {
        if (!done) {
          const int k = k0 + i;
          mk = m_[i];
          thk = th_[i];
          double a = ps_[i] * thk, g = pg_[i] * thk;
          // buoyancy of the layers below k, mass of layers 1..k
          test2 = (k == Na) ? 0.0 : ((A - (Ap + a)) * (rho_l - rho_s) + (G - (Gp + g)) * rho_l);
          double test1 = (Mp + mk) + snowmass;
          if (!(test1 < test2) || k >= Na) done = true;
          else { Ap += a; Gp += g; Mp += mk; Tp += thk; }
        }
      }
    }
    double test1 = Mp + snowmass;
    freeboard = test2 - test1 + (rho_l - mk / thk) * thk;
    freeboard = freeboard / rho_l;
    freeboard = freeboard + Tp;
  }
  return freeboard;
}

// ---------------------------------------------------------------- snow, mo_snow.f90
// snow_coupling, mo_snow.f90:61-104.  The reference passes T_snow / T as both the guess and the result of getT;
// by-reference argument passing makes the guess H/c_l (getT's first statement overwrites it).
// (core: the top layer's enthalpy, temperature and solid fraction in registers -- the fused down sweep calls it between the
// brine expulsion and the drainage of layer 1; the wrapper below works on the arrays, as the unfused order and the up sweep do)
template <class K>
__device__ RARE int snow_coupling_core(Col &c, const Ctx &x, double &H_abs, const double m, const double S_bu, double &T, double &phi) {
  const Salt &s = x.salt;
  double H;
  const double m_snow = CL(m_snow), S_abs_snow = GS(S_ABS_SNOW);
  double phi_sn = GS(PHI_S);
  int rc = 0;
  H_abs = H_abs + m_snow * latent_heat + CL(H_abs_snow);
  CL(H_abs_snow) = -m_snow * latent_heat;
  H = H_abs / m;
#define COUPLE_GETT()                                                                                      \
  do {                                                                                                     \
    double hs = CL(H_abs_snow) / m_snow;                                                                     \
    double T_sn = CL(T_snow);                                                                              \
    rc |= getT(s, hs, S_abs_snow / m_snow, hs / c_l, T_sn, phi_sn);                                        \
    CL(T_snow) = T_sn;                                                                                     \
    rc |= getT(s, H, S_bu, H / c_l, T, phi);                                                               \
  } while (0)
  COUPLE_GETT();
  if (T > 0.0 && H_abs <= -CL(H_abs_snow)) {
    CL(H_abs_snow) = CL(H_abs_snow) + H_abs;
    H_abs = 0.0;
    COUPLE_GETT();
  } else if (T > 0.0 && H_abs > -CL(H_abs_snow)) {
    H_abs = (H_abs + CL(H_abs_snow)) * m / m_snow / (1.0 + m / m_snow);
    CL(H_abs_snow) = H_abs * m_snow / m;
    COUPLE_GETT();
  } else {
    int jj = 0;
    while (fabs(T - CL(T_snow)) > (double)0.1f && jj < 201) {
      double d = CL(T_snow) - (CL(T_snow) + T) / 2.0;
      double sg = dmax(fabs(d), 0.1);
      if (signbit(d)) sg = -sg;
      CL(H_abs_snow) = CL(H_abs_snow) - sg * c_s * m_snow;
      H_abs = H_abs + sg * c_s * m_snow;
      jj = jj + 1;
      H = H_abs / m;
      COUPLE_GETT();
    }
    if (jj > 200 && fabs(T - CL(T_snow)) > 1.0) rc = 16;
  }
#undef COUPLE_GETT
  GS(PHI_S) = phi_sn;
  return rc ? (rc == 16 ? 16 : 99) : 0;
}
template <class K>
__device__ RARE void snow_coupling(Col &c, const Ctx &x) {
  double H_abs = LAY(SAMSIM_A_H_ABS, 1), T = LAY(SAMSIM_A_T, 1), phi = LAY(SAMSIM_A_PHI, 1);
  const int rc = snow_coupling_core<K>(c, x, H_abs, LAY(SAMSIM_A_M, 1), LAY(SAMSIM_A_S_BU, 1), T, phi);
  LAY(SAMSIM_A_H_ABS, 1) = H_abs;
  LAY(SAMSIM_A_T, 1) = T;
  LAY(SAMSIM_A_PHI, 1) = phi;
  if (rc) STOPC(rc, 1);
}


// snow_precip (mo_snow.f90:123-150) and snow_precip_0 (:167-192), called from mo_grotz.f90:251-265
template <class K>
__device__ __forceinline__ void snow_fall(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  if (!(dmax(CL(liquid_precip), CL(solid_precip)) > 0.0)) return;
  const double dt = g.dt, T2m = CL(T2m);
  double solid, liquid;
  if (CFG(precip_flag) == 0) { solid = CL(solid_precip); liquid = CL(liquid_precip); }
  else if (T2m > 0.0) { solid = 0.0; liquid = CL(liquid_precip); }
  else { solid = CL(liquid_precip); liquid = 0.0; }
  if (c.Na > 1) {
    double d_thick = dt * solid * rho_l / rho_snow;
    CL(m_snow) = CL(m_snow) + dt * rho_l * (liquid + solid);
    CL(thick_snow) = CL(thick_snow) + d_thick;
    CL(H_abs_snow) = CL(H_abs_snow) + dt * T2m * liquid * rho_l * c_l;
    CL(H_abs_snow) = CL(H_abs_snow) + dt * dmin(T2m, -1.0) * solid * rho_l * c_s;
    CL(H_abs_snow) = CL(H_abs_snow) - dt * solid * rho_l * latent_heat;
  } else {
    double H_abs = LAY(SAMSIM_A_H_ABS, 1), S_abs = LAY(SAMSIM_A_S_ABS, 1);
    const double m = LAY(SAMSIM_A_M, 1), T = LAY(SAMSIM_A_T, 1);
    H_abs = H_abs + (liquid + solid) * (T2m - T) * dt;
    H_abs = H_abs - solid * latent_heat * dt;
    S_abs = S_abs - (liquid + solid) * S_abs / m * dt;
    LAY(SAMSIM_A_H_ABS, 1) = H_abs;
    LAY(SAMSIM_A_S_ABS, 1) = S_abs;
  }
}

// snow_thermo (mo_snow.f90:212-320) / snow_thermo_meltwater (:331-454) wrapped in the block of
// mo_grotz.f90:273-292 and :604-624
template <class K>
__device__ RARE void snow_block(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  // (psi_l_snow, psi_g_snow and S_abs_snow are only read by this routine and by rare events -- flooding, the melting of a thin
  // cover, melt water from the snow: they live in the scalar block, which is written where a value changes)
  if (!(CL(thick_snow) > 0.0)) {
    if ((c.flags & COLF_RESTART) || CL(m_snow) != 0.0 || CL(thick_snow) != 0.0 || CL(psi_s_snow) != 0.0 || CL(H_abs_snow) != 0.0) {   // the cover has just gone (or the state is new)
      GS(PSI_L_SNOW) = 0.0; GS(PSI_G_SNOW) = 0.0; GS(S_ABS_SNOW) = 0.0;
    }
    CL(thick_snow) = 0.0; CL(m_snow) = 0.0; CL(psi_s_snow) = 0.0;
    CL(H_abs_snow) = 0.0; CL(melt_thick_snow) = 0.0;
    return;
  }
  double psi_l_sn, psi_g_sn;
  const double S_abs_sn = GS(S_ABS_SNOW);
  CL(melt_thick_snow) = 0.0;
  const bool meltwater = (CFG(snow_flush_flag) == 1);
  double m = LAY(SAMSIM_A_M, 1), thick = LAY(SAMSIM_A_THICK, 1), H_abs = LAY(SAMSIM_A_H_ABS, 1);
  bool touched = false;
  double phi_snow = 0.0, max_lwc, max_lwc_v, sat_snow;
  const double H_snow = quot(CL(H_abs_snow), CL(m_snow)), S_bu_snow = quot(S_abs_sn, CL(m_snow)), psi_s_old = CL(psi_s_snow);
  const double T_in = CL(T_snow);
  double T_sn = T_in;
  int rc = getT(x.salt, H_snow, S_bu_snow, T_in, T_sn, phi_snow);
  CL(T_snow) = T_sn;
  if (rc) STOPC(99, 0);
  CL(psi_s_snow) = quot(quot(CL(m_snow) * phi_snow, rho_s), CL(thick_snow));
  psi_l_sn = quot(quot(CL(m_snow) * (1.0 - phi_snow), rho_l), CL(thick_snow));
  if (CL(psi_s_snow) + psi_l_sn > 1.0) {
    CL(thick_snow) = CL(m_snow) * (phi_snow / rho_s + (1.0 - phi_snow) / rho_l);
    CL(psi_s_snow) = CL(m_snow) * phi_snow / rho_s / CL(thick_snow);
    psi_l_sn = CL(m_snow) * (1.0 - phi_snow) / rho_l / CL(thick_snow);
    if (fabs(CL(psi_s_snow) + psi_l_sn - 1.0) > 0.0000001) { GS(PSI_L_SNOW) = psi_l_sn; STOPC(345, 0); }
  }
  psi_g_sn = 1.0 - CL(psi_s_snow) - psi_l_sn;
  if (CL(psi_s_snow) > 0.0) max_lwc = quot(0.057 * (1.0 - CL(psi_s_snow)), CL(psi_s_snow)) + 0.017;
  else max_lwc = 0.0;

  if (psi_s_old > CL(psi_s_snow) && CL(psi_s_snow) > 0.0) {
    if ((1.0 - phi_snow) > max_lwc) CL(thick_snow) = CL(thick_snow) * (1.0 - (psi_s_old - CL(psi_s_snow)) / psi_s_old);
    double tmin = (phi_snow * CL(m_snow) / rho_s + (1.0 - phi_snow) * CL(m_snow) / rho_l);
    if (CL(thick_snow) < tmin) CL(thick_snow) = tmin;
    CL(psi_s_snow) = CL(m_snow) * phi_snow / rho_s / CL(thick_snow);
    psi_l_sn = CL(m_snow) * (1.0 - phi_snow) / rho_l / CL(thick_snow);
    psi_g_sn = 1.0 - CL(psi_s_snow) - psi_l_sn;
    psi_g_sn = fabs(psi_g_sn);
  } else if (CL(psi_s_snow) < 0.000001) {
    CL(thick_snow) = CL(m_snow) / rho_l;
    CL(psi_s_snow) = 0.0; psi_g_sn = 0.0; psi_l_sn = 1.0;
  }

  const bool wet = (1.0 - phi_snow) > max_lwc && psi_g_sn > 0.0 && (!meltwater || psi_l_sn > 0.0);
  if (wet) {
    touched = true;
    const double T_snow = CL(T_snow), pss = CL(psi_s_snow);
    max_lwc_v = max_lwc * CL(m_snow) / (rho_l * CL(thick_snow));
    if (!meltwater) {
      sat_snow = CL(thick_snow) * (psi_l_sn - max_lwc_v);
      sat_snow = sat_snow / (1.0 - pss - max_lwc_v - dmin(gas_snow_ice2, psi_g_sn));
      CL(thick_snow) = CL(thick_snow) - sat_snow;
      thick = thick + sat_snow;
      CL(m_snow) = CL(m_snow) - sat_snow * (pss * rho_s + (1.0 - pss - gas_snow_ice2) * rho_l);
      m = m + sat_snow * (pss * rho_s + (1.0 - pss - gas_snow_ice2) * rho_l);
      CL(H_abs_snow) = CL(H_abs_snow) - sat_snow * pss * rho_s * c_s * T_snow;
      H_abs = H_abs + sat_snow * pss * rho_s * c_s * T_snow;
      CL(H_abs_snow) = CL(H_abs_snow) + sat_snow * pss * rho_s * latent_heat;
      H_abs = H_abs - sat_snow * pss * rho_s * latent_heat;
      CL(H_abs_snow) = CL(H_abs_snow) - sat_snow * (1.0 - pss) * rho_l * c_l * T_snow;
      H_abs = H_abs + sat_snow * (1.0 - pss) * rho_l * c_l * T_snow;
    } else {
      const double ksf = g.k_snow_flush;
      double slush = (psi_l_sn - max_lwc_v) * (1.0 - ksf);
      double flush = (psi_l_sn - max_lwc_v) * ksf;
      CL(melt_thick_snow) = CL(thick_snow) * flush;
      sat_snow = CL(thick_snow) * (slush);
      sat_snow = sat_snow / (1.0 - pss - max_lwc_v - dmin(gas_snow_ice2, psi_g_sn));
      const double gmin = dmin(gas_snow_ice2, psi_g_sn);
      CL(thick_snow) = CL(thick_snow) - sat_snow - CL(melt_thick_snow);
      thick = thick + sat_snow;
      CL(m_snow) = CL(m_snow) - sat_snow * (pss * rho_s + (1.0 - pss - gmin) * rho_l) - CL(melt_thick_snow) * rho_l;
      m = m + sat_snow * (pss * rho_s + (1.0 - pss - gmin) * rho_l);
      CL(H_abs_snow) = CL(H_abs_snow) - sat_snow * pss * rho_s * c_s * T_snow;
      H_abs = H_abs + sat_snow * pss * rho_s * c_s * T_snow;
      CL(H_abs_snow) = CL(H_abs_snow) + sat_snow * pss * rho_s * latent_heat;
      H_abs = H_abs - sat_snow * pss * rho_s * latent_heat;
      CL(H_abs_snow) = CL(H_abs_snow) - sat_snow * (1.0 - pss - gmin) * rho_l * c_l * T_snow - CL(melt_thick_snow) * rho_l * c_l * T_snow;
      H_abs = H_abs + sat_snow * (1.0 - pss - gmin) * rho_l * c_l * T_snow;
    }
  } else if (psi_g_sn <= 0.0) {
    touched = true;
    H_abs = H_abs + CL(H_abs_snow); m = m + CL(m_snow); thick = thick + CL(thick_snow);
    CL(H_abs_snow) = 0.0; CL(m_snow) = 0.0; CL(thick_snow) = 0.0;
    psi_g_sn = 0.0; CL(psi_s_snow) = 0.0; psi_l_sn = 0.0;
  }
  if (touched) {
    LAY(SAMSIM_A_M, 1) = m;
    LAY(SAMSIM_A_THICK, 1) = thick;
    LAY(SAMSIM_A_H_ABS, 1) = H_abs;
  }
  GS(PSI_L_SNOW) = psi_l_sn;
  GS(PSI_G_SNOW) = psi_g_sn;
  if (psi_g_sn < 0.0) STOPC(9876, 0);
}

// ---------------------------------------------------------------- S1: first thermodynamic sweep, bottom -> top
// mo_grotz.f90:297-307 (S_bu, H, getT chain, S_br, Expulsion mo_thermo_functions.f90:157-187) fused with the
// permeability / Rayleigh-number part of fl_grav_drain (mo_grav_drain.f90:103-136): ray(k) needs only suffix
// quantities over k..N_active, which an upward sweep meets in the right order.
//
// RayScan carries those suffix quantities; s1_layer is the per-layer body shared by
//   - sweep_thermo_expulsion  the full sweep (first step, and after flushing / regridding changed the column),
//   - sweep_up_fused          which runs it for layers N_active..2 of the NEXT step right after the second getT of
//                             this step (same enthalpy, same guess chain => the same T and phi, computed once),
//   - prologue_top_layer      layer 1 of the current step (everything that changes between two steps touches layer 1).
struct RayScan {
  double minp, stp, st;                      // suffix min(perm), sum(thick/perm), sum(thick) over k..Na-1
  double bot, botterm, perm_bot, S_br_bot;   // bottom layer (enters linearly, mo_grav_drain.f90:119-120,128)
  double buoy_s, min_psi_s;                  // SUM(psi_s*thick), MIN(psi_s)
};
__device__ __forceinline__ void ray_scan_init(RayScan &r) {
  r.minp = 1.0e300; r.stp = 0.0; r.st = 0.0; r.bot = 0.0; r.botterm = 0.0; r.perm_bot = 0.0; r.S_br_bot = 0.0;
  r.buoy_s = 0.0; r.min_psi_s = 1.0e300;
}

// Expulsion, mo_thermo_functions.f90:157-187: volume fractions and expelled brine volume of one layer
struct Expelled { double psi_s, psi_l, psi_g, V_ex; };
// (rth = recip(thick): the fused up sweep forms it once per sweep for the two thicknesses of the grid rule; recip() is a function of
// its argument alone, so the bits are the same wherever it is formed)
__device__ __forceinline__ Expelled expulsion(double phi, double thick, double m, double rth) {
  Expelled e;
  const double V_s = m * phi * (1.0 / rho_s), V_l = m * (1.0 - phi) * (1.0 / rho_l);
  e.V_ex = dmax(V_l + V_s - thick, 0.0);   // (a sum above thick leaves a positive difference, one at or below it none)
  e.psi_s = V_s * rth;
  e.psi_l = (V_l - e.V_ex) * rth;
  e.psi_g = (thick - V_l - V_s + e.V_ex) * rth;
  e.psi_l = dmax(e.psi_l, 0.0);
  e.psi_g = dmax(e.psi_g, 0.0);
  return e;
}

// Permeability + Rayleigh number of layer k from its T, phi (Expulsion evaluated in registers).  Only PHI (by the caller)
// and ray are stored: the down sweep re-evaluates Expulsion from PHI, m and thick (same inputs, same operations) and
// writes the psi arrays itself, which is cheaper than handing psi_s, psi_l, psi_g and V_ex over through HBM.
template <class K>
__device__ __forceinline__ void s1_layer(Col &c, const Ctx &x, int k, int Na, bool do_ray, double T, double phi, double S_bu,
                                         double m, double thick, double rth, RayScan &r, bool sparse_rows = false) {
  const samsim_config &g = x.p->cfg;
  const double S_br = S_br_clamped(x.salt, T, S_bu);
  const Expelled e = expulsion(phi, thick, m, rth);
  r.min_psi_s = dmin(r.min_psi_s, e.psi_s);
  r.buoy_s += e.psi_s * thick;
  if (k == 1) c.psi_l_top = e.psi_l;
  if (do_ray) {
    const double perm = x.p17 * pow_3p1(1000.0 * fabs(e.psi_l));  // mo_grav_drain.f90:105
    if (k == Na) {
      r.S_br_bot = S_br;
      r.bot = thick * e.psi_s / psi_s_min;
      r.perm_bot = perm;
      r.botterm = r.bot / perm;
    } else {
      const double height = r.st + r.bot;  // thick(k+1..Na-1) + bottom part
      r.minp = dmin(r.minp, perm);
      r.stp = r.stp + quot(thick, perm);
      r.st = r.st + thick;
      double ray;
      const double d_S_br = S_br - r.S_br_bot;
      if (CFG(harmonic_flag) == 2) {
        const double hp = (r.minp < x.p14) ? 0.0 : quot(r.st + r.bot, r.stp + r.botterm);
        ray = grav_f * rho_l * bbeta * d_S_br * height * hp;
      } else {
        ray = grav_f * rho_l * bbeta * d_S_br * height * dmin(r.minp, r.perm_bot);
      }
      ray = ray * (1.0 / (kappa_l * mu));
      ray = dmax(ray, 0.0);
      if (!sparse_rows) {
        LAYU(SAMSIM_A_RAY, k) = ray;
      } else if (k == 1 || x.ray_rows_all || wave_any(ray > ray_crit)) {  // wave-uniform k, see Ctx::rflag (row 1 always: ray_row_valid)
        LAYU(SAMSIM_A_RAY, k) = ray;
        // (every executing lane reads the word, sets the same bit and writes the same value back -- two LDS instructions in
        // lock-step, no leader to elect: the lane number a leader test compares with was one more value carried through the loop)
        x.rflag[(k - 1) >> 6] = x.rflag[(k - 1) >> 6] | (1ull << ((k - 1) & 63));
      }
    }
  }
}

// What flooding needs of the whole column (mo_flood.f90:66-80: the harmonic-mean permeability SUM(thick) / SUM(thick/perm) with the
// bottom layer's solid part, and the total thickness) is what the first sweep's scan holds once layer 1 is in: instead of walking
// the column twice more (flood, then refresh_ray_top after flooding has changed thick(1); a power per layer each), the sweep leaves
// the two numbers in the hand-over block where the snow load makes flooding possible -- and the scan WITHOUT layer 1, which
// refresh_ray_top completes with the flooded top layer.  (sums bottom -> top where the reference's run top -> bottom: round-off)
template <class K>
__device__ __forceinline__ void flood_handover(Col &c, const Ctx &x, const RayScan &all, const RayScan &below_top, double thick_bottom) {
  const samsim_config &g = x.p->cfg;
  if (!(CFG(flood_flag) > 1 && c.Na > 1 && CL(m_snow) > all.buoy_s * (rho_l - rho_s))) return;   // (= flood_possible of column_step)
  SPEC(SP_FL_HP) = quot(all.st + all.bot, all.stp + all.botterm);
  SPEC(SP_FL_SALL) = all.st + thick_bottom;
  SPEC(SP_MINP) = below_top.minp; SPEC(SP_STP) = below_top.stp; SPEC(SP_ST) = below_top.st;
  SPEC(SP_BOT) = below_top.bot; SPEC(SP_BOTTERM) = below_top.botterm; SPEC(SP_SBR_BOT) = below_top.S_br_bot;
}

// all_phi: the solid fractions of every layer go to their array (an output point follows); otherwise only where something reads them
// before the up sweep rewrites them (layer 1, the bottom two layers: thin-snow coupling, regrid trigger).  whole_wave: every column
// of the wave runs this sweep (the normal state of a melt season, when every column flushes in every step): then the Rayleigh rows
// are stored and flagged like the fused up sweep's -- only where some column drains -- instead of all of them.
template <class K>
__device__ RARE void sweep_thermo_expulsion(Col &c, const Ctx &x, bool all_phi, bool whole_wave) {
  const samsim_config &g = x.p->cfg;
  const Salt &s = x.salt;
  const int Na = c.Na;
  const bool do_ray = (CFG(grav_flag) >= 2 && Na > 1);
  double T_test = g.T_bottom;
  RayScan r, r_below_top;
  ray_scan_init(r);
  r_below_top = r;
  double thick_bottom = 0.0;
  int rc = 0, rc_layer = 0;
  if (do_ray && Na <= c.N - 1 && (!whole_wave || x.ray_rows_all)) LAYU(SAMSIM_A_RAY, Na) = 0.0;
  if (whole_wave) {
    for (int w = 0; w <= (c.N - 1) >> 6; ++w) x.rflag[w] = 0ull;   // every lane writes the same zeros
    __builtin_amdgcn_wave_barrier();
  }
  // operands requested two layers ahead of the arithmetic, unconditionally and from a clamped row, as in sweep_up_fused.
  // The thickness rule (COLF_REGULAR) is checked against the array after samsim_set_state and after a regrid; in between -- the
  // steps of a melt season, which take this sweep because flush3 rewrites every layer -- nothing touches the thicknesses below
  // layer 1 and a regular column's come from the rule.  Decided per wave, so that the loop's requests stay unconditional.
  struct L4 { double H, m, th, S; };
  bool regular = true;
  const double th_mid_rule = LAYU(SAMSIM_A_THICK, g.n_top + 1);
  const bool check_col = (c.flags & COLF_REGULAR) == 0 || (c.flags & (COLF_RESTART | COLF_REGRID)) != 0;
  const bool check_wave = wave_any(check_col);
  const int kmax = wave_max(Na);
  auto run = [&](auto check_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
    auto ld = [&](int j) -> L4 {
      L4 r;
      r.H = LAYU(SAMSIM_A_H_ABS, j); r.m = LAYU(SAMSIM_A_M, j); r.S = LAYU(SAMSIM_A_S_ABS, j);
      r.th = (CHECK || j < 2) ? LAYU(SAMSIM_A_THICK, j) : thick_by_rule(j, g.n_top, g.n_middle, th_mid_rule, g.thick_0);
      return r;
    };
    L4 cur = ld(Na), nxt = ld(Na >= 2 ? Na - 1 : 1), nn = nxt;
    for (int k = kmax; k >= 1; --k) {
      if (k > Na) continue;
      nn = ld(k >= 3 ? k - 2 : 1);
      const double H_abs = cur.H, m = cur.m, thick = cur.th;
      if (CHECK && k >= 2 && thick != thick_by_rule(k, g.n_top, g.n_middle, th_mid_rule, g.thick_0)) regular = false;
      double S_abs = cur.S;
      cur = nxt; nxt = nn;
      if (S_abs < 0.0) {  // health check of the previous step, mo_grotz.f90:812-818 (element-wise clamp)
        S_abs = 0.0;
        LAYU(SAMSIM_A_S_ABS, k) = S_abs;
      }
      double S_bu, H;
      per_mass(S_abs, H_abs, m, S_bu, H);
      double T, phi = 0.0;
      int rr = getT_chain<true>(s, H, S_bu, T_test, T, phi);
      if (rr && !rc) { rc = rr; rc_layer = k; }
      T_test = T;
      // T and phi are the hand-over to the down sweep; S_bu / S_br are recomputed there from T, S_abs, m
      LAYU(SAMSIM_A_T, k) = T;
      if (all_phi || k == 1 || k >= Na - 1) LAYU(SAMSIM_A_PHI, k) = phi;
      if (k == 1) r_below_top = r;                                   // the scan over layers N_active..2 (flood_handover)
      if (k == Na) thick_bottom = thick;
      s1_layer<K>(c, x, k, Na, do_ray, T, phi, S_bu, m, thick, recip(thick), r, whole_wave);
    }
  };
  if (check_wave) run(std::true_type{}); else run(std::false_type{});
  if (whole_wave) { __builtin_amdgcn_wave_barrier(); c.ray_all = false; }   // the rows that hold a value are the flagged ones
  if (do_ray) flood_handover<K>(c, x, r, r_below_top, thick_bottom);
  c.neg_psi = r.min_psi_s < 0.0;
  c.buoy_s = r.buoy_s;
  c.flags = regular ? (c.flags | COLF_REGULAR) : (c.flags & ~COLF_REGULAR);
  if (rc) STOPC(rc, rc_layer);
}

// Layer 1 of the first sweep when layers N_active..2 were already done by the previous step's up sweep
// (their prognostic values have not changed since).  The scan state comes from the hand-over block.
template <class K>
__device__ __forceinline__ void prologue_top_layer(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const int Na = c.Na;
  const bool do_ray = (CFG(grav_flag) >= 2 && Na > 1);
  RayScan r;
  r.minp = SPEC(SP_MINP); r.stp = SPEC(SP_STP); r.st = SPEC(SP_ST);
  r.bot = SPEC(SP_BOT); r.botterm = SPEC(SP_BOTTERM); r.perm_bot = SPEC(SP_PERM_BOT);
  r.S_br_bot = SPEC(SP_SBR_BOT); r.buoy_s = SPEC(SP_BUOY_S); r.min_psi_s = SPEC(SP_MIN_PSI_S);
  const RayScan r_below_top = r;
  const double H_abs = LAY(SAMSIM_A_H_ABS, 1), m = LAY(SAMSIM_A_M, 1), thick = LAY(SAMSIM_A_THICK, 1);
  double S_abs = LAY(SAMSIM_A_S_ABS, 1);
  if (S_abs < 0.0) { S_abs = 0.0; LAY(SAMSIM_A_S_ABS, 1) = S_abs; }
  double S_bu, H;
  per_mass(S_abs, H_abs, m, S_bu, H);
  if (K::general && CFG(prescribe_flag) == 2) LAY(SAMSIM_A_S_BU, 1) = S_abs / m;  // read back by prescribe_salinity
  const double T_test = (Na > 1) ? LAY(SAMSIM_A_T, 2) : g.T_bottom;
  double T, phi = 0.0;
  const int rc = getT_chain(x.salt, H, S_bu, T_test, T, phi);   // (the wave's columns together, as in the sweeps)
  LAY(SAMSIM_A_T, 1) = T;
  LAY(SAMSIM_A_PHI, 1) = phi;
  s1_layer<K>(c, x, 1, Na, do_ray, T, phi, S_bu, m, thick, recip(thick), r);
  if (do_ray && CFG(flood_flag) > 1 && CL(m_snow) > r.buoy_s * (rho_l - rho_s)) {   // (flood_handover's own test: the thickness of the bottom layer is only formed where it is used)
    THICK_RULE_INIT(tr);
    flood_handover<K>(c, x, r, r_below_top, THICK_AT(tr, Na));
  }
  c.neg_psi = r.min_psi_s < 0.0;
  c.buoy_s = r.buoy_s;
  if (rc) STOPC(rc, 1);
}

// ---------------------------------------------------------------- P2: expulsion_flux + mass_transfer, top -> bottom
// expulsion_flux (mo_mass.f90:112-136): downward brine flux recurrence, m and psi_g update.  mass_transfer
// (mo_mass.f90:53-96) with these fluxes (all <= 0: brine only moves down) needs the layer above only.  Then the
// S_bu refresh of mo_grotz.f90:333-335.  mass_transfer is skipped on the first step (mo_grotz.f90:313).
// DRY: nothing is stored -- the sweep only tells what flooding needs to know before the fused down sweep runs (column_step): the
// gas-filled volume of the column after expulsion_flux (for the freeboard) and the top and bottom layers as brine expulsion and its
// mass_transfer leave them (flooding moves water between exactly these two and the snow).
struct ExpelledEnds {
  double S1, H1, m1, psi_l1, S_br1;    // layer 1 after expulsion + mass_transfer; its liquid fraction and brine salinity of the first sweep
  double SN, HN, mN, TN, psi_gN;       // layer N_active likewise (before the gas -> ocean water replacement)
  double buoy_g;                       // SUM(psi_g*thick) after expulsion_flux
};
template <class K, bool DRY = false>
__device__ RARE void sweep_expulsion_transfer(Col &c, const Ctx &x, ExpelledEnds *ends = nullptr) {
  const int Na = c.Na;
  const bool transfer = (c.step + 1 != 1);
  double flm_k = 0.0;  // fl_m(k)
  double buoy_g = 0.0;
  double fb_a2 = 0.0, fb_g2 = 0.0;   // SUM(psi_s*thick), SUM(psi_g*thick) over layers >= 2 for func_freeboard
  double T_up = 0.0, S_br_up = 0.0, S_abs_up = 0.0;  // layer k-1: snapshot T, S_br, UPDATED S_abs
  // rows are requested a chunk at a time (see RARE_CHUNK)
  constexpr int CH = RARE_CHUNK / 2;
  THICK_RULE_INIT(tr);
  for (int k0 = 1; k0 <= Na; k0 += CH) {
    double m_[CH], th_[CH], T_[CH], H_[CH], S_[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int kk = (k0 + i <= c.N) ? k0 + i : c.N;
      m_[i] = LAY(SAMSIM_A_M, kk); th_[i] = THICK_AT(tr, kk); T_[i] = LAY(SAMSIM_A_T, kk);
      H_[i] = LAY(SAMSIM_A_H_ABS, kk); S_[i] = LAY(SAMSIM_A_S_ABS, kk);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
    const int k = k0 + i;
    if (k <= Na) {
    double m = m_[i];
    const double thick = th_[i];
    // Expulsion of the first sweep (mo_grotz.f90:306), re-evaluated from its inputs phi, thick, m
    const double T = T_[i], H_abs_in = H_[i];
    double S_abs = S_[i];
    double S_bu_in, H_in;
    per_mass(S_abs, H_abs_in, m, S_bu_in, H_in);
    // S_br(k) of the first sweep = func_S_br(T, S_abs/m) with the mass BEFORE expulsion_flux: recomputed bit for bit
    // (same inputs, same operations) instead of being stored by every S1 sweep; this unfused path keeps it for P3
    const double S_br = S_br_clamped(x.salt, T, S_bu_in);
    const Expelled ex = expulsion(phi_from_T(x.salt, H_in, S_bu_in, S_br), thick, m, recip(thick));
    const double V_ex = ex.V_ex;
    double psi_g = ex.psi_g;
    double flm_next;
    if (k == 1 || psi_g < (double)0.001f) {
      flm_next = (k == 1) ? -V_ex * rho_l : -V_ex * rho_l + flm_k;
    } else {
      flm_next = -dmax((V_ex - psi_g * thick) * rho_l, 0.0);
      psi_g = dmax((psi_g * thick - V_ex) / thick, 0.0);
    }
    if (psi_g > 0.0) buoy_g += psi_g * thick;
    if (k >= 2) { fb_a2 += ex.psi_s * thick; fb_g2 += psi_g * thick; }
    if (!DRY) {
      LAY(SAMSIM_A_PSI_S, k) = ex.psi_s;
      LAY(SAMSIM_A_PSI_L, k) = ex.psi_l;
      LAY(SAMSIM_A_PSI_G, k) = psi_g;
    }
    m = m + flm_next - flm_k;
    if (!DRY) {
      LAY(SAMSIM_A_M, k) = m;
      if (HAS_BGC) BFL(BFL_E, k) = transfer ? -flm_next : 0.0;
      LAY(SAMSIM_A_S_BR, k) = S_br;
    }
    double H_abs = H_abs_in;
    if (transfer) {
      bool ch = false;
      if (flm_next < 0.0) {
        H_abs = H_abs + flm_next * T * c_l;
        S_abs = S_abs + dmax(flm_next * S_br, -S_abs);
        ch = true;
      }
      if (flm_k < 0.0) {
        H_abs = H_abs - flm_k * T_up * c_l;
        S_abs = S_abs - dmax(flm_k * S_br_up, -S_abs_up);
        ch = true;
      }
      if (ch && !DRY) {
        LAY(SAMSIM_A_H_ABS, k) = H_abs;
        LAY(SAMSIM_A_S_ABS, k) = S_abs;
      }
    }
    if (!DRY) LAY(SAMSIM_A_S_BU, k) = S_abs / m;
    if (DRY) {
      if (k == 1) { ends->S1 = S_abs; ends->H1 = H_abs; ends->m1 = m; ends->psi_l1 = ex.psi_l; ends->S_br1 = S_br; }
      if (k == Na) { ends->SN = S_abs; ends->HN = H_abs; ends->mN = m; ends->TN = T; ends->psi_gN = psi_g; }
    }
    T_up = T; S_br_up = S_br; S_abs_up = S_abs;
    flm_k = flm_next;
    }
    }
  }
  if (DRY) { ends->buoy_g = buoy_g; return; }
  c.buoy_g = buoy_g;
  SPEC(SP_FB_A2) = fb_a2; SPEC(SP_FB_G2) = fb_g2;
}

// ---------------------------------------------------------------- vital signs, mo_grotz.f90:192-223 (output only)
template <class K>
__device__ RARE void vital_signs(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const int Na = c.Na;
  // (evaluated at the top of the loop with the PREVIOUS step's volume fractions, as in the reference: the psi arrays
  // are rewritten by this step's down sweep only later)
  double sH = 0.0, sm = 0.0, sS = 0.0, resist = 0.0, sth = 0.0, sS1 = 0.0, sm1 = 0.0;
  for (int k = 1; k <= Na; ++k) {
    const double H_abs = LAY(SAMSIM_A_H_ABS, k), m = LAY(SAMSIM_A_M, k), S_abs = LAY(SAMSIM_A_S_ABS, k);
    sH += H_abs; sm += m; sS += S_abs;
    if (k <= Na - 1) {
      const double thick = LAY(SAMSIM_A_THICK, k);
      resist = resist + thick / (LAY(SAMSIM_A_PSI_L, k) * k_l + LAY(SAMSIM_A_PSI_S, k) * k_s);
      sth += thick; sS1 += S_abs; sm1 += m;
    }
  }
  const double thN = LAY(SAMSIM_A_THICK, Na), psN = LAY(SAMSIM_A_PSI_S, Na);
  c.energy_stored = CL(H_abs_snow) + sH - g.T_bottom * sm * c_l;
  c.freshwater = sm / rho_l;
  c.freshwater = c.freshwater * (1.0 - sS / sm / ref_salinity);
  c.freshwater = c.freshwater + CL(m_snow) / rho_l;
  resist = resist + thN * psN / psi_s_min * (psi_s_min * k_s + 1.0 - psi_s_min * k_l);
  if (CL(thick_snow) > g.thick_min / 110.0) resist = resist + quot(CL(thick_snow), func_k_snow(CL(m_snow), CL(thick_snow)));
  c.total_resist = resist;
  c.thickness = ((Na > 1) ? sth : 0.0) + thN * psN / psi_s_min;
  if (Na > 1) {
    const double SN = LAY(SAMSIM_A_S_ABS, Na), mN = LAY(SAMSIM_A_M, Na);
    c.bulk_salin = (sS1 + SN * psN / psi_s_min) / (sm1 + mN * psN / psi_s_min);
  } else {
    c.bulk_salin = LAY(SAMSIM_A_S_ABS, 1) / LAY(SAMSIM_A_M, 1);
  }
}

// ---------------------------------------------------------------- flood, mo_flood.f90:55-151
// The arithmetic of flood on the two layers it touches, held in registers: layer 1 (S1, H1, m1, th1) and layer N_active (SN, HN, mN,
// TN: read; its increments incS, incH are returned, applied where `deep` -- the instant flooding below neg_free), and the snow
// (in LDS).  flood() below runs it on the arrays (the unfused order); the fused order on what its dry run of the expulsion returned.
struct FloodEnds { double S1, H1, m1, th1, SN, HN, mN, TN, incS, incH; bool deep; };
template <class K>
__device__ __forceinline__ double flood_core(Col &c, const Ctx &x, double hp, double sall, FloodEnds &e) {
  const samsim_config &g = x.p->cfg;
  const double freeboard = GS(FREEBOARD), psi_g_snow = GS(PSI_G_SNOW);
  double flood_brine = -g.dt * grav_f * rho_l * rho_l * hp * (freeboard) / (mu * sall);
  const double shift_ice = flood_brine / (rho_l * psi_g_snow / ratio_flood);
  const double shift_snow = shift_ice * (1 + psi_g_snow / (1.0 - psi_g_snow) * (1.0 - 1.0 / ratio_flood));
  double S1 = e.S1, H1 = e.H1, m1 = e.m1, th1 = e.th1;
  const double SN = e.SN, HN = e.HN, mN = e.mN, TN = e.TN;
  const double S_buN = SN / mN;

  S1 = S1 + flood_brine * S_buN;
  H1 = H1 + flood_brine * HN / mN;
  m1 = m1 + flood_brine;
  th1 = th1 + shift_ice;
  H1 = H1 + shift_snow / CL(thick_snow) * CL(H_abs_snow);
  CL(H_abs_snow) = CL(H_abs_snow) - shift_snow / CL(thick_snow) * CL(H_abs_snow);
  m1 = m1 + shift_snow / CL(thick_snow) * CL(m_snow);
  CL(m_snow) = CL(m_snow) - shift_snow / CL(thick_snow) * CL(m_snow);
  CL(thick_snow) = CL(thick_snow) - shift_snow;

  e.deep = freeboard + shift_ice < neg_free;
  e.incS = 0.0; e.incH = 0.0;
  if (e.deep) {
    const double shift = neg_free - (freeboard + shift_ice);
    flood_brine = shift * (psi_g_snow) * rho_l;
    e.incS = (x.S_bu_bottom - S_buN) * flood_brine;
    e.incH = (g.T_bottom - TN) * c_l * flood_brine;
    S1 = S1 + S_buN * flood_brine;
    H1 = H1 + TN * c_l * flood_brine;
    m1 = m1 + flood_brine;
    th1 = th1 + shift;
    H1 = H1 + shift / CL(thick_snow) * CL(H_abs_snow);
    CL(H_abs_snow) = CL(H_abs_snow) - shift / CL(thick_snow) * CL(H_abs_snow);
    m1 = m1 + shift / CL(thick_snow) * CL(m_snow);
    CL(m_snow) = CL(m_snow) - shift / CL(thick_snow) * CL(m_snow);
    CL(thick_snow) = CL(thick_snow) - shift;
  }
  e.S1 = S1; e.H1 = H1; e.m1 = m1; e.th1 = th1;
  return flood_brine;
}

template <class K>
__device__ RARE void flood(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const int Na = c.Na;
  // harmonic-mean permeability of the column and its total thickness: from the first sweep of this step (flood_handover); without
  // Rayleigh-number drainage (grav_flag 1: no scan) the column is walked here
  double hp, sall;
  if (CFG(grav_flag) >= 2) {
    hp = SPEC(SP_FL_HP);
    sall = SPEC(SP_FL_SALL);
  } else {
    double sth = 0.0;
    hp = 0.0;
    for (int k = 1; k <= Na - 1; ++k) {
      const double thick = LAY(SAMSIM_A_THICK, k);
      const double perm = x.p17 * pow_3p1(1000.0 * LAY(SAMSIM_A_PSI_L, k));
      hp = hp + thick / perm;
      sth += thick;
    }
    const double thN = LAY(SAMSIM_A_THICK, Na), psN = LAY(SAMSIM_A_PSI_S, Na);
    const double permN = x.p17 * pow_3p1(1000.0 * LAY(SAMSIM_A_PSI_L, Na));
    hp = hp + (thN * psN / psi_s_min) / permN;
    hp = (sth + thN * psN / psi_s_min) / hp;
    sall = sth + thN;
  }
  FloodEnds e;
  e.S1 = LAY(SAMSIM_A_S_ABS, 1); e.H1 = LAY(SAMSIM_A_H_ABS, 1); e.m1 = LAY(SAMSIM_A_M, 1); e.th1 = LAY(SAMSIM_A_THICK, 1);
  e.SN = LAY(SAMSIM_A_S_ABS, Na); e.HN = LAY(SAMSIM_A_H_ABS, Na); e.mN = LAY(SAMSIM_A_M, Na); e.TN = LAY(SAMSIM_A_T, Na);
  c.bgc_flood = flood_core<K>(c, x, hp, sall, e);
  if (e.deep) {
    LAY(SAMSIM_A_S_ABS, Na) = e.SN + e.incS;
    LAY(SAMSIM_A_H_ABS, Na) = e.HN + e.incH;
  }
  LAY(SAMSIM_A_S_ABS, 1) = e.S1;
  LAY(SAMSIM_A_H_ABS, 1) = e.H1;
  LAY(SAMSIM_A_M, 1) = e.m1;
  LAY(SAMSIM_A_THICK, 1) = e.th1;
}

// ---------------------------------------------------------------- flood_simple, mo_flood.f90:167-210 (flood_flag 3)
template <class K>
__device__ RARE void flood_simple(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const double shift = GS(FREEBOARD) - neg_free;
  const double flood_brine = -shift * GS(PSI_G_SNOW) * rho_l;
  double S1 = LAY(SAMSIM_A_S_ABS, 1), H1 = LAY(SAMSIM_A_H_ABS, 1), m1 = LAY(SAMSIM_A_M, 1), th1 = LAY(SAMSIM_A_THICK, 1);
  th1 = th1 - shift;
  S1 = S1 + x.S_bu_bottom * flood_brine;
  H1 = H1 - shift / CL(thick_snow) * CL(H_abs_snow);
  H1 = H1 + g.T_bottom * c_l * flood_brine;
  m1 = m1 - shift / CL(thick_snow) * CL(m_snow);
  m1 = m1 + flood_brine;
  CL(H_abs_snow) = CL(H_abs_snow) + shift / CL(thick_snow) * CL(H_abs_snow);
  CL(m_snow) = CL(m_snow) + shift / CL(thick_snow) * CL(m_snow);
  CL(thick_snow) = CL(thick_snow) + shift;
  LAY(SAMSIM_A_S_ABS, 1) = S1;
  LAY(SAMSIM_A_H_ABS, 1) = H1;
  LAY(SAMSIM_A_M, 1) = m1;
  LAY(SAMSIM_A_THICK, 1) = th1;
}

// recompute ray(1) after flood changed thick(1) (thick(1) enters only the k = 1 harmonic mean)
// (thick: the flooded thick(1); psi_l, S_br: the top layer's liquid fraction and brine salinity of this step's first sweep)
template <class K>
__device__ RARE void refresh_ray_top(Col &c, const Ctx &x, double thick, double psi_l, double S_br) {
  const samsim_config &g = x.p->cfg;
  if (CFG(harmonic_flag) != 2) return;  // MINVAL variant does not depend on thick(1)
  // the scan over layers N_active..2 as the first sweep left it (flood_handover), completed with the flooded top layer exactly as
  // s1_layer completes it
  const double perm = x.p17 * pow_3p1(1000.0 * fabs(psi_l));
  const double st2 = SPEC(SP_ST), bot = SPEC(SP_BOT);
  const double height = st2 + bot;
  const double minp = dmin(SPEC(SP_MINP), perm);
  const double stp = SPEC(SP_STP) + quot(thick, perm);
  const double st = st2 + thick;
  const double hp = (minp < x.p14) ? 0.0 : quot(st + bot, stp + SPEC(SP_BOTTERM));
  double ray = grav_f * rho_l * bbeta * (S_br - SPEC(SP_SBR_BOT)) * height * hp;
  ray = ray * (1.0 / (kappa_l * mu));
  LAY(SAMSIM_A_RAY, 1) = dmax(ray, 0.0);
}

// ---------------------------------------------------------------- P3: gravity drainage, top -> bottom
// fl_grav_drain (mo_grav_drain.f90:138-200) with ray(k) from S1: drainage flux of layer k leaves straight to the
// ocean, the compensating upward flow fl_up passes through every layer below (running sum), then mass_transfer
// (mo_mass.f90:53-96) with fl_m(k+1) = fl_up(k) >= 0.  mass_transfer reads the salt of the layer BELOW after the
// drainage loop (snapshot SS_abs), so layer k+1 is drained one iteration ahead of the transfer into layer k.
// The same pass multiplies up the Beer-law transmittance for fl_rad(N_active) (mo_heat_fluxes.f90:151-155).
template <class K>
__device__ RARE void sweep_grav_drain(Col &c, const Ctx &x, bool do_beer, double beer0) {
  const samsim_config &g = x.p->cfg;
  const Salt &s = x.salt;
  const int Na = c.Na;
  const double dt = g.dt;
  double heat_loss = 0.0, cum = 0.0, sum_before = 0.0, sum_after = 0.0, minS = 1.0e300;
  // Beer law: temp2 decays layer by layer; exp() is re-evaluated only when the thickness changes
  double temp2 = beer0, e = 0.0, th_prev = -1.0;
  int stop_layer = 0;

  struct L { double T, S_bu, S_abs, H_abs, flup, fdown; bool ch; };
  struct Ops { double T, S_bu, S_abs, H_abs, thick, S_br, S_br_below; };

  // drain(j): gravity-drainage loss of layer j (mo_grav_drain.f90:144-170) and fl_up(j)
  auto drain = [&](int j, const Ops &o) -> L {
    L r;
    r.T = o.T;
    r.S_bu = o.S_bu;
    r.S_abs = o.S_abs;
    r.H_abs = o.H_abs;
    r.ch = false;
    r.fdown = 0.0;
    const double thick = o.thick;
    if (do_beer) {
      if (thick != th_prev) { e = exp(-extinc * thick); th_prev = thick; }
      if (j == Na) c.frad = temp2 - temp2 * e;
      temp2 = temp2 * e;
    }
    sum_before += r.S_abs;
    r.flup = cum;
    if (j <= Na - 1) {
      const double S_br = o.S_br;
      const double ray = ray_row_valid(c, x, j) ? LAY(SAMSIM_A_RAY, j) : 0.0;
      if (ray > ray_crit && S_br > o.S_br_below) {
        const double psi_s = LAY(SAMSIM_A_PSI_S, j), m = LAY(SAMSIM_A_M, j);
        if (psi_s > 0.001 && r.S_abs / m > 0.1) {
          const double psi_l = LAY(SAMSIM_A_PSI_L, j);
          double flux = x_grav * (ray - ray_crit) * dt * thick;
          flux = dmin(flux, psi_l * rho_l * thick);
          r.S_abs = r.S_abs - flux * S_br;
          if (r.S_abs < 0.0 && !stop_layer) stop_layer = j;
          CL(grav_temp) = CL(grav_temp) + flux * r.T;
          r.H_abs = r.H_abs - flux * c_l * r.T;
          heat_loss = heat_loss + flux * c_l * r.T;
          cum = cum + flux;
          r.flup = dmin(cum, psi_l * rho_l * thick);
          r.fdown = flux;
          r.ch = true;
        }
      }
    }
    sum_after += r.S_abs;
    return r;
  };

  // Layer j is drained, then layer j-1 -- which now knows its neighbour below -- is finished: the reference's order.  The plain
  // operands of a chunk of layers are requested together (see RARE_CHUNK); what only a draining layer reads is loaded there.
  constexpr int CH = RARE_CHUNK / 2;
  const int N = c.N;
  THICK_RULE_INIT(tr);
  L cur = {0, 0, 0, 0, 0, 0, false};
  double flup_prev = 0.0;  // fl_up(k-1) = fl_m(k)
  for (int j0 = 1; j0 <= Na + 1; j0 += CH) {
    double T_[CH], Sbu_[CH], S_[CH], H_[CH], th_[CH], Sbr_[CH + 1];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int kk = (j0 + i <= N) ? j0 + i : N;
      T_[i] = LAY(SAMSIM_A_T, kk); Sbu_[i] = LAY(SAMSIM_A_S_BU, kk); S_[i] = LAY(SAMSIM_A_S_ABS, kk);
      H_[i] = LAY(SAMSIM_A_H_ABS, kk); th_[i] = THICK_AT(tr, kk); Sbr_[i] = LAY(SAMSIM_A_S_BR, kk);
    }
    Sbr_[CH] = LAY(SAMSIM_A_S_BR, (j0 + CH <= N) ? j0 + CH : N);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int j = j0 + i;
      if (j <= Na + 1) {
        L nxt = cur;
        if (j <= Na) nxt = drain(j, Ops{T_[i], Sbu_[i], S_[i], H_[i], th_[i], Sbr_[i], Sbr_[i + 1]});
        if (j >= 2) {
          const int k = j - 1;
          double T_below, S_bu_below, SS_abs_below;
          if (k < Na) {
            T_below = nxt.T; S_bu_below = nxt.S_bu; SS_abs_below = nxt.S_abs;
          } else {
            T_below = g.T_bottom; S_bu_below = x.S_bu_bottom; SS_abs_below = x.S_bu_bottom * 2000.0;
          }
          if (cur.flup > 0.0) {  // fl_m(k+1) > 0: inflow from below
            cur.H_abs = cur.H_abs + cur.flup * T_below * c_l;
            cur.S_abs = cur.S_abs + dmin(cur.flup * S_br_clamped(s, T_below, S_bu_below), SS_abs_below);
            cur.ch = true;
          }
          if (flup_prev > 0.0) {  // fl_m(k) > 0: outflow to the layer above
            cur.H_abs = cur.H_abs - flup_prev * cur.T * c_l;
            cur.S_abs = cur.S_abs - dmin(flup_prev * S_br_clamped(s, cur.T, cur.S_bu), cur.S_abs);
            cur.ch = true;
          }
          if (k == Na) {
            CL(grav_drain) = CL(grav_drain) + cur.flup;
            if (CFG(grav_heat_flag) == 2) { cur.H_abs = cur.H_abs + heat_loss - cur.flup * c_l * g.T_bottom; cur.ch = true; }
          }
          if (cur.ch) {
            LAY(SAMSIM_A_S_ABS, k) = cur.S_abs;
            LAY(SAMSIM_A_H_ABS, k) = cur.H_abs;
          }
          if (HAS_BGC) { BFL(BFL_D, k) = cur.fdown; BFL(BFL_U, k) = cur.flup; }
          minS = dmin(minS, cur.S_abs);
          flup_prev = cur.flup;
        }
        cur = nxt;
      }
    }
  }
  CL(grav_salt) = CL(grav_salt) + sum_before;
  CL(grav_salt) = CL(grav_salt) - sum_after;
  if (stop_layer) STOPC(21234, stop_layer);
  if (minS < 0.0) STOPC(1337, 0);
}

// fl_grav_drain_simple (mo_grav_drain.f90:218-278, grav_flag 3) with ray(k) from S1: every layer above the critical
// Rayleigh number loses 1 % of its salt (`0.99` is a default-REAL literal); fused with the Beer-law pass like P3.
template <class K>
__device__ RARE void sweep_grav_drain_simple(Col &c, const Ctx &x, bool do_beer, double beer0) {
  const int Na = c.Na;
  double temp2 = beer0, e = 0.0, th_prev = -1.0;
  for (int k = 1; k <= Na; ++k) {
    if (do_beer) {
      const double thick = LAY(SAMSIM_A_THICK, k);
      if (thick != th_prev) { e = exp(-extinc * thick); th_prev = thick; }
      if (k == Na) c.frad = temp2 - temp2 * e;
      temp2 = temp2 * e;
    }
    if (k <= Na - 1 && ray_row_valid(c, x, k) && LAY(SAMSIM_A_RAY, k) > ray_crit) LAY(SAMSIM_A_S_ABS, k) = LAY(SAMSIM_A_S_ABS, k) * (double)0.99f;
  }
  CL(grav_drain) = 0.0;
}

// Beer-law absorption alone (no gravity drainage this step): fl_rad(N_active), mo_heat_fluxes.f90:151-155
template <class K>
__device__ RARE void sweep_beer(Col &c, const Ctx &x, double beer0) {
  const samsim_config &g = x.p->cfg;
  const int Na = c.Na;
  double temp2 = beer0, e = 0.0, th_prev = -1.0;
  const bool regular = (c.flags & COLF_REGULAR) != 0;
  const double th_mid = LAYU(SAMSIM_A_THICK, g.n_top + 1);
  for (int k = 1; k <= Na; ++k) {
    const double thick = (regular && k >= 2) ? thick_by_rule(k, g.n_top, g.n_middle, th_mid, g.thick_0) : LAYU(SAMSIM_A_THICK, k);
    if (thick != th_prev) { e = exp(-extinc * thick); th_prev = thick; }
    if (k == Na) c.frad = temp2 - temp2 * e;
    temp2 = temp2 * e;
  }
}

// Conductive update of sub_heat_fluxes (mo_heat_fluxes.f90:272-285) for layers 2..N_active on the unfused path, top -> bottom from
// the arrays (old temperatures, this step's volume fractions, the thickness flooding may just have changed): the fused down sweep
// applies it on the fly, so the up sweep never does.  Layer 1 is left to the top-layer block (fl_Q(1) comes from the surface
// balance); fl_Q(2) and the two energy sums are handed on in the column struct.
// sub_fl_Q, mo_thermo_functions.f90:201-223, between two layers: dT / (thick_a/(2 k_a) + thick_b/(2 k_b)).  With the half-layer
// conductance g = 2k/thick = 2k * (1/thick) -- 1/thick is at hand in the sweeps, one value per stretch of the grid -- the flux is
// dT * g_a*g_b / (g_a + g_b): one division per layer where the resistance form has two (a division is a quarter-rate reciprocal
// plus five instructions).  An ulp-level re-association like the shared reciprocals; both orders of the step use it.
__device__ __forceinline__ double heat_conductance(double psi_s, double psi_l, double rth) {
  return (2.0 * (psi_s * k_s + psi_l * k_l)) * rth;
}
__device__ __forceinline__ double heat_flux_between(double dT, double g_a, double g_b) {
  return quot(dT * (g_a * g_b), g_a + g_b);
}

template <class K>
__device__ RARE void sweep_heat_down(Col &c, const Ctx &x) {
  const int Na = c.Na, N = c.N;
  const double dt = x.p->cfg.dt;
  const double frad_dt = c.frad * dt;
  double esum = 0.0;
  c.flq2 = 0.0;
  if (Na >= 2) {
    double T_up = LAY(SAMSIM_A_T, 1);
    double g_up = heat_conductance(LAY(SAMSIM_A_PSI_S, 1), LAY(SAMSIM_A_PSI_L, 1), recip(LAY(SAMSIM_A_THICK, 1)));
    double flq_k = 0.0;   // fl_Q(k)
    constexpr int CH = RARE_CHUNK / 2;   // five operands per layer (rows requested a chunk at a time, see RARE_CHUNK)
    THICK_RULE_INIT(tr);
    for (int k0 = 2; k0 <= Na; k0 += CH) {
      double T_[CH], th_[CH], ps_[CH], pl_[CH], Hm_[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int kk = (k0 + i <= N) ? k0 + i : N;
        T_[i] = LAY(SAMSIM_A_T, kk); th_[i] = THICK_AT(tr, kk);
        ps_[i] = LAY(SAMSIM_A_PSI_S, kk); pl_[i] = LAY(SAMSIM_A_PSI_L, kk);
        Hm_[i] = LAY(SAMSIM_A_H_ABS, kk - 1);    // layer k-1 (kk >= 2), finished when layer k's flux is known
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = k0 + i;
        if (k <= Na) {
          const double T = T_[i];
          const double gk = heat_conductance(ps_[i], pl_[i], recip(th_[i]));
          const double flq = heat_flux_between(T - T_up, g_up, gk);
          if (k == 2) c.flq2 = flq;
          if (k >= 3) {   // layer k-1: both of its fluxes are known now
            const double H_b = Hm_[i];
            double H_abs = H_b + (flq - flq_k) * dt;
            H_abs = H_abs + frad_dt;
            esum += H_b - H_abs;
            LAY(SAMSIM_A_H_ABS, k - 1) = H_abs;
          }
          T_up = T; g_up = gk; flq_k = flq;
        }
      }
    }
    const double H_b = LAY(SAMSIM_A_H_ABS, Na);   // bottom layer: fl_Q(N_active+1) = fl_q_bottom
    double H_abs = H_b + (c.fl_q_bottom - flq_k) * dt;
    H_abs = H_abs + frad_dt;
    esum += H_b - H_abs;
    LAY(SAMSIM_A_H_ABS, Na) = H_abs;
  }
  c.esum = esum;
}

template <class K>
__device__ __forceinline__ void surface_flux(Col &c, const Ctx &x);
template <class K>
__device__ __forceinline__ double radiation_header(Col &c, const Ctx &x, double time, int tc);

// ---------------------------------------------------------------- D: fused down sweep (P2 + P3), top -> bottom
// One pass instead of two for the common step (not the first, not an output step, no thin-snow coupling, no flooding):
// per layer j   A(j) expulsion_flux + mass_transfer + S_bu refresh          (mo_mass.f90:112-136, 53-96; mo_grotz.f90:333)
//               [j = N_active: gas -> ocean water, bottom turbulence]        (mo_grotz.f90:405-410, 450-457)
//               B(j) gravity-drainage loss of layer j, fl_up(j)              (mo_grav_drain.f90:144-170)
//               C(j-1) return-flow mass_transfer into layer j-1, final store (mo_grav_drain.f90:174-193)
// A(j) of the reference runs for all layers before B starts, but A(j) only reads layers <= j and B/C(j-1) only layers
// j-1, j, so the interleaving computes the same values.  S_br(j) and S_br(j+1) of the first sweep are recomputed from
// T and the pre-expulsion S_abs/m (bit-identical), which needs the raw loads of layer j+1 one iteration early.
template <class K>
// store_default: whether the volume fractions of layers >= 3 are stored when the sweep does not decide itself; decide_psi: it
// decides after layer 2 (see there), storing them anyway under store_default; surface_done: the sweep evaluated the surface balance
// couple: this column has a thin snow cover (snow_coupling, mo_grotz.f90:418-420, between the brine expulsion and the drainage);
// late_rad: some column of the wave has, so the radiation header and the Beer-law pass -- which read the snow temperature the
// coupling sets -- run inside the sweep, after the top two layers (time, tc, do_beer are theirs)
// COLF_FLOODED: this column was flooded before the sweep (column_step, from a dry run of the expulsion): layer 1 takes the flooded
// salt, enthalpy, mass and thickness (hand-over block) where the unfused order's flood() would have changed the arrays -- after its
// expulsion and mass_transfer, before its drainage -- and the bottom layer the increments of an instant flooding (COLF_FLOOD_DEEP)
__device__ __forceinline__ void sweep_down_fused(Col &c, const Ctx &x, bool store_default, bool decide_psi, bool &surface_done,
                                                 bool couple, bool late_rad, double time, int tc, bool do_beer) {
  const samsim_config &g = x.p->cfg;
  const Salt &s = x.salt;
  const int Na = c.Na;
  const double dt = g.dt;
  double heat_loss = 0.0, cum = 0.0, sum_before = 0.0, sum_after = 0.0, minS = 1.0e300;
  double fb_a2 = 0.0, fb_g2 = 0.0;       // SUM(psi_s*thick), SUM(psi_g*thick) over layers >= 2 for func_freeboard (see there)
  int stop_layer = 0;
  bool store_psi = true;                 // layers 1 and 2 always; the others as decided after layer 2 (below)
  // conductive heat fluxes (sub_heat_fluxes, mo_heat_fluxes.f90:272-285): see C(j-1) below
  double g_up = 0.0, flq_up = 0.0;       // half-layer conductance 2k/thick of layer j-1, fl_Q(j-1)
  double esum = 0.0;                     // SUM(H_abs before - after) of the conductive update, for the energy assert

  // The sweeps are latency bound (a wave waits on memory for most of its life), so the loads run ahead of the arithmetic:
  // the six values of layer j+2 are requested at the top of iteration j and first touched in iteration j+1 (S_br of the
  // layer below is needed one layer early), which puts a full iteration of work between request and use.  Measured on the
  // default bench: loads at use 82.3 ms per launch, one layer ahead 76.0, two ahead at 3 waves/SIMD 73.8 (two ahead at
  // 4 waves/SIMD spills inside the loop: 93).
  // Operands run two iterations ahead of the arithmetic with two request buffers and ONE finished layer: the operands of layer
  // j+2 are requested at the top of iteration j and turned into `raw` at the END of iteration j+1.  (Round 1 and the first half of
  // round 2 finished layer j+1 at the top of iteration j, because the drainage test of B(j) compares S_br(j) with S_br(j+1): one
  // iteration of lead, a second finished layer -- 18 registers -- held for the sake of a test that is reached in a fifth of the
  // layers.  That test now forms S_br(j+1) from the request buffer on demand.)
  struct Ld { double T, S_abs, m, H_abs, ray; };
  struct Raw { double T, S_abs, m, S_bu, S_br, H_abs, ray, H; };
  // This sweep only runs on columns that follow the grid rule: the interior layers are walked in three stretches (top block,
  // elastic block, bottom block), inside each of which thick and 1/thick are one value -- the loop body neither loads nor selects
  // them (round 2 formed them per layer from the configuration, which the compiler re-read from memory inside the loop).  The
  // two values are formed where a stretch begins (the elastic block's thickness is one load per sweep), so that nothing but the
  // current pair is carried through the loop.
  auto load_ld = [&](int j) -> Ld {
    Ld r;
    r.T = LAYU_LD(SAMSIM_A_T, j);
    r.S_abs = LAYU_LD(SAMSIM_A_S_ABS, j);
    r.m = LAYU_LD(SAMSIM_A_M, j);
    r.H_abs = LAYU_LD(SAMSIM_A_H_ABS, j);
    // (the row flags are read from LDS at every layer: a word kept across iterations is one more value the allocator spills,
    // and a scratch reload drains every outstanding request of the sweep)
    r.ray = (j <= Na - 1 && ray_row_valid(c, x, j)) ? LAYU(SAMSIM_A_RAY, j) : 0.0;
    return r;
  };
  auto finish = [&](const Ld &l, int j) -> Raw {
    Raw r;
    r.T = l.T; r.S_abs = l.S_abs; r.m = l.m; r.H_abs = l.H_abs; r.ray = l.ray;
    per_mass(r.S_abs, r.H_abs, r.m, r.S_bu, r.H);   // as the first sweep formed them
    r.S_br = S_br_clamped(s, r.T, r.S_bu);
    return r;
  };
  auto S_br_below = [&](const Ld &l) -> double {    // S_br of the layer in a request buffer, exactly as finish() will form it
    double S_bu, H;
    per_mass(l.S_abs, l.H_abs, l.m, S_bu, H);
    return S_br_clamped(s, l.T, S_bu);
  };
  // (SA, mA: salt and mass right after A(j).  Their quotient, the refreshed bulk salinity of mo_grotz.f90:333-335, is only
  // read where brine actually moves -- the drainage test of B(j) and the return-flow transfers of C -- so it is formed there:
  // same operands, same quotient, one division less in the nine layers out of ten that do not drain)
  // (ch: brine moved in or out of the layer -- expulsion, drainage, return flow -- so its mass or salt changed.  In winter that
  // holds in a quarter of the layer rows of a wave; elsewhere m and S_abs would be stored with the bits they were loaded with,
  // and the stores are skipped: 16 of the 88 bytes a layer-cell moves per step.)
  struct Lay { double T, SA, mA, S_abs, H_abs, m, flup; bool ch; };

  double flm_j = 0.0;                                  // fl_m(j) of expulsion_flux
  double T_up = 0.0, S_br_up = 0.0, S_abs_up = 0.0;    // layer j-1 as mass_transfer #1 sees it
  // (requests are issued unconditionally, from a clamped row where the layer does not exist -- see sweep_up_fused)
  const int N = c.N;
  Raw raw = finish(load_ld(1), 1);
  Ld ahead = load_ld(2), ahead2 = ahead;               // layers j+1 and j+2 (nlayer >= 3, samsim_create)
  Lay prev = {0, 0, 1, 0, 0, 0, 0, true};              // layer j-1 after A and B, waiting for C
  double flup_pp = 0.0;                                // fl_up(j-2)
  // One layer of the sweep: A(j), B(j), C(j-1).  LAST = the column's bottom layer N_active, which differs from lane to lane: it
  // runs after the loop (once per wave, every lane with its own j), so that the loop body -- the interior layers -- carries
  // neither the bottom-layer work (gas -> ocean water, the bottom turbulence with its exp and two pow) nor its registers.
  auto layer = [&](const int j, const Ld &below, const double thick, const double rth, auto last_tag, auto first_tag) {   // below: the request buffer that holds layer j+1
    constexpr bool LAST = decltype(last_tag)::value, FIRST = decltype(first_tag)::value;
    if (!LAST && !FIRST) { ISA_MARK("D_LAYER_A"); }
    // ---- A(j)
    // Expulsion of the first sweep (mo_grotz.f90:306), re-evaluated from its inputs phi, thick, m
    double H_abs = raw.H_abs;
    const Expelled ex = expulsion(phi_from_T(s, raw.H, raw.S_bu, raw.S_br), thick, raw.m, rth);
    const double V_ex = ex.V_ex;
    double psi_g = ex.psi_g, m = raw.m, S_abs = raw.S_abs;
    const double T = raw.T, S_br = raw.S_br;
    double flm_next;
    if (j == 1 || psi_g < (double)0.001f) {
      flm_next = (j == 1) ? -V_ex * rho_l : -V_ex * rho_l + flm_j;
    } else {
      flm_next = -dmax((V_ex - psi_g * thick) * rho_l, 0.0);
      psi_g = dmax(quot(psi_g * thick - V_ex, thick), 0.0);
    }
    if (!FIRST) { fb_a2 += ex.psi_s * thick; fb_g2 += psi_g * thick; }
    // The up sweep only needs the layer's half resistance thick/(2k) (sub_fl_Q, mo_thermo_functions.f90:201-223); the three
    // volume fractions are stored when something reads them this step (see column_step), and always for layer 1
    if (store_psi || j == 1) {
      LAYU(SAMSIM_A_PSI_S, j) = ex.psi_s;
      LAYU(SAMSIM_A_PSI_L, j) = ex.psi_l;
      LAYU(SAMSIM_A_PSI_G, j) = psi_g;
    }
    // sub_fl_Q (mo_thermo_functions.f90:201-223): fl_Q(j) = (T(j) - T(j-1)) / (thick(j-1)/(2k(j-1)) + thick(j)/(2k(j))) with the
    // temperatures and volume fractions of the first sweep, k = psi_s*k_s + psi_l*k_l (the reference adds psi_g*0._wp: a no-op),
    // evaluated through the half-layer conductances (heat_flux_between);
    // th_l: the thickness the conduction and the drainage see (flooding changes layer 1's after the expulsion)
    const bool flooded_here = FIRST && (c.flags & COLF_FLOODED) != 0;
    const double th_l = flooded_here ? LAYU(SAMSIM_A_THICK, 1) : thick;
    const double gj = heat_conductance(ex.psi_s, ex.psi_l, flooded_here ? recip(th_l) : rth);
    const double flq = (j >= 2) ? heat_flux_between(T - prev.T, g_up, gj) : 0.0;
    if (j == 2) c.flq2 = flq;
    m = m + flm_next - flm_j;
    if (flm_next < 0.0) {
      H_abs = H_abs + flm_next * T * c_l;
      S_abs = S_abs + dmax(flm_next * S_br, -S_abs);
    }
    if (flm_j < 0.0) {
      H_abs = H_abs - flm_j * T_up * c_l;
      S_abs = S_abs - dmax(flm_j * S_br_up, -S_abs_up);
    }
    bool ch = LAST || (flm_next < 0.0) || (flm_j < 0.0);
#if SAMSIM_STAMPS == 2
    if (!LAST && !FIRST) {   // rows of the interior in which the expulsion moves no brine in any column of the wave (m and S_abs keep their bits)
      ST_COUNT(CT_ROWS, 1);
      if (__ballot(flm_next < 0.0 || flm_j < 0.0) == 0ull) ST_COUNT(CT_ROWS_STILL, 1);
    }
#endif
    const double SA = S_abs, mA = m;     // S_bu = SA / mA: refreshed bulk salinity, mo_grotz.f90:333-335 (formed where it is read)
    T_up = T; S_br_up = S_br; S_abs_up = S_abs;
    flm_j = flm_next;
    // Thin-snow coupling (mo_grotz.f90:418-420) sits between expulsion / mass_transfer and everything below in the reference.  It
    // reads and writes layer 1 only, and layer 1 is through with the expulsion here (its own flux and the one into layer 2 are
    // applied), so it runs now, on the registers: the transfers above moved brine at the temperature of the first sweep, the
    // drainage, the return flow and the conductive flux below see the coupled one -- the unfused order, operation for operation.
    double Tl = T;
    if (FIRST && couple) {
      double phi1 = LAYU(SAMSIM_A_PHI, 1);
      const double S_bu1 = S_abs / m;   // as sweep_expulsion_transfer stores it (mo_grotz.f90:333), and in the array: the second
      LAYU(SAMSIM_A_S_BU, 1) = S_bu1;   // coupling of the step (sub_heat_fluxes, in the up sweep's top-layer block) reads it there
      const int rcc = snow_coupling_core<K>(c, x, H_abs, m, S_bu1, Tl, phi1);
      LAYU(SAMSIM_A_T, 1) = Tl;
      LAYU(SAMSIM_A_PHI, 1) = phi1;
      if (rcc && !c.status) { c.status = rcc; x.err_step[c.col] = c.step + 1; x.err_layer[c.col] = 1; }
    }
    if (flooded_here) {   // flooding (mo_grotz.f90:428-445) sits here in the reference's order: flood() on the finished expulsion
      S_abs = SPEC(SP_FLD_S1); H_abs = SPEC(SP_FLD_H1); m = SPEC(SP_FLD_M1);
      ch = true;
      c.flags &= ~COLF_FLOODED;
    }
    if (LAST) {
      if (psi_g > 0.0) {  // bottom-layer gas -> ocean water
        const double t2 = psi_g * thick * rho_l;
        m = m + t2;
        S_abs = S_abs + t2 * x.S_bu_bottom;
        H_abs = H_abs + t2 * c_l * g.T_bottom;
      }
      if (c.flags & COLF_FLOOD_DEEP) {   // instant flooding below neg_free: ocean water into the bottom layer (mo_flood.f90:118-121)
        S_abs = S_abs + SPEC(SP_FL_HP);
        H_abs = H_abs + SPEC(SP_FL_SALL);
        c.flags &= ~COLF_FLOOD_DEEP;
      }
      if (CFG(turb_flag) == 2) {  // sub_turb_flux
        const double turb = Turb_A * exp(Turb_B * (-ocean_density<K>(x) + func_density(T, quot(S_abs, m)))) * dt;
        S_abs = S_abs - turb * (S_abs / m - x.S_bu_bottom);
      }
    }
    // ---- B(j)
    if (!LAST && !FIRST) { ISA_MARK("D_LAYER_B"); }
    ST_MARK(ST_D_A);
    sum_before += S_abs;
    double flup = cum;
    if (!LAST) {
      const double ray = raw.ray;
      // S_br(j+1) of the first sweep, from the request buffer of layer j+1 (same operands and operations as finish())
      if (ray > ray_crit && S_br > S_br_below(below)) {
        const double psi_s = ex.psi_s;
        if (psi_s > 0.001 && (flooded_here ? quot(S_abs, m) : quot(SA, mA)) > 0.1) {  // S_bu of this layer (j < N_active: nothing but a flooding changed it since A)
          ST_COUNT(CT_DRAIN_WAVE, 1);
          ST_COUNT(CT_DRAIN_LANE, (unsigned long long)__popcll(__ballot(1)));
          const double psi_l = ex.psi_l;
          double flux = x_grav * (ray - ray_crit) * dt * th_l;
          flux = dmin(flux, psi_l * rho_l * th_l);
          S_abs = S_abs - flux * S_br;
          if (S_abs < 0.0 && !stop_layer) stop_layer = j;
          CL(grav_temp) = CL(grav_temp) + flux * Tl;
          H_abs = H_abs - flux * c_l * Tl;
          heat_loss = heat_loss + flux * c_l * Tl;
          cum = cum + flux;
          flup = dmin(cum, psi_l * rho_l * th_l);
          ch = true;
        }
      }
    }
    sum_after += S_abs;
    if (!LAST && !FIRST) { ISA_MARK("D_LAYER_C"); }
    // ---- C(j-1): layer j-1 receives from layer j (fl_m(j) = fl_up(j-1)) and gives to j-2 (fl_m(j-1) = fl_up(j-2))
    if (j > 1) {
      if (prev.flup > 0.0) {
        prev.H_abs = prev.H_abs + prev.flup * T * c_l;
        prev.S_abs = prev.S_abs + dmin(prev.flup * S_br_clamped(s, T, quot(SA, mA)), S_abs);
        prev.ch = true;
      }
      if (flup_pp > 0.0) {
        prev.H_abs = prev.H_abs - flup_pp * prev.T * c_l;
        prev.S_abs = prev.S_abs - dmin(flup_pp * S_br_clamped(s, prev.T, quot(prev.SA, prev.mA)), prev.S_abs);
        prev.ch = true;
      }
      // The brine transports of layer j-1 are complete: what the reference does next to its enthalpy is the explicit conductive
      // update of sub_heat_fluxes, H_abs(k) += (fl_Q(k+1) - fl_Q(k))*dt, then += fl_rad(N_active)*dt (mo_heat_fluxes.f90:277-285:
      // sic, the bottom layer's absorption in every layer).  Both fluxes are at hand here -- old temperatures, this step's volume
      // fractions -- so the down sweep applies it and the up sweep neither reads T and the half resistances nor writes H_abs.
      // Layer 1 takes fl_Q(1) from the surface balance, which needs the finished layer 1: the top-layer block does it.
      if (j - 1 >= 2) {
        const double H_b = prev.H_abs;
        prev.H_abs = prev.H_abs + (flq - flq_up) * dt;
        prev.H_abs = prev.H_abs + c.frad * dt;
        esum += H_b - prev.H_abs;
      }
      if (wave_any(prev.ch)) {   // (wave-uniform: a row is stored for all its columns or for none)
        LAYU(SAMSIM_A_M, j - 1) = prev.m;
        LAYU(SAMSIM_A_S_ABS, j - 1) = prev.S_abs;
      }
      LAYU(SAMSIM_A_H_ABS, j - 1) = prev.H_abs;
      minS = dmin(minS, prev.S_abs);
      flup_pp = prev.flup;
    }
    g_up = gj; flq_up = flq;
    prev.T = Tl; prev.SA = SA; prev.mA = mA; prev.S_abs = S_abs; prev.H_abs = H_abs; prev.m = m; prev.flup = flup; prev.ch = ch;
    if (!LAST && !FIRST) { ISA_MARK("D_LAYER_END"); }
    ST_MARK(ST_D_B);
  };
  const int jmax = wave_max(Na);
  auto request = [&](const int j) { ahead2 = load_ld(j + 2 <= N ? j + 2 : N); };      // top of iteration j: layer j+2
  auto advance = [&](const int j) { raw = finish(ahead, j + 1); ahead = ahead2; };      // end of iteration j: layer j+1 becomes current
  // ---- layers 1 and 2 (where they are interior layers), volume fractions always stored
  const double thick1 = (c.flags & COLF_FLOODED) ? SPEC(SP_FLD_TH1_BEFORE) : LAYU(SAMSIM_A_THICK, 1);   // (the expulsion of layer 1 saw the unflooded thickness)
  if (1 < Na) { request(1); layer(1, ahead, thick1, recip(thick1), std::false_type{}, std::true_type{}); advance(1); }
  if (2 < Na) { request(2); layer(2, ahead, g.thick_0, recip(g.thick_0), std::false_type{}, std::false_type{}); advance(2); }   // (N_top >= 3: samsim_create)
  if (late_rad) {   // (see the head of the routine; nothing above reads fl_rad, the albedo or the short-wave flux)
    const double beer0 = radiation_header<K>(c, x, time, tc);
    c.frad = 0.0;
    if (do_beer) sweep_beer<K>(c, x, beer0);
  }
  // ---- Who reads the psi_s / psi_l / psi_g rows of the layers below?  The vital signs at the next output point and a get_state
  // after the launch (force_psi), and -- when the surface melts or the snow releases melt water -- func_freeboard and flush3
  // (mo_grotz.f90:636,670,717-725).  With N_active >= 3 layer 1 is complete by now (its return-flow transfer C(1) ran with
  // layer 2), and everything those late readers' conditions depend on can be evaluated exactly: the surface balance
  // (sub_heat_fluxes' first part reads layer 1, the snow and the forcing, none of which the rest of this sweep touches), hence
  // T_top, fl_Q(1) and fl_Q_snow; the freezing point of layer 1 (S_abs(1), m(1) stay as they are unless wet snow adds slush);
  // the snow's enthalpy after the heat fluxes, hence whether the second snow_thermo of the step can find it wet.  The rows are
  // skipped only when none of the conditions can hold, so a late reader never meets a column without them (refill_psi_rows is
  // the safety net; tools/melt_ensemble_status.py drives 4 096 columns through a melt season and freeze-up and counts its calls).
  if (decide_psi && Na >= 3) {
    surface_flux<K>(c, x);
    surface_done = true;
    const double thick_min = g.thick_min;
    const double Tf = func_T_freeze(quot(LAYU(SAMSIM_A_S_ABS, 1), LAYU(SAMSIM_A_M, 1)), CFG(salt_flag), x.tf_c3);   // as mo_grotz.f90:634 will
    bool snow_wet = false;
    if (CL(thick_snow) > 0.0) {
      // snow_thermo finds liquid water iff H_abs_snow / m_snow > -latent_heat (getT's fresh branch); the up sweep adds
      // (fl_Q(1) - fl_Q_snow)*dt to a snow cover thicker than thick_min (thinner ones take the unfused path: never here)
      const double H_new = CL(H_abs_snow) + (CL(fl_Q1) - CL(fl_Q_snow)) * dt;
      snow_wet = !(CL(thick_snow) >= thick_min) || !(H_new / CL(m_snow) <= -latent_heat);
    }
    store_psi = store_default || LAYU(SAMSIM_A_PSI_S, 1) < psi_s_top_min || CL(T_top) >= Tf || snow_wet || CL(melt_thick_snow) > 0.0;
  } else {
    store_psi = store_default || decide_psi;   // (a deciding sweep over fewer than three layers has nothing left to skip)
  }
  c.psi_full = store_psi;
  // The interior layers 3 <= j < N_active, two per trip: the two request buffers swap roles from one layer to the next, so with
  // both layers in one loop body no buffer is copied into the other (and the hand-over of layer j to C(j) of the next layer is a
  // renaming): the single-layer loop spent 35 of its 265 vector instructions on those copies.  A stretch with an odd number of
  // layers ends with one single-layer step that does copy its buffer (at most three per sweep).
  {
    const int b0 = g.n_top, b1 = g.n_top + g.n_middle;
    int j = 3;
    for (int stretch = 0; stretch < 3; ++stretch) {
      const int hi_s = stretch == 0 ? b0 : (stretch == 1 ? b1 : N);
      const int hi = hi_s < jmax - 1 ? hi_s : jmax - 1;           // last interior layer of the stretch in the longest column of the wave
      const double th_s = stretch == 1 ? LAYU(SAMSIM_A_THICK, g.n_top + 1) : g.thick_0, rth_s = recip(th_s);
      for (; j + 1 <= hi; j += 2) {
        ISA_MARK("D_ITER_BEGIN");
        ST_MARK(ST_DFUSED);
        if (j + 1 < Na) {                                  // both layers are interior layers of this column: one straight-line body
          ST_COUNT(CT_DOWN_TRIPS, 2);
          ahead2 = load_ld(j + 2 <= N ? j + 2 : N);        // layer j+2 -> second buffer
          layer(j, ahead, th_s, rth_s, std::false_type{}, std::false_type{});
          raw = finish(ahead, j + 1);
          ahead = load_ld(j + 3 <= N ? j + 3 : N);         // layer j+3 -> first buffer
          layer(j + 1, ahead2, th_s, rth_s, std::false_type{}, std::false_type{});
          raw = finish(ahead2, j + 2);
        } else if (j < Na) {                               // layer j is the column's last interior layer
          ST_COUNT(CT_DOWN_TRIPS, 1);
          ahead2 = load_ld(j + 2 <= N ? j + 2 : N);
          layer(j, ahead, th_s, rth_s, std::false_type{}, std::false_type{});
          raw = finish(ahead, j + 1);
        }
        ISA_MARK("D_ITER_END");
      }
      if (j <= hi) {                                       // odd number of layers in this stretch
        if (j < Na) {
          ahead2 = load_ld(j + 2 <= N ? j + 2 : N);
          layer(j, ahead, th_s, rth_s, std::false_type{}, std::false_type{});
          raw = finish(ahead, j + 1);
          ahead = ahead2;
        }
        ++j;
      }
    }
  }
  const double thick_Na = (Na > g.n_top && Na <= g.n_top + g.n_middle) ? LAYU(SAMSIM_A_THICK, g.n_top + 1) : g.thick_0;
  layer(Na, ahead, thick_Na, recip(thick_Na), std::true_type{}, std::false_type{});   // the bottom layer (this sweep only runs with N_active >= 2)
  // ---- C(Na): the ocean below (ghost cell of mass_transfer, mo_mass.f90:70-72)
  if (prev.flup > 0.0) {
    prev.H_abs = prev.H_abs + prev.flup * g.T_bottom * c_l;
    prev.S_abs = prev.S_abs + dmin(prev.flup * S_br_clamped(s, g.T_bottom, x.S_bu_bottom), x.S_bu_bottom * 2000.0);
  }
  if (flup_pp > 0.0) {
    prev.H_abs = prev.H_abs - flup_pp * prev.T * c_l;
    prev.S_abs = prev.S_abs - dmin(flup_pp * S_br_clamped(s, prev.T, quot(prev.SA, prev.mA)), prev.S_abs);
  }
  CL(grav_drain) = CL(grav_drain) + prev.flup;
  if (CFG(grav_heat_flag) == 2) prev.H_abs = prev.H_abs + heat_loss - prev.flup * c_l * g.T_bottom;
  // conductive update of the bottom layer: fl_Q(N_active+1) = fl_q_bottom (this sweep only runs with N_active >= 2)
  {
    const double H_b = prev.H_abs;
    prev.H_abs = prev.H_abs + (c.fl_q_bottom - flq_up) * dt;
    prev.H_abs = prev.H_abs + c.frad * dt;
    c.esum = esum + (H_b - prev.H_abs);
  }
  LAYU(SAMSIM_A_M, Na) = prev.m;
  LAYU(SAMSIM_A_S_ABS, Na) = prev.S_abs;
  LAYU(SAMSIM_A_H_ABS, Na) = prev.H_abs;
  minS = dmin(minS, prev.S_abs);
  if (store_psi) { SPEC(SP_FB_A2) = fb_a2; SPEC(SP_FB_G2) = fb_g2; }   // (read by func_freeboard, which only runs where the rows were stored)
  CL(grav_salt) = CL(grav_salt) + sum_before;
  CL(grav_salt) = CL(grav_salt) - sum_after;
  if (stop_layer) STOPC(21234, stop_layer);
  if (minS < 0.0) STOPC(1337, 0);
}

// ---------------------------------------------------------------- surface energy balance, mo_heat_fluxes.f90:77-195
// sets fl_Q(1), T_top, fl_Q_snow, albedo, fl_sw, fl_lw, T_freeze; returns the Beer-law surface value temp2
// K::general = false: the instantiation for the primary configurations (forcing tables or cooling plate, grav_flag 1/2, flush_flag
// 1/5, flood_flag 1/2, testcases without layer-array specifics); the secondary parametrisations compile away there.
template <class K>
__device__ __forceinline__ double radiation_header(Col &c, const Ctx &x, double time, int tc) {
  const samsim_config &g = x.p->cfg;
  if (CFG(boundflux_flag) != 2) return 0.0;
  CL(albedo) = func_albedo(CL(thick_snow), CL(T_snow), c.psi_l_top, g.thick_min, CFG(albedo_flag));
  if (!K::general || CFG(atmoflux_flag) == 2) {
    if (time == time_input(tc)) {
      CL(fl_sw) = x.f_sw[x.soff + tc - 1];
      CL(fl_lw) = x.f_lw[x.soff + tc - 1];
    } else {
      const double temp = (time - time_input(tc - 1)) / (time_input(tc) - time_input(tc - 1));
      CL(fl_sw) = (1.0 - temp) * x.f_sw[x.soff + tc - 2] + temp * x.f_sw[x.soff + tc - 1];
      CL(fl_lw) = (1.0 - temp) * x.f_lw[x.soff + tc - 2] + temp * x.f_lw[x.soff + tc - 1];
    }
  } else if (CFG(atmoflux_flag) == 1) {
    // sub_notzflux(time + 180 days), mo_functions.f90:270-289 (47.9, 53.1 are default-REAL literals); fl_rest lives in
    // the scalar block (atmoflux_flag 3 leaves fl_sw and fl_rest as the caller set them)
    double day = (time + 86400.0 * 180.0) / 86400.0;
    while (day > 360.0) day = day - 360.0;
    const double a = (day - 164.0) / (double)47.9f, b = (day - 206.0) / (double)53.1f;
    CL(fl_sw) = 314.0 * exp(-0.5 * (a * a));
    if (day < 60.0 || day > 300.0) CL(fl_sw) = 0.0;
    GSI(SAMSIM_S_FL_REST) = 118.0 * exp(-0.5 * (b * b)) + 179.0;
  }
  const double pen = (CL(thick_snow) < g.thick_min) ? penetr : 0.0;
  return pen * (1.0 - CL(albedo)) * CL(fl_sw);
}

// twice-iterated linearised radiative balance for the surface temperature, mo_heat_fluxes.f90:115-148: a function of the
// forcing, the albedo, and the temperature of the snow (or of the top layer under thin / no snow)
__device__ __forceinline__ double radiative_T_top(const Col &c, double fl_rest, double T1, double thick_min) {
  double T_old = (CL(thick_snow) < thick_min) ? T1 : CL(T_snow);
  const double emi = (CL(thick_snow) < thick_min) ? emissivity_ice : emissivity_snow;
  const double pen = (CL(thick_snow) < thick_min) ? penetr : 0.0;
  T_old = T_old + zeroK;
  double temp1 = (1.0 - CL(albedo)) * (1.0 - pen) * CL(fl_sw) + fl_rest;
  temp1 = temp1 + emi * 3.0 * sigma * pow_4(T_old);
  temp1 = quot(temp1, emi * 4.0 * sigma * (T_old * T_old * T_old));
  temp1 = temp1 - zeroK;
  T_old = temp1 + zeroK;
  temp1 = (1.0 - CL(albedo)) * (1.0 - pen) * CL(fl_sw) + fl_rest;
  temp1 = temp1 + emi * 3.0 * sigma * pow_4(T_old);
  temp1 = quot(temp1, emi * 4.0 * sigma * (T_old * T_old * T_old));
  temp1 = temp1 - zeroK;
  return temp1;
}

template <class K>
__device__ __forceinline__ void surface_flux(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const int Na = c.Na;
  const double psi_s1 = LAY(SAMSIM_A_PSI_S, 1), psi_l1 = LAY(SAMSIM_A_PSI_L, 1), psi_g1 = LAY(SAMSIM_A_PSI_G, 1);
  const double thick1 = LAY(SAMSIM_A_THICK, 1), T1 = LAY(SAMSIM_A_T, 1);
  const double k1 = psi_s1 * k_s + psi_l1 * k_l + psi_g1 * 0.0;
  if (CFG(boundflux_flag) == 1) {  // cooling plate, mo_heat_fluxes.f90:77-87
    double fl = (T1 - CL(T_top)) / (thick1 / (2.0 * k1));
    if (fabs(fl) > g.max_flux_plate) fl = fl / fabs(fl) * g.max_flux_plate;
    CL(fl_Q1) = fl;
    return;
  }
  if (K::general && CFG(boundflux_flag) == 3) {  // lab air temperature, mo_heat_fluxes.f90:202-219 (lab_snow_flag 0)
    GS(T_FREEZE) = dmin(func_T_freeze(LAY(SAMSIM_A_S_ABS, Na) / LAY(SAMSIM_A_M, Na), CFG(salt_flag), x.tf_c3), 0.0);
    CL(T_top) = T1;
    CL(fl_Q1) = g.alpha_flux_instable * (CL(T_top) - CL(T2m));
    if (CL(fl_Q1) < 0.0) {
      CL(T_top) = dmax(GS(T_FREEZE), T1);
      CL(fl_Q1) = g.alpha_flux_stable * (CL(T_top) - CL(T2m));
    }
    return;
  }
  // boundflux_flag 2, mo_heat_fluxes.f90:91-195
  const double thick_min = g.thick_min;
  const double fl_rest = (!K::general || CFG(atmoflux_flag) == 2) ? CL(fl_lw) + 0.0 + 0.0 : GSI(SAMSIM_S_FL_REST);
  const double emi = (CL(thick_snow) < thick_min) ? emissivity_ice : emissivity_snow;
  const double pen = (CL(thick_snow) < thick_min) ? penetr : 0.0;
  double temp1;
  CL(T_top) = radiative_T_top(c, fl_rest, T1, thick_min);

  double Tf;
  if (CL(thick_snow) >= thick_min / 100.0) Tf = 0.0;
  else Tf = func_T_freeze(quot(LAY(SAMSIM_A_S_ABS, 1), LAY(SAMSIM_A_M, 1)), CFG(salt_flag), x.tf_c3);

  GS(T_FREEZE) = Tf;

  const double k_snow = (CL(thick_snow) >= thick_min / 100.0) ? func_k_snow(CL(m_snow), CL(thick_snow)) : 0.0;
  // sub_fl_Q_snow, mo_snow.f90:498-518
  const double flq_snow_ice = quot(T1 - CL(T_snow), quot(CL(thick_snow), 2.0 * k_snow) + quot(thick1, 2.0 * (psi_s1 * k_s + psi_l1 * k_l)));
  if (CL(T_top) > Tf && Na > 1) {
    temp1 = emi * sigma * pow_4(Tf + zeroK) - (1.0 - CL(albedo)) * (1.0 - pen) * CL(fl_sw) - fl_rest;
    if (CL(thick_snow) >= thick_min) { CL(fl_Q_snow) = temp1; CL(fl_Q1) = flq_snow_ice; }
    else if (CL(thick_snow) >= thick_min / 100.0) { CL(fl_Q_snow) = temp1; CL(fl_Q1) = 0.0; }
    else CL(fl_Q1) = temp1;
    CL(T_top) = Tf;
  } else {
    if (CL(thick_snow) >= thick_min) {
      CL(fl_Q1) = flq_snow_ice;
      CL(fl_Q_snow) = quot(CL(T_snow) - CL(T_top), quot(CL(thick_snow), 2.0 * k_snow));  // sub_fl_Q_0_snow, mo_snow.f90:528-546
    } else if (CL(thick_snow) > thick_min / 100.0 && CL(thick_snow) < thick_min) {
      CL(fl_Q1) = 0.0;
      // sub_fl_Q_0_snow_thin, mo_snow.f90:466-487
      double k = CL(thick_snow) / (CL(thick_snow) + thick1) * k_snow + thick1 / (CL(thick_snow) + thick1) * k1;
      CL(fl_Q_snow) = (CL(T_snow) - CL(T_top)) / ((CL(thick_snow) + thick1) / (2.0 * k));
    } else {
      CL(fl_Q1) = (T1 - CL(T_top)) / (thick1 / (2.0 * k1));
    }
  }
}

// ---------------------------------------------------------------- U: fused up sweep (P4 + next step's S1), bottom -> top
// sweep_heat_thermo plus, for layers N_active..2, the first sweep of the NEXT time step: that sweep would divide the
// same H_abs by the same m and start Newton from the same guesses (T_bottom, then the layer below), so its T and phi
// are exactly the ones just computed.  What it adds -- S_br, Expulsion, permeability, Rayleigh number -- is done here
// from registers and written to the NEXT psi buffers (nps/npl/npg), because this step's remaining readers (melt film,
// freeboard, flush3) still need the current ones.  Layer 1 is left to prologue_top_layer: snow, melt water and the
// regrid trigger all act on it between the two steps.  If flushing or a regrid changes deeper layers afterwards, the
// column is flagged COLF_DIRTY and the next step runs the full first sweep instead.
template <class K>
__device__ __forceinline__ void sweep_up_fused(Col &c, const Ctx &x, long long col, bool next_is_output, bool store_phi) {
  const samsim_config &g = x.p->cfg;
  const Salt &s = x.salt;
  const int Na = c.Na;
  const double dt = g.dt, thick_min = g.thick_min;
  const bool thin_snow = (CL(thick_snow) >= thick_min / 100.0 && CL(thick_snow) < thick_min);
  const bool do_ray = (CFG(grav_flag) >= 2 && Na > 1);
  const bool keep_ray = next_is_output && col >= x.out_col0 && col < x.out_col0 + x.out_ncols;
  const double H_abs_snow_before = CL(H_abs_snow);
  double esum = c.esum;   // SUM(H_abs before - after the conductive update) over the layers >= 2, from the down sweep
  double T_test = g.T_bottom;
  int rc = 0, rc_layer = 0;
  RayScan r;
  ray_scan_init(r);
  if (keep_ray) {  // `output` prints the Rayleigh numbers of THIS step's fl_grav_drain at the next step's output point
    const size_t oc = (size_t)(col - x.out_col0), on = (size_t)x.out_ncols;
    for (int k = 1; k <= c.N - 1; ++k) x.out_lay[((size_t)SAMSIM_A_RAY * c.N + (k - 1)) * on + oc] = LAYU(SAMSIM_A_RAY, k);
  }
  for (int w = 0; w <= (c.N - 1) >> 6; ++w) x.rflag[w] = 0ull;   // every lane writes the same zeros
  __builtin_amdgcn_wave_barrier();
  if (do_ray && Na <= c.N - 1 && x.ray_rows_all) LAYU(SAMSIM_A_RAY, Na) = 0.0;   // (read by `output` only)
  // The conductive update of layers >= 2 has been applied by the down sweep (sweep_down_fused / sweep_heat_down), which also
  // hands over fl_Q(2) and the energy sums: this sweep reads the finished enthalpy and runs the second getT chain -- and, for
  // layers N_active..2, the first sweep of the next step.  Its operands (H_abs, m, S_abs, thick of a layer) are requested TWO
  // iterations ahead, unconditionally and from a clamped row where the layer does not exist: the hardware counts outstanding
  // memory operations in order, and the compiler can only wait for "all but the N youngest" when every path through the loop
  // body issues the same operations -- one conditional request and it falls back to draining them all.
  // The thicknesses: a wave whose columns all follow the grid rule (COLF_REGULAR: every layer but the first is thick_0, except the
  // N_middle elastic layers, which share thick(N_top+1)) walks the column in three stretches -- bottom block, elastic block, top
  // block -- inside each of which thick and 1/thick are the same for every layer: the loop body neither loads nor selects them.  A
  // wave with a hand-made column loads the array with the other operands and forms 1/thick per layer.
  struct UL { double th, H, m, S; };
  const bool regular_wave = !wave_any((c.flags & COLF_REGULAR) == 0);
  UL cur, nxt, nn;
  bool alive = true, neg_salt = false;
  // One layer of the sweep.  TOP = layer 1, which alone meets the snow (mo_heat_fluxes.f90:291-303) and takes fl_Q(1) from the
  // surface balance: it runs after the loop, so that the loop body -- the same for every other layer -- carries neither the
  // thin-snow coupling (up to 200 getT pairs) nor its registers.
  // LITE: a wave with a column that flushed in the previous step will flush again in this one, after this sweep: flush3 rewrites
  // every layer of that column, so the wave runs the full first sweep in the next step whatever this sweep prepares (the sweep costs
  // a wave the same for one column as for 64) -- it then only runs the second getT chain, and says so for all its columns
  // (COLF_DIRTY: the full first sweep gives a column the same bits as the fused one).  A wave that does not flush after all has lost
  // nothing but the fused first sweep of one step.
  auto body = [&](const int k, const double th_k, const double rth_k, auto top_tag, auto lite_tag) {
    constexpr bool TOP = decltype(top_tag)::value;
    constexpr bool LITE = decltype(lite_tag)::value;
    const double H_k = cur.H, m_k = cur.m, S_k = cur.S;
    double H_abs = H_k;
    const double m = m_k;
    if (TOP) {
      // conductive update of layer 1: fl_Q(2) from the down sweep (fl_q_bottom under a single layer), fl_Q(1) from the surface balance
      const double flq_below = (Na >= 2) ? c.flq2 : c.fl_q_bottom;
      const double H_b = H_abs;
      H_abs = H_abs + (flq_below - CL(fl_Q1)) * dt;
      H_abs = H_abs + c.frad * dt;
      // snow treatment, mo_heat_fluxes.f90:291-303
      if (thin_snow) {
        CL(H_abs_snow) = CL(H_abs_snow) - CL(fl_Q_snow) * dt;
        LAYU(SAMSIM_A_H_ABS, 1) = H_abs;
        snow_coupling<K>(c, x);
        if (c.status) { alive = false; return; }
        H_abs = LAYU(SAMSIM_A_H_ABS, 1);
      } else if (CL(thick_snow) >= thick_min) {
        CL(H_abs_snow) = CL(H_abs_snow) + (CL(fl_Q1) - CL(fl_Q_snow)) * dt;
      }
      esum += H_b - H_abs;   // (after the thin-snow coupling, which moves enthalpy between the snow and layer 1)
      LAYU(SAMSIM_A_H_ABS, 1) = H_abs;
    }
    double S_abs = S_k;
    double S_bu, H;
    per_mass(S_abs, H_abs, m, S_bu, H);
    double T, phi = 0.0;
    if (!TOP) { ISA_MARK("U_GETT_BEGIN"); }
    ST_MARK(ST_U_HEAD);
#if SAMSIM_STAMPS == 2
    int evals = 1;
    int rr = TOP ? getT(s, H, S_bu, T_test, T, phi, &evals) : getT_chain<LITE>(s, H, S_bu, T_test, T, phi, &evals);
    {
      const bool was_odd = (evals >> 30) & 1;
      const int redo = (evals >> 16) & 0x3fff;
      evals &= 0xffff;
      const unsigned long long om = __ballot(was_odd);
      if (om) { ST_COUNT(CT_ODD_LANES, (unsigned long long)__popcll(om)); ST_COUNT(CT_ODD_WAVES, 1); ST_COUNT(CT_ODD_EVALS_WAVE, (unsigned long long)wave_max(redo)); }
    }
    ST_COUNT(CT_UP_TRIPS, 1);
    ST_COUNT(CT_NEWTON_WAVE, (unsigned long long)wave_max(evals));
    { int tot = 0; unsigned long long mk = __ballot(1); while (mk) { const int ln = __ffsll((long long)mk) - 1; tot += __builtin_amdgcn_readlane(evals, ln); mk &= mk - 1; }
      ST_COUNT(CT_NEWTON_LANE, (unsigned long long)tot); }
#else
    int rr = TOP ? getT(s, H, S_bu, T_test, T, phi) : getT_chain<LITE>(s, H, S_bu, T_test, T, phi);
#endif
    if (!TOP) { ISA_MARK("U_GETT_END"); }
    ST_MARK(ST_U_GETT);
    if (rr && !rc) { rc = rr; rc_layer = k; }
    T_test = T;
    LAYU(SAMSIM_A_T, k) = T;
    // the down sweeps recompute phi from T; the array is kept for its readers: the regrid trigger and layer_dynamics (bottom
    // two active layers), layer 1, the output snapshot and get_state
    if (store_phi || TOP || k >= Na - 1) LAYU(SAMSIM_A_PHI, k) = phi;
    if (!TOP && !LITE) {
      // first sweep of the next step for this layer (its own S_abs < 0 clamp first, mo_grotz.f90:812-818)
      // (a clamped salt mass changes S_bu and therefore T: such a column is left to the full sweep, flagged after the loop -- a
      // read-modify-write of the column's flag word inside the loop is one more value for the allocator to spill there)
      neg_salt = neg_salt || (S_abs < 0.0);
      s1_layer<K>(c, x, k, Na, do_ray, T, phi, S_bu, m, th_k, rth_k, r, true);
    }
    ST_MARK(ST_U_TAIL);
  };
  const int kmax = wave_max(Na);
  auto load3 = [&](int j) -> UL { UL u; u.th = 0.0; u.H = LAYU_LD(SAMSIM_A_H_ABS, j); u.m = LAYU_LD(SAMSIM_A_M, j); u.S = LAYU_LD(SAMSIM_A_S_ABS, j); return u; };
  auto load4 = [&](int j) -> UL { UL u = load3(j); u.th = LAYU(SAMSIM_A_THICK, j); return u; };
  auto layers = [&](auto lite_tag) {
  if (regular_wave) {
    cur = load3(Na); nxt = load3(Na >= 2 ? Na - 1 : 1); nn = nxt;   // layers k, k-1, k-2

    const int b0 = g.n_top, b1 = g.n_top + g.n_middle;
    int k = kmax;
    for (int stretch = 0; stretch < 3; ++stretch) {
      const int klo = stretch == 0 ? b1 + 1 : (stretch == 1 ? b0 + 1 : 2);
      // (formed where the stretch begins -- the elastic block's thickness is one load per sweep -- so that only this pair is carried)
      const double th_s = stretch == 1 ? LAYU(SAMSIM_A_THICK, g.n_top + 1) : g.thick_0, rth_s = recip(th_s);
      for (; k >= klo; --k) {
        ISA_MARK("U_ITER_BEGIN");
        ST_MARK(ST_UP);
        if (k > Na) continue;
        nn = load3(k >= 3 ? k - 2 : 1);    // (three layers ahead: no faster, gpurun_out/r3f)
        body(k, th_s, rth_s, std::false_type{}, lite_tag);
        cur = nxt; nxt = nn;
        ISA_MARK("U_ITER_END");
      }
    }
  } else {
    cur = load4(Na); nxt = load4(Na >= 2 ? Na - 1 : 1); nn = nxt;
    for (int k = kmax; k >= 2; --k) {
      if (k > Na) continue;
      nn = load4(k >= 3 ? k - 2 : 1);
      body(k, cur.th, recip(cur.th), std::false_type{}, lite_tag);
      cur = nxt; nxt = nn;
    }
  }
  };
  const bool lite = K::fixed && CFG(flush_flag) == 5 && wave_any((c.flags & COLF_FLUSHED) != 0);
  if (lite) { ST_COUNT(CT_LITE, 1); layers(std::true_type{}); } else layers(std::false_type{});
  __builtin_amdgcn_wave_barrier();   // the row flags are complete: the next readers are the down sweeps of the next step
  if (neg_salt || lite) c.flags |= COLF_DIRTY;
  body(1, LAYU(SAMSIM_A_THICK, 1), 0.0, std::true_type{}, std::false_type{});
  if (!alive) return;
  // hand-over block for prologue_top_layer of the next step
  if (!lite) {
  SPEC(SP_MINP) = r.minp; SPEC(SP_STP) = r.stp; SPEC(SP_ST) = r.st;
  SPEC(SP_BOT) = r.bot; SPEC(SP_BOTTERM) = r.botterm; SPEC(SP_PERM_BOT) = r.perm_bot;
  SPEC(SP_SBR_BOT) = r.S_br_bot; SPEC(SP_BUOY_S) = r.buoy_s; SPEC(SP_MIN_PSI_S) = r.min_psi_s;
  }
  // energy conservation assert, mo_heat_fluxes.f90:265-310: (SUM(H_abs) + H_abs_snow) before + what went in - the same after,
  // with the two sums taken as one sum of per-layer differences
  double bal = esum + (H_abs_snow_before - CL(H_abs_snow));
  bal = bal + (double)Na * (c.frad * dt);
  if (thin_snow || CL(thick_snow) >= thick_min) bal = bal + c.fl_q_bottom * dt - CL(fl_Q_snow) * dt;
  else bal = bal + c.fl_q_bottom * dt - CL(fl_Q1) * dt;
  if (rc) STOPC(rc, rc_layer);
  if (fabs(bal / dt) > 0.00001) STOPC(431, 0);
}

// ---------------------------------------------------------------- melt film, mo_functions.f90:386-474
__device__ __forceinline__ void sub_melt_thick(double psi_l, double psi_s, double psi_g, double T, double T_freeze, double T_top, double fl_Q,
                               double thick_snow, double dt, double &melt_thick, double &thick, double thick_min) {
  melt_thick = 0.0;
  if (thick_snow < thick_min && T_top >= T_freeze) {
    melt_thick = -fl_Q - 2.0 * (psi_l * k_l + psi_s * k_s) / thick * (T_freeze - T);
    melt_thick = melt_thick * dt / dmax(latent_heat * rho_s * psi_s, 0.000000000000001);
    melt_thick = dmin(psi_l * thick, melt_thick);
  }
  if (psi_s < psi_s_top_min) melt_thick = thick * (1.0 - psi_s / psi_s_top_min);
  if (melt_thick > 0.0 && psi_g > gas_snow_ice2) {
    if (melt_thick > (psi_g - gas_snow_ice2) * thick) {
      melt_thick = melt_thick - (psi_g - gas_snow_ice2) * thick;
      thick = thick * (1.0 - (psi_g - gas_snow_ice2));
    } else {
      thick = thick - melt_thick;
      melt_thick = 0.0;
    }
  }
}

// ---------------------------------------------------------------- flush3, mo_flush.f90:70-237
template <class K>
__device__ RARE void flush3(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const Salt &s = x.salt;
  const int Na = c.Na, N = c.N;
  const double dt = g.dt;
  // horizontal flow length = total thickness (mo_flush.f90:104)
  double cnst = 0.0;
  THICK_RULE_INIT(tr);
  for (int k0 = 1; k0 <= Na; k0 += 2 * RARE_CHUNK) {   // (rows requested a chunk at a time, see RARE_CHUNK)
    double th_[2 * RARE_CHUNK];
#pragma unroll
    for (int i = 0; i < 2 * RARE_CHUNK; ++i) { const int kk = (k0 + i <= N) ? k0 + i : N; th_[i] = THICK_AT(tr, kk); }
#pragma unroll
    for (int i = 0; i < 2 * RARE_CHUNK; ++i) if (k0 + i <= Na) cnst += th_[i];
  }
  cnst = cnst * para_flush_horiz;
  const double psi_l1 = LAY(SAMSIM_A_PSI_L, 1), thick1 = LAY(SAMSIM_A_THICK, 1), T1 = LAY(SAMSIM_A_T, 1);
  CL(melt_thick) = dmin(CL(melt_thick), psi_l1 * thick1);
  CL(melt_thick) = dmin(CL(melt_thick), g.thick_0 / 3.0);

  // permeability and bottom -> top equivalent resistance R(k) (stored in the V_ex scratch rows)
  const double pfill = (CFG(snow_flush_flag) == 1) ? 0.0 : 1.0;
  for (int k = Na + 1; k <= N; ++k) LAY(SAMSIM_A_PERM, k) = pfill;
  double R_below = 0.0;  // R(k+1)
  for (int k0 = Na; k0 >= 1; k0 -= RARE_CHUNK) {
    double th_[RARE_CHUNK], pl_[RARE_CHUNK], pg_[RARE_CHUNK];
#pragma unroll
    for (int i = 0; i < RARE_CHUNK; ++i) {
      const int kk = (k0 - i >= 1) ? k0 - i : 1;
      th_[i] = THICK_AT(tr, kk); pl_[i] = LAY(SAMSIM_A_PSI_L, kk);
      pg_[i] = (CFG(snow_flush_flag) == 1) ? LAY(SAMSIM_A_PSI_G, kk) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < RARE_CHUNK; ++i) {
      const int k = k0 - i;
      if (k >= 1) {
        const double thick = th_[i];
        double perm;
        if (CFG(snow_flush_flag) == 1) {
          perm = x.p17 * pow_3p1(1000.0 * fabs(pl_[i] + 2.0 * pg_[i]));
          if (perm == 0.0) perm = 1.0;
        } else {
          perm = x.p17 * pow_3p1(1000.0 * fabs(pl_[i]));
        }
        LAY(SAMSIM_A_PERM, k) = perm;
        const double pm = dmax(perm, 0.00000000000000000000001);
        const double R_v = mu * thick / pm, R_h = mu * cnst / (thick * pm);
        double R;
        if (k == Na) R = 0.0;
        else if (k == Na - 1) R = R_v;
        else { R = R_below + R_v; R = ((R)*R_h) / (R + R_h); }
        LAY(D_V_EX, k) = R;
        R_below = R;
      }
    }
  }
  const double R1 = R_below;
  double flush_total = (GS(FREEBOARD) + CL(melt_thick)) / R1 * grav_f * dt * func_density(T1, S_br_poly(s, T1)) * rho_l;
  flush_total = dmin(flush_total, CL(melt_thick) * rho_l);
  GS(MELT_ERR) = GS(MELT_ERR) + CL(melt_thick) - dmin(flush_total / rho_l, CL(melt_thick));

  // top -> bottom: split into vertical / horizontal parts, vertical mass_transfer (fl_m(k+1) = -flush_v(k) <= 0),
  // horizontal loss of every layer goes to layer N_active
  double fv_up = 0.0;                                  // flush_v(k-1)
  double T_up = 0.0, S_bu_up = 0.0, S_abs_up = 0.0;    // layer k-1: T, local S_bu snapshot, S_abs after the vertical transfer
  double sum_fh = 0.0, accH = 0.0, accS = 0.0, minS = 1.0e300;
  double S_bu_N = 0.0;
  constexpr int CH2 = RARE_CHUNK / 2;   // nine operands per layer
  for (int k0 = 1; k0 <= Na; k0 += CH2) {
    double th_[CH2], pe_[CH2], T_[CH2], m_[CH2], S_[CH2], H_[CH2], Rn_[CH2], fvv_[CH2], fhh_[CH2];
#pragma unroll
    for (int i = 0; i < CH2; ++i) {
      const int kk = (k0 + i <= N) ? k0 + i : N, kn = (kk + 1 <= N) ? kk + 1 : N;
      th_[i] = THICK_AT(tr, kk); pe_[i] = LAY(SAMSIM_A_PERM, kk); T_[i] = LAY(SAMSIM_A_T, kk);
      m_[i] = LAY(SAMSIM_A_M, kk); S_[i] = LAY(SAMSIM_A_S_ABS, kk); H_[i] = LAY(SAMSIM_A_H_ABS, kk);
      Rn_[i] = LAY(D_V_EX, kn); fvv_[i] = LAY(SAMSIM_A_FLUSH_V, kk); fhh_[i] = LAY(SAMSIM_A_FLUSH_H, kk);
    }
#pragma unroll
    for (int i = 0; i < CH2; ++i) {
    const int k = k0 + i;
    if (k <= Na) {
    const double thick = th_[i], perm = pe_[i], T = T_[i];
    double m = m_[i], S_abs = S_[i], H_abs = H_[i];
    const double S_bu = S_abs / m;  // local S_bu of flush3 (mo_flush.f90:101)
    const double pm = dmax(perm, 0.00000000000000000000001);
    const double R_v = mu * thick / pm, R_h = mu * cnst / (thick * pm);
    double fh, fv;
    if (k <= Na - 1) {
      const double Rn = Rn_[i];
      const double src = (k == 1) ? flush_total : fv_up;
      fh = src * (Rn + R_v) / (Rn + R_v + R_h);
      fv = src * R_h / (Rn + R_v + R_h);
    } else {
      fv = fv_up;
      fh = 0.0;
    }
    LAY(SAMSIM_A_FLUSH_V, k) = fvv_[i] + fv;  // accumulated output, mo_grotz.f90:697-737
    LAY(SAMSIM_A_FLUSH_H, k) = fhh_[i] + fh;
    if (HAS_BGC) { BFL(BFL_V, k) = fv; BFL(BFL_H, k) = fh; }
    sum_fh += fh;
    const double flm_next = -fv, flm_k = -fv_up;
    if (flm_next < 0.0) {
      H_abs = H_abs + flm_next * T * c_l;
      S_abs = S_abs + dmax(flm_next * S_br_clamped(s, T, S_bu), -S_abs);
    }
    if (k > 1 && flm_k < 0.0) {
      H_abs = H_abs - flm_k * T_up * c_l;
      S_abs = S_abs - dmax(flm_k * S_br_clamped(s, T_up, S_bu_up), -S_abs_up);
    }
    T_up = T; S_bu_up = S_bu; S_abs_up = S_abs;
    fv_up = fv;
    if (k == Na) {
      S_bu_N = S_bu;
      if (CFG(flush_heat_flag) == 2) H_abs = H_abs - flm_next * T * c_l;
      // horizontal contributions of the layers above, then the loss of all horizontal brine
      H_abs = H_abs + accH;
      S_abs = S_abs + accS;
      const double loss_S = sum_fh * S_bu_N, loss_H = sum_fh * T * c_l;
      if (CFG(flush_heat_flag) == 2) H_abs = H_abs - loss_H;
      S_abs = S_abs - loss_S;
    } else {
      if (k == 1) {
        m = m - flush_total;
        LAY(SAMSIM_A_M, 1) = m;
        LAY(SAMSIM_A_THICK, 1) = thick - flush_total / rho_l;
      }
      const double loss_S = fh * S_br_clamped(s, T, S_abs / m);
      const double loss_H = fh * T * c_l;
      S_abs = S_abs - loss_S;
      H_abs = H_abs - loss_H;
      accH += loss_H;
      accS += loss_S;
    }
    LAY(SAMSIM_A_S_ABS, k) = S_abs;
    LAY(SAMSIM_A_H_ABS, k) = H_abs;
    minS = dmin(minS, S_abs);
    }
    }
  }
  if (minS < -0.00000000000000000000000001) {
    for (int k = 1; k <= Na; ++k) {
      const double v = LAY(SAMSIM_A_S_ABS, k);
      if (v < 0.0) LAY(SAMSIM_A_S_ABS, k) = 0.0;
    }
  }
  if (fabs(LAY(SAMSIM_A_M, 1)) < 0.000001) STOPC(9876, 1);
}

// ---------------------------------------------------------------- layer_dynamics, mo_layer_dynamics.f90:64-716
// ---------------------------------------------------------------- bgc_advection, mo_mass.f90:150-209
// The reference collects the step's brine fluxes in the (N+1)^2 matrix fl_brine_bgc and loops over all of it; at most
// four entries per row are ever set:   (i, i-1) fl_up(i-1)                    return flow of the gravity drainage
//                                      (i, i+1) -fl_m(i+1) + flush_v(i)       expulsion, vertical flushing
//                                      (i, N_active) flush_h(i)               horizontal flushing   [same entry for i = N_active-1]
//                                      (i, N_active+1) fl_down(i) [+ the expulsion part of (N_active-1, N_active), sic]
//                                      (N_active, 1) flood_brine,  (N_active+1, N_active) flood_brine + fl_up(N_active)
// Every flux is upwind (brine concentration of the source layer), limited to a third of the source's content.  One pass
// top -> bottom: what a layer gives to the layer above is added before that layer is stored (one layer of delay), what it
// gives to the layer below / to the bottom layer is carried along.
template <class K>
__device__ RARE void bgc_advection(Col &c, const Ctx &x) {
  const int Na = c.Na;
  for (int t = 0; t < x.n_bgc; ++t) {
    const double bottom = BGC_BOT(t);
    double carry_dn = 0.0, to_bottom = 0.0, to_top = 0.0, pend = 0.0;
    for (int i = 1; i <= Na; ++i) {
      const double q = BGC(t, i);
      const double br = q / dmax(LAY(SAMSIM_A_PSI_L, i) * LAY(SAMSIM_A_THICK, i) * rho_l, 0.000000000000001);
      const double lim = q / 3.0;
      const double E = BFL(BFL_E, i), V = BFL(BFL_V, i);
      double F_up = (i >= 2) ? BFL(BFL_U, i - 1) : 0.0;
      double F_dn = E + V, F_h = (i <= Na - 2) ? BFL(BFL_H, i) : 0.0, F_out = 0.0, F_top = 0.0;
      if (i == Na - 1) F_dn = F_dn + BFL(BFL_H, i);                       // (N_active-1, N_active) holds both
      if (i <= Na - 1) { if (c.bgc_grav) F_out = ((i == Na - 1) ? E : 0.0) + BFL(BFL_D, i); }
      else {                                                               // i = N_active: (i, i+1) leaves the domain
        double sh = 0.0;
        for (int k = 1; k <= Na - 1; ++k) sh += BFL(BFL_H, k);
        F_out = F_dn + sh; F_dn = 0.0;
        if (Na == 2) F_up = F_up + c.bgc_flood; else F_top = c.bgc_flood;  // (N_active, 1)
      }
      const double f_up = dmin(F_up * br, lim), f_dn = dmin(F_dn * br, lim), f_h = dmin(F_h * br, lim);
      const double f_out = dmin(F_out * br, lim), f_top = dmin(F_top * br, lim);
      double temp = q;
      if (i == Na && Na > 2) temp = temp - f_top;
      if (i >= 2) temp = temp - f_up;
      if (i < Na) temp = temp - f_dn;
      if (i <= Na - 2) temp = temp - f_h;
      temp = temp + carry_dn;                                              // from the layer above
      if (i == Na) {
        temp = temp + to_bottom;                                           // horizontal flushing of the layers above
        temp = temp - f_out;
        temp = temp + (c.bgc_flood + BFL(BFL_U, Na)) * bottom;            // (N_active+1, N_active): from the water below
      } else {
        temp = temp - f_out;
      }
      if (i >= 2) BGC(t, i - 1) = pend + f_up;                             // the layer above is complete now
      pend = temp;
      carry_dn = f_dn; to_bottom = to_bottom + f_h; to_top = f_top;
    }
    BGC(t, Na) = pend;
    if (Na > 2 && to_top != 0.0) BGC(t, 1) = BGC(t, 1) + to_top;
  }
  for (int r = 0; r < BFL_NROW; ++r)                                       // fl_brine_bgc = 0, mo_grotz.f90:745
    for (int k = 1; k <= Na; ++k) BFL(r, k) = 0.0;
}

// Regridding with tracers.  In the reference every statement of the regrid routines on S_abs / S_bu / S_bu_bottom has a twin
// on bgc_temp / bgc_bulk / bgc_bottom (mo_layer_dynamics.f90:205-373).  Each routine below therefore takes `tr`: tr < 0 is
// the routine proper; tr >= 0 replays it for tracer tr -- same control flow, the tracer standing in for S_abs, and nothing
// else written (m, H_abs, thick, N_active stay as they are, so every replay and then the proper pass see the old profile).
template <class K>
__device__ __forceinline__ gdouble &salt_at(Col &c, const Ctx &x, int tr, int k) {
  if (K::bgc && tr >= 0) return BGC(tr, k);
  return LAY(SAMSIM_A_S_ABS, k);
}
template <class K>
__device__ __forceinline__ double salt_below(Col &c, const Ctx &x, int tr) {
  if (K::bgc && tr >= 0) return BGC_BOT(tr);
  return x.S_bu_bottom;
}
struct LayerVals { double rho, S_bu, H; };
template <class K>
__device__ __forceinline__ LayerVals layer_vals(Col &c, const Ctx &x, int tr, int k) {
  const double m = LAY(SAMSIM_A_M, k);
  LayerVals v;
  v.rho = m / LAY(SAMSIM_A_THICK, k);
  v.S_bu = salt_at<K>(c, x, tr, k) / m;
  v.H = LAY(SAMSIM_A_H_ABS, k) / m;
  return v;
}
template <class K>
__device__ __forceinline__ void set_layer(Col &c, const Ctx &x, int tr, int k, const LayerVals &v, double thick_0) {
  if (tr < 0) LAY(SAMSIM_A_M, k) = v.rho * thick_0;
  salt_at<K>(c, x, tr, k) = v.S_bu * v.rho * thick_0;
  if (tr < 0) LAY(SAMSIM_A_H_ABS, k) = v.H * v.rho * thick_0;
}
template <class K>
__device__ __forceinline__ void zero_layer(Col &c, const Ctx &x, int tr, int k) {
  salt_at<K>(c, x, tr, k) = 0.0;
  if (tr < 0) { LAY(SAMSIM_A_M, k) = 0.0; LAY(SAMSIM_A_H_ABS, k) = 0.0; LAY(SAMSIM_A_THICK, k) = 0.0; }
}

// top_melt, mo_layer_dynamics.f90:191-327
template <class K>
__device__ __forceinline__ void top_melt(Col &c, const Ctx &x, int tr) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, N_top = g.n_top, N_middle = g.n_middle;
  const double thick_0 = g.thick_0;
  int Na = c.Na;
  // layer 1 absorbs layer 2
  salt_at<K>(c, x, tr, 1) = salt_at<K>(c, x, tr, 1) + salt_at<K>(c, x, tr, 2);
  if (tr < 0) {
    LAY(SAMSIM_A_M, 1) = LAY(SAMSIM_A_M, 1) + LAY(SAMSIM_A_M, 2);
    LAY(SAMSIM_A_H_ABS, 1) = LAY(SAMSIM_A_H_ABS, 1) + LAY(SAMSIM_A_H_ABS, 2);
    LAY(SAMSIM_A_THICK, 1) = LAY(SAMSIM_A_THICK, 1) + LAY(SAMSIM_A_THICK, 2);
  }
  // the layer values that later branches need from the OLD profile
  const bool have_mid = (Na == N);
  LayerVals old_top1 = {0, 0, 0};
  if (have_mid) old_top1 = layer_vals<K>(c, x, tr, N_top + 1);
  const int kend = (N_top - 1 < Na - 1) ? N_top - 1 : Na - 1;
  for (int k = 2; k <= kend; ++k) set_layer<K>(c, x, tr, k, layer_vals<K>(c, x, tr, k + 1), thick_0);  // reads old k+1 (not yet modified)
  if (Na <= N_top) {
    zero_layer<K>(c, x, tr, Na);
    Na = Na - 1;
  } else if (Na > N_top && Na <= N && LAY(SAMSIM_A_THICK, N_top + 1) / thick_0 < 1.00001) {
    for (int k = N_top; k <= Na - 1; ++k) set_layer<K>(c, x, tr, k, layer_vals<K>(c, x, tr, k + 1), thick_0);
    zero_layer<K>(c, x, tr, Na);
    Na = Na - 1;
  }
  if (Na == N && LAY(SAMSIM_A_THICK, N_top + 1) - thick_0 >= 0.000001) {
    double loss_m = thick_0 * old_top1.rho, loss_S = loss_m * old_top1.S_bu, loss_H = loss_m * old_top1.H;
    salt_at<K>(c, x, tr, N_top) = loss_S;
    if (tr < 0) { LAY(SAMSIM_A_M, N_top) = loss_m; LAY(SAMSIM_A_H_ABS, N_top) = loss_H; }
    for (int k = N_top + 1; k <= N_middle + N_top; ++k) {
      const LayerVals below = layer_vals<K>(c, x, tr, k + 1);  // old values of k+1
      double m = LAY(SAMSIM_A_M, k), H_abs = LAY(SAMSIM_A_H_ABS, k), S_abs = salt_at<K>(c, x, tr, k);
      m = m - loss_m; H_abs = H_abs - loss_H; S_abs = S_abs - loss_S;
      const double shift = thick_0 * (double)(float)(N_middle - k + N_top) / (double)(float)(N_middle);
      loss_m = shift * below.rho; loss_S = loss_m * below.S_bu; loss_H = loss_m * below.H;
      m = m + loss_m; H_abs = H_abs + loss_H; S_abs = S_abs + loss_S;
      salt_at<K>(c, x, tr, k) = S_abs;
      if (tr < 0) { LAY(SAMSIM_A_M, k) = m; LAY(SAMSIM_A_H_ABS, k) = H_abs; }
    }
    if (tr < 0)
      for (int k = N_top + 1; k <= N_top + N_middle; ++k) LAY(SAMSIM_A_THICK, k) = LAY(SAMSIM_A_THICK, k) - thick_0 / (double)(float)(N_middle);
  }
  if (tr >= 0) return;
  c.Na = Na;
  double sth = 0.0;
  for (int k = 1; k <= N; ++k) sth += LAY(SAMSIM_A_THICK, k);
  if (thick_0 * (Na + 0.501) <= sth && Na < N) STOPC(7889, 0);
}

// top_grow, mo_layer_dynamics.f90:607-716
template <class K>
__device__ __forceinline__ void top_grow(Col &c, const Ctx &x, int tr) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, N_top = g.n_top, N_middle = g.n_middle;
  const double thick_0 = g.thick_0;
  int Na = c.Na;
  LayerVals carry = layer_vals<K>(c, x, tr, 1);  // old values of layer k-1
  {
    const double loss_m = thick_0 * carry.rho, loss_S = loss_m * carry.S_bu, loss_H = loss_m * carry.H;
    salt_at<K>(c, x, tr, 1) = salt_at<K>(c, x, tr, 1) - loss_S;
    if (tr < 0) {
      LAY(SAMSIM_A_M, 1) = LAY(SAMSIM_A_M, 1) - loss_m;
      LAY(SAMSIM_A_H_ABS, 1) = LAY(SAMSIM_A_H_ABS, 1) - loss_H;
      LAY(SAMSIM_A_THICK, 1) = LAY(SAMSIM_A_THICK, 1) - thick_0;
    }
  }
  int kend = (N_top < Na) ? N_top : Na;
  if (Na > N_top && Na < N) kend = Na;  // second branch continues the same shift over N_top+1..Na
  for (int k = 2; k <= kend; ++k) {
    const LayerVals old_k = layer_vals<K>(c, x, tr, k);
    set_layer<K>(c, x, tr, k, carry, thick_0);
    carry = old_k;
  }
  if (Na <= N_top || (Na > N_top && Na < N)) {
    Na = Na + 1;
    set_layer<K>(c, x, tr, Na, carry, thick_0);  // S_bu*thick_0*rho and S_bu*rho*thick_0 differ in association:
    salt_at<K>(c, x, tr, Na) = carry.S_bu * thick_0 * carry.rho;  // mo_layer_dynamics.f90:660-661,674-675
    if (tr < 0) { LAY(SAMSIM_A_H_ABS, Na) = carry.H * thick_0 * carry.rho; LAY(SAMSIM_A_THICK, Na) = thick_0; }
  } else if (Na == N) {
    // carry holds the old values of layer N_top
    double loss_m = thick_0 * carry.rho, loss_S = loss_m * carry.S_bu, loss_H = loss_m * carry.H;
    for (int k = N_top + 1; k <= N_middle + N_top; ++k) {
      const LayerVals own = layer_vals<K>(c, x, tr, k);  // old values of k
      double m = LAY(SAMSIM_A_M, k), H_abs = LAY(SAMSIM_A_H_ABS, k), S_abs = salt_at<K>(c, x, tr, k);
      m = m + loss_m; H_abs = H_abs + loss_H; S_abs = S_abs + loss_S;
      const double shift = thick_0 * (double)(float)(N_middle - k + N_top) / (double)(float)(N_middle);
      loss_m = shift * own.rho; loss_S = loss_m * own.S_bu; loss_H = loss_m * own.H;
      m = m - loss_m; H_abs = H_abs - loss_H; S_abs = S_abs - loss_S;
      salt_at<K>(c, x, tr, k) = S_abs;
      if (tr < 0) { LAY(SAMSIM_A_M, k) = m; LAY(SAMSIM_A_H_ABS, k) = H_abs; }
    }
    if (tr < 0)
      for (int k = N_top + 1; k <= N_top + N_middle; ++k) LAY(SAMSIM_A_THICK, k) = LAY(SAMSIM_A_THICK, k) + thick_0 / (double)(float)(N_middle);
  }
  if (tr < 0) c.Na = Na;
}

// bottom_melt, mo_layer_dynamics.f90:341-427 (N_active == Nlayer)
template <class K>
__device__ __forceinline__ void bottom_melt(Col &c, const Ctx &x, int tr) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, N_top = g.n_top, N_middle = g.n_middle;
  const double thN = LAY(SAMSIM_A_THICK, N);
  double loss_m = 0.0, loss_S = 0.0, loss_H = 0.0;
  LayerVals carry = {0, 0, 0};
  for (int k = N_top + 1; k <= N_top + N_middle; ++k) {
    const LayerVals own = layer_vals<K>(c, x, tr, k);
    double m = LAY(SAMSIM_A_M, k), H_abs = LAY(SAMSIM_A_H_ABS, k), S_abs = salt_at<K>(c, x, tr, k);
    m = m + loss_m; H_abs = H_abs + loss_H; S_abs = S_abs + loss_S;
    const double shift = thN * (k - N_top) / (double)(float)(N_middle);
    loss_m = shift * own.rho; loss_H = loss_m * own.H; loss_S = loss_m * own.S_bu;
    m = m - loss_m; H_abs = H_abs - loss_H; S_abs = S_abs - loss_S;
    salt_at<K>(c, x, tr, k) = S_abs;
      if (tr < 0) { LAY(SAMSIM_A_M, k) = m; LAY(SAMSIM_A_H_ABS, k) = H_abs; }
    if (tr < 0) LAY(SAMSIM_A_THICK, k) = LAY(SAMSIM_A_THICK, k) - thN / (double)(float)(N_middle);
    carry = own;
  }
  for (int k = N_top + N_middle + 1; k <= N; ++k) {
    const LayerVals own = layer_vals<K>(c, x, tr, k);
    const double thick = LAY(SAMSIM_A_THICK, k);
    salt_at<K>(c, x, tr, k) = carry.rho * thick * carry.S_bu;
    if (tr < 0) { LAY(SAMSIM_A_H_ABS, k) = carry.rho * thick * carry.H; LAY(SAMSIM_A_M, k) = carry.rho * thick; }
    carry = own;
  }
}

// bottom_growth, mo_layer_dynamics.f90:438-523 (N_active == Nlayer)
template <class K>
__device__ __forceinline__ void bottom_growth(Col &c, const Ctx &x, int tr) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, N_top = g.n_top, N_middle = g.n_middle, N_bottom = g.n_bottom;
  const double thN = LAY(SAMSIM_A_THICK, N);
  double gain_m = 0.0, gain_S = 0.0, gain_H = 0.0;
  for (int k = N_top + 1; k <= N_top + N_middle; ++k) {
    const LayerVals below = layer_vals<K>(c, x, tr, k + 1);
    double m = LAY(SAMSIM_A_M, k), H_abs = LAY(SAMSIM_A_H_ABS, k), S_abs = salt_at<K>(c, x, tr, k);
    m = m - gain_m; H_abs = H_abs - gain_H; S_abs = S_abs - gain_S;
    const double shift = thN * (k - N_top) / (double)(float)(N_middle);
    gain_m = shift * below.rho; gain_H = gain_m * below.H; gain_S = gain_m * below.S_bu;
    m = m + gain_m; H_abs = H_abs + gain_H; S_abs = S_abs + gain_S;
    salt_at<K>(c, x, tr, k) = S_abs;
      if (tr < 0) { LAY(SAMSIM_A_M, k) = m; LAY(SAMSIM_A_H_ABS, k) = H_abs; }
  }
  if (tr < 0)
    for (int k = N_top + 1; k <= N_top + N_middle; ++k) LAY(SAMSIM_A_THICK, k) = LAY(SAMSIM_A_THICK, k) + thN / (double)(float)(N_middle);
  for (int k = N - N_bottom + 1; k <= N - 1; ++k) {
    salt_at<K>(c, x, tr, k) = salt_at<K>(c, x, tr, k + 1);
    if (tr < 0) { LAY(SAMSIM_A_H_ABS, k) = LAY(SAMSIM_A_H_ABS, k + 1); LAY(SAMSIM_A_M, k) = LAY(SAMSIM_A_M, k + 1); }
  }
  const double mN = thN * rho_l;
  salt_at<K>(c, x, tr, N) = mN * salt_below<K>(c, x, tr);
  if (tr < 0) { LAY(SAMSIM_A_M, N) = mN; LAY(SAMSIM_A_H_ABS, N) = mN * g.T_bottom * c_l; }
}

// layer_dynamics, mo_layer_dynamics.f90:64-175: exactly one branch per call, in priority order
template <class K>
__device__ RARE void layer_dynamics(Col &c, const Ctx &x) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, Na = c.Na, N_top = g.n_top, bf = CFG(bottom_flag);
  const double thick_0 = g.thick_0;
  const int km1 = (Na - 1 > 1) ? Na - 1 : 1;
  const double phi_Na = LAY(SAMSIM_A_PHI, Na), phi_km1 = LAY(SAMSIM_A_PHI, km1);
  const double phi_Nm1 = LAY(SAMSIM_A_PHI, N - 1), phi_N = LAY(SAMSIM_A_PHI, N);
  const double th_mid = LAY(SAMSIM_A_THICK, N_top + 1), th1 = LAY(SAMSIM_A_THICK, 1);
  const int nt = HAS_BGC ? x.n_bgc : 0;   // tracer replays (tr = nt-1 .. 0) come first, the routine proper (tr = -1) last
  if (phi_Nm1 <= psi_s_min / 2.0 && phi_Na < 0.00001 && Na == N && th_mid / thick_0 > 1.000001 && bf == 1) {
    for (int tr = nt - 1; tr >= -1; --tr) bottom_melt<K>(c, x, tr);
  } else if (Na > 1 && Na < N && phi_Na < 0.00001 && phi_km1 <= psi_s_min / 2.0 && bf == 1) {
    for (int tr = nt - 1; tr >= -1; --tr) zero_layer<K>(c, x, tr, Na);  // bottom_melt_simple, :573-591
    c.Na = Na - 1;
  } else if (Na > 1 && phi_Na < 0.00001 && phi_km1 <= psi_s_min / 2.0 && (th_mid / thick_0) < 1.01 && bf == 1) {
    for (int tr = nt - 1; tr >= -1; --tr) zero_layer<K>(c, x, tr, Na);
    c.Na = Na - 1;
  } else if (phi_Na > psi_s_min && Na < N && bf == 1) {
    // bottom_growth_simple, :537-560
    const double mnew = thick_0 * rho_l;
    c.Na = Na + 1;
    LAY(SAMSIM_A_THICK, Na + 1) = thick_0;
    LAY(SAMSIM_A_M, Na + 1) = mnew;
    LAY(SAMSIM_A_H_ABS, Na + 1) = mnew * g.T_bottom * c_l;
    for (int tr = nt - 1; tr >= -1; --tr) salt_at<K>(c, x, tr, Na + 1) = mnew * salt_below<K>(c, x, tr);
  } else if (phi_N > psi_s_min && bf == 1) {
    for (int tr = nt - 1; tr >= -1; --tr) bottom_growth<K>(c, x, tr);
  } else if (th1 > 1.5 * thick_0) {
    GS(MELT_OUT3) = GS(MELT_OUT3) - th1;
    for (int tr = nt - 1; tr >= -1; --tr) top_grow<K>(c, x, tr);
    GS(MELT_OUT3) = GS(MELT_OUT3) + LAY(SAMSIM_A_THICK, 1);
  } else if (th1 < 0.5 * thick_0) {
    GS(MELT_OUT3) = GS(MELT_OUT3) - th1;
    for (int tr = nt - 1; tr >= -1; --tr) top_melt<K>(c, x, tr);
    if (c.status) return;
    GS(MELT_OUT3) = GS(MELT_OUT3) + LAY(SAMSIM_A_THICK, 1);
  }
}


// ---------------------------------------------------------------- output snapshot, mo_grotz.f90:340-398
template <class K>
__device__ RARE void output_point(Col &c, const Ctx &x, long long col, double time) {
  const samsim_config &g = x.p->cfg;
  if (c.Na > 1) GS(FREEBOARD) = func_freeboard<K>(c, x); else GS(FREEBOARD) = 0.0;
  if (CFG(grav_flag) == 2) {
    if (CL(grav_drain) == 0.0) CL(grav_temp) = 0.0; else CL(grav_temp) = CL(grav_temp) / CL(grav_drain);
    CL(grav_salt) = CL(grav_salt) / g.time_out;
    CL(grav_drain) = CL(grav_drain) / g.time_out;
  }
  {  // the vital signs are not carried in registers between output points: their slots of the scalar block are written here
    gdouble *sc = x.scal + c.col;
    const size_t nc = c.ncol;
    sc[(size_t)SAMSIM_S_ENERGY_STORED * nc] = c.energy_stored; sc[(size_t)SAMSIM_S_FRESHWATER * nc] = c.freshwater;
    sc[(size_t)SAMSIM_S_TOTAL_RESIST * nc] = c.total_resist; sc[(size_t)SAMSIM_S_THICKNESS * nc] = c.thickness;
    sc[(size_t)SAMSIM_S_BULK_SALIN * nc] = c.bulk_salin;
  }
  if (col >= x.out_col0 && col < x.out_col0 + x.out_ncols) {
    const size_t oc = (size_t)(col - x.out_col0), on = (size_t)x.out_ncols;
    for (int a = 0; a < SAMSIM_NARR; ++a) {
      if (a == SAMSIM_A_RAY) continue;  // captured before it was overwritten (see sweep_up_fused / column_step)
      for (int k = 1; k <= c.N; ++k) {
        double v = LAY(a, k);
        if (a == SAMSIM_A_S_BU && k <= c.Na) v = LAY(SAMSIM_A_S_ABS, k) / LAY(SAMSIM_A_M, k);  // refresh of mo_grotz.f90:333-335
        x.out_lay[((size_t)a * c.N + (k - 1)) * on + oc] = v;
      }
    }
    gdouble *o = x.out_scal + oc;
#define OUT(idx, v) o[(size_t)(idx) * on] = (v)
    OUT(SAMSIM_S_M_SNOW, CL(m_snow)); OUT(SAMSIM_S_H_ABS_SNOW, CL(H_abs_snow)); OUT(SAMSIM_S_S_ABS_SNOW, GS(S_ABS_SNOW));
    OUT(SAMSIM_S_THICK_SNOW, CL(thick_snow)); OUT(SAMSIM_S_PSI_S_SNOW, CL(psi_s_snow)); OUT(SAMSIM_S_PSI_L_SNOW, GS(PSI_L_SNOW));
    OUT(SAMSIM_S_PSI_G_SNOW, GS(PSI_G_SNOW)); OUT(SAMSIM_S_T_SNOW, CL(T_snow)); OUT(SAMSIM_S_PHI_S, GS(PHI_S));
    OUT(SAMSIM_S_T_TOP, CL(T_top)); OUT(SAMSIM_S_MELT_THICK, CL(melt_thick)); OUT(SAMSIM_S_T2M, CL(T2m));
    OUT(SAMSIM_S_LIQUID_PRECIP, CL(liquid_precip)); OUT(SAMSIM_S_SOLID_PRECIP, CL(solid_precip)); OUT(SAMSIM_S_FL_Q_BOTTOM, c.fl_q_bottom);
    OUT(SAMSIM_S_GRAV_DRAIN, CL(grav_drain)); OUT(SAMSIM_S_GRAV_SALT, CL(grav_salt)); OUT(SAMSIM_S_GRAV_TEMP, CL(grav_temp));
    OUT(SAMSIM_S_MELT_OUT1, GS(MELT_OUT1)); OUT(SAMSIM_S_MELT_OUT2, GS(MELT_OUT2)); OUT(SAMSIM_S_MELT_OUT3, GS(MELT_OUT3));
    OUT(SAMSIM_S_MELT_ERR, GS(MELT_ERR)); OUT(SAMSIM_S_FREEBOARD, GS(FREEBOARD)); OUT(SAMSIM_S_T_FREEZE, GS(T_FREEZE));
    OUT(SAMSIM_S_ALBEDO, CL(albedo)); OUT(SAMSIM_S_FL_SW, CL(fl_sw)); OUT(SAMSIM_S_FL_LW, CL(fl_lw));
    OUT(SAMSIM_S_MELT_THICK_SNOW, CL(melt_thick_snow)); OUT(SAMSIM_S_FL_Q_SNOW, CL(fl_Q_snow));
    OUT(SAMSIM_S_ENERGY_STORED, c.energy_stored); OUT(SAMSIM_S_FRESHWATER, c.freshwater); OUT(SAMSIM_S_TOTAL_RESIST, c.total_resist);
    OUT(SAMSIM_S_THICKNESS, c.thickness); OUT(SAMSIM_S_BULK_SALIN, c.bulk_salin);
    OUT(SAMSIM_S_FL_REST, (CFG(boundflux_flag) == 2 && (!K::general || CFG(atmoflux_flag) == 2)) ? CL(fl_lw) + 0.0 + 0.0
                                                                        : GSI(SAMSIM_S_FL_REST));
    OUT(SAMSIM_S_S_BU_BOTTOM, x.S_bu_bottom);
    OUT(SAMSIM_S_DT2M, GSI(SAMSIM_S_DT2M));
    OUT(SAMSIM_S_PRECIP_SCALE, GSI(SAMSIM_S_PRECIP_SCALE));
#undef OUT
    x.out_n_active[oc] = c.Na;
    if (HAS_BGC) {
      for (int t = 0; t < x.n_bgc; ++t) {
        for (int k = 1; k <= c.N; ++k) x.out_bgc[((size_t)t * c.N + (k - 1)) * on + oc] = BGC(t, k);
        x.out_bgc_bot[(size_t)t * on + oc] = BGC_BOT(t);
      }
    }
  }
  CL(grav_drain) = 0.0; CL(grav_salt) = 0.0; CL(grav_temp) = 0.0;
  GS(MELT_OUT1) = 0.0; GS(MELT_OUT2) = 0.0; GS(MELT_OUT3) = 0.0;
  (void)time;
}

// prescribe_flag 2, mo_grotz.f90:482-497: bulk salinity linear from S_bu_bottom to 4 over the lowest 0.15 m and from 4 to 0
// above it.  The SUMs start afresh for every layer, in ascending order like the reference's; of S_bu only layer 1 is written (the up
// sweep refreshes the others from S_abs before anything reads them).  Layer 1 of ice thinner than 0.15 m keeps the S_bu of the first sweep.
template <class K>
__device__ RARE void prescribe_salinity(Col &c, const Ctx &x) {
  const int N = c.N, Na = c.Na;
  const double Sb = x.S_bu_bottom;
  auto thick_sum = [&](int a) { double t = 0.0; for (int j = a; j <= Na; ++j) t += LAY(SAMSIM_A_THICK, j); return t; };
  const double total = thick_sum(1);
  double S_bu1 = LAY(SAMSIM_A_S_BU, 1);
  int k = Na;
  while (k > 1) {
    const double t = thick_sum(k);
    if (!(t < 0.15)) break;
    LAY(SAMSIM_A_S_ABS, k) = (Sb - t / 0.15 * (Sb - 4.0)) * LAY(SAMSIM_A_M, k);
    k = k - 1;
  }
  while (k > 1) {
    const double t = thick_sum(k);
    if (!(t >= 0.15)) break;
    LAY(SAMSIM_A_S_ABS, k) = (4.0 - 4.0 * (t - 0.15) / (total - 0.15)) * LAY(SAMSIM_A_M, k);
    k = k - 1;
    S_bu1 = 0.0;
  }
  // Both loops ending above layer 1 takes SUMs that shrink as layers are added (a negative or NaN thickness).  The reference then
  // leaves S_bu(2..k) as the refresh of mo_grotz.f90:333 set them and forms S_abs = S_bu*m from that; the unfused order, which a
  // prescribed profile always takes, has that row in the array (sweep_expulsion_transfer).
  for (int j = k; j > 1; --j) LAY(SAMSIM_A_S_ABS, j) = LAY(SAMSIM_A_S_BU, j) * LAY(SAMSIM_A_M, j);
  if (Na > 1) LAY(SAMSIM_A_S_ABS, Na) = Sb * LAY(SAMSIM_A_M, Na);
  else S_bu1 = Sb;
  LAY(SAMSIM_A_S_ABS, 1) = S_bu1 * LAY(SAMSIM_A_M, 1);
  LAY(SAMSIM_A_S_BU, 1) = S_bu1;  // read by the thin-snow coupling of sub_heat_fluxes (mo_heat_fluxes.f90:293)
  for (int j = Na + 1; j <= N; ++j) LAY(SAMSIM_A_S_ABS, j) = 0.0;
}

// flush4, mo_flush.f90:253-296 (flush_flag 6): the melt water leaves the top layer with its brine salinity; every layer more
// liquid than the one above loses the fraction 1 - para_flush_gamma of its salt, down to the first one that is not
// (layers below N_active hold no salt, so the walk may end there).
template <class K>
__device__ RARE void flush4(Col &c, const Ctx &x) {
  const int Na = c.Na;
  const double T1 = LAY(SAMSIM_A_T, 1), m1 = LAY(SAMSIM_A_M, 1), melt = CL(melt_thick);
  double S1 = LAY(SAMSIM_A_S_ABS, 1);
  LAY(SAMSIM_A_H_ABS, 1) = LAY(SAMSIM_A_H_ABS, 1) - melt * rho_l * c_l * T1;
  S1 = S1 - melt * rho_l * S_br_clamped(x.salt, T1, S1 / m1);
  LAY(SAMSIM_A_THICK, 1) = LAY(SAMSIM_A_THICK, 1) - melt;
  LAY(SAMSIM_A_M, 1) = m1 - melt * rho_l;
  CL(melt_thick) = 0.0;
  double above = LAY(SAMSIM_A_PSI_L, 1);
  for (int k = 2; k <= Na; ++k) {
    const double here = LAY(SAMSIM_A_PSI_L, k);
    if (!(here > above)) break;
    LAY(SAMSIM_A_S_ABS, k) = para_flush_gamma * LAY(SAMSIM_A_S_ABS, k);
    above = here;
  }
  LAY(SAMSIM_A_S_ABS, 1) = dmax(S1, 0.0);
  double mn = 0.0;
  for (int k = 2; k <= Na; ++k) mn = dmin(mn, LAY(SAMSIM_A_S_ABS, k));
  if (mn < 0.0) STOPC(9876, 0);
}

// testcase specifics that only touch scalars, mo_grotz.f90:503-565
template <class K>
__device__ __forceinline__ void testcase_scalars(Col &c, const Ctx &x, const samsim_config &g, double time) {
  if (CFG(testcase) == 1) {  // sub_test1, mo_testcase_specifics.f90:42-89
    for (int n = 1; n <= 20; ++n) {
      if (fabs(time - (double)((float)(12 * n) * 3600.0f)) < (double)0.01f) { CL(T_top) = (n & 1) ? -10.0 : -5.0; break; }
    }
  } else if (K::general && CFG(testcase) == 3) {  // sub_test3, :172-187
    CL(liquid_precip) = 0.0;
    CL(solid_precip) = 0.15 / 86400.0 / 356.0;
  } else if (CFG(testcase) == 4 || CFG(testcase) == 7) {  // sub_test4, :197-202
    c.fl_q_bottom = -7.0 * sin(time * (2.0 * pi_f) / (86400.0 * 365.0)) + 7.0;
    if (K::sites) c.fl_q_bottom = c.fl_q_bottom + x.dflq;   // samsim_set_ocean: this column's offset (0 unless given)
  } else if (K::general && CFG(testcase) == 2) {  // sub_test2, :99-111
    if (time > 86400.0 * 25.0) CL(T2m) = 15.0;
    else if (time > 86400.0 * 15.0) CL(T2m) = 1.0;
  } else if (K::general && CFG(testcase) == 9) {  // sub_test9, :121-136
    if (time < 19.75 * 3600.0) CL(T2m) = 0.0;
    else if (time < 86400.0 * 3.0 + 2.25 * 3600.0) CL(T2m) = -15.0;
    else CL(T2m) = 1.0;
  } else if (K::general && CFG(testcase) == 34) {  // sub_test34, :146-162
    if (time < 2.0 * 3600.0) CL(T2m) = 0.0;
    else if (time < 86400.0 * 5.0) CL(T2m) = -15.0;
    else if (time < 86400.0 * 7.0) CL(T2m) = -5.0;
    else CL(T2m) = 1.0;
  } else if (K::general && CFG(testcase) == 6) {  // sub_test6, :211-232
    const double t[8] = {1714.0, 1676.0, 1525.0, 1483.0, 1385.0, 1349.0, 1160.0, 1100.0};
    for (int i = 0; i < 8; ++i) {
      if (time > t[i] * 60.0) { CL(T2m) = (i == 0) ? -19.0 : ((i & 1) ? -5.0 : -18.0); break; }
    }
  }
}

// The reference's order between expulsion and the heat fluxes, sweep by sweep: taken whenever something sits between
// expulsion and gravity drainage (the output block, thin-snow coupling, a possible flooding event) or no Rayleigh-number
// drainage runs at all; mo_grotz.f90:312-565.
template <class K>
__device__ RARE void down_unfused(Col &c, const Ctx &x, long long col, double time, int tc, bool out_step, bool coupling,
                                  bool do_grav, bool do_beer) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N, Na = c.Na;
    sweep_expulsion_transfer<K>(c, x);   // mo_grotz.f90:312-335

    if (out_step) output_point<K>(c, x, col, time);  // mo_grotz.f90:340-398

    // bottom-layer gas -> ocean water, mo_grotz.f90:405-410
    {
      const double psi_gN = LAY(SAMSIM_A_PSI_G, Na);
      if (psi_gN > 0.0) {
        const double temp2 = psi_gN * LAY(SAMSIM_A_THICK, Na) * rho_l;
        LAY(SAMSIM_A_M, Na) = LAY(SAMSIM_A_M, Na) + temp2;
        LAY(SAMSIM_A_S_ABS, Na) = LAY(SAMSIM_A_S_ABS, Na) + temp2 * x.S_bu_bottom;
        LAY(SAMSIM_A_H_ABS, Na) = LAY(SAMSIM_A_H_ABS, Na) + temp2 * c_l * g.T_bottom;
      }
    }
    // thin-snow coupling, mo_grotz.f90:418-420
    if (coupling) {
      snow_coupling<K>(c, x);
      if (c.status) return;
    }
    // flooding, mo_grotz.f90:428-445
    if (Na > 1 && CFG(flood_flag) > 1 && CL(m_snow) > 0.0 && CFG(freeboard_snow_flag) == 0) {
      // func_freeboard's "snow underwater" branch (mo_functions.f90:96-101) needs only the buoyancy totals, which S1
      // and P2 have accumulated; a non-negative freeboard is not read here and every later reader re-evaluates it
      const double buoy = c.buoy_s * (rho_l - rho_s) + c.buoy_g * rho_l;
      if (CL(m_snow) > buoy) {
        GS(FREEBOARD) = (buoy - CL(m_snow)) / rho_l;
        if (GS(FREEBOARD) < 0.0 && CFG(flood_flag) == 2) {
          flood<K>(c, x);
          if (CFG(grav_flag) >= 2) refresh_ray_top<K>(c, x, LAY(SAMSIM_A_THICK, 1), LAY(SAMSIM_A_PSI_L, 1), LAY(SAMSIM_A_S_BR, 1));
        } else if (K::general && CFG(flood_flag) == 3 && GS(FREEBOARD) < neg_free) {
          flood_simple<K>(c, x);
          if (CFG(grav_flag) >= 2) refresh_ray_top<K>(c, x, LAY(SAMSIM_A_THICK, 1), LAY(SAMSIM_A_PSI_L, 1), LAY(SAMSIM_A_S_BR, 1));
        }
      }
    }
    // bottom turbulence, sub_turb_flux mo_functions.f90:347-363
    if (CFG(turb_flag) == 2) {
      const double m = LAY(SAMSIM_A_M, Na), T = LAY(SAMSIM_A_T, Na);
      double S_abs = LAY(SAMSIM_A_S_ABS, Na);
      const double turb = Turb_A * exp(Turb_B * (-ocean_density<K>(x) + func_density(T, S_abs / m))) * g.dt;
      S_abs = S_abs - turb * (S_abs / m - x.S_bu_bottom);
      LAY(SAMSIM_A_S_ABS, Na) = S_abs;
      if (HAS_BGC) {  // the tracers of the bottom layer mix with the same coefficient, :358-360
        for (int t = 0; t < x.n_bgc; ++t) { const double q = BGC(t, Na); BGC(t, Na) = q - turb * (q / m - BGC_BOT(t)); }
      }
    }

    // testcase specifics, mo_grotz.f90:503-565 (the scalar ones commute with the gravity drainage sweep below)
    testcase_scalars<K>(c, x, g, time);

    // gravity drainage (mo_grotz.f90:463-477) fused with the Beer-law pass of sub_heat_fluxes
    const double beer0 = radiation_header<K>(c, x, time, tc);
    c.frad = 0.0;
    if (do_grav) {
      sweep_grav_drain<K>(c, x, do_beer, beer0);
      c.bgc_grav = true;
      if (c.status) return;
    } else if (K::general && CFG(grav_flag) == 3 && Na > 1) {
      sweep_grav_drain_simple<K>(c, x, do_beer, beer0);
    } else if (do_beer) {
      sweep_beer<K>(c, x, beer0);
    }
    if (K::general && CFG(prescribe_flag) == 2) prescribe_salinity<K>(c, x);  // mo_grotz.f90:482-497
    if (K::general && CFG(testcase) == 5 && c.step + 1 == 2) {  // mo_grotz.f90:543-544
      for (int k = 1; k <= N; ++k) LAY(SAMSIM_A_S_ABS, k) = 5.0 * LAY(SAMSIM_A_M, k);
    }
    // conductive update of layers >= 2 (sub_heat_fluxes, mo_grotz.f90:584; the tank budget in between only reads S_abs and m)
    sweep_heat_down<K>(c, x);
}

// Safety net of the stored-row decision (sweep_down_fused): a late reader of psi_s / psi_l / psi_g -- func_freeboard, flush3 -- in a
// step whose down sweep skipped the rows of layers >= 3.  The sweep evaluates those readers' conditions exactly before it skips
// (profiles/r3_melt_ensemble_status.json: a free-running ensemble through melt season and freeze-up never gets here; the stamps
// build counts the calls, CT_REFILL), so this is not on any tested trajectory; should a column ever arrive, it keeps running:
// the rows are filled by one Expulsion pass over the finished layers (temperature of the second sweep, current masses) -- the
// values the next step's first sweep will form -- instead of the column being stopped.
template <class K>
__device__ RARE void refill_psi_rows(Col &c, const Ctx &x) {
  ST_COUNT(CT_REFILL, (unsigned long long)__popcll(__ballot(1)));
  THICK_RULE_INIT(tr);
  double fb_a2 = LAY(SAMSIM_A_PSI_S, 2) * THICK_AT(tr, 2), fb_g2 = LAY(SAMSIM_A_PSI_G, 2) * THICK_AT(tr, 2);
  for (int k = 3; k <= c.Na; ++k) {
    const double m = LAY(SAMSIM_A_M, k), thick = THICK_AT(tr, k);
    double S_bu, H;
    per_mass(LAY(SAMSIM_A_S_ABS, k), LAY(SAMSIM_A_H_ABS, k), m, S_bu, H);
    const double S_br = S_br_clamped(x.salt, LAY(SAMSIM_A_T, k), S_bu);
    const Expelled e = expulsion(phi_from_T(x.salt, H, S_bu, S_br), thick, m, recip(thick));
    LAY(SAMSIM_A_PSI_S, k) = e.psi_s;
    LAY(SAMSIM_A_PSI_L, k) = e.psi_l;
    LAY(SAMSIM_A_PSI_G, k) = e.psi_g;
    fb_a2 += e.psi_s * thick; fb_g2 += e.psi_g * thick;
  }
  SPEC(SP_FB_A2) = fb_a2; SPEC(SP_FB_G2) = fb_g2;
  c.psi_full = true;
}

// ---------------------------------------------------------------- one time step, mo_grotz.f90:182-835
template <class K>
__device__ __forceinline__ void column_step(Col &c, Ctx &x, long long col, double time, int tc, bool out_step, bool next_out,
                                            bool last_step) {
  const samsim_config &g = x.p->cfg;
  const int N = c.N;

  if (out_step) {
    vital_signs<K>(c, x);  // mo_grotz.f90:192-223; only ever read by `output`
    // `output` prints the Rayleigh numbers of the PREVIOUS step's fl_grav_drain.  Normally the previous up sweep has
    // saved them before overwriting; on the first step after set_state the array itself still holds them.
    if ((c.flags & COLF_RESTART) && col >= x.out_col0 && col < x.out_col0 + x.out_ncols) {
      const size_t oc = (size_t)(col - x.out_col0), on = (size_t)x.out_ncols;
      for (int k = 1; k <= N; ++k) x.out_lay[((size_t)SAMSIM_A_RAY * N + (k - 1)) * on + oc] = LAY(SAMSIM_A_RAY, k);
    }
  }

  // forcing, mo_grotz.f90:229-241 (+ ensemble perturbation)
  if (CFG(atmoflux_flag) == 2) {
    if (time == time_input(tc)) {
      CL(T2m) = x.f_T2m[x.soff + tc - 1];
      CL(liquid_precip) = x.f_precip[x.soff + tc - 1];
    } else {
      const double temp = (time - time_input(tc - 1)) / (time_input(tc) - time_input(tc - 1));
      CL(T2m) = (1.0 - temp) * x.f_T2m[x.soff + tc - 2] + temp * x.f_T2m[x.soff + tc - 1];
      CL(liquid_precip) = (1.0 - temp) * x.f_precip[x.soff + tc - 2] + temp * x.f_precip[x.soff + tc - 1];
    }
    // the column's perturbation (samsim_set_forcing): two words of the scalar block, read where they are used
    CL(T2m) = CL(T2m) + GSI(SAMSIM_S_DT2M);
    CL(liquid_precip) = CL(liquid_precip) * GSI(SAMSIM_S_PRECIP_SCALE);
  }

  c.bgc_flood = 0.0; c.bgc_grav = false;
  snow_fall<K>(c, x);                  // mo_grotz.f90:251-265
  snow_block<K>(c, x);                 // mo_grotz.f90:273-292
  if (c.status) return;

  // first thermodynamic sweep, mo_grotz.f90:297-307 (+ Rayleigh numbers): only layer 1 is left to do unless the
  // column changed below layer 1 since the last up sweep
  c.ray_all = (c.flags & COLF_DIRTY) != 0;
  // (sparse Rayleigh rows need wave-uniform layer indices and every column of the wave in the sweep)
  const bool whole_wave = !wave_any(!c.ray_all);
  if (c.ray_all) { ST_COUNT(CT_DIRTY, 1); ST_COUNT(CT_L_DIRTY, (unsigned long long)__popcll(__ballot(1))); sweep_thermo_expulsion<K>(c, x, out_step, whole_wave); }
  else prologue_top_layer<K>(c, x);
  c.flags &= (COLF_REGULAR | COLF_FLUSHED);
  if (c.status) return;

  int Na = c.Na;
  const bool do_grav = (CFG(grav_flag) == 2 && Na > 1), do_beer = (CFG(boundflux_flag) == 2);
  // The fused down sweep covers the common step.  The reference's order is kept by the unfused path whenever something
  // sits between expulsion and gravity drainage: the output block, thin-snow coupling, a possible flooding event
  // (decided from SUM(psi_g*thick) AFTER expulsion_flux: m_snow above the solid-only buoyancy is treated as possible).
  const bool coupling = (CL(m_snow) > 0.0 && CL(thick_snow) < g.thick_min);
  const bool flood_possible = (CFG(flood_flag) > 1 && CL(m_snow) > 0.0 && CFG(freeboard_snow_flag) == 0 &&
                               CL(m_snow) > c.buoy_s * (rho_l - rho_s));
  // (a thin snow cover no longer needs the unfused order: the fused down sweep couples it to the top layer in place)
  // (a possible flooding no longer needs it either where flood_flag is 2 and no thin snow is coupled in the same step: see below)
  const bool fused_col = do_grav && !out_step && (c.step + 1 != 1) && (!flood_possible || (CFG(flood_flag) == 2 && !coupling)) &&
                     !(K::general && CFG(testcase) == 5 && c.step + 1 == 2) && !HAS_BGC &&
                     !(K::general && CFG(prescribe_flag) == 2)
                     && (c.flags & COLF_REGULAR) != 0   // the fused down sweep takes the thicknesses from the grid rule only
      ;
  // SAMSIM_PATH_MODE 2 (default): one path per wave.  A wave whose columns disagree runs both paths one after the other, each
  // with part of its lanes idle -- the normal state of a melt season, when some column of almost every wave has thin snow or a
  // flooded surface.  The unfused path is the general one (the reference's order, literally) and gives a column the same bits as
  // the fused one (tools/path_equiv.py compares the two on the GPU), so a wave in which any column needs it takes it for all.
  // SAMSIM_PATH_MODE 1 = always unfused (the build tools/path_equiv.py compares the product with).
#if SAMSIM_PATH_MODE == 1
  const bool fused = false;
#else
  const bool fused = !wave_any(!fused_col);
#endif

  ST_MARK(ST_PRO);
  ST_COUNT(CT_WAVESTEPS, 1);
  ST_COUNT(CT_LANES, (unsigned long long)__popcll(__ballot(1)));
  ST_COUNT(CT_L_COUPLING, (unsigned long long)__popcll(__ballot(coupling)));
  ST_COUNT(CT_L_FLOODP, (unsigned long long)__popcll(__ballot(flood_possible)));
  ST_COUNT(CT_L_IRREG, (unsigned long long)__popcll(__ballot((c.flags & COLF_REGULAR) == 0)));
  ST_COUNT(CT_L_UNFUSED, (unsigned long long)__popcll(__ballot(!fused_col)));
  bool surface_done = false;   // the fused down sweep has evaluated the surface balance already
  if (fused) {
    ST_COUNT(CT_FUSED, 1);
    // Flooding (mo_grotz.f90:428-445) sits between the brine expulsion and the gravity drainage.  Whether a column floods is
    // decided from the gas volume expulsion_flux leaves (freeboard) and what it moves depends on the top and bottom layers after the
    // expulsion -- both known only once the expulsion has gone through the whole column, while the fused sweep drains layer 1 long
    // before.  So a column whose snow load makes flooding possible first takes a DRY RUN of the expulsion (four loads per layer, no
    // store), floods its top and bottom layers and snow in registers exactly as flood() does on the arrays, and hands the fused sweep
    // the flooded top layer; the Rayleigh number of layer 1 is redone with the flooded thickness from the first sweep's scan.  One
    // read-only pass instead of the unfused order's three extra sweeps (and flood / refresh_ray_top no longer walk the column at
    // all): BASELINE cfg5, where every column floods in every step.
    if (flood_possible) {
      ExpelledEnds en;
      sweep_expulsion_transfer<K, true>(c, x, &en);
      THICK_RULE_INIT(tr);
      if (en.psi_gN > 0.0) {   // bottom-layer gas -> ocean water (mo_grotz.f90:405-410) precedes the flooding block
        const double temp2 = en.psi_gN * THICK_AT(tr, Na) * rho_l;
        en.mN = en.mN + temp2;
        en.SN = en.SN + temp2 * x.S_bu_bottom;
        en.HN = en.HN + temp2 * c_l * g.T_bottom;
      }
      const double buoy = c.buoy_s * (rho_l - rho_s) + en.buoy_g * rho_l;
      if (CL(m_snow) > buoy) {
        GS(FREEBOARD) = (buoy - CL(m_snow)) / rho_l;
        if (GS(FREEBOARD) < 0.0) {
          FloodEnds e;
          e.S1 = en.S1; e.H1 = en.H1; e.m1 = en.m1; e.th1 = LAY(SAMSIM_A_THICK, 1);
          e.SN = en.SN; e.HN = en.HN; e.mN = en.mN; e.TN = en.TN;
          const double th1_before = e.th1;
          flood_core<K>(c, x, SPEC(SP_FL_HP), SPEC(SP_FL_SALL), e);
          LAY(SAMSIM_A_THICK, 1) = e.th1;
          refresh_ray_top<K>(c, x, e.th1, en.psi_l1, en.S_br1);
          // (the scan rows have served: they carry the flooded top layer to the down sweep, see samsim_device.h)
          c.flags |= COLF_FLOODED;
          SPEC(SP_FLD_S1) = e.S1; SPEC(SP_FLD_H1) = e.H1; SPEC(SP_FLD_M1) = e.m1; SPEC(SP_FLD_TH1_BEFORE) = th1_before;
          if (e.deep) { c.flags |= COLF_FLOOD_DEEP; SPEC(SP_FL_HP) = e.incS; SPEC(SP_FL_SALL) = e.incH; }
        }
      }
    }
    // testcase specifics (mo_grotz.f90:503-565) and the radiation header only read time, snow scalars and psi_l(1),
    // none of which the down sweep changes, so they can run first
    testcase_scalars<K>(c, x, g, time);
    // with a thin snow cover somewhere in the wave the radiation header waits for the coupling inside the sweep (it reads T_snow)
    const bool late_rad = wave_any(coupling);
    double beer0 = 0.0;
    if (!late_rad) {
      beer0 = radiation_header<K>(c, x, time, tc);
      c.frad = 0.0;
    }
    // The volume fractions of layers >= 3 are only stored where something reads them (see sweep_down_fused): always through the
    // run-time-flag instantiation; with melt-water flushing (flush_flag 5 on the radiative surface) the sweep decides from the
    // finished top layer; without it (flush_flag 1) only the vital signs at the next output point and a get_state after the
    // launch read them
    bool store_default = true, decide_psi = false;
    if (K::fixed && K::boundflux_flag == 2 && K::flush_flag == 5) { store_default = next_out || last_step; decide_psi = true; }
    if (K::fixed && K::flush_flag == 1) store_default = next_out || last_step;
    // fl_rad(N_active) enters the conductive update of every layer (mo_heat_fluxes.f90:282-285), which the down sweep applies as
    // it goes: the Beer-law product over the layer thicknesses (a pass over one array) comes first
    if (do_beer && !late_rad) sweep_beer<K>(c, x, beer0);
    sweep_down_fused<K>(c, x, store_default, decide_psi, surface_done, coupling, late_rad, time, tc, do_beer);
    ST_MARK(ST_DFUSED);
    if (c.status) return;
  } else {
    ST_COUNT(CT_UNFUSED, 1);
    c.psi_full = true;
    down_unfused<K>(c, x, col, time, tc, out_step, coupling, do_grav, do_beer);
    ST_MARK(ST_DUNFUSED);
    if (c.status) return;
  }

  // tank: the water below holds what salt the ice does not, mo_grotz.f90:573-575
  if (K::general && CFG(tank_flag) == 2) {
    double sS = 0.0, sm = 0.0;
    for (int k = 1; k <= c.Na; ++k) { sS += LAY(SAMSIM_A_S_ABS, k); sm += LAY(SAMSIM_A_M, k); }
    x.S_bu_bottom = (g.S_total - sS) / (g.m_total - sm);
    if (HAS_BGC) {  // :575-577 (sic: the budget of tracer 1 sets the concentration of every tracer)
      double sb = 0.0;
      for (int k = 1; k <= c.Na; ++k) sb += BGC(0, k);
      const double v = (x.bgc_total0 - sb) / (g.m_total - sm);
      for (int t = 0; t < x.n_bgc; ++t) BGC_BOT(t) = v;
    }
  }

  // heat fluxes + second thermodynamic sweep (mo_grotz.f90:584-598) + first sweep of the next step for layers >= 2
  if (!surface_done) surface_flux<K>(c, x);
  ST_MARK(ST_SURF);
  sweep_up_fused<K>(c, x, col, next_out, next_out || last_step);
  ST_MARK(ST_UP);
  if (c.status) return;

  // snow thermodynamics again, mo_grotz.f90:603-625
  const double melt_thick_snow_old = CL(melt_thick_snow);
  snow_block<K>(c, x);
  if (c.status) return;
  CL(melt_thick_snow) = melt_thick_snow_old + CL(melt_thick_snow);

  // flushing preparations, mo_grotz.f90:632-664
  bool fb_valid = false;
  if (Na > 1 && CFG(flush_flag) > 2 && (CFG(boundflux_flag) == 2 || (K::general && CFG(boundflux_flag) == 3))) {
    // boundflux_flag 3 (:649-663) runs the same block on the air temperature instead of the surface temperature
    const bool lab = K::general && CFG(boundflux_flag) == 3;
    const double T_surf = lab ? CL(T2m) : CL(T_top);
    const double T_freeze = func_T_freeze(quot(LAY(SAMSIM_A_S_ABS, 1), LAY(SAMSIM_A_M, 1)), CFG(salt_flag), x.tf_c3);
    GS(T_FREEZE) = T_freeze;
    CL(melt_thick) = 0.0;
    const double psi_s1 = LAY(SAMSIM_A_PSI_S, 1);
    // the reference evaluates func_freeboard first (:636); its value is only read under the melt condition (:637)
    if (psi_s1 < psi_s_top_min || T_surf >= T_freeze) {
      if (!c.psi_full) refill_psi_rows<K>(c, x);
      ST_COUNT(CT_L_FREEBOARD, (unsigned long long)__popcll(__ballot(1)));
      GS(FREEBOARD) = func_freeboard<K>(c, x);
      fb_valid = true;
      if (GS(FREEBOARD) > 0.0000000000001) {
        double thick1 = LAY(SAMSIM_A_THICK, 1);
        const double thick1_in = thick1;
        double melt_thick = 0.0;
        sub_melt_thick(LAY(SAMSIM_A_PSI_L, 1), psi_s1, LAY(SAMSIM_A_PSI_G, 1), LAY(SAMSIM_A_T, 1), T_freeze, T_surf, CL(fl_Q1),
                       CL(thick_snow), g.dt, melt_thick, thick1, g.thick_min);
        CL(melt_thick) = melt_thick;
        if (lab) CL(melt_thick) = dmax(CL(melt_thick), 0.0);
        if (CL(thick_snow) >= g.thick_min / 100.0 && CL(melt_thick) > 0.00000000001 && CL(melt_thick_snow) == 0.0) {
          // sub_melt_snow, mo_functions.f90:443-474
          double H_abs = LAY(SAMSIM_A_H_ABS, 1), m = LAY(SAMSIM_A_M, 1);
          const double shift = 1.0 / dmax(GS(PSI_G_SNOW), 0.01) * CL(melt_thick);
          if (shift >= CL(thick_snow)) {
            CL(melt_thick) = CL(melt_thick) - CL(thick_snow) * GS(PSI_G_SNOW);
            H_abs = H_abs + CL(H_abs_snow);
            m = m + CL(m_snow);
            thick1 = thick1 + (1.0 - GS(PSI_G_SNOW)) * CL(thick_snow);
            CL(thick_snow) = 0.0; CL(m_snow) = 0.0; CL(H_abs_snow) = 0.0;
          } else {
            H_abs = H_abs + shift / CL(thick_snow) * CL(H_abs_snow);
            CL(H_abs_snow) = CL(H_abs_snow) - shift / CL(thick_snow) * CL(H_abs_snow);
            m = m + shift / CL(thick_snow) * CL(m_snow);
            CL(m_snow) = CL(m_snow) - shift / CL(thick_snow) * CL(m_snow);
            thick1 = thick1 + shift - CL(melt_thick);
            CL(thick_snow) = CL(thick_snow) - shift;
            CL(melt_thick) = 0.0;
          }
          LAY(SAMSIM_A_H_ABS, 1) = H_abs;
          LAY(SAMSIM_A_M, 1) = m;
          fb_valid = false;
        }
        if (thick1 != thick1_in) { LAY(SAMSIM_A_THICK, 1) = thick1; fb_valid = false; }
      }
    }
  }

  // flushing, mo_grotz.f90:670-737
  c.flags &= ~COLF_FLUSHED;
  // freeboard (:670) is only read when flush_flag 4 / flush3 can run (:704-716): N_active > 2 and melt water present
  const bool flush_possible = ((CFG(flush_flag) == 5 || (K::general && (CFG(flush_flag) == 4 || CFG(flush_flag) == 6))) && Na > 2 &&
                               CL(melt_thick) + CL(melt_thick_snow) > 0.000000000001);
  if (flush_possible && !c.psi_full) refill_psi_rows<K>(c, x);
  if (flush_possible && !fb_valid) GS(FREEBOARD) = func_freeboard<K>(c, x);
  // (the accumulators sit in the scalar block: x + 0 is x, so nothing is read or written while nothing melts)
  if (CL(melt_thick) != 0.0) GS(MELT_OUT1) = GS(MELT_OUT1) + CL(melt_thick);
  if (CL(melt_thick_snow) != 0.0) GS(MELT_OUT2) = GS(MELT_OUT2) + CL(melt_thick_snow);
  CL(melt_thick) = CL(melt_thick) + CL(melt_thick_snow);
  if (CL(melt_thick_snow) > 0.0) {
    const double mts = CL(melt_thick_snow);
    double H1 = LAY(SAMSIM_A_H_ABS, 1), S1 = LAY(SAMSIM_A_S_ABS, 1), m1 = LAY(SAMSIM_A_M, 1);
    H1 = H1 + mts * rho_l * c_l * CL(T_snow);
    S1 = S1 + mts * rho_l * S_br_clamped(x.salt, CL(T_snow), GS(S_ABS_SNOW) / CL(m_snow));
    m1 = m1 + mts * rho_l;
    LAY(SAMSIM_A_H_ABS, 1) = H1; LAY(SAMSIM_A_S_ABS, 1) = S1; LAY(SAMSIM_A_M, 1) = m1;
    LAY(SAMSIM_A_THICK, 1) = LAY(SAMSIM_A_THICK, 1) + mts;
    LAY(SAMSIM_A_S_BU, 1) = S1 / m1;
  }
  if (flush_possible && GS(FREEBOARD) > 0.001) {
    if (CL(melt_thick) > 0.000000000001) {
      if (K::general && CFG(flush_flag) == 4) {  // melt water simply leaves the top layer, mo_grotz.f90:704-713
        const double T1 = LAY(SAMSIM_A_T, 1), m1 = LAY(SAMSIM_A_M, 1);
        LAY(SAMSIM_A_H_ABS, 1) = LAY(SAMSIM_A_H_ABS, 1) - CL(melt_thick) * rho_l * c_l * T1;
        LAY(SAMSIM_A_S_ABS, 1) = LAY(SAMSIM_A_S_ABS, 1) * (1.0 - (CL(melt_thick) * rho_l) / m1);
        LAY(SAMSIM_A_THICK, 1) = LAY(SAMSIM_A_THICK, 1) - CL(melt_thick);
        LAY(SAMSIM_A_M, 1) = m1 - CL(melt_thick) * rho_l;
      } else if (K::general && CFG(flush_flag) == 6) {  // :729-733
        if (CL(thick_snow) < g.thick_0) {
          flush4<K>(c, x);
          c.flags |= COLF_DIRTY;
          if (c.status) return;
        }
      } else {
        if (CL(melt_thick_snow) > 0.0) GS(FREEBOARD) = func_freeboard<K>(c, x);  // layer 1 changed since the last evaluation (:717)
        ST_COUNT(CT_L_FLUSH3, (unsigned long long)__popcll(__ballot(1)));
        flush3<K>(c, x);
        c.flags |= COLF_DIRTY | COLF_FLUSHED;
        if (c.status) return;
      }
    }
  }

  // tracer advection with this step's brine fluxes, mo_grotz.f90:742-747
  if (HAS_BGC) bgc_advection<K>(c, x);

  // layer dynamics, mo_grotz.f90:755-795
  if (Na > 1) {
    const double th1 = LAY(SAMSIM_A_THICK, 1);
    if (LAY(SAMSIM_A_PHI, Na) > psi_s_min || LAY(SAMSIM_A_PHI, Na - 1) <= psi_s_min / 2.0 || th1 / g.thick_0 > 1.5 ||
        th1 / g.thick_0 < 0.5) {
      ST_COUNT(CT_L_REGRID, (unsigned long long)__popcll(__ballot(1)));
      layer_dynamics<K>(c, x);
      c.flags |= COLF_DIRTY | COLF_REGRID;
      if (c.status) return;
    }
    Na = c.Na;
    const int kn = (Na + 1 < N) ? Na + 1 : N;
    if (Na < N && LAY(SAMSIM_A_THICK, kn) == 0.0) {  // scrub, :772-783
      LAY(SAMSIM_A_T, Na + 1) = g.T_bottom;
      LAY(SAMSIM_A_S_BU, Na + 1) = x.S_bu_bottom;
      LAY(SAMSIM_A_PSI_L, Na + 1) = 1.0;
      LAY(SAMSIM_A_PSI_S, Na + 1) = 0.0;
      if (HAS_BGC) for (int t = 0; t < x.n_bgc; ++t) BGC(t, Na + 1) = 0.0;
    }
  } else {
    if (LAY(SAMSIM_A_PHI, 1) > psi_s_min) {
      layer_dynamics<K>(c, x);
      c.flags |= COLF_DIRTY | COLF_REGRID;
      if (c.status) return;
    }
  }

  ST_MARK(ST_POST);
  // health check, mo_grotz.f90:808-819 (negative S_abs is clamped element-wise at the next sweep)
  if (c.neg_psi) STOPC(1337, 0);
  if (c.Na == 1) {
    const double v = LAY(SAMSIM_A_S_ABS, 1);
    if (v < 0.0) LAY(SAMSIM_A_S_ABS, 1) = 0.0;
  }
}

#ifndef SAMSIM_WAVES
#define SAMSIM_WAVES 1
#endif
template <class K>
__global__ void __launch_bounds__(SAMSIM_BLOCK, SAMSIM_WAVES) samsim_step_kernel(const DevParams *__restrict__ pp, double *__restrict__ lay, double *__restrict__ scal,
                                                                        double *__restrict__ spec, int32_t *__restrict__ n_active,
                                                                        int32_t *__restrict__ status, int32_t *__restrict__ err_layer,
                                                                        long long *__restrict__ err_step, long long *__restrict__ work,
                                                                        int32_t *__restrict__ flags, const double *__restrict__ f_sw,
                                                                        const double *__restrict__ f_lw, const double *__restrict__ f_T2m,
                                                                        const double *__restrict__ f_precip, double *__restrict__ out_lay,
                                                                        double *__restrict__ out_scal, int32_t *__restrict__ out_n_active,
                                                                        double *__restrict__ bgc, double *__restrict__ bgc_bot,
                                                                        double *__restrict__ bfl, double *__restrict__ out_bgc,
                                                                        double *__restrict__ out_bgc_bot, const int32_t *__restrict__ site) {
  const DevParams &p = *pp;
  const long long blk = (long long)blockIdx.x + p.block0;
  const long long col = blk * blockDim.x + threadIdx.x;
  if (col >= p.ncol) return;
  Ctx x;
  x.p = pp;
  x.f_sw = (gcdouble *)f_sw; x.f_lw = (gcdouble *)f_lw; x.f_T2m = (gcdouble *)f_T2m; x.f_precip = (gcdouble *)f_precip;
  x.out_lay = (gdouble *)out_lay; x.out_scal = (gdouble *)out_scal; x.out_n_active = (gint32 *)out_n_active;
  x.scal = (gdouble *)scal;
  x.err_layer = (gint32 *)err_layer; x.err_step = (__attribute__((address_space(1))) long long *)err_step;
  x.bgc = (gdouble *)bgc; x.bgc_bot = (gdouble *)bgc_bot; x.bfl = (gdouble *)bfl;
  x.out_bgc = (gdouble *)out_bgc; x.out_bgc_bot = (gdouble *)out_bgc_bot;
  x.n_bgc = K::bgc ? p.n_bgc : 0; x.bgc_total0 = p.bgc_total0;
  x.soff = (K::sites && p.nsites > 1) ? site[col] * p.flen : 0;
  x.dflq = (K::sites && p.ocean_dflq) ? p.ocean_dflq[col] : 0.0;
  x.ocean_sbu = K::sites && p.ocean_sbu != nullptr;
  x.out_col0 = p.out_col0; x.out_ncols = p.out_ncols;
  x.p17 = p.p17; x.p14 = p.p14; x.tf_c3 = p.tf_c3;
  x.S_bu_bottom = (K::general && (K::fixed ? K::tank_flag : p.cfg.tank_flag) == 2) ? scal[(size_t)SAMSIM_S_S_BU_BOTTOM * (size_t)p.ncol + (size_t)col] : p.cfg.S_bu_bottom;
  if (K::sites && p.ocean_sbu && (K::fixed ? K::tank_flag : p.cfg.tank_flag) != 2) x.S_bu_bottom = p.ocean_sbu[col];
  x.rho_bottom = func_density(p.cfg.T_bottom, p.cfg.S_bu_bottom);
  if ((K::fixed ? K::salt_flag : p.cfg.salt_flag) == 1) x.salt = Salt{-18.7, -0.519, -0.00535, -21.4, -0.886, -0.0170};
  else x.salt = Salt{-17.6, -0.389, -0.00362, -17.6, -0.389, -0.00362};

#if SAMSIM_STAMPS
  __shared__ unsigned long long st_lds[48];
  if (threadIdx.x < 48) st_lds[threadIdx.x] = 0;
  __syncthreads();
  x.st.acc = st_lds;
  x.st.t0 = __builtin_amdgcn_s_memtime();
#endif
  Col c;
  c.lay = (gdouble *)((gchar *)lay + (size_t)blk * ((size_t)p.cfg.nlayer * DEV_ROWB) + 4096);
  c.col = (unsigned)col;
  c.coff = (unsigned)col * 8u;
  c.lcoff = threadIdx.x * 8u;
  c.rstride = (unsigned)p.ncol * 8u;
  c.astride = (size_t)p.cfg.nlayer * (size_t)p.ncol * 8u;
  c.ncol = (size_t)p.ncol;
  c.N = p.cfg.nlayer;
  c.Na = n_active[col];
  c.status = status[col];
  c.frad = 0.0; c.neg_psi = false; c.buoy_s = 0.0; c.buoy_g = 0.0; c.psi_l_top = 1.0;
  c.flags = flags[col];
  c.spec = (gdouble *)spec;
  const size_t nc = (size_t)p.ncol;
  double *sc = scal + col;
#define SLOAD(field, idx) c.field = sc[(size_t)(idx) * nc]
  SLOAD(fl_q_bottom, SAMSIM_S_FL_Q_BOTTOM);
#undef SLOAD
  // row flags of the Rayleigh-number array (Ctx::rflag): at the start of a launch every row is valid (the last up sweep of a
  // launch stores all rows, as does samsim_set_state's full first sweep)
  __shared__ unsigned long long lds_rflag[SAMSIM_MAX_NLAYER / 64];
  x.rflag = (volatile lu64 *)lds_rflag;
  for (int w = 0; w < SAMSIM_MAX_NLAYER / 64; ++w) x.rflag[w] = ~0ull;
  // the LDS-resident scalars (each lane reads and writes only its own words: no barrier needed)
  __shared__ double lds_scal[LD_NSLOT * SAMSIM_BLOCK];
  c.ld = (ldouble *)lds_scal + threadIdx.x;
#define LLOAD(field, idx) CL(field) = sc[(size_t)(idx) * nc]
  LLOAD(grav_drain, SAMSIM_S_GRAV_DRAIN); LLOAD(grav_salt, SAMSIM_S_GRAV_SALT); LLOAD(grav_temp, SAMSIM_S_GRAV_TEMP);
  LLOAD(T_top, SAMSIM_S_T_TOP); LLOAD(fl_Q_snow, SAMSIM_S_FL_Q_SNOW); LLOAD(melt_thick, SAMSIM_S_MELT_THICK);
  CL(fl_Q1) = 0.0;
  LLOAD(albedo, SAMSIM_S_ALBEDO); LLOAD(fl_sw, SAMSIM_S_FL_SW); LLOAD(fl_lw, SAMSIM_S_FL_LW);
  LLOAD(T2m, SAMSIM_S_T2M); LLOAD(liquid_precip, SAMSIM_S_LIQUID_PRECIP); LLOAD(solid_precip, SAMSIM_S_SOLID_PRECIP);
  LLOAD(m_snow, SAMSIM_S_M_SNOW); LLOAD(H_abs_snow, SAMSIM_S_H_ABS_SNOW); LLOAD(thick_snow, SAMSIM_S_THICK_SNOW);
  LLOAD(T_snow, SAMSIM_S_T_SNOW); LLOAD(psi_s_snow, SAMSIM_S_PSI_S_SNOW); LLOAD(melt_thick_snow, SAMSIM_S_MELT_THICK_SNOW);
#undef LLOAD

  // uniform clock (mo_data: time, i, n_time_out, time_counter) evolves identically in every lane
  double time = p.time0;
  long long step = p.step0;
  int n_time_out = p.n_time_out0, tc = p.time_counter0;
  long long work_done = 0;
  for (long long s = 0; s < p.nsteps; ++s) {
    if ((K::fixed ? K::atmoflux_flag : p.cfg.atmoflux_flag) == 2) {
      if (time > time_input(tc)) tc = tc + 1;
      if (tc > p.flen) tc = p.flen;
    }
    const bool out_step = (n_time_out == p.cfg.i_time_out) || (step + 1 == 1);
    if (out_step) n_time_out = 0; else n_time_out = n_time_out + 1;
    const bool next_out = (n_time_out == p.cfg.i_time_out);
    // `output` prints the Rayleigh numbers the step BEFORE the output step drained with (mo_grotz.f90:340-398 runs before
    // fl_grav_drain), i.e. those this step's up sweep writes when the step after next is an output step: then, before an
    // output step and at the end of a launch the up sweep stores every row
    const bool next2_out = ((next_out ? 0 : n_time_out + 1) == p.cfg.i_time_out);
    x.ray_rows_all = next_out || next2_out || (s + 1 == p.nsteps);
    ST_MARK(ST_HEAD);
    if (!c.status) {
      c.step = step;
      work_done += c.Na;
      // The lane's column index is the same in every step, so every address formed from it -- some forty scalar-block, hand-over
      // and top-layer words per step -- is invariant in this loop: left alone the optimiser computes all of them once, as 64-bit
      // per-lane addresses, keeps them for the whole launch and, having no registers for them, spills them at the start and
      // reloads one from scratch memory (= HBM) at every use.  Passing the index through an empty asm makes them values of the
      // step: each is formed where it is used (two or three vector instructions) and nothing is carried.
      // (Round 2 passed the stored index through an empty asm; the allocator then kept it in scratch memory and reloaded it at every
      // step.  Now the lane number is read off the hardware -- two instructions, no memory -- and the index rebuilt from it.)
      unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(lane));
      c.lcoff = lane * 8u;
      c.col = (unsigned)(blk * SAMSIM_BLOCK) + lane;
      c.coff = c.col * 8u;
      const long long col_step = blk * SAMSIM_BLOCK + (long long)lane;
      column_step<K>(c, x, col_step, time, tc, out_step, next_out, s + 1 == p.nsteps);
    }
    time = time + p.cfg.dt;
    step = step + 1;
  }

  // (the column index and the scalar block's address are rebuilt from the lane number rather than carried through the time loop)
  unsigned lane_end = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane_end));
  const long long col_end = blk * SAMSIM_BLOCK + (long long)lane_end;
  n_active[col_end] = c.Na;
  flags[col_end] = c.flags;
  status[col_end] = c.status;
  work[col_end] += work_done;
  sc = scal + col_end;
#define SSTORE(field, idx) sc[(size_t)(idx) * nc] = c.field
  SSTORE(fl_q_bottom, SAMSIM_S_FL_Q_BOTTOM);
#undef SSTORE
#define LSTORE(field, idx) sc[(size_t)(idx) * nc] = CL(field)
  LSTORE(grav_drain, SAMSIM_S_GRAV_DRAIN); LSTORE(grav_salt, SAMSIM_S_GRAV_SALT); LSTORE(grav_temp, SAMSIM_S_GRAV_TEMP);
  LSTORE(albedo, SAMSIM_S_ALBEDO); LSTORE(fl_sw, SAMSIM_S_FL_SW); LSTORE(fl_lw, SAMSIM_S_FL_LW);
  LSTORE(T2m, SAMSIM_S_T2M); LSTORE(liquid_precip, SAMSIM_S_LIQUID_PRECIP); LSTORE(solid_precip, SAMSIM_S_SOLID_PRECIP);
  LSTORE(T_top, SAMSIM_S_T_TOP); LSTORE(fl_Q_snow, SAMSIM_S_FL_Q_SNOW); LSTORE(melt_thick, SAMSIM_S_MELT_THICK);
  LSTORE(m_snow, SAMSIM_S_M_SNOW); LSTORE(H_abs_snow, SAMSIM_S_H_ABS_SNOW); LSTORE(thick_snow, SAMSIM_S_THICK_SNOW);
  LSTORE(T_snow, SAMSIM_S_T_SNOW); LSTORE(psi_s_snow, SAMSIM_S_PSI_S_SNOW); LSTORE(melt_thick_snow, SAMSIM_S_MELT_THICK_SNOW);
#undef LSTORE
  sc[(size_t)SAMSIM_S_S_BU_BOTTOM * nc] = x.S_bu_bottom;
  // fl_rest = fl_lw + sensible + latent (both zero) with the forcing tables, mo_heat_fluxes.f90:112
  if ((K::fixed ? K::boundflux_flag : p.cfg.boundflux_flag) == 2 && (!K::general || (K::fixed ? K::atmoflux_flag : p.cfg.atmoflux_flag) == 2)) sc[(size_t)SAMSIM_S_FL_REST * nc] = CL(fl_lw) + 0.0 + 0.0;
#if SAMSIM_STAMPS
  ST_MARK(ST_TAIL);
  __syncthreads();
  if (threadIdx.x < 48 && st_lds[threadIdx.x]) atomicAdd(&g_stamps[threadIdx.x], st_lds[threadIdx.x]);
#endif
}

}  // namespace

// d_params: device copy of the parameter block `hp` (host copy, used here for the direct pointer arguments)
extern "C" hipError_t samsim_launch_step(const DevParams *d_params, const DevParams *hp, long long grid, hipStream_t stream) {
  const int block = SAMSIM_BLOCK;
  const samsim_config &g = hp->cfg;
  // one instantiation per flag set; tracers (bgc_flag 2) and several forcing sets / oceans select their own
  const bool tracers = g.bgc_flag == 2, sites = hp->nsites > 1 || hp->ocean_dflq || hp->ocean_sbu;
  auto kernel = samsim_step_kernel<KGeneric>;
  if (!sites && flags_match<KSheba>(g)) kernel = tracers ? samsim_step_kernel<KShebaBgc> : samsim_step_kernel<KSheba>;
  else if (!tracers && sites && flags_match<KSheba>(g)) kernel = samsim_step_kernel<KShebaSites>;
  else if (!sites && flags_match<KPlate>(g)) kernel = tracers ? samsim_step_kernel<KPlateBgc> : samsim_step_kernel<KPlate>;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(block), 0, stream, d_params, hp->lay, hp->scal, hp->spec,
                     hp->n_active, hp->status, hp->err_layer, hp->err_step, hp->work, hp->flags, hp->f_sw, hp->f_lw, hp->f_T2m,
                     hp->f_precip, hp->out_lay, hp->out_scal, hp->out_n_active, hp->bgc, hp->bgc_bot, hp->bfl, hp->out_bgc,
                     hp->out_bgc_bot, hp->site);
  return hipGetLastError();
}
