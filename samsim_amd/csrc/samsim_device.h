// samsim_device.h -- device-side data layout shared by the kernels and the C-ABI host code.
//
// HBM layout (one allocation per handle, sized for ncol columns of nlayer layers):
//   lay  [ncol/64][nlayer][DEV_NARR][64] float64   layer arrays per 64-column block (SAMSIM_BLOCKED below): a wave's 64
//                                           lanes read 512 contiguous bytes for every (array, layer) pair, the sixteen arrays of
//                                           a layer row are 8 KiB, a wave's whole block is one contiguous piece
//   scal [SAMSIM_NSCAL][ncol]     float64   per-column scalars
//   n_active/status/err_layer [ncol] int32, err_step/work [ncol] int64
// One GPU thread owns one column for the whole launch; a launch advances every column `nsteps` time steps.
#ifndef SAMSIM_DEVICE_H
#define SAMSIM_DEVICE_H

#include "../../include/samsim.h"

// device-internal layer arrays: the public ones (enum samsim_layer_array) followed by scratch
enum dev_layer_array {
  D_V_EX = SAMSIM_NARR,   // scratch row: equivalent resistance R(k) of flush3 (mo_flush.f90:137-145), after the up sweep;
                          // before it, D_HR = the half resistance thick/(2k) the down sweep hands to the up sweep
  D_HR = D_V_EX,
  DEV_NARR
};

// SAMSIM_BLOCKED 1 (default): the layer block is stored per 64-column block -- [block][layer][array][64 lanes] -- so that a wave's
// column block is one contiguous piece (DEV_NARR * 512 B per layer): every array of a layer row sits within the immediate offset
// range of one row address, a row address is scalar arithmetic, and the lane's offset is the same register for the whole launch.
// 0 = [array][layer][column].  The boundary (samsim_set_state / samsim_get_state) keeps [array][layer][column] either way.
#define DEV_NARR_C 16                                  // = DEV_NARR
static_assert(DEV_NARR == DEV_NARR_C, "DEV_NARR_C");
#define DEV_ROWB ((size_t)DEV_NARR_C * 512)            // bytes of one layer of one 64-column block
// doubles in the layer block of a handle, and the position of element (array a, 0-based layer k0, column col)
#define DEV_LAY_DOUBLES(N, ncol) ((((size_t)(ncol) + 63) / 64) * (size_t)(N) * DEV_NARR_C * 64)
#define DEV_LAY_INDEX(a, k0, col, N, ncol) ((((size_t)(col) >> 6) * (size_t)(N) + (size_t)(k0)) * (DEV_NARR_C * 64) + (size_t)(a) * 64 + ((size_t)(col) & 63))

// per-column values handed from the up sweep of step n to the top-layer prologue of step n+1: [DEV_NSPEC][ncol]
enum dev_spec {
  SP_MINP = 0, SP_STP, SP_ST,               // suffix min(perm), sum(thick/perm), sum(thick) over layers 2..N_active-1
  SP_BOT, SP_BOTTERM, SP_PERM_BOT, SP_SBR_BOT,  // bottom-layer terms of the Rayleigh number
  SP_BUOY_S, SP_MIN_PSI_S,                  // partial SUM(psi_s*thick), MIN(psi_s) over layers 2..N_active
  // from the down sweep of a step to func_freeboard later in the same step (written where the sweep stores the volume-fraction rows):
  SP_FB_A2, SP_FB_G2,                       // SUM(psi_s*thick), SUM(psi_g*thick) over layers 2..N_active, top -> bottom
  // from the first sweep of a step to flood / refresh_ray_top of the same step (written only where the snow load makes flooding
  // possible): harmonic-mean permeability of the whole column and its total thickness, as mo_flood.f90:66-80 forms them
  SP_FL_HP, SP_FL_SALL,
  DEV_NSPEC,
  // Once a column has been flooded in the fused order (column_step) the scan rows above have served; until the up sweep rewrites
  // them they carry the flooded top layer from the flooding block to the down sweep (COLF_FLOODED), and SP_FL_HP / SP_FL_SALL the
  // bottom layer's increments of an instant flooding (COLF_FLOOD_DEEP): memory instead of six values held across the radiation
  // header and the Beer-law pass.
  SP_FLD_S1 = SP_MINP, SP_FLD_H1 = SP_STP, SP_FLD_M1 = SP_ST, SP_FLD_TH1_BEFORE = SP_BOT
};

// rows of the tracer flux block: what the reference collects in fl_brine_bgc(N+1, N+1) (mo_data.f90:183)
enum dev_bgc_flux {
  BFL_E = 0,   // -fl_m(k+1) of expulsion_flux: layer k -> k+1                       (mo_grotz.f90:316-320)
  BFL_D,       // fl_down(k) of fl_grav_drain: layer k -> ocean                      (mo_grav_drain.f90:179)
  BFL_U,       // fl_up(k) of fl_grav_drain: layer k+1 -> k (k = N_active: ocean -> N_active)  (:181-183)
  BFL_V,       // flush_v(k) of flush3: layer k -> k+1                                (mo_flush.f90:171-173)
  BFL_H,       // flush_h(k) of flush3: layer k -> N_active                           (:169)
  BFL_NROW
};

// per-column flag bits
#define COLF_DIRTY 1    // prognostic layers changed since the last up sweep: the next step runs the full S1 sweep
#define COLF_RESTART 2  // first step after samsim_set_state: RAY holds the previous Rayleigh numbers
#define COLF_FLOODED 32     // (within a step) the fused order flooded this column: the flooded top layer waits in the hand-over block
#define COLF_FLOOD_DEEP 16  // (within a step) the fused order flooded this column below neg_free: the bottom layer's increments wait in the hand-over block
#define COLF_FLUSHED 64  // flush3 ran in the previous step (a wave with such a column leaves the next step's first sweep to that step: sweep_up_fused LITE)
#define COLF_REGRID 8   // layer_dynamics changed the grid in the previous step: the full first sweep checks the thickness rule again
#define COLF_REGULAR 4  // the thicknesses of layers 2..N_active follow the grid rule (thick_0, and one common value in the middle
                        // block): the sweeps take them from two loaded values instead of the array (checked by the full first
                        // sweep after samsim_set_state and after every regrid)

struct DevParams {
  samsim_config cfg;
  double *lay;
  double *scal;
  int32_t *n_active;
  int32_t *status;
  int32_t *err_layer;
  long long *err_step;
  long long *work;
  double *spec;         // [DEV_NSPEC][ncol]
  int32_t *flags;       // [ncol] COLF_*
  const double *f_sw, *f_lw, *f_T2m, *f_precip;
  int32_t flen;
  int32_t nsites;               // forcing sets; table s starts at s*flen
  const int32_t *site;          // [ncol] set of each column (nsites > 1)
  const double *ocean_dflq;     // [ncol] offset on the oceanic heat flux of sub_test4, or null (samsim_set_ocean)
  const double *ocean_sbu;      // [ncol] salinity of the water below each column (tank_flag 1), or null
  long long ncol;
  // this launch's share of the columns: 64-column blocks [block0, block0 + grid) (a step of a large ensemble is two launches on two
  // streams, samsim_capi.cpp launch())
  long long block0;
  // uniform clock at launch (mo_data: time, i, n_time_out, time_counter)
  double time0;
  long long step0;
  int32_t n_time_out0;
  int32_t time_counter0;
  long long nsteps;
  // output snapshot window
  double *out_lay;       // [SAMSIM_NARR][nlayer][out_ncols]
  double *out_scal;      // [SAMSIM_NSCAL][out_ncols]
  int32_t *out_n_active; // [out_ncols]
  long long out_col0, out_ncols;
  // passive tracers (bgc_flag 2): bgc [n_bgc][nlayer][ncol] amounts, bgc_bot [n_bgc][ncol] concentration of the water below,
  // bfl [BFL_NROW][nlayer][ncol] this step's brine fluxes (the sparse rows of fl_brine_bgc), snapshot for the output window
  double *bgc, *bgc_bot, *bfl, *out_bgc, *out_bgc_bot;
  int32_t n_bgc;
  double bgc_total0;     // bgc_total(1): the tank budget of mo_grotz.f90:576 reads tracer 1 only
  // host-evaluated constants: 10**(-17), 10**(-14) (mo_grav_drain.f90:105,112) and the float32 product
  // 5.33*10.0**(-7.0) of func_T_freeze (mo_functions.f90:246)
  double p17, p14, tf_c3;
};

#endif
