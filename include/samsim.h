/*
 * samsim.h -- C-ABI of the MI355X-native batched sea-ice column solver (libsamsim_hip.so).
 *
 * Drop-in boundary for ONE path of pgriewank/SAMSIM V2.0: the body of the time loop
 * mo_grotz.f90:182-835 (the per-timestep 1-D thermodynamic + brine-transport + regrid update of a
 * column), evaluated for `ncol` independent columns at once.  The reference has no FFI: its physics
 * are Fortran module procedures called from `grotz` on the `mo_data` module globals.  Each entry point
 * below cites the reference interface it replaces; INTEGRATION.md shows the iso_c_binding stub a
 * maintainer adds on the Fortran side.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a HOST pointer; the library owns device memory.
 *   - layer arrays are struct-of-arrays, column fastest:  a[(k-1)*ncol + c], k = 1..nlayer (Fortran
 *     layer index), c = 0..ncol-1.  Scalars: s[idx*ncol + c].
 *   - all arithmetic is IEEE float64 (`wp = SELECTED_REAL_KIND(12,307)`, mo_parameters.f90:33).
 *   - return value: 0 = ok, negative = API error (samsim_strerror).  Physics failures (the reference's
 *     `STOP n`) never abort the process: they are recorded per column in
 *     status[] and the column is frozen.
 *   - flags and time are uniform over columns (they are scalars of mo_data in the reference), so time,
 *     step index and the forcing-table cursor live in the handle, not per column.
 */
#ifndef SAMSIM_H
#define SAMSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMSIM_ABI_VERSION 5
#define SAMSIM_MAX_NLAYER 1024

/* -------- configuration: every flag of mo_data.f90:136-155 plus the scalars mo_init sets -------- */
typedef struct samsim_config {
  int32_t struct_size;          /* = sizeof(samsim_config); ABI check                                */
  int32_t testcase;             /* selects the time-dependent forcing of mo_grotz.f90:503-565:
                                   1 sub_test1 (T_top toggles), 2/6/9/34 sub_test2/6/9/34 (T2m schedule of the
                                   tank experiments), 3 sub_test3 (snow fall), 4/7 sub_test4 (fl_q_bottom),
                                   5 (S_abs reset at step 2); 8, 44, 45, 99, 101-105, 111 (lab tables, imposed
                                   snow) are refused; any other value (0, 33, 50): none               */
  int32_t nlayer, n_top, n_middle, n_bottom;          /* mo_data.f90:62-65                           */
  int32_t atmoflux_flag;        /* 1 Notz climatology, 2 forcing tables, 3 fixed fl_sw / fl_rest     */
  int32_t grav_flag;            /* 1 none, 2 Rayleigh-number gravity drainage, 3 simple              */
  int32_t prescribe_flag;       /* 1 none, 2 prescribed salinity profile (mo_grotz.f90:482-497)      */
  int32_t grav_heat_flag;       /* 1, 2                                                              */
  int32_t flush_heat_flag;      /* 1, 2                                                              */
  int32_t turb_flag;            /* 1, 2                                                              */
  int32_t salt_flag;            /* 1 sea salt, 2 NaCl                                                */
  int32_t boundflux_flag;       /* 1 cooling plate, 2 radiative balance, 3 lab air temperature T2m   */
  int32_t flush_flag;           /* 1 none, 4 melt water removed, 5 flush3, 6 flush4                  */
  int32_t flood_flag;           /* 1 none, 2 flood, 3 flood_simple                                   */
  int32_t bottom_flag;          /* 1, 2                                                              */
  int32_t debug_flag;           /* 1 (ignored)                                                       */
  int32_t precip_flag;          /* 0, 1                                                              */
  int32_t harmonic_flag;        /* 1, 2                                                              */
  int32_t tank_flag;            /* 1 ocean of fixed salinity, 2 tank: S_bu_bottom from the salt budget */
  int32_t albedo_flag;          /* 1, 2                                                              */
  int32_t lab_snow_flag;        /* 0 (1, snow in the lab with boundflux_flag 3, not supported)       */
  int32_t freeboard_snow_flag;  /* 0, 1                                                              */
  int32_t snow_flush_flag;      /* 0, 1                                                              */
  int32_t snow_precip_flag;     /* (echo only)                                                       */
  int32_t bgc_flag;             /* 1 none, 2 passive tracers advected with the brine (samsim_set_tracers)  */
  int32_t i_time_out;           /* INT(time_out/dt), mo_init.f90:2001                                */
  double  dt, thick_0, thick_min;            /* thick_min = thick_0/2, mo_init.f90:1994              */
  double  T_bottom, S_bu_bottom;
  double  k_snow_flush, max_flux_plate;      /* mo_parameters.f90:107,110                            */
  double  time_out, time_total;              /* echo / grav_* normalisation mo_grotz.f90:355-356     */
  double  alpha_flux_instable, alpha_flux_stable;   /* boundflux_flag 3, mo_heat_fluxes.f90:208-214   */
  double  m_total, S_total;                  /* tank_flag 2: water and salt in the tank, mo_init.f90:996-997 */
} samsim_config;

/* -------- per-column scalar slots (state + accumulators + output-only), s[idx*ncol + c] -------- */
enum samsim_scalar {
  SAMSIM_S_M_SNOW = 0, SAMSIM_S_H_ABS_SNOW, SAMSIM_S_S_ABS_SNOW, SAMSIM_S_THICK_SNOW,
  SAMSIM_S_PSI_S_SNOW, SAMSIM_S_PSI_L_SNOW, SAMSIM_S_PSI_G_SNOW, SAMSIM_S_T_SNOW, SAMSIM_S_PHI_S,
  SAMSIM_S_T_TOP, SAMSIM_S_MELT_THICK, SAMSIM_S_T2M, SAMSIM_S_LIQUID_PRECIP, SAMSIM_S_SOLID_PRECIP,
  SAMSIM_S_FL_Q_BOTTOM,
  SAMSIM_S_GRAV_DRAIN, SAMSIM_S_GRAV_SALT, SAMSIM_S_GRAV_TEMP,            /* accumulators            */
  SAMSIM_S_MELT_OUT1, SAMSIM_S_MELT_OUT2, SAMSIM_S_MELT_OUT3, SAMSIM_S_MELT_ERR,
  SAMSIM_S_FREEBOARD, SAMSIM_S_T_FREEZE, SAMSIM_S_ALBEDO, SAMSIM_S_FL_SW, SAMSIM_S_FL_LW,
  SAMSIM_S_MELT_THICK_SNOW, SAMSIM_S_FL_Q_SNOW,
  SAMSIM_S_ENERGY_STORED, SAMSIM_S_FRESHWATER, SAMSIM_S_TOTAL_RESIST,     /* vital signs             */
  SAMSIM_S_THICKNESS, SAMSIM_S_BULK_SALIN,
  SAMSIM_S_FL_REST,             /* bundled long-wave + turbulent flux (constant for atmoflux_flag 3) */
  SAMSIM_S_S_BU_BOTTOM,         /* salinity of the water below: evolves with tank_flag 2 (mo_grotz.f90:573-575);
                                   with tank_flag 1 an echo of cfg.S_bu_bottom that set_state ignores      */
  SAMSIM_S_DT2M, SAMSIM_S_PRECIP_SCALE,                                   /* ensemble perturbation   */
  SAMSIM_NSCAL
};

/* -------- per-column layer arrays, a[idx][k][c] -------- */
enum samsim_layer_array {
  SAMSIM_A_H_ABS = 0, SAMSIM_A_S_ABS, SAMSIM_A_M, SAMSIM_A_THICK,         /* prognostic (mo_data.f90:34-45) */
  SAMSIM_A_T, SAMSIM_A_PHI, SAMSIM_A_PSI_S, SAMSIM_A_PSI_L, SAMSIM_A_PSI_G,
  SAMSIM_A_S_BU, SAMSIM_A_S_BR, SAMSIM_A_RAY, SAMSIM_A_PERM,
  SAMSIM_A_FLUSH_V, SAMSIM_A_FLUSH_H,
  SAMSIM_NARR
};
#define SAMSIM_NPROG 4

/* Whole-column state as SoA blocks; used for initialisation, checkpoint/restart and parity tests.
 * (mo_data.f90:34-133; the reference keeps it in module globals allocated by sub_allocate,
 * mo_init.f90:2040-2090.) */
typedef struct samsim_state_soa {
  int64_t  ncol;
  int32_t  nlayer;
  int32_t  narr;      /* number of layer arrays present in `lay`: SAMSIM_NPROG (prognostic only; the carried
                         diagnostics are then initialised as mo_init.f90:1982-1990 does) or SAMSIM_NARR */
  double  *lay;       /* [narr][nlayer][ncol]                                                        */
  double  *scal;      /* [SAMSIM_NSCAL][ncol]                                                        */
  int32_t *n_active;  /* [ncol]                                                                      */
} samsim_state_soa;

/* uniform clock of the ensemble (mo_data: time, i, n_time_out, time_counter) */
typedef struct samsim_clock {
  double  time;          /* model time [s]                                                           */
  int64_t step;          /* number of completed time steps (= i-1 of mo_grotz.f90:182)               */
  int32_t n_time_out;    /* mo_grotz.f90:340,395-398                                                 */
  int32_t time_counter;  /* 1-based cursor into the 3-hourly tables, mo_grotz.f90:229-241            */
  int64_t n_outputs;     /* number of output points passed so far                                    */
} samsim_clock;

/* Snapshot taken at the reference's output point (mo_grotz.f90:340-398) for the column window
 * [col0, col0+ncols) chosen with samsim_set_output_window: exactly what mo_output.f90:129-144 writes. */
typedef struct samsim_output_soa {
  int64_t  ncols;
  int32_t  nlayer;
  int32_t  reserved;
  double  *lay;       /* [SAMSIM_NARR][nlayer][ncols]  (T, psi_s, psi_l, psi_g, S_bu, thick, ray, perm, flush_v, flush_h ...) */
  double  *scal;      /* [SAMSIM_NSCAL][ncols]         (freeboard, snow, vital signs, grav_*, T2m, T_top, melt_out) */
  int32_t *n_active;  /* [ncols]                                                                     */
  double   time;      /* model time of the snapshot                                                  */
  int64_t  step;      /* step index i (1-based) of the snapshot                                      */
} samsim_output_soa;

typedef struct samsim_handle samsim_handle;

/* sub_allocate (mo_init.f90:2040-2090) + the flag/scalar part of init (mo_init.f90:83-132, 1981-2031).
 * device: HIP device ordinal.  ncol columns of nlayer layers are allocated on it.  One handle holds at most
 * SAMSIM_MAX_NCOL columns (its [SAMSIM_NSCAL][ncol] scalar block is addressed with 32-bit byte offsets: SAMSIM_ERR_ARG
 * beyond), whatever nlayer is; larger ensembles take several handles (column ranges), which is also how they are spread
 * over GPUs. */
#define SAMSIM_MAX_NCOL ((int64_t)(((1ull << 32) - 1) / (8ull * SAMSIM_NSCAL)))
int samsim_create(const samsim_config *cfg, int64_t ncol, int32_t device, samsim_handle **h);

/* sub_input (mo_functions.f90:304-327): 3-hourly tables, time_input(k) = (k-1)*10800 s.
 * dT2m_col / precip_scale_col ([ncol] or NULL) perturb the ensemble: T2m_c = T2m + dT2m_c,
 * precip_c = precip * precip_scale_c (SURVEY.md section 8 d, cfg3). */
int samsim_set_forcing(samsim_handle *h, int32_t len, const double *fl_sw, const double *fl_lw,
                       const double *T2m, const double *precip,
                       const double *dT2m_col, const double *precip_scale_col);
/* Forcing for a grid of columns: nsites sets of the four tables (e.g. the nine ERA-interim sites
 * under input/ERA-interim of the reference), each array holding set s at [s*len, (s+1)*len); site_of_column[c] in
 * [0, nsites) selects the set column c reads.  samsim_set_forcing is the one-site case. */
int samsim_set_forcing_sites(samsim_handle *h, int32_t nsites, int32_t len, const double *fl_sw, const double *fl_lw,
                             const double *T2m, const double *precip, const int32_t *site_of_column,
                             const double *dT2m_col, const double *precip_scale_col);

/* The water below a grid of columns (the reference has one column and sets both as scalars of mo_data):
 * dfl_q_bottom_col[c] ([ncol] or NULL) is added to the oceanic heat flux sub_test4 assigns every step (testcases 4 and 7,
 * mo_testcase_specifics.f90:197-202: fl_q_bottom_c = -7 sin(2 pi t / year) + 7 + dfl_q_bottom_c); S_bu_bottom_col[c] ([ncol] or
 * NULL) replaces cfg.S_bu_bottom for column c wherever the reference reads S_bu_bottom (mass_transfer's ghost cell, flooding,
 * bottom turbulence, bottom growth; tank_flag 1 -- with tank_flag 2 the tank budget owns it, mo_grotz.f90:573-575).  A column
 * with offset 0 and cfg.S_bu_bottom is the reference's column.  NULL, NULL switches both off. */
int samsim_set_ocean(samsim_handle *h, const double *dfl_q_bottom_col, const double *S_bu_bottom_col);

/* initial state of init(testcase) (mo_init.f90:141-1978) or a checkpoint; col0 and s->ncol select a window.
 * The two perturbation slots SAMSIM_S_DT2M / SAMSIM_S_PRECIP_SCALE are owned by samsim_set_forcing: set_state
 * ignores them, get_state returns them. */
int samsim_set_state(samsim_handle *h, const samsim_state_soa *s, int64_t col0);
int samsim_get_state(samsim_handle *h, samsim_state_soa *s, int64_t col0);
int samsim_set_clock(samsim_handle *h, const samsim_clock *c);
int samsim_get_clock(samsim_handle *h, samsim_clock *c);

/* The time loop body, mo_grotz.f90:182-835, nsteps times for every column.  Asynchronous on the
 * handle's HIP stream; every getter synchronises. */
int samsim_step(samsim_handle *h, int64_t nsteps);
/* same, and returns the device time of the launch(es) measured with HIP events on the handle's stream */
int samsim_step_timed(samsim_handle *h, int64_t nsteps, double *kernel_ms);
/* nlaunches launches of nsteps steps each, enqueued back to back (no wait in between: the tail of one launch is filled by the
 * head of the next), and the device time of the whole sequence, measured with HIP events on the handle's streams (ABI 4) */
int samsim_steps_timed(samsim_handle *h, int64_t nsteps, int32_t nlaunches, double *device_ms);
int samsim_synchronize(samsim_handle *h);
/* which GPU the handle lives on: the ordinal passed to samsim_create and the PCI bus id HIP reports for it (pci_bus_id: buffer
 * of len >= 16 bytes, or NULL).  A multi-process run records it per rank so that a scaling record shows no GPU was shared.  ABI 5. */
int samsim_get_device(samsim_handle *h, int32_t *device, char *pci_bus_id, int32_t len);
/* How a step of a large ensemble is launched (no reference counterpart; results do not depend on it, bit for bit).  From
 * min_blocks 64-column blocks up (default 8 192 = 524 288 columns; 0 = never) a step runs as two concurrent launches on the
 * handle's two HIP streams, the first taking first_part_eighths/8 of the blocks (default 4): the workgroups of one launch finish
 * raggedly and a second launch in flight tops the chip up.  Every other entry point waits for both.  ABI 5. */
int samsim_set_launch_split(samsim_handle *h, int64_t min_blocks, int32_t first_part_eighths);

/* number of steps until (and including) the next output point of mo_grotz.f90:340 */
int64_t samsim_steps_to_output(samsim_handle *h);
int samsim_set_output_window(samsim_handle *h, int64_t col0, int64_t ncols);
/* output (mo_output.f90:116-146): latest snapshot; returns SAMSIM_ERR_NO_OUTPUT if none was taken */
int samsim_get_output(samsim_handle *h, samsim_output_soa *o);

/* the reference's STOP codes: status[c] = 0 or code; step/layer of first failure.  Every code is one of the
 * reference's own (16, 99, 345, 431, 1337, 7889, 9876, 21234, ...): the library stops no column the reference would not. */
int samsim_get_status(samsim_handle *h, int32_t *status, int64_t *step, int32_t *layer);
/* restart only: puts back what samsim_get_status returned for the columns [col0, col0+ncols) (samsim_set_state clears the
 * status of the columns it uploads), so that a column frozen by a STOP code stays frozen -- and reported -- after a
 * checkpoint / restart.  step and layer may be NULL. */
int samsim_set_status(samsim_handle *h, const int32_t *status, const int64_t *step, const int32_t *layer, int64_t col0,
                      int64_t ncols);
/* sum over columns of N_active accumulated over all steps taken (layer-cell updates) */
int samsim_get_work(samsim_handle *h, int64_t *layer_cell_updates, int64_t *column_steps);

/* Passive biogeochemical tracers (bgc_flag 2; mo_data.f90:181-193, bgc_advection mo_mass.f90:150-209): n_bgc tracers,
 * bgc_abs[t][k][c] = amount of tracer t in layer k of column c.  samsim_set_tracers fixes their number, the concentration
 * of the water below the ice (bgc_bottom, mo_init.f90:934-935) and -- tank_flag 2 only, else NULL -- the totals in the tank
 * (bgc_total, mo_init.f90:1013); it must be called before the first step.  The tracer state is zero until set
 * (init: bgc_abs(1,:) = bgc_bottom(:)*m(1), mo_init.f90:940).  samsim_get_tracer_output returns the snapshot taken at the
 * reference's output point for the output window: bgc_abs[n_bgc][nlayer][ncols], bgc_bottom[n_bgc][ncols]. */
#define SAMSIM_MAX_NBGC 8
int samsim_set_tracers(samsim_handle *h, int32_t n_bgc, const double *bgc_bottom, const double *bgc_total);
int samsim_set_tracer_state(samsim_handle *h, const double *bgc_abs, int64_t col0, int64_t ncols);
int samsim_get_tracer_state(samsim_handle *h, double *bgc_abs, double *bgc_bottom, int64_t col0, int64_t ncols);
/* restart only: the per-column concentration of the water below, bgc_bottom[n_bgc][ncols], as samsim_get_tracer_state
 * returned it (it evolves with the tank budget, mo_grotz.f90:575-577; samsim_set_tracers sets one value for all columns) */
int samsim_set_tracer_bottom(samsim_handle *h, const double *bgc_bottom, int64_t col0, int64_t ncols);
int samsim_get_tracer_output(samsim_handle *h, double *bgc_abs, double *bgc_bottom);

/* Ensemble statistics (what replaces "one column per .dat row", mo_output.f90:129-144, when the
 * run holds 10^5..10^6 columns): for each requested per-column scalar (enum samsim_scalar, or SAMSIM_STAT_N_ACTIVE) the
 * number of columns that have not failed, their mean, minimum, maximum and population standard deviation, reduced on the
 * device.  The vital signs and the freeboard hold the values of the last output point (mo_grotz.f90:192-223, 340-347). */
#define SAMSIM_STAT_N_ACTIVE (-1)
typedef struct samsim_stat {
  int64_t count;
  double  mean, min, max, std;
} samsim_stat;
int samsim_get_ensemble_stats(samsim_handle *h, int32_t nslots, const int32_t *slots, samsim_stat *out);

void samsim_destroy(samsim_handle *h);
const char *samsim_strerror(int code);
int samsim_abi_version(void);
int samsim_device_count(void);

enum samsim_error {
  SAMSIM_OK = 0,
  SAMSIM_ERR_ARG = -1, SAMSIM_ERR_UNSUPPORTED = -2, SAMSIM_ERR_HIP = -3, SAMSIM_ERR_NO_DEVICE = -4,
  SAMSIM_ERR_NO_OUTPUT = -5, SAMSIM_ERR_ABI = -6, SAMSIM_ERR_NOMEM = -7
};

#ifdef __cplusplus
}
#endif
#endif /* SAMSIM_H */
