/*
 * samsim_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement (plain C, float64, one column at a time) of
 * the reference hot path pgriewank/SAMSIM mo_grotz.f90:182-835 and everything it calls.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product (libsamsim_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it against (i) the reference's committed
 * known answers reference_output/Reference_testcase1_with_Version_2/dat_*.dat (3-decimal, all 72 rows),
 * and (ii) full-precision dumps of the unmodified reference physics built with flang under oracle/_ref
 * (oracle/build_ref.sh), committed as fixtures under tests/golden/.
 *
 * The batch API mirrors include/samsim.h one to one (oracle_* for samsim_*), so a parity test feeds
 * identical SoA inputs to both and compares the outputs.
 */
#ifndef SAMSIM_ORACLE_H
#define SAMSIM_ORACLE_H

#include "../include/samsim.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_handle oracle_handle;

int  oracle_create(const samsim_config *cfg, int64_t ncol, oracle_handle **h);
int  oracle_set_forcing(oracle_handle *h, int32_t len, const double *fl_sw, const double *fl_lw,
                        const double *T2m, const double *precip,
                        const double *dT2m_col, const double *precip_scale_col);
int  oracle_set_forcing_sites(oracle_handle *h, int32_t nsites, int32_t len, const double *fl_sw, const double *fl_lw,
                              const double *T2m, const double *precip, const int32_t *site_of_column,
                              const double *dT2m_col, const double *precip_scale_col);
int  oracle_set_state(oracle_handle *h, const samsim_state_soa *s, int64_t col0);
int  oracle_get_state(oracle_handle *h, samsim_state_soa *s, int64_t col0);
int  oracle_set_clock(oracle_handle *h, const samsim_clock *c);
int  oracle_get_clock(oracle_handle *h, samsim_clock *c);
int  oracle_step(oracle_handle *h, int64_t nsteps);
/* resume in the middle of a step: run only the part of the loop body after the output point
 * (mo_grotz.f90:405-835) once, for teacher forcing from a reference dump taken inside `output` */
int  oracle_step_part_b(oracle_handle *h);
int64_t oracle_steps_to_output(oracle_handle *h);
int  oracle_set_output_window(oracle_handle *h, int64_t col0, int64_t ncols);
int  oracle_get_output(oracle_handle *h, samsim_output_soa *o);
int  oracle_get_status(oracle_handle *h, int32_t *status, int64_t *step, int32_t *layer);
int  oracle_set_ocean(oracle_handle *h, const double *dfl_q_bottom_col, const double *S_bu_bottom_col);
int  oracle_set_status(oracle_handle *h, const int32_t *status, const int64_t *step, const int32_t *layer, int64_t col0, int64_t ncols);
int  oracle_get_work(oracle_handle *h, int64_t *layer_cell_updates, int64_t *column_steps);
int  oracle_set_tracers(oracle_handle *h, int32_t n_bgc, const double *bgc_bottom, const double *bgc_total);
int  oracle_set_tracer_state(oracle_handle *h, const double *bgc_abs, int64_t col0, int64_t ncols);
int  oracle_set_tracer_bottom(oracle_handle *h, const double *bgc_bottom, int64_t col0, int64_t ncols);
int  oracle_get_tracer_state(oracle_handle *h, double *bgc_abs, double *bgc_bottom, int64_t col0, int64_t ncols);
int  oracle_get_tracer_output(oracle_handle *h, double *bgc_abs, double *bgc_bottom);
int  oracle_get_ensemble_stats(oracle_handle *h, int32_t nslots, const int32_t *slots, samsim_stat *out);
void oracle_set_threads(oracle_handle *h, int nthreads);   /* columns over OpenMP threads (cpu_baseline) */
void oracle_destroy(oracle_handle *h);

/* function-level entry points (unit-level golden vectors) */
void   oracle_getT(int salt_flag, double H, double S_bu, double T_in, double *T, double *phi, int *status);
double oracle_func_S_br(int salt_flag, double T, double S_bu, int has_S_bu);
double oracle_func_ddT_S_br(int salt_flag, double T);
double oracle_func_T_freeze(double S_bu, int salt_flag);
double oracle_func_density(double T, double S);
double oracle_func_albedo(double thick_snow, double T_snow, double psi_l, double thick_min, int albedo_flag);
double oracle_func_k_snow(double m_snow, double thick_snow);
double oracle_func_freeboard(int N_active, const double *psi_s, const double *psi_g, const double *m,
                             const double *thick, double m_snow, int freeboard_snow_flag);
void   oracle_Expulsion(double phi, double thick, double m, double *psi_s, double *psi_l, double *psi_g, double *V_ex);
/* 1-based arrays of length N+1 (index 0 unused), as above */
void   oracle_flood_simple(double freeboard, double *S_abs1, double *H_abs1, double *m1, double *thick1, double T_bottom,
                           double S_bu_bottom, double psi_g_snow, double *H_abs_snow, double *m_snow, double *thick_snow);
void   oracle_fl_grav_drain_simple(int N, int N_active, int harmonic_flag, const double *psi_s, const double *psi_l,
                                   const double *thick, const double *S_br, double *S_abs, double *ray);
void   oracle_sub_notzflux(double time, double *fl_sw, double *fl_rest);

#ifdef __cplusplus
}
#endif
#endif
