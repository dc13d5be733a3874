/*
 * samsim_oracle.c -- TEST INFRASTRUCTURE (see samsim_oracle.h).  Parity status: PINNED (header).
 *
 * Plain-C, float64, single-column restatement of the reference's per-timestep update
 * (pgriewank/SAMSIM V2.0, Fortran 90).  It follows the reference statement by statement: same
 * operation order, same O(N^2) loops, same float32-literal quirks, so that it agrees with the
 * flang-built reference to round-off.  Each function cites the reference file:line it follows.
 * Compile with -O2 -ffp-contract=off (no FMA contraction, like flang -O2 on baseline x86-64).
 *
 * Layer arrays are 1-based (index 0 unused) to keep the Fortran indices.
 */
#include "samsim_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants: mo_parameters.f90:38-112 */
/* default-REAL (float32) parameters, mo_parameters.f90:38-39 */
static const double pi_f   = (double)3.1415f;
static const double grav_f = (double)9.8061f;
static const double k_s = 2.2, k_l = 0.523;
static const double c_s = 2020.0, c_s_beta = 7.6973, c_l = 3400.0;
static const double rho_s = 920.0, rho_l = 1028.0, latent_heat = 333500.0, zeroK = 273.15;
/* `0.8_wp*1e-3` etc. multiply by a float32 literal, mo_parameters.f90:56,57,59 */
#define bbeta   (0.8 * (double)1e-3f)
#define mu      (2.55 * (double)1e-3f)
#define kappa_l (k_l / rho_l / c_l)
#define sigma   (5.6704 * (double)1e-8f)
static const double psi_s_min = 0.05, neg_free = -0.05;
static const double x_grav = 0.000584, ray_crit = 4.89;
static const double para_flush_horiz = 1.0;
static const double para_flush_gamma = 0.9;                 /* mo_parameters.f90:81 */
static const double psi_s_top_min = 0.40, ratio_flood = 1.50, ref_salinity = 34.0;
static const double rho_snow = 330.0, gas_snow_ice2 = 0.20;
static const double emissivity_ice = 0.95, emissivity_snow = 1.00, penetr = 0.30, extinc = 2.00;
#define Turb_A (0.1 * 0.05 * rho_l / 86400.0)
static const double Turb_B = 0.05;

typedef struct column {
  const samsim_config *cfg;
  int N;                         /* Nlayer */
  int N_active;
  double *H_abs, *S_abs, *m, *thick;
  double *T, *phi, *psi_s, *psi_l, *psi_g, *S_bu, *S_br, *H, *V_ex;
  double *fl_Q, *fl_m, *fl_rad, *ray, *perm, *flush_v, *flush_h;
  /* snow, mo_data.f90:92-103 */
  double m_snow, H_abs_snow, S_abs_snow, thick_snow, psi_s_snow, psi_l_snow, psi_g_snow, T_snow, phi_s;
  double liquid_precip, solid_precip, fl_Q_snow;
  double T_top, T2m, fl_q_bottom, melt_thick, melt_thick_snow, melt_thick_snow_old;
  double melt_thick_output[3], melt_err;
  double freeboard, T_freeze, albedo, fl_sw, fl_lw, fl_rest;
  double grav_drain, grav_salt, grav_temp;
  double energy_stored, freshwater, total_resist, thickness, bulk_salin;
  double dT2m, precip_scale;
  double S_bu_bottom;            /* salinity of the water below the ice: cfg value, or the tank budget (tank_flag 2) */
  double dflq;                   /* samsim_set_ocean: offset on the oceanic heat flux of sub_test4 (0 unless given) */
  int ocean_sbu;                 /* samsim_set_ocean: S_bu_bottom is this column's own value (survives set_state) */
  double ocean_sbu_value;
  /* passive tracers, mo_data.f90:181-193 */
  int n_bgc;
  double *bgc[SAMSIM_MAX_NBGC];  /* bgc_abs(:, t), 1-based */
  double bgc_bottom[SAMSIM_MAX_NBGC];
  const double *bgc_total;       /* tank totals (shared) */
  double *flb;                   /* fl_brine_bgc(N+1, N+1): FLB(i, j) = brine from layer i to layer j this step */
  double *snap_bgc;              /* [n_bgc][N] */
  double snap_bgc_bottom[SAMSIM_MAX_NBGC];
  /* clock */
  double time;
  int64_t step;                  /* completed steps; i = step+1 */
  int n_time_out, time_counter;
  int64_t n_outputs;
  /* forcing (shared) */
  int flen;
  const double *fl_sw_input, *fl_lw_input, *T2m_input, *precip_input;
  /* error state */
  int status; int64_t err_step; int err_layer;
  int64_t work;                  /* sum of N_active over steps */
  /* output snapshot */
  int snap_valid; double snap_time; int64_t snap_step; int snap_N_active;
  double *snap_lay;              /* [SAMSIM_NARR][N] */
  double snap_scal[SAMSIM_NSCAL];
} column;

#define FLB(i, j) c->flb[(size_t)(i) * (size_t)(c->N + 2) + (size_t)(j)]
#define STOP(code, layer) do { if (!c->status) { c->status = (code); c->err_step = c->step + 1; c->err_layer = (layer); } return; } while (0)
#define CHECK() do { if (c->status) return; } while (0)

#ifndef POW4
#define POW4(x) pow((x), 4.0)
#endif
#ifndef POW3
#define POW3(x) ((x) * (x) * (x))
#endif
static double dmax(double a, double b) { return a > b ? a : b; }   /* Fortran MAX/MIN for non-NaN args */
static double dmin(double a, double b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ mo_thermo_functions.f90 */

/* func_S_br, mo_thermo_functions.f90:308-360 (T**2._wp, T**3._wp are real powers -> pow) */
double oracle_func_S_br(int salt_flag, double T, double S_bu, int has_S_bu) {
  double c1 = 0.0, c2, c3, c4, S_br;
  if (salt_flag == 1) { c2 = -18.7; c3 = -0.519; c4 = -0.00535; }
  else                { c2 = -17.6; c3 = -0.389; c4 = -0.00362; }
  S_br = c1 + c2 * T + c3 * (T * T) + c4 * POW3(T);
  if (has_S_bu) { if (S_br < S_bu) S_br = S_bu; }
  return S_br;
}
#define S_BR(T)        oracle_func_S_br(c->cfg->salt_flag, (T), 0.0, 0)
#define S_BR2(T, Sbu)  oracle_func_S_br(c->cfg->salt_flag, (T), (Sbu), 1)

/* func_ddT_S_br, mo_thermo_functions.f90:380-414: sea salt uses the OLD (Notz) coefficients */
double oracle_func_ddT_S_br(int salt_flag, double T) {
  double c2, c3, c4, T_crit = -20.0, d;
  if (salt_flag == 1) { c2 = -21.4; c3 = -0.886; c4 = -0.0170; }
  else                { c2 = -17.6; c3 = -0.389; c4 = -0.00362; }
  d = c2 + 2.0 * c3 * T + 3.0 * c4 * pow(T, 2.0);
  if (T < T_crit) d = c2 + 2.0 * c3 * T_crit + 3.0 * c4 * pow(T_crit, 2.0);
  return d;
}

/* getT, mo_thermo_functions.f90:62-143.  *status = 99 on non-convergence (STOP 99, :110-123). */
void oracle_getT(int sf, double H, double S_bu, double T_in, double *T_out, double *phi_out, int *status) {
  double T, phi = *phi_out, T_0, f, ddT_f, T_fr, sb;
  int i;
  T = H / c_l;                                                                        /* :80 */
  if (oracle_func_S_br(sf, T, S_bu, 1) > S_bu && S_bu > 0.001) {                      /* :82 */
    T_fr = -1.0;
    /* :87  ABS(func_S_br(T_fr)/S_bu-1.0)>0.0001 : float32 literals */
    while (fabs(oracle_func_S_br(sf, T_fr, 0, 0) / S_bu - 1.0) > (double)0.0001f) {
      T_0 = T_fr;
      f = oracle_func_S_br(sf, T_0, 0, 0) - S_bu;
      ddT_f = oracle_func_ddT_S_br(sf, T_0);
      T_fr = T_0 - f / ddT_f;
    }
    T_0 = T_in;                                                                       /* :94 */
    sb = oracle_func_S_br(sf, T_0, 0, 0);
    f = -latent_heat - H + latent_heat * S_bu / dmax(sb, 0.000000001) + c_s * T_0 + c_s_beta * T_0 * T_0 / 2.0;
    ddT_f = c_s + c_s_beta * T_0 - latent_heat * S_bu * oracle_func_ddT_S_br(sf, T_0) / dmax(pow(sb, 2.0), 0.0000000001);
    T = T_0 - f / ddT_f;
    i = 0;
    while (fabs(f) > 1.0) {                                                           /* :99 */
      T_0 = T;
      if (T_0 > 0.0 || T_0 < -200.0) T_0 = T_fr;
      sb = oracle_func_S_br(sf, T_0, 0, 0);
      f = -latent_heat - H + latent_heat * S_bu / dmax(sb, 0.0000000001) + c_s * T_0 + c_s_beta * T_0 * T_0 / 2.0;
      ddT_f = c_s + c_s_beta * T_0 - latent_heat * S_bu * oracle_func_ddT_S_br(sf, T_0) / dmax(sb * sb, 0.0000000001);
      T = T_0 - f / ddT_f;
      i = i + 1;
      if (i == 260) { if (status) *status = 99; break; }
    }
    phi = 1.0 - S_bu / oracle_func_S_br(sf, T, S_bu, 1);                              /* :125 */
  } else if (S_bu < 0.001) {                                                          /* :127 */
    if (H > 0.0) { phi = 0.0; T = H / c_l; }
    else if (H <= -latent_heat) { phi = 1.0; T = (H + latent_heat) / c_s; }
    else if (H <= 0.0 && -latent_heat < H) { T = 0.0; phi = -H / latent_heat; }
  } else {
    phi = 0.0;                                                                        /* :139 */
  }
  *T_out = T; *phi_out = phi;
}

static void getT(column *c, double H, double S_bu, double T_in, double *T, double *phi, int layer) {
  int st = 0;
  oracle_getT(c->cfg->salt_flag, H, S_bu, T_in, T, phi, &st);
  if (st && !c->status) { c->status = st; c->err_step = c->step + 1; c->err_layer = layer; }
}

/* Expulsion, mo_thermo_functions.f90:157-187 */
void oracle_Expulsion(double phi, double thick, double m, double *psi_s, double *psi_l, double *psi_g, double *V_ex) {
  double V_s = m * phi / rho_s;
  double V_l = m * (1.0 - phi) / rho_l;
  if (V_s + V_l > thick) *V_ex = V_l + V_s - thick; else *V_ex = 0.0;
  *psi_s = V_s / thick;
  *psi_l = (V_l - *V_ex) / thick;
  *psi_g = (thick - V_l - V_s + *V_ex) / thick;
  if (*psi_l < 0.0) *psi_l = 0.0;
  if (*psi_g < 0.0) *psi_g = 0.0;
}

/* sub_fl_Q, mo_thermo_functions.f90:201-223 */
static double sub_fl_Q(double psi_s_1, double psi_l_1, double psi_g_1, double thick_1, double T_1,
                       double psi_s_2, double psi_l_2, double psi_g_2, double thick_2, double T_2) {
  double k_1 = psi_s_1 * k_s + psi_l_1 * k_l + psi_g_1 * 0.0;
  double k_2 = psi_s_2 * k_s + psi_l_2 * k_l + psi_g_2 * 0.0;
  double R = thick_1 / (2.0 * k_1) + thick_2 / (2.0 * k_2);
  return (T_2 - T_1) / R;
}

/* sub_fl_Q_0, mo_thermo_functions.f90:238-266 */
static double sub_fl_Q_0(double psi_s, double psi_l, double psi_g, double thick, double T, double T_bound, int direct_flag) {
  double k = psi_s * k_s + psi_l * k_l + psi_g * 0.0;
  double R = thick / (2.0 * k);
  if (direct_flag == 1) return (T_bound - T) / R;
  return (T - T_bound) / R;
}

/* ------------------------------------------------------------------ mo_functions.f90 */

/* func_density, mo_functions.f90:51-62 */
double oracle_func_density(double T, double S) {
  double density_0 = 999.842594 + 6.8 / 100.0 * T;
  double A = 0.825, B = -5.7 / 1000.0;
  return density_0 + A * S + B * pow(dmax(S, 0.0), 1.5);
}

/* func_freeboard, mo_functions.f90:79-130 (O(N^2) as written); arrays 1-based */
double oracle_func_freeboard(int N_active, const double *psi_s, const double *psi_g, const double *m,
                             const double *thick, double m_snow, int freeboard_snow_flag) {
  double freeboard, snowmass, test1, test2, s1, s2, sm;
  int k, j;
  snowmass = (freeboard_snow_flag == 0) ? m_snow : 0.0;
  s1 = 0.0; for (j = 1; j <= N_active; j++) s1 += psi_s[j] * thick[j];
  s2 = 0.0; for (j = 1; j <= N_active; j++) s2 += psi_g[j] * thick[j];
  if (snowmass > s1 * (rho_l - rho_s) + s2 * rho_l) {                                /* :96 */
    test2 = s1 * (rho_l - rho_s) + s2 * rho_l;
    freeboard = test2 - snowmass;
    freeboard = freeboard / rho_l;
  } else {
    test1 = 0.0; test2 = 1.0; k = 0;
    while (test1 < test2) {                                                          /* :113 */
      k = k + 1;
      s1 = 0.0; for (j = k + 1; j <= N_active; j++) s1 += psi_s[j] * thick[j];
      s2 = 0.0; for (j = k + 1; j <= N_active; j++) s2 += psi_g[j] * thick[j];
      test2 = s1 * (rho_l - rho_s) + s2 * rho_l;
      sm = 0.0; for (j = 1; j <= k; j++) sm += m[j];
      test1 = sm + snowmass;
      if (k >= N_active) break;  /* reference terminates here because test2 = 0 <= test1 */
    }
    sm = 0.0; for (j = 1; j <= k - 1; j++) sm += m[j];
    test1 = sm + snowmass;
    freeboard = test2 - test1 + (rho_l - m[k] / thick[k]) * thick[k];               /* :124 */
    freeboard = freeboard / rho_l;
    sm = 0.0; for (j = 1; j <= k - 1; j++) sm += thick[j];
    freeboard = freeboard + sm;
  }
  return freeboard;
}

/* func_albedo, mo_functions.f90:157-208 (float32 literals :163-167, :170, :199) */
double oracle_func_albedo(double thick_snow, double T_snow, double psi_l, double thick_min, int albedo_flag) {
  double albedo;
  const double ice_dry = (double)0.75f, ice_wet = (double)0.6f, snow_dry = (double)0.85f,
               snow_wet = (double)0.75f, water = (double)0.2f;
  if (thick_snow > thick_min) {
    if (T_snow < (double)(-0.01f)) albedo = snow_dry; else albedo = snow_wet;
    albedo = ice_dry + (albedo - ice_dry) * dmin(1.0, thick_snow / 0.3);
  } else {
    if (psi_l > 0.9) albedo = water;
    else if (psi_l > 0.6) albedo = ice_wet + (water - ice_wet) * ((psi_l - 0.6) / 0.3);
    else if (psi_l > 0.2) albedo = ice_wet;
    else albedo = ice_dry;
  }
  if (albedo_flag == 1) {
    if (thick_snow > thick_min) {
      if (T_snow < (double)(-0.01f)) albedo = snow_dry; else albedo = snow_wet;
    } else {
      if (psi_l < (double)0.8f) albedo = ice_dry; else albedo = water;
    }
  }
  return albedo;
}

/* func_T_freeze, mo_functions.f90:239-250: the default-REAL products are evaluated in float32 */
double oracle_func_T_freeze(double S_bu, int salt_flag) {
  if (salt_flag == 2) {
    const float c3 = 5.33f * powf(10.0f, -7.0f);
    return -0.0592 * S_bu - (double)9.37f * pow(S_bu, 2.0) - (double)c3 * pow(S_bu, 3.0);
  } else {
    const float a = 1.710523f * 1e-3f, b = 2.154996f * 1e-4f;
    return -0.0575 * S_bu + (double)a * pow(S_bu, 1.5) - (double)b * pow(S_bu, 2.0);
  }
}

/* sub_notzflux, mo_functions.f90:270-289 (47.9, 53.1, 60., 300., 360 are default-REAL / integer literals) */
static void sub_notzflux(double time, double *fl_sw, double *fl_rest) {
  double day = time / 86400.0, a, b;
  while (day > 360.0) day = day - 360.0;
  a = (day - 164.0) / (double)47.9f;
  b = (day - 206.0) / (double)53.1f;
  *fl_sw = 314.0 * exp(-0.5 * (a * a));
  *fl_rest = 118.0 * exp(-0.5 * (b * b)) + 179.0;
  if (day < 60.0 || day > 300.0) *fl_sw = 0.0;
}

/* sub_test2, sub_test9, sub_test6: air-temperature schedules of the tank experiments, mo_testcase_specifics.f90:99-136,211-232 */
static void sub_test2(double time, double *T2m) {
  if (time > 86400.0 * 25.0) *T2m = 15.0;
  else if (time > 86400.0 * 15.0) *T2m = 1.0;
}
static void sub_test9(double time, double *T2m) {
  if (time < 19.75 * 3600.0) *T2m = 0.0;
  else if (time < 86400.0 * 3.0 + 2.25 * 3600.0) *T2m = -15.0;
  else *T2m = 1.0;
}
static void sub_test34(double time, double *T2m) {                                   /* mo_testcase_specifics.f90:146-162 */
  if (time < 2.0 * 3600.0) *T2m = 0.0;
  else if (time < 86400.0 * 5.0) *T2m = -15.0;
  else if (time < 86400.0 * 7.0) *T2m = -5.0;
  else *T2m = 1.0;
}
static void sub_test6(double time, double *T2m) {
  static const double t[8] = {1714.0, 1676.0, 1525.0, 1483.0, 1385.0, 1349.0, 1160.0, 1100.0};
  static const double v[8] = {-19.0, -5.0, -18.0, -5.0, -18.0, -5.0, -18.0, -5.0};
  for (int i = 0; i < 8; i++) if (time > t[i] * 60.0) { *T2m = v[i]; return; }
}

/* sub_turb_flux, mo_functions.f90:347-363 */
static double sub_turb_flux(double T_bottom, double S_bu_bottom, double T, double *S_abs, double m, double dt) {
  double turb = Turb_A * exp(Turb_B * (-oracle_func_density(T_bottom, S_bu_bottom) + oracle_func_density(T, *S_abs / m))) * dt;
  *S_abs = *S_abs - turb * (*S_abs / m - S_bu_bottom);
  return turb;   /* the tracers of the bottom layer mix with the same coefficient, :358-360 */
}

/* sub_melt_thick, mo_functions.f90:386-428 */
static void sub_melt_thick(double psi_l, double psi_s, double psi_g, double T, double T_freeze, double T_top,
                           double fl_Q, double thick_snow, double dt, double *melt_thick, double *thick, double thick_min) {
  *melt_thick = 0.0;
  if (thick_snow < thick_min && T_top >= T_freeze) {
    *melt_thick = -fl_Q - 2.0 * (psi_l * k_l + psi_s * k_s) / *thick * (T_freeze - T);
    *melt_thick = *melt_thick * dt / dmax(latent_heat * rho_s * psi_s, 0.000000000000001);
    *melt_thick = dmin(psi_l * *thick, *melt_thick);
  }
  if (psi_s < psi_s_top_min) *melt_thick = *thick * (1.0 - psi_s / psi_s_top_min);
  if (*melt_thick > 0.0 && psi_g > gas_snow_ice2) {
    if (*melt_thick > (psi_g - gas_snow_ice2) * *thick) {
      *melt_thick = *melt_thick - (psi_g - gas_snow_ice2) * *thick;
      *thick = *thick * (1.0 - (psi_g - gas_snow_ice2));
    } else {
      *thick = *thick - *melt_thick;
      *melt_thick = 0.0;
    }
  }
}

/* sub_melt_snow, mo_functions.f90:443-474 */
static void sub_melt_snow(double *melt_thick, double *thick, double *thick_snow, double *H_abs, double *H_abs_snow,
                          double *m, double *m_snow, double *psi_g_snow) {
  double shift = 1.0 / dmax(*psi_g_snow, 0.01) * *melt_thick;
  if (shift >= *thick_snow) {
    *melt_thick = *melt_thick - *thick_snow * *psi_g_snow;
    *H_abs = *H_abs + *H_abs_snow;
    *m = *m + *m_snow;
    *thick = *thick + (1.0 - *psi_g_snow) * *thick_snow;
    *thick_snow = 0.0; *m_snow = 0.0; *H_abs_snow = 0.0;
  } else {
    *H_abs = *H_abs + shift / *thick_snow * *H_abs_snow;
    *H_abs_snow = *H_abs_snow - shift / *thick_snow * *H_abs_snow;
    *m = *m + shift / *thick_snow * *m_snow;
    *m_snow = *m_snow - shift / *thick_snow * *m_snow;
    *thick = *thick + shift - *melt_thick;
    *thick_snow = *thick_snow - shift;
    *melt_thick = 0.0;
  }
}

/* ------------------------------------------------------------------ mo_mass.f90 */

/* mass_transfer, mo_mass.f90:53-96.  fl_m has N+1 entries (1-based). */
static void mass_transfer(column *c, const double *fl_m) {
  int N = c->N, Na = c->N_active, k;
  double TT[SAMSIM_MAX_NLAYER + 2], SS_bu[SAMSIM_MAX_NLAYER + 2], SS_abs[SAMSIM_MAX_NLAYER + 2];
  double *H_abs = c->H_abs, *S_abs = c->S_abs;
  (void)N;
  for (k = 1; k <= Na; k++) { TT[k] = c->T[k]; SS_bu[k] = c->S_bu[k]; SS_abs[k] = S_abs[k]; }
  TT[Na + 1] = c->cfg->T_bottom;
  SS_bu[Na + 1] = c->S_bu_bottom;
  SS_abs[Na + 1] = c->S_bu_bottom * 2000.0;
  for (k = 1; k <= Na; k++) {
    if (fl_m[k + 1] > 0.0) {
      H_abs[k] = H_abs[k] + fl_m[k + 1] * TT[k + 1] * c_l;
      S_abs[k] = S_abs[k] + dmin(fl_m[k + 1] * S_BR2(TT[k + 1], SS_bu[k + 1]), SS_abs[k + 1]);
    } else if (fl_m[k + 1] < 0.0) {
      H_abs[k] = H_abs[k] + fl_m[k + 1] * TT[k] * c_l;
      S_abs[k] = S_abs[k] + dmax(fl_m[k + 1] * S_BR2(TT[k], SS_bu[k]), -S_abs[k]);
    }
    if (fl_m[k] > 0.0) {
      H_abs[k] = H_abs[k] - fl_m[k] * TT[k] * c_l;
      S_abs[k] = S_abs[k] - dmin(fl_m[k] * S_BR2(TT[k], SS_bu[k]), S_abs[k]);
    } else if (fl_m[k] < 0.0) {
      H_abs[k] = H_abs[k] - fl_m[k] * TT[k - 1] * c_l;
      S_abs[k] = S_abs[k] - dmax(fl_m[k] * S_BR2(TT[k - 1], SS_bu[k - 1]), -S_abs[k - 1]);
    }
  }
}

/* expulsion_flux, mo_mass.f90:112-136 (psi_g<0.001: float32 literal) */
static void expulsion_flux(column *c) {
  int N = c->N, Na = c->N_active, k;
  double *fl_m = c->fl_m, *V_ex = c->V_ex, *psi_g = c->psi_g, *thick = c->thick, *m = c->m;
  for (k = 1; k <= N + 1; k++) fl_m[k] = 0.0;
  fl_m[2] = -V_ex[1] * rho_l;
  for (k = 2; k <= Na; k++) {
    if (psi_g[k] < (double)0.001f) {
      fl_m[k + 1] = -V_ex[k] * rho_l + fl_m[k];
    } else {
      fl_m[k + 1] = -dmax((V_ex[k] - psi_g[k] * thick[k]) * rho_l, 0.0);
      psi_g[k] = dmax((psi_g[k] * thick[k] - V_ex[k]) / thick[k], 0.0);
    }
  }
  for (k = 1; k <= Na; k++) m[k] = m[k] + fl_m[k + 1] - fl_m[k];
}

/* ------------------------------------------------------------------ mo_grav_drain.f90 */

/* fl_grav_drain, mo_grav_drain.f90:74-201 */
static void fl_grav_drain(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, k, kk;
  double fl_up[SAMSIM_MAX_NLAYER + 2], fl_down[SAMSIM_MAX_NLAYER + 2], perm[SAMSIM_MAX_NLAYER + 2];
  double harmonic_perm[SAMSIM_MAX_NLAYER + 2], fl_m[SAMSIM_MAX_NLAYER + 3];
  double flux, test1, d_S_br, height, ray_mini = ray_crit, heat_loss = 0.0, s;
  double *S_br = c->S_br, *psi_l = c->psi_l, *psi_s = c->psi_s, *thick = c->thick;
  double *S_abs = c->S_abs, *H_abs = c->H_abs, *T = c->T, *m = c->m, *ray = c->ray;
  double dt = g->dt;
  const double p17 = pow(10.0, -17.0), p14 = pow(10.0, -14.0);

  for (k = 1; k <= N; k++) { perm[k] = 0.0; harmonic_perm[k] = 0.0; }
  perm[Na] = 9999999.0;
  for (k = 1; k <= N - 1; k++) ray[k] = 0.0;
  for (k = 1; k <= Na; k++) { fl_up[k] = 0.0; fl_down[k] = 0.0; }

  for (k = 1; k <= Na; k++) perm[k] = p17 * pow(1000.0 * fabs(psi_l[k]), 3.10);     /* :105 */

  if (g->harmonic_flag == 2) {                                                       /* :109-123 */
    for (k = 1; k <= Na - 1; k++) {
      test1 = perm[k]; for (kk = k; kk <= Na - 1; kk++) if (perm[kk] < test1) test1 = perm[kk];
      if (test1 < p14) {
        harmonic_perm[k] = 0.0;
      } else {
        for (kk = k; kk <= Na - 1; kk++) harmonic_perm[k] = harmonic_perm[k] + thick[kk] / perm[kk];
        harmonic_perm[k] = harmonic_perm[k] + (thick[Na] * psi_s[Na] / psi_s_min) / perm[Na];
        s = 0.0; for (kk = k; kk <= Na - 1; kk++) s += thick[kk];
        harmonic_perm[k] = (s + thick[Na] * psi_s[Na] / psi_s_min) / harmonic_perm[k];
      }
    }
  }

  for (k = 1; k <= Na - 1; k++) {                                                    /* :126-136 */
    d_S_br = S_br[k] - S_br[Na];
    s = 0.0; for (kk = k + 1; kk <= Na - 1; kk++) s += thick[kk];
    height = s + thick[Na] * psi_s[Na] / psi_s_min;
    if (g->harmonic_flag == 1) {
      test1 = perm[k]; for (kk = k; kk <= Na; kk++) if (perm[kk] < test1) test1 = perm[kk];
      ray[k] = grav_f * rho_l * bbeta * d_S_br * height * test1;
    } else if (g->harmonic_flag == 2) {
      ray[k] = grav_f * rho_l * bbeta * d_S_br * height * harmonic_perm[k];
    }
    ray[k] = ray[k] / (kappa_l * mu);
    ray[k] = dmax(ray[k], 0.0);
  }

  s = 0.0; for (k = 1; k <= N; k++) s += S_abs[k];
  c->grav_salt = c->grav_salt + s;                                                   /* :141 */

  for (k = 1; k <= Na - 1; k++) {
    if (ray[k] > ray_mini && psi_s[k] > 0.001 && S_abs[k] / m[k] > 0.1 && S_br[k] > S_br[k + 1]) {
      flux = x_grav * (ray[k] - ray_mini) * dt * thick[k];
      flux = dmin(flux, psi_l[k] * rho_l * thick[k]);
      S_abs[k] = S_abs[k] - flux * S_br[k];
      if (S_abs[k] < 0.0) STOP(21234, k);
      c->grav_temp = c->grav_temp + flux * T[k];
      H_abs[k] = H_abs[k] - flux * c_l * T[k];
      heat_loss = heat_loss + flux * c_l * T[k];
      fl_down[k] = flux;
      for (kk = k; kk <= Na; kk++) fl_up[kk] = fl_up[kk] + flux;
      fl_up[k] = dmin(fl_up[k], psi_l[k] * rho_l * thick[k]);
    }
  }

  s = 0.0; for (k = 1; k <= N; k++) s += S_abs[k];
  c->grav_salt = c->grav_salt - s;                                                   /* :172 */

  for (k = 1; k <= N + 1; k++) fl_m[k] = 0.0;  /* local fl_m is uninitialised beyond N_active+1 in the reference; unused */
  fl_m[1] = 0.0;
  for (k = 1; k <= Na; k++) fl_m[k + 1] = fl_up[k];
  if (c->n_bgc > 0) {                                                                /* :178-185 (sic: column N_active is read) */
    for (k = 1; k <= Na - 1; k++) FLB(k, Na + 1) = FLB(k, Na) + fl_down[k];
    for (k = 1; k <= Na; k++) FLB(k + 1, k) = FLB(k + 1, k) + fl_up[k];
  }

  mass_transfer(c, fl_m);                                                            /* :187 */

  c->grav_drain = c->grav_drain + fl_m[Na + 1];

  if (g->grav_heat_flag == 2) H_abs[Na] = H_abs[Na] + heat_loss - fl_up[Na] * c_l * g->T_bottom;

  for (k = 1; k <= N; k++) if (S_abs[k] < 0.0) STOP(1337, k);                        /* :197-200 */
}

/* fl_grav_drain_simple, mo_grav_drain.f90:218-278 (grav_flag 3): Rayleigh numbers as above, then every layer above the
 * critical value loses 1 % of its salt (`0.99` is a float32 literal).  harmonic_perm is not initialised in the reference;
 * the flang build reads zeros there. */
static void fl_grav_drain_simple(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, k, kk;
  double perm[SAMSIM_MAX_NLAYER + 2], harmonic_perm[SAMSIM_MAX_NLAYER + 2];
  double temp, d_S_br, height, ray_mini = ray_crit, s;
  double *S_br = c->S_br, *psi_l = c->psi_l, *psi_s = c->psi_s, *thick = c->thick, *S_abs = c->S_abs, *ray = c->ray;
  const double p17 = pow(10.0, -17.0), p14 = pow(10.0, -14.0);
  for (k = 1; k <= N; k++) { perm[k] = 0.0; harmonic_perm[k] = 0.0; }
  perm[Na] = 9999999.0;
  for (k = 1; k <= N - 1; k++) ray[k] = 0.0;
  for (k = 1; k <= Na; k++) perm[k] = p17 * pow(1000.0 * fabs(psi_l[k]), 3.10);
  if (g->harmonic_flag == 2) {
    for (k = 1; k <= Na - 1; k++) {
      temp = perm[k]; for (kk = k; kk <= Na - 1; kk++) if (perm[kk] < temp) temp = perm[kk];
      if (temp < p14) {
        harmonic_perm[k] = 0.0;
      } else {
        for (kk = k; kk <= Na - 1; kk++) harmonic_perm[k] = harmonic_perm[k] + thick[kk] / perm[kk];
        harmonic_perm[k] = harmonic_perm[k] + (thick[Na] * psi_s[Na] / psi_s_min) / perm[Na];
        s = 0.0; for (kk = k; kk <= Na - 1; kk++) s += thick[kk];
        harmonic_perm[k] = (s + thick[Na] * psi_s[Na] / psi_s_min) / harmonic_perm[k];
      }
    }
  }
  for (k = 1; k <= Na - 1; k++) {
    d_S_br = S_br[k] - S_br[Na];
    s = 0.0; for (kk = k + 1; kk <= Na - 1; kk++) s += thick[kk];
    height = s + thick[Na] * psi_s[Na] / psi_s_min;
    if (g->harmonic_flag == 1) {
      temp = perm[k]; for (kk = k; kk <= Na; kk++) if (perm[kk] < temp) temp = perm[kk];
      ray[k] = grav_f * rho_l * bbeta * d_S_br * height * temp;
    } else if (g->harmonic_flag == 2) {
      ray[k] = grav_f * rho_l * bbeta * d_S_br * height * harmonic_perm[k];
    }
    ray[k] = ray[k] / (kappa_l * mu);
    ray[k] = dmax(ray[k], 0.0);
  }
  for (k = Na - 1; k >= 1; k--) if (ray[k] > ray_mini) S_abs[k] = S_abs[k] * (double)0.99f;
  c->grav_drain = 0.0;
}

/* ------------------------------------------------------------------ mo_snow.f90 */

/* func_k_snow, mo_snow.f90:560-573 (`+0.15` float32 literal) */
double oracle_func_k_snow(double m_snow, double thick_snow) {
  const double c0 = 0.138, c1 = -1.01 / 1000.0, c2 = 3.233 / 1000000.0;
  double k_snow = c0 + c1 * m_snow / thick_snow + c2 * pow(m_snow / thick_snow, 2.0);
  return k_snow + (double)0.15f;
}

/* snow_coupling, mo_snow.f90:61-104.
 * The reference passes T_snow (resp. T) as BOTH the INTENT(in) guess T_in and the INTENT(out) result of getT
 * (:74-75 etc.).  Arguments are by reference, so getT's first statement `T = H/c_l` also overwrites T_in:
 * the Newton first guess of these calls is H/c_l, not the previous temperature. */
static void getT_aliased(column *c, double H, double S_bu, double *T, double *phi, int tag) {
  getT(c, H, S_bu, H / c_l, T, phi, tag);
}
static void snow_coupling(column *c) {
  double *H_abs_snow = &c->H_abs_snow, *phi_s = &c->phi_s, *T_snow = &c->T_snow;
  double *H_abs = &c->H_abs[1], *H = &c->H[1], *phi = &c->phi[1], *T = &c->T[1];
  double m_snow = c->m_snow, S_abs_snow = c->S_abs_snow, m = c->m[1], S_bu = c->S_bu[1];
  double d, sg;
  int jj;
  *H_abs = *H_abs + m_snow * latent_heat + *H_abs_snow;
  *H_abs_snow = -m_snow * latent_heat;
  *H = *H_abs / m;
  getT_aliased(c, *H_abs_snow / m_snow, S_abs_snow / m_snow, T_snow, phi_s, 5701);
  getT_aliased(c, *H, S_bu, T, phi, 5702);
  if (*T > 0.0 && *H_abs <= -*H_abs_snow) {
    *H_abs_snow = *H_abs_snow + *H_abs;
    *H_abs = 0.0;
    getT_aliased(c, *H_abs_snow / m_snow, S_abs_snow / m_snow, T_snow, phi_s, 5701);
    getT_aliased(c, *H, S_bu, T, phi, 5702);
  } else if (*T > 0.0 && *H_abs > -*H_abs_snow) {
    *H_abs = (*H_abs + *H_abs_snow) * m / m_snow / (1.0 + m / m_snow);
    *H_abs_snow = *H_abs * m_snow / m;
    getT_aliased(c, *H_abs_snow / m_snow, S_abs_snow / m_snow, T_snow, phi_s, 5701);
    getT_aliased(c, *H, S_bu, T, phi, 5702);
  } else {
    jj = 0;
    while (fabs(*T - *T_snow) > (double)0.1f && jj < 201) {
      d = *T_snow - (*T_snow + *T) / 2.0;
      sg = dmax(fabs(d), 0.1); if (signbit(d)) sg = -sg;                               /* SIGN(a,b) */
      *H_abs_snow = *H_abs_snow - sg * c_s * m_snow;
      *H_abs = *H_abs + sg * c_s * m_snow;
      jj = jj + 1;
      *H = *H_abs / m;
      getT_aliased(c, *H_abs_snow / m_snow, S_abs_snow / m_snow, T_snow, phi_s, 5701);
      getT_aliased(c, *H, S_bu, T, phi, 5702);
    }
    if (jj > 200 && fabs(*T - *T_snow) > 1.0) STOP(16, 1);
  }
}

/* snow_precip, mo_snow.f90:123-150 */
static void snow_precip(column *c, int has_solid) {
  double solid_precip, liquid_precip, d_thick, dt = c->cfg->dt, T2m = c->T2m;
  if (has_solid) { solid_precip = c->solid_precip; liquid_precip = c->liquid_precip; }
  else if (T2m > 0.0) { solid_precip = 0.0; liquid_precip = c->liquid_precip; }
  else { solid_precip = c->liquid_precip; liquid_precip = 0.0; }
  d_thick = dt * solid_precip * rho_l / rho_snow;
  c->m_snow = c->m_snow + dt * rho_l * (liquid_precip + solid_precip);
  c->thick_snow = c->thick_snow + d_thick;
  c->H_abs_snow = c->H_abs_snow + dt * T2m * liquid_precip * rho_l * c_l;
  c->H_abs_snow = c->H_abs_snow + dt * dmin(T2m, -1.0) * solid_precip * rho_l * c_s;
  c->H_abs_snow = c->H_abs_snow - dt * solid_precip * rho_l * latent_heat;
}

/* snow_precip_0, mo_snow.f90:167-192 */
static void snow_precip_0(column *c, int has_solid) {
  double solid_precip, liquid_precip, dt = c->cfg->dt, T2m = c->T2m;
  if (has_solid) { solid_precip = c->solid_precip; liquid_precip = c->liquid_precip; }
  else if (T2m > 0.0) { solid_precip = 0.0; liquid_precip = c->liquid_precip; }
  else { solid_precip = c->liquid_precip; liquid_precip = 0.0; }
  c->H_abs[1] = c->H_abs[1] + (liquid_precip + solid_precip) * (T2m - c->T[1]) * dt;
  c->H_abs[1] = c->H_abs[1] - solid_precip * latent_heat * dt;
  c->S_abs[1] = c->S_abs[1] - (liquid_precip + solid_precip) * c->S_abs[1] / c->m[1] * dt;
}

/* snow_thermo (meltwater==0), mo_snow.f90:212-320, and snow_thermo_meltwater (meltwater==1), :331-454 */
static void snow_thermo(column *c, int meltwater) {
  double *psi_l_snow = &c->psi_l_snow, *psi_s_snow = &c->psi_s_snow, *psi_g_snow = &c->psi_g_snow;
  double *thick_snow = &c->thick_snow, *H_abs_snow = &c->H_abs_snow, *m_snow = &c->m_snow, *T_snow = &c->T_snow;
  double *m = &c->m[1], *thick = &c->thick[1], *H_abs = &c->H_abs[1];
  double sat_snow, psi_s_old, phi_snow = 0.0, T_in, H_snow, S_bu_snow, max_lwc, max_lwc_v;
  double psi_l_snow_slush, psi_l_snow_flush, gmin;
  double ksf = c->cfg->k_snow_flush;

  H_snow = *H_abs_snow / *m_snow;
  S_bu_snow = c->S_abs_snow / *m_snow;
  psi_s_old = *psi_s_snow;
  T_in = *T_snow;
  getT(c, H_snow, S_bu_snow, T_in, T_snow, &phi_snow, 5700);
  *psi_s_snow = *m_snow * phi_snow / rho_s / *thick_snow;
  *psi_l_snow = *m_snow * (1.0 - phi_snow) / rho_l / *thick_snow;
  if (*psi_s_snow + *psi_l_snow > 1.0) {
    *thick_snow = *m_snow * (phi_snow / rho_s + (1.0 - phi_snow) / rho_l);
    *psi_s_snow = *m_snow * phi_snow / rho_s / *thick_snow;
    *psi_l_snow = *m_snow * (1.0 - phi_snow) / rho_l / *thick_snow;
    if (fabs(*psi_s_snow + *psi_l_snow - 1.0) > 0.0000001) STOP(345, 0);
  }
  *psi_g_snow = 1.0 - *psi_s_snow - *psi_l_snow;
  if (*psi_s_snow > 0.0) max_lwc = 0.057 * (1.0 - *psi_s_snow) / (*psi_s_snow) + 0.017;
  else max_lwc = 0.0;

  if (psi_s_old > *psi_s_snow && *psi_s_snow > 0.0) {
    if ((1.0 - phi_snow) > max_lwc) *thick_snow = *thick_snow * (1.0 - (psi_s_old - *psi_s_snow) / psi_s_old);
    if (*thick_snow < (phi_snow * *m_snow / rho_s + (1.0 - phi_snow) * *m_snow / rho_l))
      *thick_snow = (phi_snow * *m_snow / rho_s + (1.0 - phi_snow) * *m_snow / rho_l);
    *psi_s_snow = *m_snow * phi_snow / rho_s / *thick_snow;
    *psi_l_snow = *m_snow * (1.0 - phi_snow) / rho_l / *thick_snow;
    *psi_g_snow = 1.0 - *psi_s_snow - *psi_l_snow;
    *psi_g_snow = fabs(*psi_g_snow);
  } else if (*psi_s_snow < 0.000001) {
    *thick_snow = *m_snow / rho_l;
    *psi_s_snow = 0.0; *psi_g_snow = 0.0; *psi_l_snow = 1.0;
  }

  if (!meltwater) {
    if ((1.0 - phi_snow) > max_lwc && *psi_g_snow > 0.0) {                            /* :279 */
      max_lwc_v = max_lwc * *m_snow / (rho_l * *thick_snow);
      sat_snow = *thick_snow * (*psi_l_snow - max_lwc_v);
      sat_snow = sat_snow / (1.0 - *psi_s_snow - max_lwc_v - dmin(gas_snow_ice2, *psi_g_snow));
      *thick_snow = *thick_snow - sat_snow;
      *thick = *thick + sat_snow;
      *m_snow = *m_snow - sat_snow * (*psi_s_snow * rho_s + (1.0 - *psi_s_snow - gas_snow_ice2) * rho_l);
      *m = *m + sat_snow * (*psi_s_snow * rho_s + (1.0 - *psi_s_snow - gas_snow_ice2) * rho_l);
      *H_abs_snow = *H_abs_snow - sat_snow * *psi_s_snow * rho_s * c_s * *T_snow;
      *H_abs = *H_abs + sat_snow * *psi_s_snow * rho_s * c_s * *T_snow;
      *H_abs_snow = *H_abs_snow + sat_snow * *psi_s_snow * rho_s * latent_heat;
      *H_abs = *H_abs - sat_snow * *psi_s_snow * rho_s * latent_heat;
      *H_abs_snow = *H_abs_snow - sat_snow * (1.0 - *psi_s_snow) * rho_l * c_l * *T_snow;
      *H_abs = *H_abs + sat_snow * (1.0 - *psi_s_snow) * rho_l * c_l * *T_snow;
    } else if (*psi_g_snow <= 0.0) {
      *H_abs = *H_abs + *H_abs_snow; *m = *m + *m_snow; *thick = *thick + *thick_snow;
      *H_abs_snow = 0.0; *m_snow = 0.0; *thick_snow = 0.0;
      *psi_g_snow = 0.0; *psi_s_snow = 0.0; *psi_l_snow = 0.0;
    }
  } else {
    if ((1.0 - phi_snow) > max_lwc && *psi_l_snow > 0.0 && *psi_g_snow > 0.0) {       /* :398 */
      max_lwc_v = max_lwc * *m_snow / (rho_l * *thick_snow);
      psi_l_snow_slush = (*psi_l_snow - max_lwc_v) * (1.0 - ksf);
      psi_l_snow_flush = (*psi_l_snow - max_lwc_v) * ksf;
      c->melt_thick_snow = *thick_snow * psi_l_snow_flush;
      sat_snow = *thick_snow * (psi_l_snow_slush);
      sat_snow = sat_snow / (1.0 - *psi_s_snow - max_lwc_v - dmin(gas_snow_ice2, *psi_g_snow));
      gmin = dmin(gas_snow_ice2, *psi_g_snow);
      *thick_snow = *thick_snow - sat_snow - c->melt_thick_snow;
      *thick = *thick + sat_snow;
      *m_snow = *m_snow - sat_snow * (*psi_s_snow * rho_s + (1.0 - *psi_s_snow - gmin) * rho_l) - c->melt_thick_snow * rho_l;
      *m = *m + sat_snow * (*psi_s_snow * rho_s + (1.0 - *psi_s_snow - gmin) * rho_l);
      *H_abs_snow = *H_abs_snow - sat_snow * *psi_s_snow * rho_s * c_s * *T_snow;
      *H_abs = *H_abs + sat_snow * *psi_s_snow * rho_s * c_s * *T_snow;
      *H_abs_snow = *H_abs_snow + sat_snow * *psi_s_snow * rho_s * latent_heat;
      *H_abs = *H_abs - sat_snow * *psi_s_snow * rho_s * latent_heat;
      *H_abs_snow = *H_abs_snow - sat_snow * (1.0 - *psi_s_snow - gmin) * rho_l * c_l * *T_snow
                    - c->melt_thick_snow * rho_l * c_l * *T_snow;
      *H_abs = *H_abs + sat_snow * (1.0 - *psi_s_snow - gmin) * rho_l * c_l * *T_snow;
    } else if (*psi_g_snow <= 0.0) {
      *H_abs = *H_abs + *H_abs_snow; *m = *m + *m_snow; *thick = *thick + *thick_snow;
      *H_abs_snow = 0.0; *m_snow = 0.0; *thick_snow = 0.0;
      *psi_g_snow = 0.0; *psi_s_snow = 0.0; *psi_l_snow = 0.0;
    }
  }
  if (*psi_g_snow < 0.0) STOP(9876, 0);   /* `stop 09876` */
}

/* the snow block of mo_grotz.f90:273-292 and :607-625 */
static void snow_block(column *c) {
  if (c->thick_snow > 0.0) {
    if (c->cfg->snow_flush_flag == 0) { snow_thermo(c, 0); c->melt_thick_snow = 0.0; }
    else if (c->cfg->snow_flush_flag == 1) { c->melt_thick_snow = 0.0; snow_thermo(c, 1); }
  } else {
    c->thick_snow = 0.0; c->m_snow = 0.0; c->psi_s_snow = 0.0; c->psi_l_snow = 0.0; c->psi_g_snow = 0.0;
    c->H_abs_snow = 0.0; c->S_abs_snow = 0.0; c->melt_thick_snow = 0.0;
  }
}

/* sub_fl_Q_0_snow_thin, mo_snow.f90:466-487 */
static double sub_fl_Q_0_snow_thin(double m_snow, double thick_snow, double T_snow, double psi_s, double psi_l,
                                   double psi_g, double thick, double T_bound) {
  double k_snow = oracle_func_k_snow(m_snow, thick_snow);
  double k = psi_s * k_s + psi_l * k_l + psi_g * 0.0;
  double R;
  k = thick_snow / (thick_snow + thick) * k_snow + thick / (thick_snow + thick) * k;
  R = (thick_snow + thick) / (2.0 * k);
  return (T_snow - T_bound) / R;
}
/* sub_fl_Q_snow, mo_snow.f90:498-518 */
static double sub_fl_Q_snow(double m_snow, double thick_snow, double T_snow, double psi_s_2, double psi_l_2,
                            double thick_2, double T_2) {
  double k_snow = oracle_func_k_snow(m_snow, thick_snow);
  double k_2 = psi_s_2 * k_s + psi_l_2 * k_l;
  double R = thick_snow / (2.0 * k_snow) + thick_2 / (2.0 * k_2);
  return (T_2 - T_snow) / R;
}
/* sub_fl_Q_0_snow, mo_snow.f90:528-546 */
static double sub_fl_Q_0_snow(double m_snow, double thick_snow, double T_snow, double T_bound) {
  double k = oracle_func_k_snow(m_snow, thick_snow);
  double R = thick_snow / (2.0 * k);
  return (T_snow - T_bound) / R;
}

/* ------------------------------------------------------------------ mo_flood.f90 */

/* flood, mo_flood.f90:55-151 */
static void flood(column *c) {
  const samsim_config *g = c->cfg;
  int Na = c->N_active, k;
  double perm[SAMSIM_MAX_NLAYER + 2], S_bu[SAMSIM_MAX_NLAYER + 2];
  double flood_brine, shift_ice, shift_snow, shift, harmonic_perm, s;
  double *psi_s = c->psi_s, *psi_l = c->psi_l, *S_abs = c->S_abs, *H_abs = c->H_abs, *m = c->m, *T = c->T, *thick = c->thick;
  double dt = g->dt, freeboard = c->freeboard, psi_g_snow = c->psi_g_snow;
  const double p17 = pow(10.0, -17.0);

  for (k = 1; k <= Na; k++) perm[k] = p17 * pow(1000.0 * psi_l[k], 3.10);
  harmonic_perm = 0.0;
  for (k = 1; k <= Na - 1; k++) harmonic_perm = harmonic_perm + thick[k] / perm[k];
  harmonic_perm = harmonic_perm + (thick[Na] * psi_s[Na] / psi_s_min) / perm[Na];
  s = 0.0; for (k = 1; k <= Na - 1; k++) s += thick[k];
  harmonic_perm = (s + thick[Na] * psi_s[Na] / psi_s_min) / harmonic_perm;

  s = 0.0; for (k = 1; k <= Na; k++) s += thick[k];
  flood_brine = -dt * grav_f * rho_l * rho_l * harmonic_perm * (freeboard) / (mu * s);

  shift_ice = flood_brine / (rho_l * psi_g_snow / ratio_flood);
  shift_snow = shift_ice * (1 + psi_g_snow / (1.0 - psi_g_snow) * (1.0 - 1.0 / ratio_flood));

  for (k = 1; k <= Na; k++) S_bu[k] = S_abs[k] / m[k];

  S_abs[1] = S_abs[1] + flood_brine * S_bu[Na];
  H_abs[1] = H_abs[1] + flood_brine * H_abs[Na] / m[Na];
  m[1] = m[1] + flood_brine;

  thick[1] = thick[1] + shift_ice;
  H_abs[1] = H_abs[1] + shift_snow / c->thick_snow * c->H_abs_snow;
  c->H_abs_snow = c->H_abs_snow - shift_snow / c->thick_snow * c->H_abs_snow;
  m[1] = m[1] + shift_snow / c->thick_snow * c->m_snow;
  c->m_snow = c->m_snow - shift_snow / c->thick_snow * c->m_snow;
  c->thick_snow = c->thick_snow - shift_snow;

  if (freeboard + shift_ice < neg_free) {
    shift = neg_free - (freeboard + shift_ice);
    flood_brine = shift * (psi_g_snow) * rho_l;
    S_abs[Na] = S_abs[Na] + (c->S_bu_bottom - S_bu[Na]) * flood_brine;
    H_abs[Na] = H_abs[Na] + (g->T_bottom - T[Na]) * c_l * flood_brine;
    S_abs[1] = S_abs[1] + S_bu[Na] * flood_brine;
    H_abs[1] = H_abs[1] + T[Na] * c_l * flood_brine;
    m[1] = m[1] + flood_brine;
    thick[1] = thick[1] + shift;
    H_abs[1] = H_abs[1] + shift / c->thick_snow * c->H_abs_snow;
    c->H_abs_snow = c->H_abs_snow - shift / c->thick_snow * c->H_abs_snow;
    m[1] = m[1] + shift / c->thick_snow * c->m_snow;
    c->m_snow = c->m_snow - shift / c->thick_snow * c->m_snow;
    c->thick_snow = c->thick_snow - shift;
  }
  if (c->n_bgc > 0) {                                                                /* :140-143 */
    FLB(Na, 1) = FLB(Na, 1) + flood_brine;
    FLB(Na + 1, Na) = FLB(Na + 1, Na) + flood_brine;
  }
}

/* flood_simple, mo_flood.f90:167-210 (flood_flag 3) */
static void flood_simple(column *c) {
  const samsim_config *g = c->cfg;
  double shift = c->freeboard - neg_free;
  double flood_brine = -shift * c->psi_g_snow * rho_l;
  double *S_abs = c->S_abs, *H_abs = c->H_abs, *m = c->m, *thick = c->thick;
  thick[1] = thick[1] - shift;
  S_abs[1] = S_abs[1] + c->S_bu_bottom * flood_brine;
  H_abs[1] = H_abs[1] - shift / c->thick_snow * c->H_abs_snow;
  H_abs[1] = H_abs[1] + g->T_bottom * c_l * flood_brine;
  m[1] = m[1] - shift / c->thick_snow * c->m_snow;
  m[1] = m[1] + flood_brine;
  c->H_abs_snow = c->H_abs_snow + shift / c->thick_snow * c->H_abs_snow;
  c->m_snow = c->m_snow + shift / c->thick_snow * c->m_snow;
  c->thick_snow = c->thick_snow + shift;
}

/* ------------------------------------------------------------------ mo_flush.f90 */

/* flush3, mo_flush.f90:70-237 */
static void flush3(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, k;
  double R_h[SAMSIM_MAX_NLAYER + 2], R_v[SAMSIM_MAX_NLAYER + 2], R[SAMSIM_MAX_NLAYER + 2];
  double S_bu[SAMSIM_MAX_NLAYER + 2], fl_m[SAMSIM_MAX_NLAYER + 3];
  double cnst, flush_total, loss_S_abs, loss_H_abs, s, saveS[SAMSIM_MAX_NLAYER + 2];
  double *psi_l = c->psi_l, *psi_g = c->psi_g, *thick = c->thick, *S_abs = c->S_abs, *H_abs = c->H_abs, *m = c->m, *T = c->T;
  double *perm = c->perm, *flush_v = c->flush_v, *flush_h = c->flush_h;
  double dt = g->dt, freeboard = c->freeboard;
  const double p17 = pow(10.0, -17.0);

  for (k = 1; k <= N; k++) S_bu[k] = 0.0;
  for (k = 1; k <= N + 1; k++) fl_m[k] = 0.0;
  for (k = 1; k <= Na; k++) { flush_v[k] = 0.0; flush_h[k] = 0.0; }
  for (k = 1; k <= Na; k++) S_bu[k] = S_abs[k] / m[k];

  s = 0.0; for (k = 1; k <= Na; k++) s += thick[k];
  cnst = s * para_flush_horiz;

  c->melt_thick = dmin(c->melt_thick, psi_l[1] * thick[1]);
  c->melt_thick = dmin(c->melt_thick, g->thick_0 / 3.0);

  if (g->snow_flush_flag == 1) {
    for (k = 1; k <= N; k++) perm[k] = 0.0;
    for (k = 1; k <= Na; k++) perm[k] = p17 * pow(1000.0 * fabs(psi_l[k] + 2.0 * psi_g[k]), 3.10);
    for (k = 1; k <= Na; k++) if (perm[k] == 0.0) perm[k] = 1.0;
  } else if (g->snow_flush_flag == 0) {
    for (k = 1; k <= N; k++) perm[k] = 1.0;
    for (k = 1; k <= Na; k++) perm[k] = p17 * pow(1000.0 * fabs(psi_l[k]), 3.10);
  }

  for (k = 1; k <= Na; k++) {
    R_v[k] = mu * thick[k] / dmax(perm[k], 0.00000000000000000000001);
    R_h[k] = mu * cnst / (thick[k] * dmax(perm[k], 0.00000000000000000000001));
  }
  R[Na] = 0.0;
  R[Na - 1] = R_v[Na - 1];
  if (Na > 2) {
    for (k = Na - 2; k >= 1; k--) {
      R[k] = R[k + 1] + R_v[k];
      R[k] = ((R[k]) * R_h[k]) / (R[k] + R_h[k]);
    }
  }

  flush_total = (freeboard + c->melt_thick) / R[1] * grav_f * dt * oracle_func_density(T[1], S_BR(T[1])) * rho_l;
  flush_total = dmin(flush_total, c->melt_thick * rho_l);
  c->melt_err = c->melt_err + c->melt_thick - dmin(flush_total / rho_l, c->melt_thick);

  flush_h[1] = flush_total * (R[2] + R_v[1]) / (R[2] + R_v[1] + R_h[1]);
  flush_v[1] = flush_total * R_h[1] / (R[2] + R_v[1] + R_h[1]);
  for (k = 2; k <= Na - 1; k++) {
    flush_h[k] = flush_v[k - 1] * (R[k + 1] + R_v[k]) / (R[k + 1] + R_v[k] + R_h[k]);
    flush_v[k] = flush_v[k - 1] * R_h[k] / (R[k + 1] + R_v[k] + R_h[k]);
  }
  flush_v[Na] = flush_v[Na - 1];
  flush_h[Na] = 0.0;

  if (c->n_bgc > 0) {                                                                /* :168-175 */
    double sh = 0.0;
    for (k = 1; k <= Na - 1; k++) FLB(k, Na) = FLB(k, Na) + flush_h[k];
    for (k = 1; k <= N; k++) sh += flush_h[k];
    FLB(Na, Na + 1) = FLB(Na, Na + 1) + sh;
    for (k = 1; k <= Na; k++) FLB(k, k + 1) = FLB(k, k + 1) + flush_v[k];
  }

  fl_m[1] = 0.0;
  for (k = 1; k <= Na; k++) fl_m[k + 1] = -flush_v[k];

  /* mass_transfer is called with the LOCAL S_bu (:181) */
  for (k = 1; k <= Na; k++) { saveS[k] = c->S_bu[k]; c->S_bu[k] = S_bu[k]; }
  mass_transfer(c, fl_m);
  for (k = 1; k <= Na; k++) c->S_bu[k] = saveS[k];

  if (g->flush_heat_flag == 2) H_abs[Na] = H_abs[Na] - fl_m[Na + 1] * T[Na] * c_l;

  m[1] = m[1] - flush_total;
  thick[1] = thick[1] - flush_total / rho_l;

  for (k = 1; k <= Na - 1; k++) {
    loss_S_abs = flush_h[k] * S_BR2(T[k], S_abs[k] / m[k]);
    loss_H_abs = flush_h[k] * T[k] * c_l;
    S_abs[k] = S_abs[k] - loss_S_abs;
    H_abs[k] = H_abs[k] - loss_H_abs;
    H_abs[Na] = H_abs[Na] + loss_H_abs;
    S_abs[Na] = S_abs[Na] + loss_S_abs;
  }
  s = 0.0; for (k = 1; k <= Na; k++) s += flush_h[k];
  loss_S_abs = s * S_bu[Na];
  loss_H_abs = s * T[Na] * c_l;
  if (g->flush_heat_flag == 2) H_abs[Na] = H_abs[Na] - loss_H_abs;
  S_abs[Na] = S_abs[Na] - loss_S_abs;

  s = S_abs[1]; for (k = 1; k <= N; k++) if (S_abs[k] < s) s = S_abs[k];
  if (s < -0.00000000000000000000000001) for (k = 1; k <= Na; k++) S_abs[k] = dmax(S_abs[k], 0.0);

  if (fabs(m[1]) < 0.000001) STOP(9876, 1);
}

/* ------------------------------------------------------------------ mo_layer_dynamics.f90 */

/* top_melt, mo_layer_dynamics.f90:191-327 */
static void top_melt(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, N_top = g->n_top, N_middle = g->n_middle, k;
  double rho[SAMSIM_MAX_NLAYER + 2], H[SAMSIM_MAX_NLAYER + 2], S_bu[SAMSIM_MAX_NLAYER + 2];
  double loss_m, loss_S_abs, loss_H_abs, shift, thick_0 = g->thick_0, s;
  double *m = c->m, *S_abs = c->S_abs, *H_abs = c->H_abs, *thick = c->thick;
  int Na = c->N_active, kend;

  for (k = 1; k <= Na; k++) { rho[k] = m[k] / thick[k]; S_bu[k] = S_abs[k] / m[k]; H[k] = H_abs[k] / m[k]; }

  loss_m = thick_0 * rho[1]; loss_S_abs = loss_m * S_bu[1]; loss_H_abs = loss_m * H[1];
  (void)loss_S_abs; (void)loss_H_abs;
  m[1] = m[1] + m[2]; S_abs[1] = S_abs[1] + S_abs[2]; H_abs[1] = H_abs[1] + H_abs[2]; thick[1] = thick[1] + thick[2];

  kend = (N_top - 1 < Na - 1) ? N_top - 1 : Na - 1;
  for (k = 2; k <= kend; k++) {
    m[k] = rho[k + 1] * thick_0;
    S_abs[k] = S_bu[k + 1] * rho[k + 1] * thick_0;
    H_abs[k] = H[k + 1] * rho[k + 1] * thick_0;
  }

  if (Na <= N_top) {
    m[Na] = 0.0; S_abs[Na] = 0.0; H_abs[Na] = 0.0; thick[Na] = 0.0;
    Na = Na - 1;
  } else if (Na > N_top && Na <= N && thick[N_top + 1] / thick_0 < 1.00001) {
    for (k = N_top; k <= Na - 1; k++) {
      m[k] = rho[k + 1] * thick_0;
      S_abs[k] = S_bu[k + 1] * rho[k + 1] * thick_0;
      H_abs[k] = H[k + 1] * rho[k + 1] * thick_0;
    }
    m[Na] = 0.0; S_abs[Na] = 0.0; H_abs[Na] = 0.0; thick[Na] = 0.0;
    Na = Na - 1;
  }

  if (Na == N && thick[N_top + 1] - thick_0 >= 0.000001) {
    loss_m = thick_0 * rho[N_top + 1];
    loss_S_abs = loss_m * S_bu[N_top + 1];
    loss_H_abs = loss_m * H[N_top + 1];
    m[N_top] = loss_m; S_abs[N_top] = loss_S_abs; H_abs[N_top] = loss_H_abs;
    for (k = N_top + 1; k <= N_middle + N_top; k++) {
      m[k] = m[k] - loss_m; H_abs[k] = H_abs[k] - loss_H_abs; S_abs[k] = S_abs[k] - loss_S_abs;
      shift = thick_0 * (double)(float)(N_middle - k + N_top) / (double)(float)(N_middle);
      loss_m = shift * rho[k + 1];
      loss_S_abs = loss_m * S_bu[k + 1];
      loss_H_abs = loss_m * H[k + 1];
      m[k] = m[k] + loss_m; H_abs[k] = H_abs[k] + loss_H_abs; S_abs[k] = S_abs[k] + loss_S_abs;
    }
    for (k = N_top + 1; k <= N_top + N_middle; k++) thick[k] = thick[k] - thick_0 / (double)(float)(N_middle);
  }
  c->N_active = Na;

  s = 0.0; for (k = 1; k <= N; k++) s += thick[k];
  if (thick_0 * (Na + 0.501) <= s && Na < N) STOP(7889, 0);
}

/* bottom_melt, mo_layer_dynamics.f90:341-427 */
static void bottom_melt(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, N_top = g->n_top, N_middle = g->n_middle, k;
  double rho[SAMSIM_MAX_NLAYER + 2], H[SAMSIM_MAX_NLAYER + 2], S_bu[SAMSIM_MAX_NLAYER + 2];
  double loss_m = 0.0, loss_S_abs = 0.0, loss_H_abs = 0.0, shift;
  double *m = c->m, *S_abs = c->S_abs, *H_abs = c->H_abs, *thick = c->thick;
  for (k = N_top + 1; k <= N; k++) { rho[k] = m[k] / thick[k]; S_bu[k] = S_abs[k] / m[k]; H[k] = H_abs[k] / m[k]; }
  for (k = N_top + 1; k <= N_top + N_middle; k++) {
    m[k] = m[k] + loss_m; H_abs[k] = H_abs[k] + loss_H_abs; S_abs[k] = S_abs[k] + loss_S_abs;
    shift = thick[N] * (k - N_top) / (double)(float)(N_middle);
    loss_m = shift * rho[k]; loss_H_abs = loss_m * H[k]; loss_S_abs = loss_m * S_bu[k];
    m[k] = m[k] - loss_m; H_abs[k] = H_abs[k] - loss_H_abs; S_abs[k] = S_abs[k] - loss_S_abs;
  }
  for (k = N_top + 1; k <= N_top + N_middle; k++) thick[k] = thick[k] - thick[N] / (double)(float)(N_middle);
  for (k = N_top + N_middle + 1; k <= N; k++) {
    H_abs[k] = rho[k - 1] * thick[k] * H[k - 1];
    S_abs[k] = rho[k - 1] * thick[k] * S_bu[k - 1];
    m[k] = rho[k - 1] * thick[k];
  }
}

/* bottom_growth, mo_layer_dynamics.f90:438-523 */
static void bottom_growth(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, N_top = g->n_top, N_middle = g->n_middle, N_bottom = g->n_bottom, k;
  double rho[SAMSIM_MAX_NLAYER + 2], H[SAMSIM_MAX_NLAYER + 2], S_bu[SAMSIM_MAX_NLAYER + 2];
  double gain_m = 0.0, gain_S_abs = 0.0, gain_H_abs = 0.0, shift;
  double *m = c->m, *S_abs = c->S_abs, *H_abs = c->H_abs, *thick = c->thick;
  for (k = N_top + 1; k <= N_top + N_middle + 1; k++) { rho[k] = m[k] / thick[k]; S_bu[k] = S_abs[k] / m[k]; H[k] = H_abs[k] / m[k]; }
  for (k = N_top + 1; k <= N_top + N_middle; k++) {
    m[k] = m[k] - gain_m; H_abs[k] = H_abs[k] - gain_H_abs; S_abs[k] = S_abs[k] - gain_S_abs;
    shift = thick[N] * (k - N_top) / (double)(float)(N_middle);
    gain_m = shift * rho[k + 1]; gain_H_abs = gain_m * H[k + 1]; gain_S_abs = gain_m * S_bu[k + 1];
    m[k] = m[k] + gain_m; H_abs[k] = H_abs[k] + gain_H_abs; S_abs[k] = S_abs[k] + gain_S_abs;
  }
  for (k = N_top + 1; k <= N_top + N_middle; k++) thick[k] = thick[k] + thick[N] / (double)(float)(N_middle);
  for (k = N - N_bottom + 1; k <= N - 1; k++) { H_abs[k] = H_abs[k + 1]; S_abs[k] = S_abs[k + 1]; m[k] = m[k + 1]; }
  m[N] = thick[N] * rho_l;
  H_abs[N] = m[N] * g->T_bottom * c_l;
  S_abs[N] = m[N] * c->S_bu_bottom;
}

/* bottom_growth_simple, mo_layer_dynamics.f90:537-560 */
static void bottom_growth_simple(column *c) {
  const samsim_config *g = c->cfg;
  int Na = c->N_active + 1;
  c->N_active = Na;
  c->thick[Na] = g->thick_0;
  c->m[Na] = c->thick[Na] * rho_l;
  c->H_abs[Na] = c->m[Na] * g->T_bottom * c_l;
  c->S_abs[Na] = c->m[Na] * c->S_bu_bottom;
}

/* bottom_melt_simple, mo_layer_dynamics.f90:573-591 */
static void bottom_melt_simple(column *c) {
  int Na = c->N_active;
  c->thick[Na] = 0.0; c->m[Na] = 0.0; c->S_abs[Na] = 0.0; c->H_abs[Na] = 0.0;
  c->N_active = Na - 1;
}

/* top_grow, mo_layer_dynamics.f90:607-716 */
static void top_grow(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, N_top = g->n_top, N_middle = g->n_middle, k, kend;
  double rho[SAMSIM_MAX_NLAYER + 2], H[SAMSIM_MAX_NLAYER + 2], S_bu[SAMSIM_MAX_NLAYER + 2];
  double loss_m, loss_S_abs, loss_H_abs, shift, thick_0 = g->thick_0;
  double *m = c->m, *S_abs = c->S_abs, *H_abs = c->H_abs, *thick = c->thick;
  int Na = c->N_active;
  for (k = 1; k <= Na; k++) { rho[k] = m[k] / thick[k]; S_bu[k] = S_abs[k] / m[k]; H[k] = H_abs[k] / m[k]; }
  loss_m = thick_0 * rho[1]; loss_S_abs = loss_m * S_bu[1]; loss_H_abs = loss_m * H[1];
  m[1] = m[1] - loss_m; S_abs[1] = S_abs[1] - loss_S_abs; H_abs[1] = H_abs[1] - loss_H_abs; thick[1] = thick[1] - thick_0;
  kend = (N_top < Na) ? N_top : Na;
  for (k = 2; k <= kend; k++) {
    m[k] = rho[k - 1] * thick_0;
    S_abs[k] = S_bu[k - 1] * rho[k - 1] * thick_0;
    H_abs[k] = H[k - 1] * rho[k - 1] * thick_0;
  }
  if (Na <= N_top) {
    Na = Na + 1;
    m[Na] = rho[Na - 1] * thick_0;
    S_abs[Na] = S_bu[Na - 1] * thick_0 * rho[Na - 1];
    H_abs[Na] = H[Na - 1] * thick_0 * rho[Na - 1];
    thick[Na] = thick_0;
  } else if (Na > N_top && Na < N) {
    for (k = N_top + 1; k <= Na; k++) {
      m[k] = rho[k - 1] * thick_0;
      S_abs[k] = S_bu[k - 1] * rho[k - 1] * thick_0;
      H_abs[k] = H[k - 1] * rho[k - 1] * thick_0;
    }
    Na = Na + 1;
    m[Na] = rho[Na - 1] * thick_0;
    S_abs[Na] = S_bu[Na - 1] * thick_0 * rho[Na - 1];
    H_abs[Na] = H[Na - 1] * thick_0 * rho[Na - 1];
    thick[Na] = thick_0;
  } else if (Na == N) {
    loss_m = thick_0 * rho[N_top]; loss_S_abs = loss_m * S_bu[N_top]; loss_H_abs = loss_m * H[N_top];
    for (k = N_top + 1; k <= N_middle + N_top; k++) {
      m[k] = m[k] + loss_m; H_abs[k] = H_abs[k] + loss_H_abs; S_abs[k] = S_abs[k] + loss_S_abs;
      shift = thick_0 * (double)(float)(N_middle - k + N_top) / (double)(float)(N_middle);
      loss_m = shift * rho[k]; loss_S_abs = loss_m * S_bu[k]; loss_H_abs = loss_m * H[k];
      m[k] = m[k] - loss_m; H_abs[k] = H_abs[k] - loss_H_abs; S_abs[k] = S_abs[k] - loss_S_abs;
    }
    for (k = N_top + 1; k <= N_top + N_middle; k++) thick[k] = thick[k] + thick_0 / (double)(float)(N_middle);
  }
  c->N_active = Na;
}

/* bgc_advection, mo_mass.f90:150-209: upwind advection of the tracers with the brine fluxes of this step, every flux limited
 * to a third of the layer's content; the (N+1)^2 loop of the reference as written */
static void bgc_advection(column *c) {
  int N = c->N, Na = c->N_active, i, j, t;
  double bgc_temp[SAMSIM_MAX_NLAYER + 2], bgc_br[SAMSIM_MAX_NLAYER + 2], flux;
  for (t = 0; t < c->n_bgc; t++) {
    double *q = c->bgc[t];
    for (i = 1; i <= N; i++) bgc_temp[i] = q[i];
    for (i = 1; i <= Na; i++) bgc_br[i] = q[i] / dmax(c->psi_l[i] * c->thick[i] * rho_l, 0.000000000000001);
    for (i = 1; i <= Na; i++) {
      for (j = 1; j <= Na; j++) {
        flux = dmin(FLB(i, j) * bgc_br[i], q[i] / 3.0);
        bgc_temp[i] = bgc_temp[i] - flux;
        bgc_temp[j] = bgc_temp[j] + flux;
      }
    }
    for (i = 1; i <= Na; i++) {
      flux = dmin(FLB(i, Na + 1) * bgc_br[i], q[i] / 3.0);
      bgc_temp[i] = bgc_temp[i] - flux;
    }
    for (j = 1; j <= Na; j++) {
      flux = FLB(Na + 1, j) * c->bgc_bottom[t];
      bgc_temp[j] = bgc_temp[j] + flux;
    }
    for (i = 1; i <= N; i++) q[i] = bgc_temp[i];
  }
  for (i = 0; i < (N + 2) * (N + 2); i++) c->flb[i] = 0.0;                           /* mo_grotz.f90:745 */
}

/* One regrid routine, tracers included.  In the reference every statement on S_abs / S_bu / S_bu_bottom has a twin on
 * bgc_temp / bgc_bulk / bgc_bottom (mo_layer_dynamics.f90:205-373); the twin is obtained here by running the same
 * routine on a scratch copy of the column in which the tracer stands in for S_abs. */
static void regrid(column *c, void (*fn)(column *)) {
  int N = c->N, k, t;
  if (c->n_bgc == 0) { fn(c); return; }
  double m0[SAMSIM_MAX_NLAYER + 3], th0[SAMSIM_MAX_NLAYER + 3], H0[SAMSIM_MAX_NLAYER + 3];
  double m2[SAMSIM_MAX_NLAYER + 3], th2[SAMSIM_MAX_NLAYER + 3], H2[SAMSIM_MAX_NLAYER + 3], q2[SAMSIM_MAX_NLAYER + 3];
  int Na0 = c->N_active;
  for (k = 0; k <= N + 2; k++) { m0[k] = c->m[k]; th0[k] = c->thick[k]; H0[k] = c->H_abs[k]; }
  fn(c);
  for (t = 0; t < c->n_bgc; t++) {
    column cc = *c;
    for (k = 0; k <= N + 2; k++) { m2[k] = m0[k]; th2[k] = th0[k]; H2[k] = H0[k]; q2[k] = c->bgc[t][k]; }
    cc.m = m2; cc.thick = th2; cc.H_abs = H2; cc.S_abs = q2;
    cc.N_active = Na0; cc.S_bu_bottom = c->bgc_bottom[t]; cc.status = 0;
    fn(&cc);
    for (k = 0; k <= N + 2; k++) c->bgc[t][k] = q2[k];
  }
}

/* layer_dynamics, mo_layer_dynamics.f90:64-175: exactly one branch per call */
static void layer_dynamics(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, N_top = g->n_top, bf = g->bottom_flag;
  double *phi = c->phi, *thick = c->thick, thick_0 = g->thick_0;
  int km1 = (Na - 1 > 1) ? Na - 1 : 1;
  if (phi[N - 1] <= psi_s_min / 2.0 && phi[Na] < 0.00001 && Na == N && thick[N_top + 1] / thick_0 > 1.000001 && bf == 1) {
    regrid(c, bottom_melt);
  } else if (Na > 1 && Na < N && phi[Na] < 0.00001 && phi[km1] <= psi_s_min / 2.0 && bf == 1) {
    regrid(c, bottom_melt_simple);
  } else if (Na > 1 && phi[Na] < 0.00001 && phi[km1] <= psi_s_min / 2.0 && (thick[N_top + 1] / thick_0) < 1.01 && bf == 1) {
    regrid(c, bottom_melt_simple);
  } else if (phi[Na] > psi_s_min && Na < N && bf == 1) {
    regrid(c, bottom_growth_simple);
  } else if (phi[N] > psi_s_min && bf == 1) {
    regrid(c, bottom_growth);
  } else if (thick[1] > 1.5 * thick_0) {
    c->melt_thick_output[2] = c->melt_thick_output[2] - thick[1];
    regrid(c, top_grow);
    c->melt_thick_output[2] = c->melt_thick_output[2] + thick[1];
  } else if (thick[1] < 0.5 * thick_0) {
    c->melt_thick_output[2] = c->melt_thick_output[2] - thick[1];
    regrid(c, top_melt);
    c->melt_thick_output[2] = c->melt_thick_output[2] + thick[1];
  }
}

/* ------------------------------------------------------------------ mo_testcase_specifics.f90 */

/* sub_test1, mo_testcase_specifics.f90:42-89: |time - n*12h| < 0.01 (float32 literals) */
static void sub_test1(double time, double *T_top) {
  int n;
  for (n = 1; n <= 20; n++) {
    if (fabs(time - (double)((float)(12 * n) * 3600.0f)) < (double)0.01f) {
      *T_top = (n % 2 == 1) ? -10.0 : -5.0;
      return;
    }
  }
}
/* sub_test4, mo_testcase_specifics.f90:197-202: pi is float32 */
static void sub_test4(double time, double *fl_q_bottom) {
  *fl_q_bottom = -7.0 * sin(time * (2.0 * pi_f) / (86400.0 * 365.0)) + 7.0;
}

/* ------------------------------------------------------------------ mo_heat_fluxes.f90 */

/* linear interpolation in the 3-hourly tables; time_input(k) = (k-1)*3600*3, mo_functions.f90:323-325 */
static double time_input(int k) { return ((double)(float)k - 1.0) * 3600.0 * 3.0; }

/* sub_heat_fluxes, mo_heat_fluxes.f90:69-312 (boundflux_flag 1, 2, and 3 without lab snow) */
static void sub_heat_fluxes(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, k, tc = c->time_counter;
  double T_old, emi, pen, temp, temp1, temp2, dt = g->dt, thick_min = g->thick_min, s, e;
  double *psi_s = c->psi_s, *psi_l = c->psi_l, *psi_g = c->psi_g, *thick = c->thick, *T = c->T;
  double *fl_Q = c->fl_Q, *H_abs = c->H_abs, *fl_rad = c->fl_rad;

  if (g->boundflux_flag == 1) {                                                       /* :77-87 */
    fl_Q[1] = sub_fl_Q_0(psi_s[1], psi_l[1], psi_g[1], thick[1], T[1], c->T_top, -1);
    if (fabs(fl_Q[1]) > g->max_flux_plate) fl_Q[1] = fl_Q[1] / fabs(fl_Q[1]) * g->max_flux_plate;
  }

  if (g->boundflux_flag == 2) {                                                       /* :91-195 */
    c->albedo = oracle_func_albedo(c->thick_snow, c->T_snow, psi_l[1], thick_min, g->albedo_flag);
    if (g->atmoflux_flag == 1) {
      sub_notzflux(c->time + 86400.0 * 180.0, &c->fl_sw, &c->fl_rest);
    } else if (g->atmoflux_flag == 2) {
      if (c->time == time_input(tc)) {
        c->fl_sw = c->fl_sw_input[tc - 1];
        c->fl_lw = c->fl_lw_input[tc - 1];
      } else {
        temp = (c->time - time_input(tc - 1)) / (time_input(tc) - time_input(tc - 1));
        c->fl_sw = (1.0 - temp) * c->fl_sw_input[tc - 2] + temp * c->fl_sw_input[tc - 1];
        c->fl_lw = (1.0 - temp) * c->fl_lw_input[tc - 2] + temp * c->fl_lw_input[tc - 1];
      }
      c->fl_rest = c->fl_lw + 0.0 + 0.0;
    }
    if (c->thick_snow < thick_min) T_old = T[1]; else T_old = c->T_snow;
    if (c->thick_snow < thick_min) { emi = emissivity_ice; pen = penetr; }
    else { emi = emissivity_snow; pen = 0.0; }
    T_old = T_old + zeroK;

    temp1 = (1.0 - c->albedo) * (1.0 - pen) * c->fl_sw + c->fl_rest;
    temp1 = temp1 + emi * 3.0 * sigma * POW4(T_old);
    temp1 = temp1 / (emi * 4.0 * sigma * POW3(T_old));
    temp1 = temp1 - zeroK;

    T_old = temp1 + zeroK;
    temp1 = (1.0 - c->albedo) * (1.0 - pen) * c->fl_sw + c->fl_rest;
    temp1 = temp1 + emi * 3.0 * sigma * POW4(T_old);
    temp1 = temp1 / (emi * 4.0 * sigma * POW3(T_old));
    temp1 = temp1 - zeroK;

    c->T_top = temp1;

    temp2 = pen * (1.0 - c->albedo) * c->fl_sw;                                       /* :151-155 */
    for (k = 1; k <= Na; k++) {
      e = exp(-extinc * thick[k]);
      fl_rad[k] = temp2 - temp2 * e;
      temp2 = temp2 * e;
    }

    if (c->thick_snow >= thick_min / 100.0) c->T_freeze = 0.0;
    else c->T_freeze = oracle_func_T_freeze(c->S_abs[1] / c->m[1], g->salt_flag);

    if (c->T_top > c->T_freeze && Na > 1) {                                           /* :167-181 */
      temp1 = emi * sigma * POW4(c->T_freeze + zeroK) - (1.0 - c->albedo) * (1.0 - pen) * c->fl_sw - c->fl_rest;
      if (c->thick_snow >= thick_min) {
        c->fl_Q_snow = temp1;
        fl_Q[1] = sub_fl_Q_snow(c->m_snow, c->thick_snow, c->T_snow, psi_s[1], psi_l[1], thick[1], T[1]);
      } else if (c->thick_snow >= thick_min / 100.0) {
        c->fl_Q_snow = temp1;
        fl_Q[1] = 0.0;
      } else {
        fl_Q[1] = temp1;
      }
      c->T_top = c->T_freeze;
    } else {                                                                          /* :183-194 */
      if (c->thick_snow >= thick_min) {
        fl_Q[1] = sub_fl_Q_snow(c->m_snow, c->thick_snow, c->T_snow, psi_s[1], psi_l[1], thick[1], T[1]);
        c->fl_Q_snow = sub_fl_Q_0_snow(c->m_snow, c->thick_snow, c->T_snow, c->T_top);
      } else if (c->thick_snow > thick_min / 100.0 && c->thick_snow < thick_min) {
        fl_Q[1] = 0.0;
        c->fl_Q_snow = sub_fl_Q_0_snow_thin(c->m_snow, c->thick_snow, c->T_snow, psi_s[1], psi_l[1], psi_g[1], thick[1], c->T_top);
      } else {
        fl_Q[1] = sub_fl_Q_0(psi_s[1], psi_l[1], psi_g[1], thick[1], T[1], c->T_top, -1);
      }
    }
  }

  if (g->boundflux_flag == 3) {                                                       /* :202-219, lab_snow_flag 0 */
    c->T_freeze = dmin(oracle_func_T_freeze(c->S_abs[Na] / c->m[Na], g->salt_flag), 0.0);
    c->T_top = T[1];
    fl_Q[1] = g->alpha_flux_instable * (c->T_top - c->T2m);
    if (fl_Q[1] < 0.0) {
      c->T_top = dmax(c->T_freeze, T[1]);
      fl_Q[1] = g->alpha_flux_stable * (c->T_top - c->T2m);
    }
  }

  fl_Q[Na + 1] = c->fl_q_bottom;                                                      /* :262 */

  s = 0.0; for (k = 1; k <= N; k++) s += H_abs[k];
  temp1 = s + c->H_abs_snow;                                                          /* :269 */

  for (k = 2; k <= Na; k++)
    fl_Q[k] = sub_fl_Q(psi_s[k - 1], psi_l[k - 1], psi_g[k - 1], thick[k - 1], T[k - 1], psi_s[k], psi_l[k], psi_g[k], thick[k], T[k]);

  for (k = 1; k <= Na; k++) H_abs[k] = H_abs[k] + (fl_Q[k + 1] - fl_Q[k]) * dt;

  for (k = 1; k <= Na; k++) {                                                         /* :282-285: fl_rad(N_active) */
    H_abs[k] = H_abs[k] + fl_rad[Na] * dt;
    temp1 = temp1 + fl_rad[Na] * dt;
  }

  if (c->thick_snow >= thick_min / 100.0 && c->thick_snow < thick_min) {              /* :291-303 */
    c->H_abs_snow = c->H_abs_snow - c->fl_Q_snow * dt;
    snow_coupling(c); CHECK();
    temp1 = temp1 + c->fl_q_bottom * dt - c->fl_Q_snow * dt;
  } else if (c->thick_snow >= thick_min) {
    c->H_abs_snow = c->H_abs_snow + (fl_Q[1] - c->fl_Q_snow) * dt;
    temp1 = temp1 + c->fl_q_bottom * dt - c->fl_Q_snow * dt;
  } else {
    temp1 = temp1 + c->fl_q_bottom * dt - fl_Q[1] * dt;
  }

  s = 0.0; for (k = 1; k <= N; k++) s += H_abs[k];
  temp2 = s + c->H_abs_snow;
  if (fabs((temp1 - temp2) / dt) > 0.00001) STOP(431, 0);
}

/* function-level entry points for the secondary parametrisations (unit-level golden vectors) */
void oracle_flood_simple(double freeboard, double *S_abs1, double *H_abs1, double *m1, double *thick1, double T_bottom,
                         double S_bu_bottom, double psi_g_snow, double *H_abs_snow, double *m_snow, double *thick_snow) {
  samsim_config g; column c; double S[2], H[2], m[2], th[2];
  memset(&g, 0, sizeof g); memset(&c, 0, sizeof c);
  g.T_bottom = T_bottom; g.S_bu_bottom = S_bu_bottom; c.S_bu_bottom = S_bu_bottom;
  S[1] = *S_abs1; H[1] = *H_abs1; m[1] = *m1; th[1] = *thick1;
  c.cfg = &g; c.S_abs = S; c.H_abs = H; c.m = m; c.thick = th;
  c.freeboard = freeboard; c.psi_g_snow = psi_g_snow; c.H_abs_snow = *H_abs_snow; c.m_snow = *m_snow; c.thick_snow = *thick_snow;
  flood_simple(&c);
  *S_abs1 = S[1]; *H_abs1 = H[1]; *m1 = m[1]; *thick1 = th[1];
  *H_abs_snow = c.H_abs_snow; *m_snow = c.m_snow; *thick_snow = c.thick_snow;
}

void oracle_fl_grav_drain_simple(int N, int N_active, int harmonic_flag, const double *psi_s, const double *psi_l,
                                 const double *thick, const double *S_br, double *S_abs, double *ray) {
  samsim_config g; column c;
  memset(&g, 0, sizeof g); memset(&c, 0, sizeof c);
  g.harmonic_flag = harmonic_flag;
  c.cfg = &g; c.N = N; c.N_active = N_active;
  c.psi_s = (double *)psi_s; c.psi_l = (double *)psi_l; c.thick = (double *)thick; c.S_br = (double *)S_br;
  c.S_abs = S_abs; c.ray = ray;
  fl_grav_drain_simple(&c);
}

void oracle_sub_notzflux(double time, double *fl_sw, double *fl_rest) { sub_notzflux(time, fl_sw, fl_rest); }

/* ------------------------------------------------------------------ mo_grotz.f90:182-835 */

static double freeboard_now(column *c) {
  return oracle_func_freeboard(c->N_active, c->psi_s, c->psi_g, c->m, c->thick, c->m_snow, c->cfg->freeboard_snow_flag);
}

/* first part of the loop body, up to (not including) the output block: mo_grotz.f90:192-335 */
static void step_part_a(column *c) {
  const samsim_config *g = c->cfg;
  int Na = c->N_active, k, jj;
  double sH, sm, sS, temp, T_test, s1, s2;
  double *H_abs = c->H_abs, *S_abs = c->S_abs, *m = c->m, *thick = c->thick, *psi_l = c->psi_l, *psi_s = c->psi_s;

  /* vital signs :192-223 */
  sH = 0.0; sm = 0.0; sS = 0.0;
  for (k = 1; k <= Na; k++) sH += H_abs[k];
  for (k = 1; k <= Na; k++) sm += m[k];
  for (k = 1; k <= Na; k++) sS += S_abs[k];
  c->energy_stored = c->H_abs_snow + sH - g->T_bottom * sm * c_l;
  c->freshwater = sm / rho_l;
  c->freshwater = c->freshwater * (1.0 - sS / sm / ref_salinity);
  c->freshwater = c->freshwater + c->m_snow / rho_l;
  c->total_resist = 0.0;
  for (jj = 1; jj <= Na - 1; jj++) c->total_resist = c->total_resist + thick[jj] / (psi_l[jj] * k_l + psi_s[jj] * k_s);
  c->total_resist = c->total_resist + thick[Na] * psi_s[Na] / psi_s_min * (psi_s_min * k_s + 1.0 - psi_s_min * k_l);
  if (c->thick_snow > g->thick_min / 110.0) c->total_resist = c->total_resist + c->thick_snow / oracle_func_k_snow(c->m_snow, c->thick_snow);
  if (Na > 1) { s1 = 0.0; for (k = 1; k <= Na - 1; k++) s1 += thick[k]; c->thickness = s1; }
  else c->thickness = 0.0;
  c->thickness = c->thickness + thick[Na] * psi_s[Na] / psi_s_min;
  if (Na > 1) {
    s1 = 0.0; for (k = 1; k <= Na - 1; k++) s1 += S_abs[k];
    s2 = 0.0; for (k = 1; k <= Na - 1; k++) s2 += m[k];
    c->bulk_salin = s1 + S_abs[Na] * psi_s[Na] / psi_s_min;
    c->bulk_salin = c->bulk_salin / (s2 + m[Na] * psi_s[Na] / psi_s_min);
  } else {
    c->bulk_salin = S_abs[1] / m[1];
  }

  /* forcing :229-241 */
  if (g->atmoflux_flag == 2) {
    int tc = c->time_counter;
    if (c->time > time_input(tc)) tc = tc + 1;
    if (tc > c->flen) tc = c->flen;           /* the reference would read past the table; clamp */
    c->time_counter = tc;
    if (c->time == time_input(tc)) {
      c->T2m = c->T2m_input[tc - 1];
      c->liquid_precip = c->precip_input[tc - 1];
    } else {
      temp = (c->time - time_input(tc - 1)) / (time_input(tc) - time_input(tc - 1));
      c->T2m = (1.0 - temp) * c->T2m_input[tc - 2] + temp * c->T2m_input[tc - 1];
      c->liquid_precip = (1.0 - temp) * c->precip_input[tc - 2] + temp * c->precip_input[tc - 1];
    }
    /* ensemble perturbation; identity for dT2m = 0, precip_scale = 1 */
    c->T2m = c->T2m + c->dT2m;
    c->liquid_precip = c->liquid_precip * c->precip_scale;
  }

  /* snow fall :251-265 */
  if (g->precip_flag == 1) {
    if (dmax(c->liquid_precip, c->solid_precip) > 0.0 && Na > 1) snow_precip(c, 0);
    else if (dmax(c->liquid_precip, c->solid_precip) > 0.0 && Na == 1) snow_precip_0(c, 0);
  } else if (g->precip_flag == 0) {
    if (dmax(c->liquid_precip, c->solid_precip) > 0.0 && Na > 1) snow_precip(c, 1);
    else if (dmax(c->liquid_precip, c->solid_precip) > 0.0 && Na == 1) snow_precip_0(c, 1);
  }

  /* snow thermodynamics :273-292 */
  snow_block(c); CHECK();

  /* inner layer thermodynamics and expulsion :297-307 */
  T_test = g->T_bottom;
  for (k = Na; k >= 1; k--) {
    c->S_bu[k] = S_abs[k] / m[k];
    c->H[k] = H_abs[k] / m[k];
    getT(c, c->H[k], c->S_bu[k], T_test, &c->T[k], &c->phi[k], k); CHECK();
    T_test = c->T[k];
    c->S_br[k] = S_BR2(c->T[k], c->S_bu[k]);
    oracle_Expulsion(c->phi[k], thick[k], m[k], &c->psi_s[k], &c->psi_l[k], &c->psi_g[k], &c->V_ex[k]);
  }

  /* brine flux due to expulsion :312-321 */
  expulsion_flux(c);
  if (c->step + 1 != 1) {
    mass_transfer(c, c->fl_m);
    for (k = 1; k <= Na && c->n_bgc > 0; k++) FLB(k, k + 1) = -c->fl_m[k + 1];       /* :316-320 */
  }

  for (k = Na; k >= 1; k--) c->S_bu[k] = S_abs[k] / m[k];                             /* :333-335 */
}

static void take_snapshot(column *c) {
  int N = c->N, k;
  double *L = c->snap_lay;
  const double *src[SAMSIM_NARR];
  if (!L) return;
  src[SAMSIM_A_H_ABS] = c->H_abs; src[SAMSIM_A_S_ABS] = c->S_abs; src[SAMSIM_A_M] = c->m; src[SAMSIM_A_THICK] = c->thick;
  src[SAMSIM_A_T] = c->T; src[SAMSIM_A_PHI] = c->phi; src[SAMSIM_A_PSI_S] = c->psi_s; src[SAMSIM_A_PSI_L] = c->psi_l;
  src[SAMSIM_A_PSI_G] = c->psi_g; src[SAMSIM_A_S_BU] = c->S_bu; src[SAMSIM_A_S_BR] = c->S_br; src[SAMSIM_A_RAY] = c->ray;
  src[SAMSIM_A_PERM] = c->perm; src[SAMSIM_A_FLUSH_V] = c->flush_v; src[SAMSIM_A_FLUSH_H] = c->flush_h;
  for (int a = 0; a < SAMSIM_NARR; a++) for (k = 1; k <= N; k++) L[(size_t)a * N + (k - 1)] = src[a][k];
  double *s = c->snap_scal;
  memset(s, 0, sizeof(c->snap_scal));
  s[SAMSIM_S_M_SNOW] = c->m_snow; s[SAMSIM_S_H_ABS_SNOW] = c->H_abs_snow; s[SAMSIM_S_S_ABS_SNOW] = c->S_abs_snow;
  s[SAMSIM_S_THICK_SNOW] = c->thick_snow; s[SAMSIM_S_PSI_S_SNOW] = c->psi_s_snow; s[SAMSIM_S_PSI_L_SNOW] = c->psi_l_snow;
  s[SAMSIM_S_PSI_G_SNOW] = c->psi_g_snow; s[SAMSIM_S_T_SNOW] = c->T_snow; s[SAMSIM_S_PHI_S] = c->phi_s;
  s[SAMSIM_S_T_TOP] = c->T_top; s[SAMSIM_S_MELT_THICK] = c->melt_thick; s[SAMSIM_S_T2M] = c->T2m;
  s[SAMSIM_S_LIQUID_PRECIP] = c->liquid_precip; s[SAMSIM_S_SOLID_PRECIP] = c->solid_precip; s[SAMSIM_S_FL_Q_BOTTOM] = c->fl_q_bottom;
  s[SAMSIM_S_GRAV_DRAIN] = c->grav_drain; s[SAMSIM_S_GRAV_SALT] = c->grav_salt; s[SAMSIM_S_GRAV_TEMP] = c->grav_temp;
  s[SAMSIM_S_MELT_OUT1] = c->melt_thick_output[0]; s[SAMSIM_S_MELT_OUT2] = c->melt_thick_output[1];
  s[SAMSIM_S_MELT_OUT3] = c->melt_thick_output[2]; s[SAMSIM_S_MELT_ERR] = c->melt_err;
  s[SAMSIM_S_FREEBOARD] = c->freeboard; s[SAMSIM_S_T_FREEZE] = c->T_freeze; s[SAMSIM_S_ALBEDO] = c->albedo;
  s[SAMSIM_S_FL_SW] = c->fl_sw; s[SAMSIM_S_FL_LW] = c->fl_lw; s[SAMSIM_S_MELT_THICK_SNOW] = c->melt_thick_snow;
  s[SAMSIM_S_FL_Q_SNOW] = c->fl_Q_snow;
  s[SAMSIM_S_ENERGY_STORED] = c->energy_stored; s[SAMSIM_S_FRESHWATER] = c->freshwater; s[SAMSIM_S_TOTAL_RESIST] = c->total_resist;
  s[SAMSIM_S_THICKNESS] = c->thickness; s[SAMSIM_S_BULK_SALIN] = c->bulk_salin; s[SAMSIM_S_FL_REST] = c->fl_rest; s[SAMSIM_S_S_BU_BOTTOM] = c->S_bu_bottom;
  s[SAMSIM_S_DT2M] = c->dT2m; s[SAMSIM_S_PRECIP_SCALE] = c->precip_scale;
  for (int t = 0; t < c->n_bgc && c->snap_bgc; t++) {
    for (k = 1; k <= N; k++) c->snap_bgc[(size_t)t * N + (k - 1)] = c->bgc[t][k];
    c->snap_bgc_bottom[t] = c->bgc_bottom[t];
  }
  c->snap_valid = 1; c->snap_time = c->time; c->snap_step = c->step + 1; c->snap_N_active = c->N_active;
}

/* the output block, mo_grotz.f90:340-398 */
static void output_point(column *c) {
  const samsim_config *g = c->cfg;
  if (c->n_time_out == g->i_time_out || c->step + 1 == 1) {
    if (c->N_active > 1) c->freeboard = freeboard_now(c); else c->freeboard = 0.0;
    if (g->grav_flag == 2) {
      if (c->grav_drain == 0.0) c->grav_temp = 0.0; else c->grav_temp = c->grav_temp / c->grav_drain;
      c->grav_salt = c->grav_salt / g->time_out;
      c->grav_drain = c->grav_drain / g->time_out;
    }
    take_snapshot(c);
    c->n_outputs++;
    c->grav_drain = 0.0; c->grav_salt = 0.0; c->grav_temp = 0.0;
    c->melt_thick_output[0] = 0.0; c->melt_thick_output[1] = 0.0; c->melt_thick_output[2] = 0.0;
    c->n_time_out = 0;
  } else {
    c->n_time_out = c->n_time_out + 1;
  }
}

/* SUM(thick(a:b)) as the reference forms it: in ascending index order */
static double thick_sum(const column *c, int a, int b) {
  double s = 0.0;
  for (int k = a; k <= b; k++) s += c->thick[k];
  return s;
}

/* prescribe_flag 2, mo_grotz.f90:482-497: linear from S_bu_bottom to 4 over the lowest 0.15 m, from 4 to 0 above it.
 * Layers the two loops do not reach keep the S_bu of the first sweep (layer 1 of ice thinner than 0.15 m). */
static void prescribe_salinity(column *c) {
  int N = c->N, Na = c->N_active, k = Na;
  double Sb = c->S_bu_bottom;
  while (k > 1 && thick_sum(c, k, Na) < 0.15) {
    c->S_bu[k] = Sb - thick_sum(c, k, Na) / 0.15 * (Sb - 4.0);
    k = k - 1;
  }
  while (k > 1 && thick_sum(c, k, Na) >= 0.15) {
    c->S_bu[k] = 4.0 - 4.0 * (thick_sum(c, k, Na) - 0.15) / (thick_sum(c, 1, Na) - 0.15);
    k = k - 1;
    c->S_bu[1] = 0.0;
  }
  c->S_bu[Na] = Sb;
  for (k = 1; k <= N; k++) c->S_abs[k] = c->S_bu[k] * c->m[k];
}

/* flush4, mo_flush.f90:253-296 (flush_flag 6): the melt water leaves the top layer with its brine salinity; every layer
 * that is more liquid than the one above it loses the fraction 1 - para_flush_gamma of its salt, down to the first one
 * that is not.  The reference's loop has no upper bound on k (it would read psi_l(Nlayer+1) on a column whose liquid
 * fraction rises all the way down); here it ends at Nlayer. */
static void flush4(column *c) {
  int N = c->N, k;
  double S_bu1 = c->S_abs[1] / c->m[1], mn;
  c->H_abs[1] = c->H_abs[1] - c->melt_thick * rho_l * c_l * c->T[1];
  c->S_abs[1] = c->S_abs[1] - c->melt_thick * rho_l * S_BR2(c->T[1], S_bu1);
  c->thick[1] = c->thick[1] - c->melt_thick;
  c->m[1] = c->m[1] - c->melt_thick * rho_l;
  c->melt_thick = 0.0;
  k = 2;
  while (k <= N && c->psi_l[k] > c->psi_l[k - 1]) {
    c->S_abs[k] = para_flush_gamma * c->S_abs[k];
    k = k + 1;
  }
  c->S_abs[1] = (c->S_abs[1] > 0.0) ? c->S_abs[1] : 0.0;
  mn = c->S_abs[1]; for (k = 1; k <= N; k++) if (c->S_abs[k] < mn) mn = c->S_abs[k];
  if (mn < 0.0) STOP(9876, 0);
}

/* rest of the loop body: mo_grotz.f90:405-819 */
static void step_part_b(column *c) {
  const samsim_config *g = c->cfg;
  int N = c->N, Na = c->N_active, k;
  double temp2, T_test, mn;
  double *H_abs = c->H_abs, *S_abs = c->S_abs, *m = c->m, *thick = c->thick;

  /* bottom-layer gas -> ocean water :405-410 */
  if (c->psi_g[Na] > 0.0) {
    temp2 = c->psi_g[Na] * thick[Na] * rho_l;
    m[Na] = m[Na] + temp2;
    S_abs[Na] = S_abs[Na] + temp2 * c->S_bu_bottom;
    H_abs[Na] = H_abs[Na] + temp2 * c_l * g->T_bottom;
  }

  /* thin-snow coupling :418-420 */
  if (c->m_snow > 0.0 && c->thick_snow < g->thick_min) { snow_coupling(c); CHECK(); }

  /* flooding :428-445 */
  if (Na > 1 && g->flood_flag > 1) {
    c->freeboard = freeboard_now(c);
    if (c->freeboard < 0.0) {
      if (g->flood_flag == 2) flood(c);
      else if (g->flood_flag == 3 && c->freeboard < neg_free) flood_simple(c);
    }
  }

  /* bottom turbulence :450-457 */
  if (g->turb_flag == 2) {
    double turb = sub_turb_flux(g->T_bottom, c->S_bu_bottom, c->T[Na], &S_abs[Na], m[Na], g->dt);
    for (int t = 0; t < c->n_bgc; t++) c->bgc[t][Na] = c->bgc[t][Na] - turb * (c->bgc[t][Na] / m[Na] - c->bgc_bottom[t]);
  }

  /* gravity drainage :463-477 */
  if (g->grav_flag == 2 && Na > 1) { fl_grav_drain(c); CHECK(); }
  else if (g->grav_flag == 3 && Na > 1) fl_grav_drain_simple(c);

  /* prescribed salinity profile :482-497 */
  if (g->prescribe_flag == 2) prescribe_salinity(c);

  /* testcase specifics :503-565 */
  if (g->testcase == 1) sub_test1(c->time, &c->T_top);
  else if (g->testcase == 3) { c->liquid_precip = 0.0; c->solid_precip = 0.15 / 86400.0 / 356.0; }  /* sub_test3, :172-187 */
  else if (g->testcase == 4 || g->testcase == 7) {
    sub_test4(c->time, &c->fl_q_bottom);
    if (c->dflq != 0.0) c->fl_q_bottom = c->fl_q_bottom + c->dflq;   /* a grid of columns: samsim_set_ocean */
  }
  else if (g->testcase == 5 && c->step + 1 == 2) { for (k = 1; k <= N; k++) S_abs[k] = 5.0 * m[k]; }   /* mo_grotz.f90:543-544 */
  else if (g->testcase == 2) sub_test2(c->time, &c->T2m);
  else if (g->testcase == 6) sub_test6(c->time, &c->T2m);
  else if (g->testcase == 9) sub_test9(c->time, &c->T2m);
  else if (g->testcase == 34) sub_test34(c->time, &c->T2m);

  /* tank: the water below the ice holds what salt the ice does not, mo_grotz.f90:573-575 */
  if (g->tank_flag == 2) {
    double sS = 0.0, sm = 0.0;
    for (k = 1; k <= N; k++) sS += S_abs[k];
    for (k = 1; k <= N; k++) sm += m[k];
    c->S_bu_bottom = (g->S_total - sS) / (g->m_total - sm);
    if (c->n_bgc > 0) {                                                                /* :575-577 (sic: tracer 1 sets them all) */
      double sb = 0.0;
      for (k = 1; k <= N; k++) sb += c->bgc[0][k];
      for (int t = 0; t < c->n_bgc; t++) c->bgc_bottom[t] = (c->bgc_total[0] - sb) / (g->m_total - sm);
    }
  }

  /* heat fluxes :584 */
  sub_heat_fluxes(c); CHECK();

  /* second thermodynamic sweep :592-598 */
  T_test = g->T_bottom;
  for (k = Na; k >= 1; k--) {
    c->S_bu[k] = S_abs[k] / m[k];
    c->H[k] = H_abs[k] / m[k];
    getT(c, c->H[k], c->S_bu[k], T_test, &c->T[k], &c->phi[k], k); CHECK();
    T_test = c->T[k];
  }

  c->melt_thick_snow_old = c->melt_thick_snow;                                        /* :603 */
  snow_block(c); CHECK();                                                             /* :604-624 */
  c->melt_thick_snow = c->melt_thick_snow_old + c->melt_thick_snow;                   /* :625 */

  /* flushing preparations :632-664 */
  if (Na > 1 && g->flush_flag > 2) {
    if (g->boundflux_flag == 3) {                                                     /* :649-663: T2m in place of T_top */
      c->T_freeze = oracle_func_T_freeze(S_abs[1] / m[1], g->salt_flag);
      c->melt_thick = 0.0;
      if (freeboard_now(c) > 0.0000000000001) {
        if (c->psi_s[1] < psi_s_top_min || c->T2m >= c->T_freeze) {
          sub_melt_thick(c->psi_l[1], c->psi_s[1], c->psi_g[1], c->T[1], c->T_freeze, c->T2m, c->fl_Q[1], c->thick_snow,
                         g->dt, &c->melt_thick, &thick[1], g->thick_min);
          c->melt_thick = dmax(c->melt_thick, 0.0);
          if (c->thick_snow >= g->thick_min / 100.0 && c->melt_thick > 0.00000000001 && c->melt_thick_snow == 0.0)
            sub_melt_snow(&c->melt_thick, &thick[1], &c->thick_snow, &H_abs[1], &c->H_abs_snow, &m[1], &c->m_snow, &c->psi_g_snow);
        }
      }
    }
    if (g->boundflux_flag == 2) {
      c->T_freeze = oracle_func_T_freeze(S_abs[1] / m[1], g->salt_flag);
      c->melt_thick = 0.0;
      if (freeboard_now(c) > 0.0000000000001) {
        if (c->psi_s[1] < psi_s_top_min || c->T_top >= c->T_freeze) {
          sub_melt_thick(c->psi_l[1], c->psi_s[1], c->psi_g[1], c->T[1], c->T_freeze, c->T_top, c->fl_Q[1], c->thick_snow,
                         g->dt, &c->melt_thick, &thick[1], g->thick_min);
          if (c->thick_snow >= g->thick_min / 100.0 && c->melt_thick > 0.00000000001 && c->melt_thick_snow == 0.0)
            sub_melt_snow(&c->melt_thick, &thick[1], &c->thick_snow, &H_abs[1], &c->H_abs_snow, &m[1], &c->m_snow, &c->psi_g_snow);
        }
      }
    }
  }

  /* flushing :670-737 */
  c->freeboard = freeboard_now(c);
  c->melt_thick_output[0] = c->melt_thick_output[0] + c->melt_thick;
  c->melt_thick_output[1] = c->melt_thick_output[1] + c->melt_thick_snow;
  c->melt_thick = c->melt_thick + c->melt_thick_snow;
  if (c->melt_thick_snow > 0.0) {
    H_abs[1] = H_abs[1] + c->melt_thick_snow * rho_l * c_l * c->T_snow;
    S_abs[1] = S_abs[1] + c->melt_thick_snow * rho_l * S_BR2(c->T_snow, c->S_abs_snow / c->m_snow);
    thick[1] = thick[1] + c->melt_thick_snow;
    m[1] = m[1] + c->melt_thick_snow * rho_l;
    c->S_bu[1] = S_abs[1] / m[1];
    c->H[1] = H_abs[1] / m[1];
  }
  {
    double fv_old[SAMSIM_MAX_NLAYER + 2], fh_old[SAMSIM_MAX_NLAYER + 2];
    for (k = 1; k <= N; k++) { fv_old[k] = c->flush_v[k]; fh_old[k] = c->flush_h[k]; c->flush_v[k] = 0.0; c->flush_h[k] = 0.0; }
    if (Na > 1 && c->freeboard > 0.001) {
      if (g->flush_flag == 4) {                                                        /* mo_grotz.f90:704-713 */
        if (c->melt_thick > 0.000000000001 && Na > 2) {
          H_abs[1] = H_abs[1] - c->melt_thick * rho_l * c_l * c->T[1];
          S_abs[1] = S_abs[1] * (1.0 - (c->melt_thick * rho_l) / m[1]);
          thick[1] = thick[1] - c->melt_thick;
          m[1] = m[1] - c->melt_thick * rho_l;
        }
      } else if (g->flush_flag == 5) {
        if (c->melt_thick > 0.000000000001 && Na > 2 && c->freeboard > 0.0) {
          c->freeboard = freeboard_now(c);
          flush3(c); CHECK();
        }
      } else if (g->flush_flag == 6) {                                                 /* :729-733 */
        if (c->melt_thick > 0.000000000001 && Na > 2 && c->thick_snow < g->thick_0) { flush4(c); CHECK(); }
      }
    }
    for (k = 1; k <= N; k++) { c->flush_v[k] = c->flush_v[k] + fv_old[k]; c->flush_h[k] = c->flush_h[k] + fh_old[k]; }
  }

  /* tracer advection with this step's brine fluxes :742-747 */
  if (c->n_bgc > 0) bgc_advection(c);

  /* layer dynamics :755-795 */
  if (Na > 1) {
    if (c->phi[Na] > psi_s_min || c->phi[Na - 1] <= psi_s_min / 2.0 || thick[1] / g->thick_0 > 1.5 || thick[1] / g->thick_0 < 0.5) {
      layer_dynamics(c); CHECK();
    }
    Na = c->N_active;
    if (Na < N && thick[(Na + 1 < N) ? Na + 1 : N] == 0.0) {                          /* :772-783 scrub */
      c->T[Na + 1] = g->T_bottom; c->S_bu[Na + 1] = c->S_bu_bottom; c->H[Na + 1] = 0.0;
      c->psi_l[Na + 1] = 1.0; c->psi_s[Na + 1] = 0.0;
      for (int t = 0; t < c->n_bgc; t++) c->bgc[t][Na + 1] = 0.0;
    }
  } else {
    if (c->phi[1] > psi_s_min) { layer_dynamics(c); CHECK(); }
  }
  Na = c->N_active;

  c->time = c->time + g->dt;                                                          /* :802 */

  /* health check :808-819 */
  mn = c->psi_s[1]; for (k = 1; k <= Na; k++) if (c->psi_s[k] < mn) mn = c->psi_s[k];
  if (mn < 0.0) STOP(1337, 0);
  mn = S_abs[1]; for (k = 1; k <= Na; k++) if (S_abs[k] < mn) mn = S_abs[k];
  if (mn < 0.0) for (k = 1; k <= Na; k++) S_abs[k] = dmax(S_abs[k], 0.0);
}

static void column_step(column *c) {
  if (c->status) return;
  c->work += c->N_active;
  step_part_a(c); if (c->status) return;
  output_point(c);
  step_part_b(c); if (c->status) return;
  c->step++;
}

/* ------------------------------------------------------------------ batch API (mirrors include/samsim.h) */

struct oracle_handle {
  samsim_config cfg;
  int64_t ncol;
  column *cols;
  double *f_sw, *f_lw, *f_T2m, *f_precip;
  int flen;
  int64_t out_col0, out_ncols;
  int nthreads;
  int n_bgc;
  double *bgc_total;
};

static double *dupd(const double *p, int n);
static double *lay_alloc(int N) { return (double *)calloc((size_t)N + 3, sizeof(double)); }

int oracle_create(const samsim_config *cfg, int64_t ncol, oracle_handle **out) {
  if (!cfg || !out || ncol <= 0) return SAMSIM_ERR_ARG;
  if (cfg->struct_size != (int32_t)sizeof(samsim_config)) return SAMSIM_ERR_ABI;
  int N = cfg->nlayer;
  if (N < 3 || N > SAMSIM_MAX_NLAYER || cfg->n_top + cfg->n_middle + cfg->n_bottom != N) return SAMSIM_ERR_ARG;
  oracle_handle *h = (oracle_handle *)calloc(1, sizeof(*h));
  h->cfg = *cfg; h->ncol = ncol; h->nthreads = 1;
  h->cols = (column *)calloc((size_t)ncol, sizeof(column));
  for (int64_t i = 0; i < ncol; i++) {
    column *c = &h->cols[i];
    c->cfg = &h->cfg; c->N = N; c->N_active = 1;
    c->H_abs = lay_alloc(N); c->S_abs = lay_alloc(N); c->m = lay_alloc(N); c->thick = lay_alloc(N);
    c->T = lay_alloc(N); c->phi = lay_alloc(N); c->psi_s = lay_alloc(N); c->psi_l = lay_alloc(N); c->psi_g = lay_alloc(N);
    c->S_bu = lay_alloc(N); c->S_br = lay_alloc(N); c->H = lay_alloc(N); c->V_ex = lay_alloc(N);
    c->fl_Q = lay_alloc(N); c->fl_m = lay_alloc(N); c->fl_rad = lay_alloc(N); c->ray = lay_alloc(N); c->perm = lay_alloc(N);
    c->flush_v = lay_alloc(N); c->flush_h = lay_alloc(N);
    /* mo_init.f90:1982-1990 */
    for (int k = 1; k <= N; k++) { c->T[k] = cfg->T_bottom; c->S_bu[k] = cfg->S_bu_bottom; c->psi_l[k] = 1.0; }
    c->time_counter = 1; c->precip_scale = 1.0; c->S_bu_bottom = cfg->S_bu_bottom;
  }
  h->out_col0 = 0; h->out_ncols = 1;
  h->cols[0].snap_lay = (double *)calloc((size_t)SAMSIM_NARR * N, sizeof(double));
  *out = h;
  return SAMSIM_OK;
}

void oracle_destroy(oracle_handle *h) {
  if (!h) return;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    free(c->H_abs); free(c->S_abs); free(c->m); free(c->thick); free(c->T); free(c->phi); free(c->psi_s); free(c->psi_l);
    free(c->psi_g); free(c->S_bu); free(c->S_br); free(c->H); free(c->V_ex); free(c->fl_Q); free(c->fl_m); free(c->fl_rad);
    free(c->ray); free(c->perm); free(c->flush_v); free(c->flush_h); free(c->snap_lay);
    for (int t = 0; t < SAMSIM_MAX_NBGC; t++) free(c->bgc[t]);
    free(c->flb); free(c->snap_bgc);
  }
  free(h->bgc_total);
  free(h->cols); free(h->f_sw); free(h->f_lw); free(h->f_T2m); free(h->f_precip);
  free(h);
}

/* tracers: mirror of samsim_set_tracers / set_tracer_state / get_tracer_state / get_tracer_output (include/samsim.h) */
int oracle_set_tracers(oracle_handle *h, int32_t n_bgc, const double *bgc_bottom, const double *bgc_total) {
  if (!h || h->cfg.bgc_flag != 2 || n_bgc < 1 || n_bgc > SAMSIM_MAX_NBGC || !bgc_bottom) return SAMSIM_ERR_ARG;
  if (h->cfg.tank_flag == 2 && !bgc_total) return SAMSIM_ERR_ARG;
  int N = h->cfg.nlayer;
  free(h->bgc_total);
  h->bgc_total = bgc_total ? dupd(bgc_total, n_bgc) : NULL;
  h->n_bgc = n_bgc;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    c->n_bgc = n_bgc; c->bgc_total = h->bgc_total;
    for (int t = 0; t < n_bgc; t++) {
      if (!c->bgc[t]) c->bgc[t] = lay_alloc(N);
      c->bgc_bottom[t] = bgc_bottom[t];
    }
    if (!c->flb) c->flb = (double *)calloc((size_t)(N + 2) * (size_t)(N + 2), sizeof(double));
    if (c->snap_lay && !c->snap_bgc) c->snap_bgc = (double *)calloc((size_t)SAMSIM_MAX_NBGC * N, sizeof(double));
  }
  return SAMSIM_OK;
}

int oracle_set_tracer_state(oracle_handle *h, const double *bgc_abs, int64_t col0, int64_t ncols) {
  if (!h || !bgc_abs || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  int N = h->cfg.nlayer;
  for (int64_t i = 0; i < ncols; i++)
    for (int t = 0; t < h->n_bgc; t++)
      for (int k = 1; k <= N; k++) h->cols[col0 + i].bgc[t][k] = bgc_abs[((size_t)t * N + (k - 1)) * ncols + i];
  return SAMSIM_OK;
}

int oracle_set_tracer_bottom(oracle_handle *h, const double *bgc_bottom, int64_t col0, int64_t ncols) {
  if (!h || !bgc_bottom || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < ncols; i++)
    for (int t = 0; t < h->n_bgc; t++) h->cols[col0 + i].bgc_bottom[t] = bgc_bottom[(size_t)t * ncols + i];
  return SAMSIM_OK;
}

int oracle_get_tracer_state(oracle_handle *h, double *bgc_abs, double *bgc_bottom, int64_t col0, int64_t ncols) {
  if (!h || !bgc_abs || h->n_bgc < 1 || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  int N = h->cfg.nlayer;
  for (int64_t i = 0; i < ncols; i++)
    for (int t = 0; t < h->n_bgc; t++) {
      for (int k = 1; k <= N; k++) bgc_abs[((size_t)t * N + (k - 1)) * ncols + i] = h->cols[col0 + i].bgc[t][k];
      if (bgc_bottom) bgc_bottom[(size_t)t * ncols + i] = h->cols[col0 + i].bgc_bottom[t];
    }
  return SAMSIM_OK;
}

int oracle_get_tracer_output(oracle_handle *h, double *bgc_abs, double *bgc_bottom) {
  if (!h || !bgc_abs || h->n_bgc < 1) return SAMSIM_ERR_ARG;
  int N = h->cfg.nlayer; int64_t w = h->out_ncols;
  for (int64_t i = 0; i < w; i++) {
    column *c = &h->cols[h->out_col0 + i];
    if (!c->snap_valid || !c->snap_bgc) return SAMSIM_ERR_NO_OUTPUT;
    for (int t = 0; t < h->n_bgc; t++) {
      for (int k = 0; k < N; k++) bgc_abs[((size_t)t * N + k) * w + i] = c->snap_bgc[(size_t)t * N + k];
      if (bgc_bottom) bgc_bottom[(size_t)t * w + i] = c->snap_bgc_bottom[t];
    }
  }
  return SAMSIM_OK;
}

void oracle_set_threads(oracle_handle *h, int n) { if (h && n > 0) h->nthreads = n; }

static double *dupd(const double *p, int n) { double *q = (double *)malloc(sizeof(double) * (size_t)n); memcpy(q, p, sizeof(double) * (size_t)n); return q; }

int oracle_set_forcing(oracle_handle *h, int32_t len, const double *fl_sw, const double *fl_lw, const double *T2m,
                       const double *precip, const double *dT2m_col, const double *precip_scale_col) {
  return oracle_set_forcing_sites(h, 1, len, fl_sw, fl_lw, T2m, precip, NULL, dT2m_col, precip_scale_col);
}

/* mirror of samsim_set_forcing_sites: every column reads the tables of its own site */
int oracle_set_forcing_sites(oracle_handle *h, int32_t nsites, int32_t len, const double *fl_sw, const double *fl_lw,
                             const double *T2m, const double *precip, const int32_t *site_of_column,
                             const double *dT2m_col, const double *precip_scale_col) {
  if (!h || len < 2 || nsites < 1 || !fl_sw || !fl_lw || !T2m || !precip || (nsites > 1 && !site_of_column)) return SAMSIM_ERR_ARG;
  free(h->f_sw); free(h->f_lw); free(h->f_T2m); free(h->f_precip);
  h->f_sw = dupd(fl_sw, len * nsites); h->f_lw = dupd(fl_lw, len * nsites); h->f_T2m = dupd(T2m, len * nsites);
  h->f_precip = dupd(precip, len * nsites);
  h->flen = len;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    size_t off = (nsites > 1) ? (size_t)site_of_column[i] * (size_t)len : 0;
    if (nsites > 1 && (site_of_column[i] < 0 || site_of_column[i] >= nsites)) return SAMSIM_ERR_ARG;
    c->flen = len; c->fl_sw_input = h->f_sw + off; c->fl_lw_input = h->f_lw + off; c->T2m_input = h->f_T2m + off;
    c->precip_input = h->f_precip + off;
    c->dT2m = dT2m_col ? dT2m_col[i] : 0.0;
    c->precip_scale = precip_scale_col ? precip_scale_col[i] : 1.0;
  }
  return SAMSIM_OK;
}

static double *scal_slot(column *c, int idx) {
  switch (idx) {
    case SAMSIM_S_M_SNOW: return &c->m_snow; case SAMSIM_S_H_ABS_SNOW: return &c->H_abs_snow;
    case SAMSIM_S_S_ABS_SNOW: return &c->S_abs_snow; case SAMSIM_S_THICK_SNOW: return &c->thick_snow;
    case SAMSIM_S_PSI_S_SNOW: return &c->psi_s_snow; case SAMSIM_S_PSI_L_SNOW: return &c->psi_l_snow;
    case SAMSIM_S_PSI_G_SNOW: return &c->psi_g_snow; case SAMSIM_S_T_SNOW: return &c->T_snow;
    case SAMSIM_S_PHI_S: return &c->phi_s; case SAMSIM_S_T_TOP: return &c->T_top;
    case SAMSIM_S_MELT_THICK: return &c->melt_thick; case SAMSIM_S_T2M: return &c->T2m;
    case SAMSIM_S_LIQUID_PRECIP: return &c->liquid_precip; case SAMSIM_S_SOLID_PRECIP: return &c->solid_precip;
    case SAMSIM_S_FL_Q_BOTTOM: return &c->fl_q_bottom;
    case SAMSIM_S_GRAV_DRAIN: return &c->grav_drain; case SAMSIM_S_GRAV_SALT: return &c->grav_salt;
    case SAMSIM_S_GRAV_TEMP: return &c->grav_temp;
    case SAMSIM_S_MELT_OUT1: return &c->melt_thick_output[0]; case SAMSIM_S_MELT_OUT2: return &c->melt_thick_output[1];
    case SAMSIM_S_MELT_OUT3: return &c->melt_thick_output[2]; case SAMSIM_S_MELT_ERR: return &c->melt_err;
    case SAMSIM_S_FREEBOARD: return &c->freeboard; case SAMSIM_S_T_FREEZE: return &c->T_freeze;
    case SAMSIM_S_ALBEDO: return &c->albedo; case SAMSIM_S_FL_SW: return &c->fl_sw; case SAMSIM_S_FL_LW: return &c->fl_lw;
    case SAMSIM_S_MELT_THICK_SNOW: return &c->melt_thick_snow; case SAMSIM_S_FL_Q_SNOW: return &c->fl_Q_snow;
    case SAMSIM_S_ENERGY_STORED: return &c->energy_stored; case SAMSIM_S_FRESHWATER: return &c->freshwater;
    case SAMSIM_S_TOTAL_RESIST: return &c->total_resist; case SAMSIM_S_THICKNESS: return &c->thickness;
    case SAMSIM_S_BULK_SALIN: return &c->bulk_salin;
    case SAMSIM_S_FL_REST: return &c->fl_rest;
    case SAMSIM_S_S_BU_BOTTOM: return &c->S_bu_bottom;
    case SAMSIM_S_DT2M: return &c->dT2m; case SAMSIM_S_PRECIP_SCALE: return &c->precip_scale;
  }
  return NULL;
}

static double *lay_slot(column *c, int a) {
  switch (a) {
    case SAMSIM_A_H_ABS: return c->H_abs; case SAMSIM_A_S_ABS: return c->S_abs; case SAMSIM_A_M: return c->m;
    case SAMSIM_A_THICK: return c->thick; case SAMSIM_A_T: return c->T; case SAMSIM_A_PHI: return c->phi;
    case SAMSIM_A_PSI_S: return c->psi_s; case SAMSIM_A_PSI_L: return c->psi_l; case SAMSIM_A_PSI_G: return c->psi_g;
    case SAMSIM_A_S_BU: return c->S_bu; case SAMSIM_A_S_BR: return c->S_br; case SAMSIM_A_RAY: return c->ray;
    case SAMSIM_A_PERM: return c->perm; case SAMSIM_A_FLUSH_V: return c->flush_v; case SAMSIM_A_FLUSH_H: return c->flush_h;
  }
  return NULL;
}

int oracle_set_state(oracle_handle *h, const samsim_state_soa *s, int64_t col0) {
  if (!h || !s || !s->lay || !s->scal || !s->n_active) return SAMSIM_ERR_ARG;
  if (s->nlayer != h->cfg.nlayer || col0 < 0 || col0 + s->ncol > h->ncol) return SAMSIM_ERR_ARG;
  if (s->narr != SAMSIM_NPROG && s->narr != SAMSIM_NARR) return SAMSIM_ERR_ARG;
  int N = s->nlayer; int64_t nc = s->ncol;
  for (int64_t i = 0; i < nc; i++) {
    column *c = &h->cols[col0 + i];
    for (int a = 0; a < s->narr; a++) {
      double *dst = lay_slot(c, a);
      for (int k = 1; k <= N; k++) dst[k] = s->lay[((size_t)a * N + (k - 1)) * nc + i];
    }
    /* the perturbation slots (>= SAMSIM_S_DT2M) belong to the forcing and are not touched by set_state */
    for (int j = 0; j < SAMSIM_S_DT2M; j++) *scal_slot(c, j) = s->scal[(size_t)j * nc + i];
    if (h->cfg.tank_flag != 2) c->S_bu_bottom = c->ocean_sbu ? c->ocean_sbu_value : h->cfg.S_bu_bottom;
    c->N_active = s->n_active[i];
    if (c->N_active < 1 || c->N_active > N) return SAMSIM_ERR_ARG;
  }
  return SAMSIM_OK;
}

int oracle_get_state(oracle_handle *h, samsim_state_soa *s, int64_t col0) {
  if (!h || !s || !s->lay || !s->scal || !s->n_active) return SAMSIM_ERR_ARG;
  if (s->nlayer != h->cfg.nlayer || col0 < 0 || col0 + s->ncol > h->ncol) return SAMSIM_ERR_ARG;
  if (s->narr != SAMSIM_NPROG && s->narr != SAMSIM_NARR) return SAMSIM_ERR_ARG;
  int N = s->nlayer; int64_t nc = s->ncol;
  for (int64_t i = 0; i < nc; i++) {
    column *c = &h->cols[col0 + i];
    for (int a = 0; a < s->narr; a++) {
      const double *src = lay_slot(c, a);
      for (int k = 1; k <= N; k++) s->lay[((size_t)a * N + (k - 1)) * nc + i] = src[k];
    }
    for (int j = 0; j < SAMSIM_NSCAL; j++) s->scal[(size_t)j * nc + i] = *scal_slot(c, j);
    s->n_active[i] = c->N_active;
  }
  return SAMSIM_OK;
}

int oracle_set_clock(oracle_handle *h, const samsim_clock *k) {
  if (!h || !k) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    c->time = k->time; c->step = k->step; c->n_time_out = k->n_time_out; c->time_counter = k->time_counter; c->n_outputs = k->n_outputs;
  }
  return SAMSIM_OK;
}
int oracle_get_clock(oracle_handle *h, samsim_clock *k) {
  if (!h || !k) return SAMSIM_ERR_ARG;
  column *c = &h->cols[0];
  k->time = c->time; k->step = c->step; k->n_time_out = c->n_time_out; k->time_counter = c->time_counter; k->n_outputs = c->n_outputs;
  return SAMSIM_OK;
}

int oracle_step(oracle_handle *h, int64_t nsteps) {
  if (!h || nsteps < 0) return SAMSIM_ERR_ARG;
  if (h->cfg.boundflux_flag == 2 && h->cfg.atmoflux_flag == 2 && !h->f_sw) return SAMSIM_ERR_ARG;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(h->nthreads)
#endif
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    for (int64_t s = 0; s < nsteps; s++) column_step(c);
  }
  return SAMSIM_OK;
}

/* mirror of samsim_get_ensemble_stats (include/samsim.h): plain two-pass statistics over the columns without a STOP code */
int oracle_get_ensemble_stats(oracle_handle *h, int32_t nslots, const int32_t *slots, samsim_stat *out) {
  if (!h || nslots < 0 || (nslots > 0 && (!slots || !out))) return SAMSIM_ERR_ARG;
  for (int s = 0; s < nslots; s++) {
    int slot = slots[s];
    if (slot != SAMSIM_STAT_N_ACTIVE && (slot < 0 || slot >= SAMSIM_NSCAL)) return SAMSIM_ERR_ARG;
    samsim_stat st = {0, 0.0, 1.0e300, -1.0e300, 0.0};
    double sum = 0.0, ssq = 0.0;
    for (int64_t i = 0; i < h->ncol; i++) {
      column *c = &h->cols[i];
      if (c->status) continue;
      double v = (slot == SAMSIM_STAT_N_ACTIVE) ? (double)c->N_active : *scal_slot(c, slot);
      sum += v; st.count++;
      if (v < st.min) st.min = v;
      if (v > st.max) st.max = v;
    }
    if (st.count > 0) {
      st.mean = sum / (double)st.count;
      for (int64_t i = 0; i < h->ncol; i++) {
        column *c = &h->cols[i];
        if (c->status) continue;
        double v = (slot == SAMSIM_STAT_N_ACTIVE) ? (double)c->N_active : *scal_slot(c, slot);
        ssq += (v - st.mean) * (v - st.mean);
      }
      st.std = sqrt(ssq / (double)st.count);
    } else { st.min = st.max = 0.0; }
    out[s] = st;
  }
  return SAMSIM_OK;
}

int oracle_step_part_b(oracle_handle *h) {
  if (!h) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    if (c->status) continue;
    step_part_b(c);
    if (!c->status) c->step++;
  }
  return SAMSIM_OK;
}

int64_t oracle_steps_to_output(oracle_handle *h) {
  column *c = &h->cols[0];
  if (c->step == 0) return 1;
  return (int64_t)(h->cfg.i_time_out - c->n_time_out) + 1;
}

int oracle_set_output_window(oracle_handle *h, int64_t col0, int64_t ncols) {
  if (!h || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  int N = h->cfg.nlayer;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    int in = (i >= col0 && i < col0 + ncols);
    if (in && !c->snap_lay) c->snap_lay = (double *)calloc((size_t)SAMSIM_NARR * N, sizeof(double));
    if (in && h->n_bgc > 0 && !c->snap_bgc) c->snap_bgc = (double *)calloc((size_t)SAMSIM_MAX_NBGC * N, sizeof(double));
    if (!in && c->snap_lay) { free(c->snap_lay); c->snap_lay = NULL; c->snap_valid = 0; }
  }
  h->out_col0 = col0; h->out_ncols = ncols;
  return SAMSIM_OK;
}

int oracle_get_output(oracle_handle *h, samsim_output_soa *o) {
  if (!h || !o || !o->lay || !o->scal || !o->n_active) return SAMSIM_ERR_ARG;
  if (o->ncols != h->out_ncols || o->nlayer != h->cfg.nlayer) return SAMSIM_ERR_ARG;
  int N = o->nlayer; int64_t nc = o->ncols;
  if (nc == 0 || !h->cols[h->out_col0].snap_valid) return SAMSIM_ERR_NO_OUTPUT;
  for (int64_t i = 0; i < nc; i++) {
    column *c = &h->cols[h->out_col0 + i];
    for (int a = 0; a < SAMSIM_NARR; a++)
      for (int k = 0; k < N; k++) o->lay[((size_t)a * N + k) * nc + i] = c->snap_lay[(size_t)a * N + k];
    for (int j = 0; j < SAMSIM_NSCAL; j++) o->scal[(size_t)j * nc + i] = c->snap_scal[j];
    o->n_active[i] = c->snap_N_active;
  }
  o->time = h->cols[h->out_col0].snap_time; o->step = h->cols[h->out_col0].snap_step;
  return SAMSIM_OK;
}

int oracle_get_status(oracle_handle *h, int32_t *status, int64_t *step, int32_t *layer) {
  if (!h) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < h->ncol; i++) {
    if (status) status[i] = h->cols[i].status;
    if (step) step[i] = h->cols[i].err_step;
    if (layer) layer[i] = h->cols[i].err_layer;
  }
  return SAMSIM_OK;
}

/* the water below a grid of columns: the checker mirrors samsim_set_ocean */
int oracle_set_ocean(oracle_handle *h, const double *dfl_q_bottom_col, const double *S_bu_bottom_col) {
  if (!h) return SAMSIM_ERR_ARG;
  if (dfl_q_bottom_col && h->cfg.testcase != 4 && h->cfg.testcase != 7) return SAMSIM_ERR_UNSUPPORTED;
  if (S_bu_bottom_col && h->cfg.tank_flag == 2) return SAMSIM_ERR_UNSUPPORTED;
  for (int64_t i = 0; S_bu_bottom_col && i < h->ncol; i++)   /* as samsim_set_ocean: nothing is changed by a rejected call */
    if (!(S_bu_bottom_col[i] >= 0.0)) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < h->ncol; i++) {
    column *c = &h->cols[i];
    c->dflq = dfl_q_bottom_col ? dfl_q_bottom_col[i] : 0.0;
    c->ocean_sbu = S_bu_bottom_col != NULL;
    c->ocean_sbu_value = S_bu_bottom_col ? S_bu_bottom_col[i] : 0.0;
    if (h->cfg.tank_flag != 2) c->S_bu_bottom = c->ocean_sbu ? c->ocean_sbu_value : h->cfg.S_bu_bottom;
  }
  return SAMSIM_OK;
}

/* restart: the checker mirrors samsim_set_status */
int oracle_set_status(oracle_handle *h, const int32_t *status, const int64_t *step, const int32_t *layer, int64_t col0, int64_t ncols) {
  if (!h || !status || col0 < 0 || ncols < 0 || col0 + ncols > h->ncol) return SAMSIM_ERR_ARG;
  for (int64_t i = 0; i < ncols; i++) {
    h->cols[col0 + i].status = status[i];
    if (step) h->cols[col0 + i].err_step = step[i];
    if (layer) h->cols[col0 + i].err_layer = layer[i];
  }
  return SAMSIM_OK;
}

int oracle_get_work(oracle_handle *h, int64_t *lcu, int64_t *cs) {
  if (!h) return SAMSIM_ERR_ARG;
  int64_t w = 0, s = 0;
  for (int64_t i = 0; i < h->ncol; i++) { w += h->cols[i].work; s += h->cols[i].step; }
  if (lcu) *lcu = w; if (cs) *cs = s;
  return SAMSIM_OK;
}
