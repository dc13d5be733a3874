!> iso_c_binding view of the C-ABI in include/samsim.h (libsamsim_hip.so).
!! This is the stub a maintainer of the reference adds on the Fortran side (see INTEGRATION.md): the derived types
!! mirror `samsim_config`, `samsim_state_soa`, `samsim_clock`, `samsim_output_soa` field by field.
MODULE mo_samsim_capi
  USE, INTRINSIC :: iso_c_binding
  IMPLICIT NONE
  PUBLIC

  ! enum samsim_layer_array (0-based in C, used 1-based here: index = value + 1)
  INTEGER, PARAMETER :: A_H_ABS = 1, A_S_ABS = 2, A_M = 3, A_THICK = 4, A_T = 5, A_PHI = 6, A_PSI_S = 7, A_PSI_L = 8, &
                        A_PSI_G = 9, A_S_BU = 10, A_S_BR = 11, A_RAY = 12, A_PERM = 13, A_FLUSH_V = 14, A_FLUSH_H = 15, &
                        SAMSIM_NARR = 15, SAMSIM_NPROG = 4
  ! enum samsim_scalar (index = value + 1)
  INTEGER, PARAMETER :: S_M_SNOW = 1, S_H_ABS_SNOW = 2, S_S_ABS_SNOW = 3, S_THICK_SNOW = 4, S_PSI_S_SNOW = 5, &
                        S_PSI_L_SNOW = 6, S_PSI_G_SNOW = 7, S_T_SNOW = 8, S_PHI_S = 9, S_T_TOP = 10, S_MELT_THICK = 11, &
                        S_T2M = 12, S_LIQUID_PRECIP = 13, S_SOLID_PRECIP = 14, S_FL_Q_BOTTOM = 15, S_GRAV_DRAIN = 16, &
                        S_GRAV_SALT = 17, S_GRAV_TEMP = 18, S_MELT_OUT1 = 19, S_MELT_OUT2 = 20, S_MELT_OUT3 = 21, &
                        S_MELT_ERR = 22, S_FREEBOARD = 23, S_T_FREEZE = 24, S_ALBEDO = 25, S_FL_SW = 26, S_FL_LW = 27, &
                        S_MELT_THICK_SNOW = 28, S_FL_Q_SNOW = 29, S_ENERGY_STORED = 30, S_FRESHWATER = 31, &
                        S_TOTAL_RESIST = 32, S_THICKNESS = 33, S_BULK_SALIN = 34, S_FL_REST = 35, S_S_BU_BOTTOM = 36, S_DT2M = 37, &
                        S_PRECIP_SCALE = 38, &
                        SAMSIM_NSCAL = 38

  TYPE, BIND(C) :: samsim_config
     INTEGER(c_int32_t) :: struct_size, testcase, nlayer, n_top, n_middle, n_bottom
     INTEGER(c_int32_t) :: atmoflux_flag, grav_flag, prescribe_flag, grav_heat_flag, flush_heat_flag, turb_flag, salt_flag, &
                           boundflux_flag, flush_flag, flood_flag, bottom_flag, debug_flag, precip_flag, harmonic_flag, &
                           tank_flag, albedo_flag, lab_snow_flag, freeboard_snow_flag, snow_flush_flag, snow_precip_flag, &
                           bgc_flag, i_time_out
     REAL(c_double)     :: dt, thick_0, thick_min, T_bottom, S_bu_bottom, k_snow_flush, max_flux_plate, time_out, time_total
     REAL(c_double)     :: alpha_flux_instable, alpha_flux_stable, m_total, S_total
  END TYPE samsim_config

  TYPE, BIND(C) :: samsim_state_soa
     INTEGER(c_int64_t) :: ncol
     INTEGER(c_int32_t) :: nlayer, narr
     TYPE(c_ptr)        :: lay, scal, n_active
  END TYPE samsim_state_soa

  TYPE, BIND(C) :: samsim_clock
     REAL(c_double)     :: time
     INTEGER(c_int64_t) :: step
     INTEGER(c_int32_t) :: n_time_out, time_counter
     INTEGER(c_int64_t) :: n_outputs
  END TYPE samsim_clock

  TYPE, BIND(C) :: samsim_output_soa
     INTEGER(c_int64_t) :: ncols
     INTEGER(c_int32_t) :: nlayer, reserved
     TYPE(c_ptr)        :: lay, scal, n_active
     REAL(c_double)     :: time
     INTEGER(c_int64_t) :: step
  END TYPE samsim_output_soa

  TYPE, BIND(C) :: samsim_stat               ! samsim_stat: ensemble statistics of one per-column scalar
     INTEGER(c_int64_t) :: count
     REAL(c_double)     :: mean, min, max, std
  END TYPE

  INTERFACE
     INTEGER(c_int) FUNCTION samsim_create(cfg, ncol, device, h) BIND(C, name='samsim_create')
       IMPORT
       TYPE(samsim_config), INTENT(in) :: cfg
       INTEGER(c_int64_t), VALUE :: ncol
       INTEGER(c_int32_t), VALUE :: device
       TYPE(c_ptr), INTENT(out) :: h
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_forcing(h, len, fl_sw, fl_lw, T2m, precip, dT2m_col, precip_scale_col) &
          BIND(C, name='samsim_set_forcing')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), VALUE :: len
       REAL(c_double), INTENT(in) :: fl_sw(*), fl_lw(*), T2m(*), precip(*)
       TYPE(c_ptr), VALUE :: dT2m_col, precip_scale_col
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_forcing_sites(h, nsites, len, fl_sw, fl_lw, T2m, precip, site_of_column, dT2m_col, &
          precip_scale_col) BIND(C, name='samsim_set_forcing_sites')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), VALUE :: nsites, len
       REAL(c_double), INTENT(in) :: fl_sw(*), fl_lw(*), T2m(*), precip(*)      ! (len, nsites)
       INTEGER(c_int32_t), INTENT(in) :: site_of_column(*)                      ! 0-based
       TYPE(c_ptr), VALUE :: dT2m_col, precip_scale_col
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_state(h, s, col0) BIND(C, name='samsim_set_state')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(samsim_state_soa), INTENT(in) :: s
       INTEGER(c_int64_t), VALUE :: col0
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_state(h, s, col0) BIND(C, name='samsim_get_state')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(samsim_state_soa), INTENT(inout) :: s
       INTEGER(c_int64_t), VALUE :: col0
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_clock(h, c) BIND(C, name='samsim_set_clock')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(samsim_clock), INTENT(in) :: c
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_clock(h, c) BIND(C, name='samsim_get_clock')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(samsim_clock), INTENT(out) :: c
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_step(h, nsteps) BIND(C, name='samsim_step')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int64_t), VALUE :: nsteps
     END FUNCTION
     INTEGER(c_int64_t) FUNCTION samsim_steps_to_output(h) BIND(C, name='samsim_steps_to_output')
       IMPORT
       TYPE(c_ptr), VALUE :: h
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_output_window(h, col0, ncols) BIND(C, name='samsim_set_output_window')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int64_t), VALUE :: col0, ncols
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_output(h, o) BIND(C, name='samsim_get_output')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(samsim_output_soa), INTENT(inout) :: o
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_status(h, status, step, layer) BIND(C, name='samsim_get_status')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), INTENT(out) :: status(*)
       INTEGER(c_int64_t), INTENT(out) :: step(*)
       INTEGER(c_int32_t), INTENT(out) :: layer(*)
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_ocean(h, dfl_q_bottom_col, S_bu_bottom_col) BIND(C, name='samsim_set_ocean')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       TYPE(c_ptr), VALUE :: dfl_q_bottom_col, S_bu_bottom_col   ! [ncol] each, or c_null_ptr
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_status(h, status, step, layer, col0, ncols) BIND(C, name='samsim_set_status')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), INTENT(in) :: status(*)
       INTEGER(c_int64_t), INTENT(in) :: step(*)
       INTEGER(c_int32_t), INTENT(in) :: layer(*)
       INTEGER(c_int64_t), VALUE :: col0, ncols
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_tracers(h, n_bgc, bgc_bottom, bgc_total) BIND(C, name='samsim_set_tracers')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), VALUE :: n_bgc
       REAL(c_double), INTENT(in) :: bgc_bottom(*)
       TYPE(c_ptr), VALUE :: bgc_total                    ! [n_bgc] with tank_flag 2, else c_null_ptr
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_tracer_state(h, bgc_abs, col0, ncols) BIND(C, name='samsim_set_tracer_state')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       REAL(c_double), INTENT(in) :: bgc_abs(*)           ! (ncols, nlayer, n_bgc), column fastest
       INTEGER(c_int64_t), VALUE :: col0, ncols
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_set_tracer_bottom(h, bgc_bottom, col0, ncols) BIND(C, name='samsim_set_tracer_bottom')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       REAL(c_double), INTENT(in) :: bgc_bottom(*)        ! (ncols, n_bgc), column fastest
       INTEGER(c_int64_t), VALUE :: col0, ncols
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_tracer_state(h, bgc_abs, bgc_bottom, col0, ncols) BIND(C, name='samsim_get_tracer_state')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       REAL(c_double), INTENT(out) :: bgc_abs(*), bgc_bottom(*)
       INTEGER(c_int64_t), VALUE :: col0, ncols
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_tracer_output(h, bgc_abs, bgc_bottom) BIND(C, name='samsim_get_tracer_output')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       REAL(c_double), INTENT(out) :: bgc_abs(*), bgc_bottom(*)
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_ensemble_stats(h, nslots, slots, out) BIND(C, name='samsim_get_ensemble_stats')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int32_t), VALUE :: nslots
       INTEGER(c_int32_t), INTENT(in) :: slots(*)        ! enum samsim_scalar values (0-based), -1 = N_active
       TYPE(samsim_stat), INTENT(out) :: out(*)
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_get_work(h, cells, colsteps) BIND(C, name='samsim_get_work')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int64_t), INTENT(out) :: cells, colsteps
     END FUNCTION
     INTEGER(c_int) FUNCTION samsim_synchronize(h) BIND(C, name='samsim_synchronize')
       IMPORT
       TYPE(c_ptr), VALUE :: h
     END FUNCTION
     !> nlaunches launches of nsteps steps, enqueued back to back, and the device time of the sequence [ms] (ABI 4)
     INTEGER(c_int) FUNCTION samsim_steps_timed(h, nsteps, nlaunches, device_ms) BIND(C, name='samsim_steps_timed')
       IMPORT
       TYPE(c_ptr), VALUE :: h
       INTEGER(c_int64_t), VALUE :: nsteps
       INTEGER(c_int32_t), VALUE :: nlaunches
       REAL(c_double), INTENT(out) :: device_ms
     END FUNCTION
     SUBROUTINE samsim_destroy(h) BIND(C, name='samsim_destroy')
       IMPORT
       TYPE(c_ptr), VALUE :: h
     END SUBROUTINE
     TYPE(c_ptr) FUNCTION samsim_strerror(code) BIND(C, name='samsim_strerror')
       IMPORT
       INTEGER(c_int), VALUE :: code
     END FUNCTION
  END INTERFACE

CONTAINS

  !> aborts with the library's message when a C-ABI call fails (the reference aborts with STOP as well)
  SUBROUTINE samsim_check(rc, what)
    INTEGER(c_int), INTENT(in)   :: rc
    CHARACTER(len=*), INTENT(in) :: what
    CHARACTER(kind=c_char), POINTER :: msg(:)
    INTEGER :: n
    IF (rc == 0) RETURN
    CALL c_f_pointer(samsim_strerror(rc), msg, (/ 200 /))
    n = 1
    DO WHILE (n < 200 .AND. msg(n) /= c_null_char)
       n = n + 1
    END DO
    PRINT *, 'samsim C-ABI error in ', what, ': code', rc, ' ', msg(1:n-1)
    STOP 3
  END SUBROUTINE samsim_check

END MODULE mo_samsim_capi
