!> Fortran host of the MI355X-native solver: keeps the reference's driver surface -- `grotz(testcase, description)`
!! calling `init`, `output_begin`, `output_settings`, `sub_input`, the time loop and `output` (mo_grotz.f90:83-877) -- and
!! the reference's `.dat` formats (mo_output.f90:116-146, 276-339), while the time-loop BODY (mo_grotz.f90:182-835)
!! runs on the GPU behind the C-ABI of include/samsim.h for `ncol` columns at once.
!!
!! New surface the reference does not have (SURVEY.md, introduction): a namelist file `samsim.nml`
!!   &samsim_run   testcase, ncol, col0, ncol_total, device, out_col, perturb, description, max_steps, restart_in, restart_out, sites /
!! (col0 / ncol_total: this process owns the global columns col0 .. col0+ncol-1 of an ensemble of ncol_total -- one host
!! process per GPU, contiguous column ranges, no exchange between them, SURVEY.md section 8e)
!!   &samsim_flags <any flag of mo_data.f90:136-155 or scalar set by mo_init> /       (overrides init(testcase))
!! `dat_settings.dat` stays the echo of what was actually used.
!!
!! Builder-written; no reference source is reused.  Module and routine names follow the reference so that the call
!! structure reads the same.
MODULE mo_data
  USE, INTRINSIC :: iso_c_binding
  USE mo_samsim_capi
  IMPLICIT NONE
  INTEGER, PARAMETER :: wp = SELECTED_REAL_KIND(12, 307)   ! mo_parameters.f90:33
  REAL(wp), PARAMETER :: rho_l = 1028.0_wp, c_l = 3400._wp, k_s = 2.2_wp, rho_s = 920._wp, c_s = 2020.0_wp
  INTEGER(c_int64_t), PARAMETER :: restart_magic = INT(z'32304B48434D4153', c_int64_t)   ! the bytes "SAMCHK02" (chunks carry the status block)
  REAL(wp), PARAMETER :: sigma = 5.6704_wp*1e-8   !< Stefan Boltzmann constant as written in mo_parameters.f90:59

  TYPE(samsim_config) :: cfg                 !< every flag / scalar that crosses the C-ABI
  INTEGER             :: testcase_id = 1
  INTEGER(c_int64_t)  :: ncol = 1            !< number of columns this process holds
  INTEGER(c_int64_t)  :: col0 = 0            !< global index (0-based) of its first column: perturbation and forcing site follow the global index
  INTEGER(c_int64_t)  :: ncol_total = -1     !< size of the whole ensemble (echo; default ncol)
  INTEGER             :: device = 0, out_col = 1
  LOGICAL             :: perturb = .FALSE.   !< per-column T2m / precipitation perturbation (SURVEY.md 8d cfg3)
  INTEGER(c_int64_t)  :: max_steps = -1
  CHARACTER(len=256)  :: sites(16) = ' '      !< directories with flux_sw/flux_lw/T2m/precip.txt.input; column c reads sites(MOD(c-1,n)+1)
  CHARACTER(len=1024) :: restart_in = ' ', restart_out = ' '   !< binary checkpoint files (samsim_amd/checkpoint.py format)
  INTEGER             :: i_time, i_time_out
  REAL(wp)            :: fl_q_bottom = 0._wp, T_top = 0._wp, fl_sw = 0._wp, fl_rest = 0._wp, T2m = 0._wp, tank_depth = 0._wp
  INTEGER             :: N_bgc = 1
  REAL(c_double), TARGET :: bgc_bottom(8) = 0._wp, bgc_total(8) = 0._wp   !< tracers, mo_data.f90:187,193
  REAL(c_double), ALLOCATABLE, TARGET :: bgc_abs(:, :, :)                  !< (ncol, Nlayer, N_bgc)
  CHARACTER*12000     :: format_bgc
  ! host copies: SoA blocks, column fastest (= C layout [array][layer][column])
  REAL(c_double), ALLOCATABLE, TARGET    :: lay(:, :, :), scal(:, :)
  INTEGER(c_int32_t), ALLOCATABLE, TARGET :: n_active(:)
  ! forcing tables (atmoflux_flag 2), mo_data.f90:166-171
  INTEGER :: Length_Input = 13148, nsites = 1
  REAL(c_double), ALLOCATABLE, TARGET :: fl_sw_input(:), fl_lw_input(:), T2m_input(:), precip_input(:)
  REAL(c_double), ALLOCATABLE, TARGET :: dT2m_col(:), precip_scale_col(:)
  CHARACTER*12000 :: format_T, format_psi, format_thick, format_snow, format_perm, format_melt
END MODULE mo_data

MODULE mo_init
  USE mo_data
  IMPLICIT NONE
CONTAINS
  !> flag defaults of the reference, mo_init.f90:83-109
  SUBROUTINE default_flags()
    cfg%struct_size = INT(c_sizeof(cfg), c_int32_t)
    cfg%boundflux_flag = 1; cfg%atmoflux_flag = 1; cfg%albedo_flag = 2
    cfg%grav_heat_flag = 1; cfg%flush_heat_flag = 1; cfg%flood_flag = 2; cfg%flush_flag = 5; cfg%grav_flag = 2
    cfg%harmonic_flag = 2; cfg%prescribe_flag = 1; cfg%salt_flag = 1
    cfg%turb_flag = 2; cfg%bottom_flag = 1; cfg%tank_flag = 1
    cfg%precip_flag = 0; cfg%freeboard_snow_flag = 0; cfg%snow_flush_flag = 1; cfg%snow_precip_flag = 1
    cfg%debug_flag = 1; cfg%bgc_flag = 1; cfg%lab_snow_flag = 0
    cfg%k_snow_flush = 0.75_wp; cfg%max_flux_plate = 10000.0_wp
  END SUBROUTINE default_flags

  !> per-testcase settings and initial state: testcases 1 (mo_init.f90:865-945), 2/6/9/33/34 (tanks), 3 (:1045-1080), 4 (:1127-1207),
  !! 5 (:1210-1273) and 7 (:1360-1395); common tail mo_init.f90:1981-2031.  Namelist group &samsim_flags (unit nml_unit, if > 0) overrides the settings.
  SUBROUTINE init(testcase, nml_unit)
    INTEGER, INTENT(in) :: testcase, nml_unit
    INTEGER :: Nlayer, N_top, N_bottom, ios, k_bgc
    INTEGER :: boundflux_flag, atmoflux_flag, albedo_flag, grav_flag, flush_flag, flood_flag, grav_heat_flag, &
               flush_heat_flag, harmonic_flag, salt_flag, turb_flag, bottom_flag, precip_flag, freeboard_snow_flag, &
               snow_flush_flag, bgc_flag, prescribe_flag
    REAL(wp) :: dt, thick_0, time_out, time_total, T_bottom, S_bu_bottom, k_snow_flush
    NAMELIST /samsim_flags/ Nlayer, N_top, N_bottom, boundflux_flag, atmoflux_flag, albedo_flag, grav_flag, flush_flag, &
         flood_flag, grav_heat_flag, flush_heat_flag, harmonic_flag, salt_flag, turb_flag, bottom_flag, precip_flag, &
         freeboard_snow_flag, snow_flush_flag, bgc_flag, prescribe_flag, dt, thick_0, time_out, time_total, T_bottom, S_bu_bottom, k_snow_flush, &
         fl_q_bottom, T_top, T2m, tank_depth

    CALL default_flags()
    cfg%testcase = testcase
    IF (testcase == 1) THEN
       cfg%nlayer = 90; cfg%n_top = 5; cfg%n_bottom = 5
       cfg%turb_flag = 1; cfg%boundflux_flag = 1; cfg%grav_heat_flag = 1; cfg%flush_flag = 1; cfg%salt_flag = 2
       T_top = -5.0_wp
       cfg%T_bottom = -1._wp; cfg%S_bu_bottom = 34._wp; fl_q_bottom = 0._wp
       cfg%thick_0 = 0.002_wp; cfg%dt = 1.0_wp; cfg%time_out = 3600._wp; cfg%time_total = cfg%time_out*72._wp
    ELSE IF (testcase == 4) THEN
       cfg%nlayer = 100; cfg%n_top = 20; cfg%n_bottom = 20
       cfg%atmoflux_flag = 2; cfg%precip_flag = 1; cfg%boundflux_flag = 2
       cfg%snow_flush_flag = 1; cfg%flush_heat_flag = 2; cfg%snow_precip_flag = 1
       cfg%T_bottom = -1.0_wp; cfg%S_bu_bottom = 34._wp
       cfg%thick_0 = 0.01_wp; cfg%time_out = 86400._wp; cfg%time_total = cfg%time_out*365._wp*4.5_wp; cfg%dt = 10._wp
    ELSE IF (testcase == 2 .OR. testcase == 6 .OR. testcase == 9 .OR. testcase == 33 .OR. testcase == 34) THEN
       ! tank experiments, mo_init.f90:948-1003, 1278-1330, 1684-1740, 1779-1873, 1876-1970 (bgc off)
       cfg%tank_flag = 2; cfg%boundflux_flag = 3; cfg%grav_heat_flag = 1
       cfg%alpha_flux_instable = 22.0_wp
       IF (testcase == 2) THEN
          fl_q_bottom = 10._wp; cfg%alpha_flux_stable = 15._wp; tank_depth = 1._wp
          cfg%nlayer = 100; cfg%n_bottom = 10; cfg%n_top = 3
          T2m = -20._wp; T_top = -18._wp; cfg%T_bottom = 0.0_wp; cfg%S_bu_bottom = 31.2_wp
          cfg%thick_0 = 0.01_wp; cfg%time_out = 3600._wp*6._wp; cfg%time_total = cfg%time_out*4._wp*30._wp; cfg%dt = 30._wp
       ELSE IF (testcase == 6) THEN
          fl_q_bottom = 35._wp; cfg%alpha_flux_stable = 11._wp; tank_depth = 0.159_wp
          cfg%nlayer = 40; cfg%n_bottom = 3; cfg%n_top = 3
          T2m = -18._wp; T_top = -18._wp; cfg%T_bottom = 0.0_wp; cfg%S_bu_bottom = 31.2_wp
          cfg%thick_0 = 0.0025_wp; cfg%time_out = 1800._wp/2._wp; cfg%time_total = cfg%time_out*39._wp*2._wp*2._wp; cfg%dt = 0.5
       ELSE IF (testcase == 33 .OR. testcase == 34) THEN
          fl_q_bottom = 10._wp; cfg%alpha_flux_stable = 15._wp; tank_depth = 0.94_wp
          cfg%nlayer = 100; cfg%n_bottom = 10; cfg%n_top = 3
          T2m = -15._wp; T_top = -10._wp; cfg%T_bottom = 0.5_wp; cfg%thick_0 = 0.005_wp; cfg%dt = 10._wp
          IF (testcase == 33) THEN
             cfg%S_bu_bottom = 0.13_wp; cfg%time_out = 60._wp*5._wp; cfg%time_total = cfg%time_out*12._wp*6._wp
          ELSE
             cfg%S_bu_bottom = 34.9_wp; cfg%time_out = 60._wp*10._wp; cfg%time_total = 86400._wp*10._wp
          END IF
       ELSE
          fl_q_bottom = 10._wp; cfg%alpha_flux_stable = 15._wp; tank_depth = 0.8_wp
          cfg%nlayer = 100; cfg%n_bottom = 10; cfg%n_top = 3
          T2m = -15._wp; T_top = -10._wp; cfg%T_bottom = -0.07_wp; cfg%S_bu_bottom = 34.6_wp
          cfg%thick_0 = 0.005_wp; cfg%time_out = 3600._wp*2._wp; cfg%time_total = cfg%time_out*12._wp*6._wp; cfg%dt = 10._wp
       END IF
    ELSE IF (testcase == 3) THEN
       cfg%nlayer = 20; cfg%n_top = 5; cfg%n_bottom = 5
       cfg%atmoflux_flag = 1; cfg%precip_flag = 0; cfg%boundflux_flag = 2
       fl_q_bottom = 8._wp
       cfg%T_bottom = -1.0_wp; cfg%S_bu_bottom = 34._wp
       cfg%thick_0 = 0.03_wp; cfg%time_out = 86400._wp*3.5_wp; cfg%time_total = cfg%time_out*54._wp*2._wp*2._wp
       cfg%dt = 60._wp
    ELSE IF (testcase == 50) THEN
       ! mo_init.f90:1497-1531: the reference's default flags with boundflux_flag 2, 70 layers, three years of growth
       cfg%nlayer = 70; cfg%n_top = 5; cfg%n_bottom = 5
       cfg%boundflux_flag = 2
       fl_q_bottom = 20._wp; T_top = -20._wp
       cfg%T_bottom = -1.72_wp; cfg%S_bu_bottom = 34._wp
       cfg%thick_0 = 0.005_wp; cfg%time_out = 3600._wp*24._wp*30._wp; cfg%time_total = cfg%time_out*12._wp*3._wp; cfg%dt = 10._wp
    ELSE IF (testcase == 5) THEN
       cfg%nlayer = 100; cfg%n_top = 20; cfg%n_bottom = 10
       cfg%boundflux_flag = 2; cfg%atmoflux_flag = 3; cfg%flush_heat_flag = 2; cfg%flush_flag = 5
       cfg%grav_flag = 1; cfg%flood_flag = 1
       fl_sw = 0._wp; fl_rest = 290._wp**4*sigma; fl_q_bottom = 15._wp
       cfg%S_bu_bottom = 5._wp; cfg%T_bottom = 0._wp
       cfg%thick_0 = 0.01_wp; cfg%time_out = 3600._wp*3._wp; cfg%time_total = cfg%time_out*24._wp*10._wp; cfg%dt = 10._wp
    ELSE IF (testcase == 7) THEN
       cfg%nlayer = 100; cfg%n_top = 20; cfg%n_bottom = 20
       cfg%atmoflux_flag = 2; cfg%precip_flag = 1; cfg%boundflux_flag = 2
       cfg%albedo_flag = 1; cfg%grav_heat_flag = 2; cfg%flush_heat_flag = 2
       cfg%flush_flag = 4; cfg%grav_flag = 3; cfg%flood_flag = 3
       cfg%T_bottom = -1.0_wp; cfg%S_bu_bottom = 34._wp
       cfg%thick_0 = 0.01_wp; cfg%time_out = 86400._wp/2._wp; cfg%time_total = cfg%time_out*365._wp*9._wp; cfg%dt = 10._wp
    ELSE
       PRINT *, 'selected testcase does not exist'   ! mo_init.f90:1973-1976
       STOP 4321
    END IF

    ! tracers as init(testcase) switches them on (mo_init.f90:921-942, 1006-1020, 1333-1345); &samsim_flags bgc_flag = 1 turns them off
    IF (testcase == 1) THEN
       cfg%bgc_flag = 2; N_bgc = 2; bgc_bottom(1) = 400._wp; bgc_bottom(2) = 500._wp
    ELSE IF (testcase == 2) THEN
       cfg%bgc_flag = 2; N_bgc = 2; bgc_bottom(1:2) = 385._wp
    ELSE IF (testcase == 6) THEN
       cfg%bgc_flag = 2; N_bgc = 1; bgc_bottom(1) = 385._wp
    END IF

    IF (nml_unit > 0) THEN
       Nlayer = cfg%nlayer; N_top = cfg%n_top; N_bottom = cfg%n_bottom
       boundflux_flag = cfg%boundflux_flag; atmoflux_flag = cfg%atmoflux_flag; albedo_flag = cfg%albedo_flag
       grav_flag = cfg%grav_flag; flush_flag = cfg%flush_flag; flood_flag = cfg%flood_flag
       grav_heat_flag = cfg%grav_heat_flag; flush_heat_flag = cfg%flush_heat_flag; harmonic_flag = cfg%harmonic_flag
       prescribe_flag = cfg%prescribe_flag
       salt_flag = cfg%salt_flag; turb_flag = cfg%turb_flag; bottom_flag = cfg%bottom_flag; precip_flag = cfg%precip_flag
       freeboard_snow_flag = cfg%freeboard_snow_flag; snow_flush_flag = cfg%snow_flush_flag; bgc_flag = cfg%bgc_flag
       dt = cfg%dt; thick_0 = cfg%thick_0; time_out = cfg%time_out; time_total = cfg%time_total
       T_bottom = cfg%T_bottom; S_bu_bottom = cfg%S_bu_bottom; k_snow_flush = cfg%k_snow_flush
       REWIND(nml_unit)
       READ(nml_unit, NML=samsim_flags, IOSTAT=ios)
       IF (ios > 0) THEN
          PRINT *, 'error in namelist group samsim_flags'
          STOP 4
       END IF
       cfg%nlayer = Nlayer; cfg%n_top = N_top; cfg%n_bottom = N_bottom
       cfg%boundflux_flag = boundflux_flag; cfg%atmoflux_flag = atmoflux_flag; cfg%albedo_flag = albedo_flag
       cfg%grav_flag = grav_flag; cfg%flush_flag = flush_flag; cfg%flood_flag = flood_flag
       cfg%grav_heat_flag = grav_heat_flag; cfg%flush_heat_flag = flush_heat_flag; cfg%harmonic_flag = harmonic_flag
       cfg%prescribe_flag = prescribe_flag
       cfg%salt_flag = salt_flag; cfg%turb_flag = turb_flag; cfg%bottom_flag = bottom_flag; cfg%precip_flag = precip_flag
       cfg%freeboard_snow_flag = freeboard_snow_flag; cfg%snow_flush_flag = snow_flush_flag; cfg%bgc_flag = bgc_flag
       cfg%dt = dt; cfg%thick_0 = thick_0; cfg%time_out = time_out; cfg%time_total = time_total
       cfg%T_bottom = T_bottom; cfg%S_bu_bottom = S_bu_bottom; cfg%k_snow_flush = k_snow_flush
    END IF

    cfg%n_middle = cfg%nlayer - cfg%n_top - cfg%n_bottom
    IF (cfg%n_top < 3) THEN
       PRINT *, 'Problem occurs when N_top smaller then 3, so just change it to 3 or more'   ! mo_init.f90:2016-2019
       STOP 666
    END IF
    cfg%thick_min = cfg%thick_0/2._wp
    IF (cfg%tank_flag == 2) THEN            ! water and salt in the tank, mo_init.f90:996-997
       cfg%m_total = rho_l*tank_depth
       cfg%S_total = rho_l*cfg%S_bu_bottom*tank_depth
    END IF
    i_time = INT(cfg%time_total/cfg%dt)
    i_time_out = INT(cfg%time_out/cfg%dt)
    cfg%i_time_out = i_time_out

    CALL sub_allocate(cfg%nlayer)
    ! defaults of the carried diagnostics, mo_init.f90:1982-1990
    lay(:, :, A_T) = cfg%T_bottom
    lay(:, :, A_S_BU) = cfg%S_bu_bottom
    lay(:, :, A_PSI_L) = 1.0_wp
    scal(:, S_PRECIP_SCALE) = 1.0_wp
    scal(:, S_T_TOP) = T_top
    scal(:, S_FL_Q_BOTTOM) = fl_q_bottom
    scal(:, S_FL_SW) = fl_sw
    scal(:, S_FL_REST) = fl_rest
    scal(:, S_T2M) = T2m
    scal(:, S_S_BU_BOTTOM) = cfg%S_bu_bottom
    IF (testcase == 5) THEN
       ! a slab: every layer active, thick = thick_0, 5 g/kg, -90 c_l J/kg (mo_init.f90:1217,1270-1273)
       n_active = cfg%nlayer
       lay(:, :, A_THICK) = cfg%thick_0
       lay(:, :, A_M) = lay(:, :, A_THICK)*rho_l
       lay(:, :, A_S_ABS) = lay(:, :, A_M)*cfg%S_bu_bottom
       lay(:, :, A_H_ABS) = lay(:, :, A_M)*(-90.0)*c_l
    ELSE
       n_active = 1
       ! the single initial water layer
       lay(:, 1, A_THICK) = cfg%thick_0
       lay(:, 1, A_M) = lay(:, 1, A_THICK)*rho_l
       lay(:, 1, A_S_ABS) = cfg%S_bu_bottom*lay(:, 1, A_M)
       IF (testcase == 1 .OR. testcase == 50) THEN
          lay(:, 1, A_H_ABS) = lay(:, 1, A_M)*cfg%T_bottom*c_l
       ELSE IF (cfg%tank_flag == 2) THEN
          lay(:, 1, A_H_ABS) = lay(:, 1, A_M)*cfg%T_bottom          ! as written in mo_init.f90:1002
       ELSE
          lay(:, 1, A_H_ABS) = 0._wp
       END IF
    END IF
    IF (cfg%bgc_flag == 2) THEN
       IF (cfg%tank_flag == 2) bgc_total(1:N_bgc) = bgc_bottom(1:N_bgc)*rho_l*tank_depth
       ALLOCATE(bgc_abs(ncol, cfg%nlayer, N_bgc))
       bgc_abs = 0._wp
       DO k_bgc = 1, N_bgc
          bgc_abs(:, 1, k_bgc) = bgc_bottom(k_bgc)*lay(:, 1, A_M)        ! bgc_abs(1,:) = bgc_bottom(:)*m(1)
       END DO
    ELSE
       N_bgc = 1
    END IF
    PRINT *, 'Initialization of testcase complete, testcase:', testcase
  END SUBROUTINE init

  !> sub_allocate, mo_init.f90:2040-2090 (host buffers; the device buffers are owned by the library)
  SUBROUTINE sub_allocate(Nlayer)
    INTEGER(c_int32_t), INTENT(in) :: Nlayer
    ALLOCATE(lay(ncol, Nlayer, SAMSIM_NARR), scal(ncol, SAMSIM_NSCAL), n_active(ncol))
    lay = 0._wp
    scal = 0._wp
  END SUBROUTINE sub_allocate

  SUBROUTINE sub_deallocate()
    DEALLOCATE(lay, scal, n_active)
  END SUBROUTINE sub_deallocate

  !> sub_input, mo_functions.f90:304-327: list-directed read of the four ERA-interim tables in the working directory
  SUBROUTINE sub_input()
    INTEGER :: s, n0
    CHARACTER(len=300) :: dir
    nsites = COUNT(LEN_TRIM(sites) > 0)
    IF (nsites == 0) THEN            ! as the reference: the four files of the working directory
       nsites = 1; sites(1) = '.'
    END IF
    ALLOCATE(fl_sw_input(Length_Input*nsites), fl_lw_input(Length_Input*nsites), T2m_input(Length_Input*nsites), &
         precip_input(Length_Input*nsites))
    DO s = 1, nsites
       dir = TRIM(sites(s))//'/'
       n0 = (s - 1)*Length_Input
       OPEN(1234, file=TRIM(dir)//'flux_lw.txt.input', status='old'); READ(1234, *) fl_lw_input(n0+1:n0+Length_Input); CLOSE(1234)
       OPEN(1234, file=TRIM(dir)//'flux_sw.txt.input', status='old'); READ(1234, *) fl_sw_input(n0+1:n0+Length_Input); CLOSE(1234)
       OPEN(1234, file=TRIM(dir)//'T2m.txt.input', status='old');     READ(1234, *) T2m_input(n0+1:n0+Length_Input);   CLOSE(1234)
       OPEN(1234, file=TRIM(dir)//'precip.txt.input', status='old');  READ(1234, *) precip_input(n0+1:n0+Length_Input); CLOSE(1234)
    END DO
  END SUBROUTINE sub_input

  !> counter-based ensemble perturbation (SURVEY.md 8d cfg3): splitmix64(column_id xor 0x5A5A2026); column 0 unperturbed
  SUBROUTINE sub_perturbation()
    INTEGER(c_int64_t) :: c, h1, h2
    ALLOCATE(dT2m_col(ncol), precip_scale_col(ncol))
    DO c = 0, ncol - 1          ! counter-based: a function of the GLOBAL column index, whichever process holds the column
       h1 = splitmix64(IEOR(c + col0, INT(z'5A5A2026', c_int64_t)))
       h2 = splitmix64(h1)
       dT2m_col(c + 1) = -2.0_wp + 4.0_wp*u01(h1)
       precip_scale_col(c + 1) = 1.0_wp + (-0.3_wp + 0.6_wp*u01(h2))
    END DO
    IF (col0 == 0) THEN         ! global column 0 stays unperturbed (the reference run)
       dT2m_col(1) = 0._wp
       precip_scale_col(1) = 1._wp
    END IF
  CONTAINS
    FUNCTION splitmix64(x) RESULT(z)
      INTEGER(c_int64_t), INTENT(in) :: x
      INTEGER(c_int64_t) :: z
      z = x + INT(z'9E3779B97F4A7C15', c_int64_t)
      z = IEOR(z, SHIFTR(z, 30))*INT(z'BF58476D1CE4E5B9', c_int64_t)
      z = IEOR(z, SHIFTR(z, 27))*INT(z'94D049BB133111EB', c_int64_t)
      z = IEOR(z, SHIFTR(z, 31))
    END FUNCTION
    FUNCTION u01(h) RESULT(u)
      INTEGER(c_int64_t), INTENT(in) :: h
      REAL(wp) :: u
      u = REAL(SHIFTR(h, 11), wp)/9007199254740992._wp
    END FUNCTION
  END SUBROUTINE sub_perturbation
END MODULE mo_init

MODULE mo_output
  USE mo_data
  IMPLICIT NONE
CONTAINS
  !> output_begin, mo_output.f90:276-339: same files, same edit descriptors (F9.3 / F9.5 / ES14.7, two blanks)
  SUBROUTINE output_begin(Nlayer)
    INTEGER(c_int32_t), INTENT(in) :: Nlayer
    CHARACTER(len=16) :: n
    WRITE(n, '(I0)') Nlayer
    format_T     = '('//TRIM(n)//'(F9.3,2x))'
    format_psi   = '('//TRIM(n)//'(F9.3,2x))'
    format_thick = '('//TRIM(n)//'(F9.5,2x))'
    format_perm  = '('//TRIM(n)//'(ES14.7,2x))'
    format_snow  = '(F9.3,2x,F9.3,2x,F9.3,2x,F9.3)'
    format_melt  = '(ES14.7,2x,ES14.7,2x,ES14.7)'
    OPEN(30, file='./output/dat_T.dat',           STATUS='replace', Recl=12288)
    OPEN(31, file='./output/dat_psi_s.dat',       STATUS='replace', Recl=12288)
    OPEN(32, file='./output/dat_thick.dat',       STATUS='replace', Recl=12288)
    OPEN(33, file='./output/dat_S_bu.dat',        STATUS='replace', Recl=12288)
    OPEN(34, file='./output/dat_ray.dat',         STATUS='replace', Recl=12288)
    OPEN(35, file='./output/dat_psi_l.dat',       STATUS='replace', Recl=12288)
    OPEN(40, file='./output/dat_freeboard.dat',   STATUS='replace', Recl=12288)
    OPEN(41, file='./output/dat_snow.dat',        STATUS='replace', Recl=12288)
    OPEN(51, file='./output/dat_ensemble.dat',    STATUS='replace', Recl=12288)
    OPEN(42, file='./output/dat_vital_signs.dat', STATUS='replace', Recl=12288)
    OPEN(43, file='./output/dat_grav_drain.dat',  STATUS='replace', Recl=12288)
    OPEN(45, file='./output/dat_T2m_T_top.dat',   STATUS='replace', Recl=12288)
    OPEN(46, file='./output/dat_perm.dat',        STATUS='replace', Recl=12288)
    OPEN(47, file='./output/dat_flush_v.dat',     STATUS='replace', Recl=12288)
    OPEN(48, file='./output/dat_flush_h.dat',     STATUS='replace', Recl=12288)
    OPEN(49, file='./output/dat_psi_g.dat',       STATUS='replace', Recl=12288)
    OPEN(50, file='./output/dat_melt.dat',        STATUS='replace', Recl=12288)
  END SUBROUTINE output_begin

  !> output_settings, mo_output.f90:41-106 (same keys, same edit descriptors) + the new run parameters
  SUBROUTINE output_settings(description, testcase)
    CHARACTER*12000, INTENT(in) :: description
    INTEGER, INTENT(in) :: testcase
    OPEN(1234, file='./output/dat_settings.dat', STATUS='replace')
    WRITE(1234, *) '################  Description  ###############'
    WRITE(1234, *) TRIM(description)
    WRITE(1234, *) '#################  Testcase  #################'
    WRITE(1234, '(A16,I9)')    'testcase        =', testcase
    WRITE(1234, *) '##############  Basic settings  ##############'
    WRITE(1234, '(A16,F15.3)') 'dt              =', cfg%dt
    WRITE(1234, '(A16,F15.3)') 'thick_0         =', cfg%thick_0
    WRITE(1234, '(A16,F15.3)') 'time_out        =', cfg%time_out
    WRITE(1234, '(A16,F15.3)') 'time_total      =', cfg%time_total
    WRITE(1234, '(A16,F15.3)') 'fl_q_bottom     =', fl_q_bottom
    WRITE(1234, '(A16,F15.3)') 'T_bottom        =', cfg%T_bottom
    WRITE(1234, '(A16,F15.3)') 'S_bu_bottom     =', cfg%S_bu_bottom
    WRITE(1234, '(A16,I9.0)')  'N_top           =', cfg%n_top
    WRITE(1234, '(A16,I9.0)')  'N_middle        =', cfg%n_middle
    WRITE(1234, '(A16,I9.0)')  'N_bottom        =', cfg%n_bottom
    WRITE(1234, '(A16,I9.0)')  'Nlayer          =', cfg%nlayer
    WRITE(1234, *) '##################  Flags  ###################'
    WRITE(1234, '(A16,I9.0)')  'boundflux_flag  =', cfg%boundflux_flag
    WRITE(1234, '(A16,I9.0)')  'atmoflux_flag   =', cfg%atmoflux_flag
    WRITE(1234, '(A16,I9.0)')  'albedo_flag     =', cfg%albedo_flag
    WRITE(1234, '(A16,I9.0)')  'grav_flag       =', cfg%grav_flag
    WRITE(1234, '(A16,I9.0)')  'flush_flag      =', cfg%flush_flag
    WRITE(1234, '(A16,I9.0)')  'flood_flag      =', cfg%flood_flag
    WRITE(1234, '(A16,I9.0)')  'grav_heat_flag  =', cfg%grav_heat_flag
    WRITE(1234, '(A16,I9.0)')  'flush_heat_flag =', cfg%flush_heat_flag
    WRITE(1234, '(A16,I9.0)')  'harmonic_flag   =', cfg%harmonic_flag
    WRITE(1234, '(A16,F15.3)') 'k_snow_flush    =', cfg%k_snow_flush
    WRITE(1234, '(A16,I9.0)')  'prescribe_flag  =', cfg%prescribe_flag
    WRITE(1234, '(A16,I9.0)')  'salt_flag       =', cfg%salt_flag
    WRITE(1234, '(A16,I9.0)')  'turb_flag       =', cfg%turb_flag
    WRITE(1234, '(A16,I9.0)')  'bottom_flag     =', cfg%bottom_flag
    WRITE(1234, '(A16,I9.0)')  'tank_flag       =', cfg%tank_flag
    WRITE(1234, '(A16,I9.0)')  'precip_flag     =', cfg%precip_flag
    WRITE(1234, '(A16,I9.0)')  'bgc_flag        =', cfg%bgc_flag
    WRITE(1234, '(A16,I9.0)')  'N_bgc           =', N_bgc
    WRITE(1234, *) '#############  Ensemble (MI355X)  ############'
    WRITE(1234, '(A16,I12)')   'ncol            =', ncol
    WRITE(1234, '(A16,I12)')   'col0            =', col0
    WRITE(1234, '(A16,I12)')   'ncol_total      =', MERGE(ncol, ncol_total, ncol_total < 0)
    WRITE(1234, '(A16,I9)')    'out_col         =', out_col
    WRITE(1234, '(A16,L9)')    'perturb         =', perturb
    CLOSE(1234)
  END SUBROUTINE output_settings

  !> output, mo_output.f90:116-146: one row per output point for column `out_col` of the snapshot (olay, oscal)
  SUBROUTINE output(Nlayer, olay, oscal)
    INTEGER(c_int32_t), INTENT(in) :: Nlayer
    REAL(c_double), INTENT(in) :: olay(:, :), oscal(:)      !< (Nlayer, SAMSIM_NARR), (SAMSIM_NSCAL)
    WRITE(30, format_T)     olay(:, A_T)
    WRITE(31, format_psi)   olay(:, A_PSI_S)
    WRITE(32, format_thick) olay(:, A_THICK)
    WRITE(33, format_T)     olay(:, A_S_BU)
    WRITE(34, format_T)     olay(1:Nlayer - 1, A_RAY)
    WRITE(35, format_psi)   olay(:, A_PSI_L)
    WRITE(40, '(F9.3)')     oscal(S_FREEBOARD)
    WRITE(41, format_snow)  oscal(S_THICK_SNOW), oscal(S_T_SNOW), oscal(S_PSI_L_SNOW), oscal(S_PSI_S_SNOW)
    WRITE(42, '(F15.1,2x,F10.5,2x,F10.5,2x,F10.5,2x,F10.5)') oscal(S_ENERGY_STORED), oscal(S_FRESHWATER), &
         oscal(S_TOTAL_RESIST), oscal(S_THICKNESS), oscal(S_BULK_SALIN)
    WRITE(43, '(F9.6,2X,F9.5,2X,F7.3)') oscal(S_GRAV_DRAIN), oscal(S_GRAV_SALT), oscal(S_GRAV_TEMP)
    WRITE(45, *)            oscal(S_T2M), oscal(S_T_TOP)
    WRITE(46, format_perm)  olay(:, A_PERM)
    WRITE(47, format_perm)  olay(:, A_FLUSH_V)
    WRITE(48, format_perm)  olay(:, A_FLUSH_H)
    WRITE(49, format_psi)   olay(:, A_PSI_G)
    WRITE(50, format_melt)  oscal(S_MELT_OUT1), oscal(S_MELT_OUT2), oscal(S_MELT_OUT3)
  END SUBROUTINE output

  !> output_begin_bgc, mo_output.f90:354-384: two files per tracer, F16.8
  SUBROUTINE output_begin_bgc(Nlayer)
    INTEGER(c_int32_t), INTENT(in) :: Nlayer
    CHARACTER(len=16)  :: n
    CHARACTER(len=64)  :: name
    INTEGER :: k
    WRITE(n, '(I0)') Nlayer + 1
    format_bgc = '('//TRIM(n)//'(F16.8,2x))'
    DO k = 1, N_bgc
       WRITE(name, '(A,I2.2,A)') './output/dat_bgc', k, '.bu.dat'
       OPEN(2*k + 400, file=TRIM(name), STATUS='replace', Recl=12288)
       WRITE(name, '(A,I2.2,A)') './output/dat_bgc', k, '.br.dat'
       OPEN(2*k + 401, file=TRIM(name), STATUS='replace', Recl=12288)
    END DO
  END SUBROUTINE output_begin_bgc

  !> output_bgc, mo_output.f90:156-188: bulk and brine concentration per layer, the water's below N_active
  SUBROUTINE output_bgc(Nlayer, N_active, obgc, obot, olay)
    INTEGER(c_int32_t), INTENT(in) :: Nlayer, N_active
    REAL(c_double),     INTENT(in) :: obgc(Nlayer, N_bgc), obot(N_bgc), olay(:, :)
    REAL(wp) :: bgc_bu(Nlayer), bgc_br(Nlayer)
    INTEGER :: k, kk
    DO k = 1, N_bgc
       DO kk = 1, Nlayer
          IF (kk <= N_active) THEN
             IF (olay(kk, A_M) .NE. 0._wp) THEN
                bgc_bu(kk) = obgc(kk, k)/olay(kk, A_M)
                IF (olay(kk, A_PSI_L) .NE. 0._wp .AND. olay(kk, A_THICK) .NE. 0) THEN
                   bgc_br(kk) = obgc(kk, k)/olay(kk, A_PSI_L)/olay(kk, A_THICK)/rho_l
                ELSE
                   bgc_br(kk) = 0._wp
                END IF
             ELSE
                bgc_bu(kk) = 0._wp
                bgc_br(kk) = 0._wp
             END IF
          ELSE
             bgc_bu(kk) = obot(k)
             bgc_br(kk) = obot(k)
          END IF
       END DO
       WRITE(2*k + 400, format_bgc) bgc_bu
       WRITE(2*k + 401, format_bgc) bgc_br
    END DO
  END SUBROUTINE output_bgc

  !> One row per output point with the ensemble statistics (count, then mean / min / max / std of thickness, snow thickness,
  !! bulk salinity, freeboard, surface temperature and N_active): what stands in for "one .dat row per column" when the
  !! run holds 10^5..10^6 columns (SURVEY.md section 8 f.1).  Column out_col keeps its own dat_*.dat files.
  SUBROUTINE output_ensemble(h, time)
    TYPE(c_ptr), INTENT(in) :: h
    REAL(wp),    INTENT(in) :: time
    INTEGER(c_int32_t) :: slots(6)
    TYPE(samsim_stat)  :: q(6)
    INTEGER :: j
    slots = (/ S_THICKNESS - 1, S_THICK_SNOW - 1, S_BULK_SALIN - 1, S_FREEBOARD - 1, S_T_TOP - 1, -1 /)
    CALL samsim_check(samsim_get_ensemble_stats(h, 6_c_int32_t, slots, q), 'samsim_get_ensemble_stats')
    WRITE(51, '(F14.1,I10,24ES16.8)') time, q(1)%count, (q(j)%mean, q(j)%min, q(j)%max, q(j)%std, j = 1, 6)
  END SUBROUTINE output_ensemble

  SUBROUTINE output_end()
    INTEGER :: u
    DO u = 30, 35
       CLOSE(u)
    END DO
    DO u = 40, 51
       IF (u /= 44) CLOSE(u)
    END DO
    IF (cfg%bgc_flag == 2) THEN
       DO u = 1, N_bgc
          CLOSE(2*u + 400); CLOSE(2*u + 401)
       END DO
    END IF
  END SUBROUTINE output_end
END MODULE mo_output

MODULE mo_grotz
  USE mo_data
  USE mo_init
  USE mo_output
  IMPLICIT NONE
CONTAINS
  !> Binary checkpoint of the resident ensemble (the reference has no restart files; SURVEY.md section 8 f.1).  Same stream
  !! format as samsim_amd/checkpoint.py: header, then chunks of columns (col0, ncols, lay, scal, n_active, and -- header word
  !! has_status = 1 -- the STOP code, layer and step of a frozen column, so that it stays frozen after the restart).
  SUBROUTINE write_restart(h, path)
    TYPE(c_ptr), INTENT(in) :: h
    CHARACTER(len=*), INTENT(in) :: path
    INTEGER(c_int64_t), PARAMETER :: chunk = 65536
    TYPE(samsim_state_soa) :: st
    TYPE(samsim_clock)     :: clk
    REAL(c_double), ALLOCATABLE, TARGET :: blay(:, :, :), bscal(:, :), bbgc(:, :, :), bbot(:, :)
    INTEGER(c_int32_t), ALLOCATABLE, TARGET :: bna(:), fstat(:), flay(:)
    INTEGER(c_int64_t), ALLOCATABLE :: fstep(:)
    INTEGER(c_int64_t) :: c0, n, nb
    INTEGER :: u
    nb = 0
    IF (cfg%bgc_flag == 2) nb = N_bgc
    CALL samsim_check(samsim_get_clock(h, clk), 'samsim_get_clock')
    ALLOCATE(fstat(ncol), fstep(ncol), flay(ncol))
    CALL samsim_check(samsim_get_status(h, fstat, fstep, flay), 'samsim_get_status')
    OPEN(NEWUNIT=u, file=TRIM(path), STATUS='replace', ACCESS='stream', FORM='unformatted')
    WRITE(u) restart_magic, ncol, INT(cfg%nlayer, c_int64_t), INT(SAMSIM_NARR, c_int64_t), INT(SAMSIM_NSCAL, c_int64_t), &
         INT(cfg%testcase, c_int64_t), clk%time, clk%step, INT(clk%n_time_out, c_int64_t), INT(clk%time_counter, c_int64_t), &
         clk%n_outputs, nb, 1_c_int64_t, 0_c_int64_t, 0_c_int64_t, 0_c_int64_t
    c0 = 0
    DO WHILE (c0 < ncol)
       n = MIN(chunk, ncol - c0)
       ALLOCATE(blay(n, cfg%nlayer, SAMSIM_NARR), bscal(n, SAMSIM_NSCAL), bna(n))
       st%ncol = n; st%nlayer = cfg%nlayer; st%narr = SAMSIM_NARR
       st%lay = c_loc(blay); st%scal = c_loc(bscal); st%n_active = c_loc(bna)
       CALL samsim_check(samsim_get_state(h, st, c0), 'samsim_get_state')
       WRITE(u) c0, n
       WRITE(u) blay
       WRITE(u) bscal
       WRITE(u) bna
       WRITE(u) fstat(c0 + 1:c0 + n)
       WRITE(u) flay(c0 + 1:c0 + n)
       WRITE(u) fstep(c0 + 1:c0 + n)
       DEALLOCATE(blay, bscal, bna)
       IF (nb > 0) THEN                     ! tracer amounts and the concentration of the water below
          ALLOCATE(bbgc(n, cfg%nlayer, nb), bbot(n, nb))
          CALL samsim_check(samsim_get_tracer_state(h, bbgc, bbot, c0, n), 'samsim_get_tracer_state')
          WRITE(u) bbgc
          WRITE(u) bbot
          DEALLOCATE(bbgc, bbot)
       END IF
       c0 = c0 + n
    END DO
    CLOSE(u)
    PRINT '(A,A,A,I0)', ' restart file written: ', TRIM(path), '  step ', clk%step
  END SUBROUTINE write_restart

  SUBROUTINE read_restart(h, path)
    TYPE(c_ptr), INTENT(in) :: h
    CHARACTER(len=*), INTENT(in) :: path
    TYPE(samsim_state_soa) :: st
    TYPE(samsim_clock)     :: clk
    REAL(c_double), ALLOCATABLE, TARGET :: blay(:, :, :), bscal(:, :), bbgc(:, :, :), bbot(:, :)
    INTEGER(c_int32_t), ALLOCATABLE, TARGET :: bna(:), fstat(:), flay(:)
    INTEGER(c_int64_t), ALLOCATABLE :: fstep(:)
    INTEGER(c_int64_t) :: hdr(6), tail(3), pad(5), c0, n, done, nb
    INTEGER :: u
    nb = 0
    IF (cfg%bgc_flag == 2) nb = N_bgc
    OPEN(NEWUNIT=u, file=TRIM(path), STATUS='old', ACCESS='stream', FORM='unformatted')
    READ(u) hdr, clk%time, clk%step, tail, pad
    IF (hdr(1) /= restart_magic .OR. hdr(2) /= ncol .OR. hdr(3) /= cfg%nlayer .OR. hdr(5) /= SAMSIM_NSCAL .OR. &
         (hdr(4) /= SAMSIM_NARR .AND. hdr(4) /= SAMSIM_NPROG)) THEN
       PRINT *, 'restart file does not fit this run (magic, ncol, Nlayer, narr, nscal):', hdr(1:5)
       STOP 5
    END IF
    IF (pad(1) /= nb) THEN
       PRINT *, 'restart file holds', pad(1), 'tracers, this run', nb
       STOP 6
    END IF
    clk%n_time_out = INT(tail(1), c_int32_t); clk%time_counter = INT(tail(2), c_int32_t); clk%n_outputs = tail(3)
    done = 0
    DO WHILE (done < ncol)
       READ(u) c0, n
       ALLOCATE(blay(n, cfg%nlayer, hdr(4)), bscal(n, SAMSIM_NSCAL), bna(n))
       READ(u) blay
       READ(u) bscal
       READ(u) bna
       st%ncol = n; st%nlayer = cfg%nlayer; st%narr = INT(hdr(4), c_int32_t)
       st%lay = c_loc(blay); st%scal = c_loc(bscal); st%n_active = c_loc(bna)
       CALL samsim_check(samsim_set_state(h, st, c0), 'samsim_set_state')
       DEALLOCATE(blay, bscal, bna)
       IF (pad(2) == 1) THEN                ! has_status: set_state cleared the STOP codes, put them back
          ALLOCATE(fstat(n), flay(n), fstep(n))
          READ(u) fstat
          READ(u) flay
          READ(u) fstep
          CALL samsim_check(samsim_set_status(h, fstat, fstep, flay, c0, n), 'samsim_set_status')
          DEALLOCATE(fstat, flay, fstep)
       END IF
       IF (nb > 0) THEN
          ALLOCATE(bbgc(n, cfg%nlayer, nb), bbot(n, nb))
          READ(u) bbgc
          READ(u) bbot
          CALL samsim_check(samsim_set_tracer_state(h, bbgc, c0, n), 'samsim_set_tracer_state')
          CALL samsim_check(samsim_set_tracer_bottom(h, bbot, c0, n), 'samsim_set_tracer_bottom')
          DEALLOCATE(bbgc, bbot)
       END IF
       done = done + n
    END DO
    CLOSE(u)
    CALL samsim_check(samsim_set_clock(h, clk), 'samsim_set_clock')
    PRINT '(A,A,A,I0)', ' restarted from ', TRIM(path), '  step ', clk%step
  END SUBROUTINE read_restart

  !> grotz, mo_grotz.f90:83-877: initialisation, forcing read-in, time loop, final output.  The loop body is one
  !! samsim_step call per output interval; `output` is fed from the snapshot the kernel takes at the reference's output point.
  SUBROUTINE grotz(testcase, description, nml_unit)
    INTEGER,         INTENT(in) :: testcase, nml_unit
    CHARACTER*12000, INTENT(in) :: description
    TYPE(c_ptr) :: h
    TYPE(samsim_state_soa)  :: st
    TYPE(samsim_output_soa) :: o
    TYPE(samsim_clock)      :: clk
    REAL(c_double), ALLOCATABLE, TARGET :: olay(:, :, :), oscal(:, :), obgc(:, :, :), obot(:, :)
    INTEGER(c_int32_t), ALLOCATABLE, TARGET :: ona(:), status(:), err_layer(:), site_of_column(:)
    INTEGER(c_int64_t), ALLOCATABLE :: err_step(:)
    INTEGER(c_int64_t) :: n, done, total, cells, colsteps
    INTEGER :: nfail, count0, count1, rate
    REAL(wp) :: time, thick1

    CALL init(testcase, nml_unit)
    CALL output_begin(cfg%nlayer)
    IF (cfg%bgc_flag == 2) CALL output_begin_bgc(cfg%nlayer)
    CALL output_settings(description, testcase)

    CALL samsim_check(samsim_create(cfg, ncol, INT(device, c_int32_t), h), 'samsim_create')
    IF (cfg%atmoflux_flag == 2) THEN
       CALL sub_input()
       ALLOCATE(site_of_column(ncol))
       DO n = 1, ncol
          site_of_column(n) = INT(MOD(col0 + n - 1, INT(nsites, c_int64_t)), c_int32_t)
       END DO
       IF (perturb) THEN
          CALL sub_perturbation()
          CALL samsim_check(samsim_set_forcing_sites(h, INT(nsites, c_int32_t), INT(Length_Input, c_int32_t), fl_sw_input, &
               fl_lw_input, T2m_input, precip_input, site_of_column, c_loc(dT2m_col), c_loc(precip_scale_col)), &
               'samsim_set_forcing_sites')
       ELSE
          CALL samsim_check(samsim_set_forcing_sites(h, INT(nsites, c_int32_t), INT(Length_Input, c_int32_t), fl_sw_input, &
               fl_lw_input, T2m_input, precip_input, site_of_column, c_null_ptr, c_null_ptr), 'samsim_set_forcing_sites')
       END IF
    END IF
    st%ncol = ncol; st%nlayer = cfg%nlayer; st%narr = SAMSIM_NARR
    st%lay = c_loc(lay); st%scal = c_loc(scal); st%n_active = c_loc(n_active)
    IF (cfg%bgc_flag == 2) THEN
       IF (cfg%tank_flag == 2) THEN
          CALL samsim_check(samsim_set_tracers(h, INT(N_bgc, c_int32_t), bgc_bottom, c_loc(bgc_total)), 'samsim_set_tracers')
       ELSE
          CALL samsim_check(samsim_set_tracers(h, INT(N_bgc, c_int32_t), bgc_bottom, c_null_ptr), 'samsim_set_tracers')
       END IF
       CALL samsim_check(samsim_set_tracer_state(h, bgc_abs, 0_c_int64_t, ncol), 'samsim_set_tracer_state')
       ALLOCATE(obgc(1, cfg%nlayer, N_bgc), obot(1, N_bgc))
    END IF
    CALL samsim_check(samsim_set_state(h, st, 0_c_int64_t), 'samsim_set_state')
    IF (LEN_TRIM(restart_in) > 0) CALL read_restart(h, restart_in)
    CALL samsim_check(samsim_set_output_window(h, INT(out_col - 1, c_int64_t), 1_c_int64_t), 'samsim_set_output_window')
    ALLOCATE(olay(1, cfg%nlayer, SAMSIM_NARR), oscal(1, SAMSIM_NSCAL), ona(1))
    o%ncols = 1; o%nlayer = cfg%nlayer; o%reserved = 0
    o%lay = c_loc(olay); o%scal = c_loc(oscal); o%n_active = c_loc(ona)

    total = i_time
    IF (max_steps >= 0) total = MIN(total, max_steps)
    CALL samsim_check(samsim_get_clock(h, clk), 'samsim_get_clock')
    done = clk%step            ! > 0 after a restart: the run continues to the same end
    CALL SYSTEM_CLOCK(count0, rate)
    DO WHILE (done < total)
       n = MIN(samsim_steps_to_output(h), total - done)
       CALL samsim_check(samsim_step(h, n), 'samsim_step')
       done = done + n
       IF (samsim_steps_to_output(h) == cfg%i_time_out + 1 .OR. done == 1) THEN   ! an output point was just passed
          CALL samsim_check(samsim_get_output(h, o), 'samsim_get_output')
          CALL output(cfg%nlayer, olay(1, :, :), oscal(1, :))
          IF (cfg%bgc_flag == 2) THEN
             CALL samsim_check(samsim_get_tracer_output(h, obgc, obot), 'samsim_get_tracer_output')
             CALL output_bgc(cfg%nlayer, ona(1), obgc(1, :, :), obot(1, :), olay(1, :, :))
          END IF
          CALL output_ensemble(h, o%time)
          time = o%time
          thick1 = olay(1, 1, A_THICK)
          ! console progress line, mo_grotz.f90:371-381
          WRITE(*, '(A10,I3,A15,F6.3,A14,F7.3,A30,F3.1,A14,F6.4,A10,F7.3,A7,F7.3)') &
               'progress: ', INT(100._wp*(time + cfg%dt)/cfg%time_total), &
               '%,  thickness: ', oscal(1, S_THICKNESS), &
               ',  surface T: ', oscal(1, S_T_TOP), &
               'C,  thermal stability (<0.5): ', k_s*cfg%dt/rho_s/c_s/MIN(thick1, cfg%thick_0)**2._wp, &
               ',  snow_thick:', oscal(1, S_THICK_SNOW), &
               ',  T_snow:', oscal(1, S_T_SNOW), &
               ',  T2m:', oscal(1, S_T2M)
       END IF
    END DO
    CALL samsim_check(samsim_synchronize(h), 'samsim_synchronize')
    CALL SYSTEM_CLOCK(count1)

    ! the reference aborts with STOP n; here failed columns are reported (SURVEY.md section 5)
    ALLOCATE(status(ncol), err_step(ncol), err_layer(ncol))
    CALL samsim_check(samsim_get_status(h, status, err_step, err_layer), 'samsim_get_status')
    nfail = COUNT(status /= 0)
    IF (nfail > 0) PRINT *, 'columns stopped by a reference STOP code:', nfail, ' first code:', &
         status(MINLOC(MERGE(0, 1, status /= 0), 1))
    CALL samsim_check(samsim_get_state(h, st, 0_c_int64_t), 'samsim_get_state')
    CALL samsim_check(samsim_get_work(h, cells, colsteps), 'samsim_get_work')
    CALL samsim_check(samsim_get_clock(h, clk), 'samsim_get_clock')
    WRITE(*, *) 'Run completed, total ice thickness at end of run:', &
         SUM(lay(out_col, 1:MAX(n_active(out_col) - 1, 0), A_THICK)), ' melt_err= ', scal(out_col, S_MELT_ERR)
    WRITE(*, '(A,I0,A,I0,A,F9.3,A,ES10.3,A)') ' MI355X: ', ncol, ' columns x ', clk%step, ' steps in ', &
         REAL(count1 - count0, wp)/REAL(rate, wp), ' s  (', REAL(colsteps, wp)/(REAL(count1 - count0, wp)/REAL(rate, wp)), &
         ' column-timesteps/s)'
    IF (LEN_TRIM(restart_out) > 0) CALL write_restart(h, restart_out)
    CALL output_end()
    CALL samsim_destroy(h)
    CALL sub_deallocate()
    IF (nfail > 0 .AND. status(out_col) /= 0) STOP 1337
  END SUBROUTINE grotz
END MODULE mo_grotz

!> Program entry, as SAMSIM.f90:84-106, with the run parameters read from `samsim.nml` instead of being compiled in.
PROGRAM SAMSIM
  USE mo_grotz
  IMPLICIT NONE
  INTEGER         :: testcase, ios, nml_unit
  CHARACTER*12000 :: description
  LOGICAL         :: have_nml
  NAMELIST /samsim_run/ testcase, ncol, col0, ncol_total, device, out_col, perturb, description, max_steps, restart_in, &
       restart_out, sites

  testcase    = 1
  description = 'MI355X-native batched column solver'
  nml_unit = -1
  INQUIRE(file='samsim.nml', exist=have_nml)
  IF (have_nml) THEN
     nml_unit = 1230
     OPEN(nml_unit, file='samsim.nml', status='old')
     READ(nml_unit, NML=samsim_run, IOSTAT=ios)
     IF (ios > 0) THEN
        PRINT *, 'error in namelist group samsim_run'
        STOP 4
     END IF
  END IF
  testcase_id = testcase
  PRINT *, 'SAMSIM is getting ready'
  CALL grotz(testcase, description, nml_unit)
  IF (have_nml) CLOSE(nml_unit)
  PRINT *, 'SAMSIM is finished'
END PROGRAM SAMSIM
