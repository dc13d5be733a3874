!> TEST INFRASTRUCTURE (oracle side) -- builder-written, not reference code.
!!
!! Calls individual procedures of the UNMODIFIED reference modules (mo_thermo_functions, mo_functions,
!! mo_snow) on deterministic input grids and writes (inputs, outputs) as raw float64 records, so that the
!! CPU oracle can be pinned function by function (SURVEY.md section 4, "function-level golden vectors").
!! Output: stream file func_golden.bin = sequence of blocks [int32 tag, int32 ncols, int32 nrows, int32 0] + data(ncols,nrows).
PROGRAM func_harness
  USE mo_parameters
  USE mo_data, ONLY: salt_flag
  USE mo_thermo_functions
  USE mo_functions
  USE mo_snow, ONLY: func_k_snow
  USE mo_flood, ONLY: flood_simple
  USE mo_grav_drain, ONLY: fl_grav_drain_simple
  IMPLICIT NONE
  INTEGER, PARAMETER :: u = 55
  INTEGER :: sf, i, j, g, n
  REAL(wp) :: H, S, Tin, T, phi, x, y, ps, pl, pg, vex, m, th
  REAL(wp), ALLOCATABLE :: buf(:,:)
  REAL(wp) :: guesses(6), sal(8)
  REAL(wp) :: a_ps(12), a_pg(12), a_m(12), a_th(12), r
  REAL(wp) :: a_pl(12), a_S(12), a_H(12), a_Sbr(12), a_ray(11), hs, ms, ts, gd
  INTEGER  :: na

  OPEN(u, file='func_golden.bin', STATUS='replace', ACCESS='stream', FORM='unformatted')

  guesses = (/ -0.5_wp, -1.8_wp, -5._wp, -12._wp, -25._wp, -60._wp /)
  sal = (/ 0.0005_wp, 0.5_wp, 2._wp, 5._wp, 10._wp, 20._wp, 34._wp, 60._wp /)
  ! tag 1/2: getT for salt_flag 1/2 : columns H, S_bu, T_in, T, phi
  DO sf = 1, 2
     salt_flag = sf
     n = 8*400*6
     ALLOCATE(buf(5, n))
     n = 0
     DO i = 1, 8
        DO j = 1, 400
           DO g = 1, 6
              S = sal(i)
              H = -330000._wp + 340000._wp*REAL(j-1, wp)/399._wp
              Tin = guesses(g)
              phi = -9._wp
              CALL getT(H, S, Tin, T, phi, 1)
              n = n + 1
              buf(:, n) = (/ H, S, Tin, T, phi /)
           END DO
        END DO
     END DO
     WRITE(u) sf, 5, n, 0
     WRITE(u) buf
     DEALLOCATE(buf)
  END DO
  ! tag 3/4: func_S_br (no clamp), func_S_br (clamp at 30), func_ddT_S_br, func_T_freeze : columns x, S_br, S_br30, ddT, T_freeze(x as salinity)
  DO sf = 1, 2
     salt_flag = sf
     ALLOCATE(buf(5, 500))
     DO j = 1, 500
        x = -40._wp + 45._wp*REAL(j-1, wp)/499._wp
        y = 80._wp*REAL(j-1, wp)/499._wp
        buf(:, j) = (/ x, func_S_br(x), func_S_br(x, 30._wp), func_ddT_S_br(x), func_T_freeze(y, sf) /)
     END DO
     WRITE(u) 2+sf, 5, 500, 0
     WRITE(u) buf
     DEALLOCATE(buf)
  END DO
  ! tag 5: func_density(T,S), func_k_snow(m, th): columns T, S, density, m_snow, thick_snow, k_snow
  ALLOCATE(buf(6, 400))
  DO j = 1, 400
     x = -5._wp + 7._wp*REAL(MOD(j*7, 400), wp)/399._wp
     y = 70._wp*REAL(j-1, wp)/399._wp
     m = 0.5_wp + 300._wp*REAL(j-1, wp)/399._wp
     th = 0.002_wp + 0.9_wp*REAL(MOD(j*13, 400), wp)/399._wp
     buf(:, j) = (/ x, y, func_density(x, y), m, th, func_k_snow(m, th) /)
  END DO
  WRITE(u) 5, 6, 400, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 6: func_albedo: columns thick_snow, T_snow, psi_l, thick_min, flag, albedo
  ALLOCATE(buf(6, 2*6*4*12))
  n = 0
  DO sf = 1, 2
     DO i = 1, 6
        DO g = 1, 4
           DO j = 1, 12
              x = 0.001_wp*REAL(i-1, wp)**2 + 0.0001_wp*REAL(i-1,wp)
              y = -0.03_wp + 0.01_wp*REAL(g, wp)
              pl = REAL(j-1, wp)/11._wp
              n = n + 1
              buf(:, n) = (/ x, y, pl, 0.005_wp, REAL(sf, wp), func_albedo(x, y, pl, 0.005_wp, sf) /)
           END DO
        END DO
     END DO
  END DO
  WRITE(u) 6, 6, n, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 7: Expulsion: columns phi, thick, m, psi_s, psi_l, psi_g, V_ex
  ALLOCATE(buf(7, 300))
  DO j = 1, 300
     phi = REAL(MOD(j*17, 300), wp)/299._wp
     th = 0.01_wp
     m = th*(850._wp + 250._wp*REAL(j-1, wp)/299._wp)
     CALL Expulsion(phi, th, m, ps, pl, pg, vex)
     buf(:, j) = (/ phi, th, m, ps, pl, pg, vex /)
  END DO
  WRITE(u) 7, 7, 300, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 8: func_freeboard on 40 pseudo-random 12-layer profiles: columns psi_s(12), psi_g(12), m(12), thick(12), m_snow, flag, freeboard
  ALLOCATE(buf(51, 40))
  r = 0.37_wp
  DO j = 1, 40
     DO i = 1, 12
        r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
        a_ps(i) = 0.05_wp + 0.9_wp*r
        r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
        a_pg(i) = 0.05_wp*r
        a_th(i) = 0.01_wp*(1._wp + REAL(MOD(i, 3), wp))
        a_m(i)  = a_th(i)*(a_ps(i)*rho_s + (1._wp - a_ps(i) - a_pg(i))*rho_l)
     END DO
     x = 40._wp*REAL(MOD(j, 5), wp)
     g = MOD(j, 2)
     buf(1:12, j)  = a_ps
     buf(13:24, j) = a_pg
     buf(25:36, j) = a_m
     buf(37:48, j) = a_th
     buf(49, j) = x
     buf(50, j) = REAL(g, wp)
     buf(51, j) = func_freeboard(12, 12, a_ps, a_pg, a_m, a_th, x, g)
  END DO
  WRITE(u) 8, 51, 40, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 9: flood_simple on 60 pseudo-random top layers: columns freeboard, S_abs, H_abs, m, thick, T_bottom, S_bu_bottom,
  !        H_abs_snow, m_snow, thick_snow, psi_g_snow (inputs) | S_abs, H_abs, m, thick, H_abs_snow, m_snow, thick_snow (outputs)
  ALLOCATE(buf(18, 60))
  DO j = 1, 60
     r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
     x = -0.05_wp - 0.04_wp*r                         ! freeboard below neg_free
     r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
     a_th = 0.01_wp; a_m = 9.3_wp + r; a_S = a_m*(3._wp + 10._wp*r); a_H = -a_m*(50000._wp + 200000._wp*r)
     r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
     ts = 0.2_wp + 0.5_wp*r; ms = ts*330._wp; hs = -ms*(333500._wp + 30000._wp*r); pg = 1._wp - 330._wp/920._wp
     y = -1.0_wp - r
     buf(1:11, j) = (/ x, a_S(1), a_H(1), a_m(1), a_th(1), y, 34._wp, hs, ms, ts, pg /)
     CALL flood_simple(x, a_S, a_H, a_m, a_th, y, 34._wp, hs, ms, ts, pg, 12, 12, 1)
     buf(12:18, j) = (/ a_S(1), a_H(1), a_m(1), a_th(1), hs, ms, ts /)
  END DO
  WRITE(u) 9, 18, 60, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 10: fl_grav_drain_simple on 60 pseudo-random 12-layer profiles, harmonic_flag 1 and 2.  The routine reads its local
  !         harmonic_perm without initialising it; scrub_stack leaves zeros where that local will live, which is what the
  !         expression was written to start from.  columns psi_s(12) psi_l(12) thick(12) S_abs(12) S_br(12) N_active flag | ray(11) S_abs(12)
  ALLOCATE(buf(85, 120))
  DO j = 1, 120
     na = 2 + MOD(j, 11)
     DO i = 1, 12
        r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
        a_pl(i) = 0.02_wp + 0.45_wp*r*REAL(i, wp)/12._wp
        a_ps(i) = 1._wp - a_pl(i)
        a_th(i) = 0.01_wp
        r = MOD(r*997._wp + 0.1234567_wp, 1._wp)
        a_S(i)  = 9._wp*(2._wp + 8._wp*r)
        a_Sbr(i) = 34._wp + 150._wp*REAL(12-i, wp)/12._wp*(0.5_wp + r)
     END DO
     a_ps(na) = 0.04_wp*r; a_pl(na) = 1._wp - a_ps(na); a_Sbr(na) = 34._wp
     g = 1 + MOD(j, 2)
     buf(1:12, j) = a_ps; buf(13:24, j) = a_pl; buf(25:36, j) = a_th; buf(37:48, j) = a_S; buf(49:60, j) = a_Sbr
     buf(61, j) = REAL(na, wp); buf(62, j) = REAL(g, wp)
     gd = 1._wp
     IF (j == 1) THEN          ! resolve the symbol once so that no loader frame runs between scrub and call
        a_H = a_S
        CALL fl_grav_drain_simple(a_ps, a_pl, a_th, a_H, a_Sbr, 12, na, a_ray, gd, g)
     END IF
     CALL scrub_stack()
     CALL fl_grav_drain_simple(a_ps, a_pl, a_th, a_S, a_Sbr, 12, na, a_ray, gd, g)
     buf(63:73, j) = a_ray; buf(74:85, j) = a_S
  END DO
  WRITE(u) 10, 85, 120, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  ! tag 11: sub_notzflux: columns time, fl_sw, fl_rest
  ALLOCATE(buf(3, 800))
  DO j = 1, 800
     x = 86400._wp*1.3_wp*REAL(j-1, wp) + 60._wp*REAL(MOD(j*37, 1440), wp)
     CALL sub_notzflux(x, y, m)
     buf(:, j) = (/ x, y, m /)
  END DO
  WRITE(u) 11, 3, 800, 0
  WRITE(u) buf
  DEALLOCATE(buf)
  CLOSE(u)
CONTAINS
  SUBROUTINE scrub_stack()
    CALL scrub_level(32)
  END SUBROUTINE scrub_stack
  RECURSIVE SUBROUTINE scrub_level(n)     ! recursion keeps the compiler from folding the pad into the caller's frame
    INTEGER, INTENT(in) :: n
    REAL(wp), VOLATILE  :: pad(512)
    pad = 0._wp
    IF (n > 0) CALL scrub_level(n - 1)
    IF (pad(1 + MOD(n, 512)) /= 0._wp) PRINT *, 'scrub'
  END SUBROUTINE scrub_level
END PROGRAM func_harness
