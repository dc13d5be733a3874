!> TEST INFRASTRUCTURE (oracle side) -- builder-written, not reference code.
!!
!! Stand-in for the reference's writer module `mo_output` (interface restated from
!! /root/reference/mo_output.f90:41-384: output_settings, output, output_bgc, output_raw,
!! output_raw_snow, output_raw_lay, output_begin, output_begin_bgc).  It is linked with the
!! UNMODIFIED reference physics/driver modules (oracle/build_ref.sh compiles those where they lie
!! under /root/reference) so that the reference run can be observed at full float64 precision
!! instead of the 3-decimal F9.3 `.dat` files (mo_output.f90:300-313).
!!
!! Every call of `output` (output steps, mo_grotz.f90:363) and -- when tracing is requested --
!! every call of `output_raw` (once per time step, mo_grotz.f90:328) appends one binary record
!! to the stream file named by SAMSIM_REF_DUMP (default ./ref_dump.bin).  The record holds the
!! complete `mo_data` column state, so it doubles as the teacher-forcing checkpoint SURVEY.md
!! section 4 asks for.  Record layout (little endian, see tests/refdump.py):
!!   int32 magic=0x53414D53, kind(1=output,2=trace), step i, N_active, Nlayer, time_counter, 2 pad
!!   float64 scalars(NSCAL), then NARR arrays of Nlayer float64 each.
!!
!! Environment knobs (all optional; the physics modules are never edited):
!!   SAMSIM_REF_DUMP       path of the dump file
!!   SAMSIM_REF_MAXSTEPS   overrides mo_data::i_time before the time loop starts (bounded runs)
!!   SAMSIM_REF_BGC        0 -> sets bgc_flag=1 (tracers off; T/phi/S are unaffected, SURVEY.md 2 row 9)
!!   SAMSIM_REF_TRACE_FROM / SAMSIM_REF_TRACE_TO   step window (inclusive) of per-step trace records
!!   SAMSIM_REF_QUIET      1 -> no dump at all (timing runs)
!!   SAMSIM_REF_INIT       1 -> a first record of kind 3 holds the state init(testcase) left (initial profiles typed into init)
!!   SAMSIM_REF_FLUSH / _GRAV / _FLOOD / _PRESCRIBE   override flush_flag, grav_flag, flood_flag, prescribe_flag after init
!!   SAMSIM_REF_HARMONIC / _BOTTOM / _FREEBOARD_SNOW / _SNOW_FLUSH   override harmonic_flag, bottom_flag, freeboard_snow_flag, snow_flush_flag
MODULE mo_output

  USE mo_parameters, ONLY: wp
  IMPLICIT NONE

  INTEGER, PARAMETER :: dump_unit = 77
  INTEGER, PARAMETER :: nscal = 48
  INTEGER, SAVE      :: trace_from = 0, trace_to = -1
  LOGICAL, SAVE      :: bgc_open = .FALSE.
  LOGICAL, SAVE      :: quiet = .FALSE.

CONTAINS

  SUBROUTINE env_int(name, val, found)
    CHARACTER(len=*), INTENT(in)  :: name
    INTEGER,          INTENT(out) :: val
    LOGICAL,          INTENT(out) :: found
    CHARACTER(len=64) :: buf
    INTEGER :: stat, length
    val = 0
    found = .FALSE.
    CALL GET_ENVIRONMENT_VARIABLE(name, buf, length, stat)
    IF (stat == 0 .AND. length > 0) THEN
       READ(buf(1:length), *, IOSTAT=stat) val
       found = (stat == 0)
    END IF
  END SUBROUTINE env_int

  SUBROUTINE dump_record(kind)
    USE mo_data
    INTEGER, INTENT(in) :: kind
    REAL(wp) :: s(nscal)
    REAL(wp), ALLOCATABLE :: pad(:)
    IF (quiet) RETURN
    s = 0._wp
    s(1)  = time
    s(2)  = dt
    s(3)  = thick_0
    s(4)  = T_bottom
    s(5)  = S_bu_bottom
    s(6)  = T_top
    s(7)  = T2m
    s(8)  = fl_q_bottom
    s(9)  = m_snow
    s(10) = H_abs_snow
    s(11) = S_abs_snow
    s(12) = thick_snow
    s(13) = psi_s_snow
    s(14) = psi_l_snow
    s(15) = psi_g_snow
    s(16) = T_snow
    s(17) = phi_s
    s(18) = liquid_precip
    s(19) = solid_precip
    s(20) = fl_q_snow
    s(21) = melt_thick
    s(22) = melt_thick_snow
    s(23) = melt_thick_output(1)
    s(24) = melt_thick_output(2)
    s(25) = melt_thick_output(3)
    s(26) = freeboard
    s(27) = T_freeze
    s(28) = albedo
    s(29) = fl_sw
    s(30) = fl_lw
    s(31) = fl_rest
    s(32) = grav_drain
    s(33) = grav_salt
    s(34) = grav_temp
    s(35) = melt_err
    s(36) = energy_stored
    s(37) = freshwater
    s(38) = total_resist
    s(39) = thickness
    s(40) = bulk_salin
    s(41) = fl_Q(1)
    s(42) = fl_Q(MIN(N_active+1, Nlayer+1))
    s(43) = thick_min
    s(44) = REAL(n_time_out, wp)
    s(45) = REAL(i_time_out, wp)
    s(46) = REAL(i_time, wp)
    WRITE(dump_unit) INT(z'53414D53'), kind, i, N_active, Nlayer, time_counter, 0, 0
    WRITE(dump_unit) s
    WRITE(dump_unit) H_abs, S_abs, m, thick
    WRITE(dump_unit) T, phi, psi_s, psi_l, psi_g, S_bu, S_br, V_ex
    ALLOCATE(pad(Nlayer))
    pad = 0._wp
    pad(1:Nlayer-1) = ray
    WRITE(dump_unit) pad
    WRITE(dump_unit) perm, flush_v, flush_h, fl_rad
    pad(1:Nlayer) = fl_Q(1:Nlayer)
    WRITE(dump_unit) pad
    DEALLOCATE(pad)
  END SUBROUTINE dump_record

  SUBROUTINE output_settings(description,testcase,N_top,N_bottom,Nlayer,fl_q_bottom,T_bottom,S_bu_bottom,thick_0,time_out,    &
    time_total,dt,boundflux_flag,atmoflux_flag,albedo_flag,grav_flag,flush_flag,flood_flag,grav_heat_flag,flush_heat_flag,    &
    harmonic_flag,prescribe_flag,salt_flag,turb_flag,bottom_flag,tank_flag,precip_flag,bgc_flag,N_bgc,k_snow_flush)
    INTEGER,         INTENT(in) :: testcase,N_top,N_bottom,Nlayer, boundflux_flag,atmoflux_flag,albedo_flag,grav_flag,flush_flag, &
                                   flood_flag,grav_heat_flag , flush_heat_flag,harmonic_flag,prescribe_flag,salt_flag,turb_flag,  &
                                   bottom_flag,tank_flag,precip_flag,bgc_flag,N_bgc
    REAL(wp),        INTENT(in) :: fl_q_bottom,T_bottom,S_bu_bottom,thick_0,time_out,time_total,dt,k_snow_flush
    CHARACTER*12000, INTENT(in) :: description
    PRINT '(A,I4,A,I5,A,F8.3,A,F8.4)', ' ref-hook: testcase ', testcase, ' Nlayer ', Nlayer, ' dt ', dt, ' thick_0 ', thick_0
  END SUBROUTINE output_settings

  SUBROUTINE output(Nlayer,T,psi_s,psi_l,thick,S_bu,ray,format_T,format_psi, &
       format_thick,format_snow,freeboard,thick_snow,T_snow,psi_l_snow,psi_s_snow,           &
       energy_stored,freshwater,total_resist,thickness,bulk_salin,                &
       grav_drain,grav_salt,grav_temp,T2m,T_top,perm,format_perm,flush_v,flush_h,psi_g,melt_thick_output,format_melt)
    INTEGER,                       INTENT(in) :: Nlayer
    REAL(wp), DIMENSION(Nlayer),   INTENT(in) :: T,psi_s,psi_l,thick,S_bu,perm,flush_v,flush_h,psi_g
    REAL(wp), DIMENSION(Nlayer-1), INTENT(in) :: ray
    REAL(wp),                      INTENT(in) :: freeboard,thick_snow,T_snow,psi_l_snow,psi_s_snow,energy_stored,&
                                                 freshwater,thickness,bulk_salin, &
                                                 total_resist,grav_drain,grav_salt,grav_temp,T2m,T_top
    REAL(wp), DIMENSION(3),        INTENT(in) :: melt_thick_output
    CHARACTER*12000,               INTENT(in) :: format_T,format_psi,format_thick,format_snow,format_perm,format_melt
    CALL dump_record(1)
  END SUBROUTINE output

  !> tracer records go to a second stream file, <dump>.bgc: int32 step, N_active, N_bgc, Nlayer; bgc_bottom(N_bgc); bgc_abs(Nlayer,N_bgc)
  SUBROUTINE output_bgc(Nlayer,N_active,bgc_bottom,N_bgc,bgc_abs,psi_l,thick,m,format_bgc)
    USE mo_data, ONLY: i
    INTEGER,                             INTENT(in) :: Nlayer, N_bgc, N_active
    REAL(wp), DIMENSION(N_bgc),          INTENT(in) :: bgc_bottom
    REAL(wp), DIMENSION(Nlayer),         INTENT(in) :: psi_l,m,thick
    REAL(wp), DIMENSION(Nlayer,N_bgc),   INTENT(in) :: bgc_abs
    CHARACTER*12000,                     INTENT(in) :: format_bgc
    CHARACTER(len=1024) :: path
    INTEGER :: stat, length
    IF (quiet) RETURN
    IF (.NOT. bgc_open) THEN
       CALL GET_ENVIRONMENT_VARIABLE('SAMSIM_REF_DUMP', path, length, stat)
       IF (stat /= 0 .OR. length == 0) path = './ref_dump.bin'
       OPEN(dump_unit + 1, file=TRIM(path)//'.bgc', STATUS='replace', ACCESS='stream', FORM='unformatted')
       bgc_open = .TRUE.
    END IF
    WRITE(dump_unit + 1) i, N_active, N_bgc, Nlayer
    WRITE(dump_unit + 1) bgc_bottom
    WRITE(dump_unit + 1) bgc_abs
  END SUBROUTINE output_bgc

  SUBROUTINE output_raw(Nlayer,N_active,time,T,thick,S_bu,psi_s,psi_l,psi_g)
    USE mo_data, ONLY: i
    INTEGER,                     INTENT(in) :: Nlayer,N_active
    REAL(wp),                    INTENT(in) :: time
    REAL(wp), DIMENSION(Nlayer), INTENT(in) :: T,thick,S_bu,psi_s,psi_l,psi_g
    IF (i >= trace_from .AND. i <= trace_to) CALL dump_record(2)
  END SUBROUTINE output_raw

  SUBROUTINE output_raw_snow(time,T_snow,thick_snow,S_abs_snow,m_snow,psi_s_snow,psi_l_snow,psi_g_snow)
    REAL(wp), INTENT(in) :: time
    REAL(wp), INTENT(in) :: T_snow,thick_snow,S_abs_snow,m_snow,psi_s_snow,psi_l_snow,psi_g_snow
  END SUBROUTINE output_raw_snow

  SUBROUTINE output_raw_lay(Nlayer,N_active,H_abs,m,S_abs,thick,string)
    INTEGER,                     INTENT(in) :: Nlayer,N_active
    REAL(wp), DIMENSION(Nlayer), INTENT(in) :: H_abs,S_abs,thick,m
    CHARACTER*6,                 INTENT(in) :: string
  END SUBROUTINE output_raw_lay

  SUBROUTINE output_begin(Nlayer,debug_flag,format_T,format_psi,format_thick,format_snow,format_T2m_top,format_perm,&
                          &format_melt)
    USE mo_data, ONLY: i_time, bgc_flag, dbg => debug_flag, i_time_out, flush_flag, grav_flag, flood_flag, prescribe_flag, &
         harmonic_flag, bottom_flag, freeboard_snow_flag, snow_flush_flag
    INTEGER,         INTENT(in)  :: Nlayer,debug_flag
    CHARACTER*12000, INTENT(out) :: format_T,format_psi,format_thick,format_snow,format_T2m_top,format_perm,&
                                    &format_melt
    CHARACTER(len=1024) :: path
    INTEGER :: v, stat, length
    LOGICAL :: found
    format_T = ' '; format_psi = ' '; format_thick = ' '; format_snow = ' '
    format_T2m_top = ' '; format_perm = ' '; format_melt = ' '
    CALL env_int('SAMSIM_REF_QUIET', v, found)
    quiet = found .AND. v == 1
    CALL env_int('SAMSIM_REF_MAXSTEPS', v, found)
    IF (found .AND. v > 0) i_time = MIN(i_time, v)
    CALL env_int('SAMSIM_REF_I_TIME_OUT', v, found)      ! output cadence in steps (diagnosis of a first difference)
    IF (found .AND. v > 0) i_time_out = v
    CALL env_int('SAMSIM_REF_BGC', v, found)
    IF (found .AND. v == 0) bgc_flag = 1
    ! flag variants the reference's init holds as commented-out lines (mo_init.f90:1068-1071, 1386-1390)
    CALL env_int('SAMSIM_REF_FLUSH', v, found)
    IF (found) flush_flag = v
    CALL env_int('SAMSIM_REF_GRAV', v, found)
    IF (found) grav_flag = v
    CALL env_int('SAMSIM_REF_FLOOD', v, found)
    IF (found) flood_flag = v
    CALL env_int('SAMSIM_REF_PRESCRIBE', v, found)
    IF (found) prescribe_flag = v
    ! flag values no shipped testcase uses, pinned on testcase 4 (tests/golden/make_flag_fixtures.py)
    CALL env_int('SAMSIM_REF_HARMONIC', v, found)
    IF (found) harmonic_flag = v
    CALL env_int('SAMSIM_REF_BOTTOM', v, found)
    IF (found) bottom_flag = v
    CALL env_int('SAMSIM_REF_FREEBOARD_SNOW', v, found)
    IF (found) freeboard_snow_flag = v
    CALL env_int('SAMSIM_REF_SNOW_FLUSH', v, found)
    IF (found) snow_flush_flag = v
    CALL env_int('SAMSIM_REF_TRACE_FROM', v, found)
    IF (found) trace_from = v
    CALL env_int('SAMSIM_REF_TRACE_TO', v, found)
    IF (found) trace_to = v
    IF (trace_to >= trace_from) dbg = 2
    IF (.NOT. quiet) THEN
       CALL GET_ENVIRONMENT_VARIABLE('SAMSIM_REF_DUMP', path, length, stat)
       IF (stat /= 0 .OR. length == 0) path = './ref_dump.bin'
       OPEN(dump_unit, file=TRIM(path), STATUS='replace', ACCESS='stream', FORM='unformatted')
       CALL env_int('SAMSIM_REF_INIT', v, found)          ! 1 -> first record (kind 3) = the state init(testcase) left
       IF (found .AND. v == 1) CALL dump_record(3)
    END IF
  END SUBROUTINE output_begin

  SUBROUTINE output_begin_bgc(Nlayer,N_bgc,format_bgc)
    INTEGER,         INTENT(in)  :: Nlayer,N_bgc
    CHARACTER*12000, INTENT(out) :: format_bgc
    format_bgc = ' '
  END SUBROUTINE output_begin_bgc

END MODULE mo_output
