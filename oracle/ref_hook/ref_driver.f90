!> TEST INFRASTRUCTURE (oracle side) -- builder-written, not reference code.
!!
!! Entry program for the reference build under oracle/_ref/.  The reference's own entry
!! (SAMSIM.f90:84-106) hard-codes `testcase` and needs recompiling to change it; this one takes the
!! testcase number from the command line and otherwise does exactly what that program does:
!! one call of the reference driver `grotz(testcase, description)` (mo_grotz.f90:83).
PROGRAM ref_driver
  USE mo_grotz
  IMPLICIT NONE
  INTEGER           :: testcase, stat
  CHARACTER(len=32) :: arg
  CHARACTER*12000   :: description

  testcase = 1
  IF (COMMAND_ARGUMENT_COUNT() >= 1) THEN
     CALL GET_COMMAND_ARGUMENT(1, arg)
     READ(arg, *, IOSTAT=stat) testcase
     IF (stat /= 0) STOP 2
  END IF
  description = 'oracle/_ref run of the unmodified reference physics'
  CALL grotz(testcase, description)
END PROGRAM ref_driver
