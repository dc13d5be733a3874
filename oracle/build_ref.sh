#!/usr/bin/env bash
# TEST INFRASTRUCTURE: builds the reference implementation (pgriewank/SAMSIM V2.0, Fortran 90) from the
# sources WHERE THEY LIE under /root/reference into oracle/_ref/ (git-ignored; nothing is copied into
# the repo).  Only runs in the build container; the GPU box uses the prebuilt binaries.
#
#   oracle/_ref/samsim_ref_dat   all reference modules unmodified + oracle/ref_hook/ref_driver.f90
#                                (writes the reference's own ./output/dat_*.dat, F9.3)
#   oracle/_ref/samsim_ref_dump  same, but the writer module mo_output is replaced by
#                                oracle/ref_hook/ref_output_hook.f90 (full-precision binary dumps)
#
# Module order follows the reference makefile (makefile:11).
set -euo pipefail
REF=${SAMSIM_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
FFLAGS=${FFLAGS:--O2}
if [ ! -d "$REF" ]; then echo "build_ref: $REF absent, keeping prebuilt oracle/_ref"; exit 0; fi
if [ ! -x "$FC" ]; then echo "build_ref: no Fortran compiler ($FC)"; exit 0; fi
MODS_A="mo_parameters mo_data mo_functions mo_init mo_thermo_functions mo_mass mo_grav_drain"
MODS_B="mo_layer_dynamics mo_flush mo_snow mo_flood mo_heat_fluxes mo_testcase_specifics mo_grotz"
for variant in dat dump; do
  B=$OUT/build_$variant
  mkdir -p "$B"
  objs=""
  for m in $MODS_A; do
    "$FC" $FFLAGS -c "$REF/$m.f90" -module-dir "$B" -o "$B/$m.o" 2>"$B/$m.log"; objs="$objs $B/$m.o"
  done
  if [ $variant = dat ]; then
    "$FC" $FFLAGS -c "$REF/mo_output.f90" -module-dir "$B" -o "$B/mo_output.o" 2>"$B/mo_output.log"
  else
    "$FC" $FFLAGS -c "$HERE/ref_hook/ref_output_hook.f90" -module-dir "$B" -o "$B/mo_output.o"
  fi
  objs="$objs $B/mo_output.o"
  for m in $MODS_B; do
    "$FC" $FFLAGS -c "$REF/$m.f90" -module-dir "$B" -o "$B/$m.o" 2>"$B/$m.log"; objs="$objs $B/$m.o"
  done
  "$FC" $FFLAGS -module-dir "$B" "$HERE/ref_hook/ref_driver.f90" $objs -o "$OUT/samsim_ref_$variant"
done
# function-level harness against the unmodified reference modules (objects of the dump variant)
B=$OUT/build_dump
FOBJS=""
for m in mo_parameters mo_data mo_functions mo_init mo_thermo_functions mo_mass mo_grav_drain mo_output mo_layer_dynamics mo_flush mo_snow mo_flood; do FOBJS="$FOBJS $B/$m.o"; done
"$FC" $FFLAGS -module-dir "$B" "$HERE/ref_hook/func_harness.f90" $FOBJS -o "$OUT/samsim_ref_func"
# run directory: forcing tables are symlinked (not copied) next to an output/ directory
mkdir -p "$OUT/run/output"
for f in flux_lw flux_sw T2m precip; do ln -sf "$REF/$f.txt.input" "$OUT/run/$f.txt.input"; done
echo "build_ref: built $OUT/samsim_ref_dat and $OUT/samsim_ref_dump"
