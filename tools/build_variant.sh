#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS" [WAVES] -- tuning / profiling build of the HIP library under samsim_amd/csrc/variants/
# (selected at run time with SAMSIM_HIP_LIB=...; never the product library)
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
cd "$HERE/samsim_amd/csrc"
mkdir -p variants
W=${3:-3}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -std=c++17 -DSAMSIM_BLOCK=64 \
  -DSAMSIM_WAVES=$W $2 -Wall -Wno-unused-function -shared -x hip samsim_kernels.hip -x hip samsim_capi.cpp -o variants/libsamsim_hip_$1.so
