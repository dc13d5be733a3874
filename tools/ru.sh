#!/bin/bash
# ru.sh [extra flags] -- resource usage (VGPRs, scratch, spills) of the KSheba instantiation with the product flags
cd "$(dirname "$0")/../samsim_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -std=c++17 -DSAMSIM_BLOCK=64 -DSAMSIM_WAVES=${WAVES:-4} "$@" -c --cuda-device-only -x hip samsim_kernels.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|Scratch|Spill" | sed 's/.*remark: //' | cut -c1-100 | tail -5
