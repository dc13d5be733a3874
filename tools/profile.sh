#!/bin/bash
# profile.sh OUTDIR [bench args...] -- rocprofv3 passes over `python3 bench.py --no-cpu-baseline --no-extra <args>` on the GPU box:
# kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and two SQ passes, each in its own run (--pmc never combined with traces).
set -e
OUT="$1"; shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$REPO/$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/trace" -- python3 $B > "$REPO/$OUT/bench_under_rocprof.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$REPO/$OUT/pmc_fetch" -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$REPO/$OUT/pmc_write" -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d "$REPO/$OUT/pmc_sq1" -- python3 $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d "$REPO/$OUT/pmc_sq2" -- python3 $B > /dev/null 2>&1
cd "$REPO"
