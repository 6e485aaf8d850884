#!/bin/bash
# SQ / clock counters only (one pass) for a bench.py variant.
# Usage: bash profiles/collect_sq.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --live-counters none $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- $CMD > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/pmc_grbm" -- $CMD > "$OUT/pmc_grbm.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d "$OUT/pmc_sq2" -- $CMD > "$OUT/pmc_sq2.log" 2>&1
tail -3 "$OUT/pmc_sq2.log"
