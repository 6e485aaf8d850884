#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default workload on the GPU
# box.  Usage (from the repo root, via gpurun):  bash profiles/collect.sh r01
# Writes raw output under gpurun_out/prof_<tag>/ ; summarise with
# profiles/summarize.py and commit the summaries under profiles/.
set -u
TAG=${1:-r00}
shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-power-soak --live-counters none $EXTRA"
# 1) kernel trace + stats (own run, no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
# 2) HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit
#    one pass; MI355X_MICROARCH.md "rocprofv3 PMC slots")
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
# 3) VALU / occupancy counters
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- $CMD > "$OUT/pmc_sq.log" 2>&1
find "$OUT" -name "*.csv" | head -50
tail -2 "$OUT/trace.log"
