#!/bin/bash
# Instruction-cache counters of the tiered kernels (several kernels of one
# launch share a CU's instruction cache when they run side by side).
# Usage (via gpurun): bash profiles/icache.sh <tag> <bench.py arguments>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/icache_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-power-soak --live-counters none $*"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc" -- $CMD > "$OUT/pmc.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    n = max(cnt[k], 1)
    req = v.get("SQC_ICACHE_REQ", 0) or 1
    print("%-70s launches %d" % (k, n))
    for c in sorted(v): print("    %-28s %.4g per launch" % (c, v[c] / n))
    print("    icache miss rate %.4f  wait_inst/wave_cycles %.3f" % (
        v.get("SQC_ICACHE_MISSES", 0) / req,
        v.get("SQ_WAIT_INST_ANY", 0) / (v.get("SQ_WAVE_CYCLES", 0) or 1)))
PY
