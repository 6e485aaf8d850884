#!/bin/bash
# SQ counters for the four main kernels (one rocprofv3 pass each).
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for spec in "hbv:--model hbvedu" "gr4j:--model gr4j" "cema:--model cemaneige" "fused:--model cemaneigegr4j --mode metric"; do
  tag=${spec%%:*}; args=${spec#*:}
  OUT=$ROOT/gpurun_out/prof_r02g_$tag
  mkdir -p $OUT
  CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-spot $args"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
for tag in ("hbv","gr4j","cema","fused"):
    acc=collections.defaultdict(list); name=None
    for f in glob.glob("/root/repo/gpurun_out/prof_r02g_%s/pmc_sq*/*/*_counter_collection.csv"%tag):
        for r in csv.DictReader(open(f)):
            if float(r["Grid_Size"])>=1000000 and "kernel" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "scan" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name=r["Kernel_Name"][:60]
    print(tag, name)
    for k,v in sorted(acc.items()): print("   %-22s %.4g" % (k, sum(v)/len(v)))
PY
