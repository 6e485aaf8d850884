#!/bin/bash
# Round 3 evidence: rocprofv3 kernel trace + stats, FETCH/WRITE traffic and SQ
# counters (separate passes, profiles/collect.sh) for the headline workload,
# GR4J (1M and one GPU's 125k shard) and the fused shard, plus the shard-size
# table and the bench line of the same build.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
bash profiles/collect.sh r03 > /dev/null 2>&1
bash profiles/collect.sh r03_gr4j --model gr4j > /dev/null 2>&1
bash profiles/collect.sh r03_gr4j125k --model gr4j --mode metric --sets 125000 > /dev/null 2>&1
bash profiles/collect.sh r03_fused125k --model cemaneigegr4j --mode metric --sets 125000 > /dev/null 2>&1
bash profiles/collect.sh r03_cema --model cemaneige > /dev/null 2>&1
cd $ROOT
for t in r03 r03_gr4j r03_gr4j125k r03_fused125k r03_cema; do
  # keep the raw per-pass CSVs small enough to come back: counters + traces
  find gpurun_out/prof_$t -name "*.db" -delete 2>/dev/null
done
bash profiles/shard_sizes.sh > gpurun_out/r03_shard_sizes.txt 2>&1
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
bash profiles/clock_power.sh > gpurun_out/r03_clock_power.txt 2>&1
