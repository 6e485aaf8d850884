#!/bin/bash
# Round 6 evidence: rocprofv3 kernel trace + stats, FETCH/WRITE traffic and SQ
# counters (separate passes, profiles/collect.sh) for the headline workload,
# for every extra configuration of the bench line -- the three hysteresis /
# ice couplings included -- and the shard-size table, clock / power, every
# model and mode, and the bench line of the same build.
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
run() { tag=$1; shift; bash profiles/collect.sh $tag "$@" > /dev/null 2>&1;
        find gpurun_out/prof_$tag -name "*.db" -delete 2>/dev/null; }
run r06
run r06_hbv125k --sets 125000
run r06_hbv100k --sets 100000
run r06_hbv400ks --sets 400000 --mode storages
run r06_hbvcat --sets 10000 --catchments 125 --mode metric
run r06_gr4j --model gr4j
run r06_gr4j125k --model gr4j --mode metric --sets 125000
run r06_fused125k --model cemaneigegr4j --mode metric --sets 125000 --score nse
run r06_cema --model cemaneige
run r06_abc --model abc
run r06_hyst --model cemaneigehystgr4j --mode metric
run r06_ice --model cemaneigegr4jice --mode metric
run r06_hystice --model cemaneigehystgr4jice --mode metric
bash profiles/shard_sizes.sh > gpurun_out/r06_shard_sizes.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
cp gpurun_out/bench_detail.json gpurun_out/r06_bench_detail.json
bash profiles/clock_power.sh > gpurun_out/r06_clock_power.txt 2>&1
bash profiles/all_models.sh > gpurun_out/r06_all_models.txt 2>&1
